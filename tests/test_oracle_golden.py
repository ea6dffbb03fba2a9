"""The oracle (oracle/*.py) against outputs of the reference itself
(tests/golden/*.npz, produced by tools/make_golden.py).  CPU only."""
import warnings

import numpy as np
import pytest
import torch

from oracle import nets, scorers, synth


def test_decode_matches_reference(golden_scorers):
    g = golden_scorers
    for i in range(g["hm"].shape[0]):
        for fn in (scorers.decode_heatmaps, scorers.decode_closed_form):
            d = fn(g["hm"][i], g["bbox"][i])
            assert np.array_equal(d["idx"], g["idx"][i])                      # integer: bit-exact
            assert np.array_equal(d["maxvals"], g["maxvals"][i])              # copied values: bit-exact
            np.testing.assert_allclose(d["coords"], g["coords"][i], rtol=1e-6, atol=1e-4)
    # the 3-point-solve route must be bit-identical to the reference's float32 result
    for i in range(g["hm"].shape[0]):
        assert np.array_equal(scorers.decode_heatmaps(g["hm"][i], g["bbox"][i])["coords"], g["coords"][i])


def test_decode_appendix_b_known_answers(golden_scorers):
    """SURVEY.md Appendix B, box [100,50,196,178] (item 5 of the fixture)."""
    g = golden_scorers
    d = scorers.decode_heatmaps(g["hm"][5], g["bbox"][5])
    expect = {0: (140.0, 70.0), 1: (100.0, 50.0), 2: (102.0, 52.0), 3: (104.5, 53.5), 4: (191.5, 174.0), 5: (194.0, 176.0)}
    for j, xy in expect.items():
        assert tuple(d["coords"][j]) == xy, j
    assert d["idx"][0] == 10 * 48 + 20


@pytest.mark.parametrize("nt", ["softmax", "sigmoid", "divide_sum"])
def test_softargmax_decode(golden_scorers, nt):
    g = golden_scorers
    for i in range(5):
        src = g["hm"][i] if nt != "divide_sum" else np.abs(g["hm"][i]) + 1e-3
        d = scorers.softargmax_decode(src, g["bbox"][i], nt)
        np.testing.assert_allclose(d["coords"], g[f"soft_{nt}_coords"][i], rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(d["maxvals"], g[f"soft_{nt}_scores"][i], rtol=1e-5)


def test_localpeak(golden_scorers):
    g = golden_scorers
    assert np.array_equal(scorers.localpeak_values(g["toy"]), g["toy_vals"])        # [4 3]
    assert scorers.localpeak_mean(np.stack([g["toy"]] * 3)) == g["toy_mean"] == 3.5
    for i in range(g["hm"].shape[0]):
        cnt, _, mean = scorers.localpeak_stats(g["hm"][i])
        assert np.array_equal(cnt, g["lp_cnt"][i])                                 # integer: bit-exact
        if np.isnan(g["lp_mean"][i]):
            assert np.isnan(mean)
        else:
            np.testing.assert_allclose(mean, g["lp_mean"][i], rtol=1e-6)
    assert np.isnan(g["lp_mean"][6])            # all-negative item -> nan (Appendix B)


def test_thc_tpc(golden_scorers):
    g = golden_scorers
    hm, bb = g["hm"], g["bbox"]
    for i in range(hm.shape[0] - 1):
        np.testing.assert_allclose(scorers.thc_pair(hm[i], hm[i + 1], "L1"), g["thc_l1"][i], rtol=1e-6)
        np.testing.assert_allclose(scorers.thc_pair(hm[i], hm[i + 1], "L2"), g["thc_l2"][i], rtol=1e-6)
        thr = 0.01 * np.sqrt((bb[i][2] - bb[i][0]) * (bb[i][3] - bb[i][1]))
        cur = scorers.decode_heatmaps(hm[i], bb[i])["coords"]
        assert scorers.tpc_pair(cur, hm[i + 1], bb[i], thr) == g["tpc"][i]
    # neighbour rule: doubled with exactly one neighbour
    assert scorers.combine_neighbours(3, 5, True, True) == 8
    assert scorers.combine_neighbours(3, 5, True, False) == 6
    assert scorers.combine_neighbours(3, 5, False, True) == 10
    assert scorers.combine_neighbours(3, 5, False, False) == 0


def test_hybrid_and_wpu(golden_scorers):
    g = golden_scorers
    np.testing.assert_allclose(scorers.hybrid_feature(g["lit_bbox"], g["lit_kp"]), g["lit_feat"], rtol=1e-12, atol=1e-12)
    w42 = {k[5:]: g[k] for k in g.files if k.startswith("ae42.")}
    w38 = {k[5:]: g[k] for k in g.files if k.startswith("ae38.")}
    for i in range(g["hm"].shape[0]):
        if not g["kp_ok"][i]:
            with pytest.raises(AssertionError):
                scorers.hybrid_feature(scorers.xyxy_to_xywh(g["bbox"][i]), g["kp"][i])
            continue
        f = scorers.hybrid_feature(scorers.xyxy_to_xywh(g["bbox"][i].tolist()), g["kp"][i])
        np.testing.assert_allclose(f, g["hybrid"][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(scorers.wpu_item(g["bbox"][i].tolist(), g["kp"][i], w42), g["wpu42"][i], rtol=2e-5)
        np.testing.assert_allclose(scorers.wpu_item(g["bbox"][i].tolist(), g["kp"][i], w42, only38=True), g["wpu38"][i], rtol=2e-5)
        f38 = f[np.r_[0:3, 5:20, 22:42]].astype(np.float32)
        r38 = scorers.ae_forward(f38, w38)
        np.testing.assert_allclose(float(np.mean((r38 - f38) ** 2)), g["wpu38cls"][i], rtol=2e-5)


def test_oks_and_accuracy(golden_scorers):
    g = golden_scorers
    for i in range(g["hm"].shape[0]):
        np.testing.assert_allclose(scorers.oks(g["bb_ann"][i], g["kp"][i], g["gt"][i]), g["oks"][i], rtol=1e-12)
    tgt, mask = synth.gaussian_targets(g["hm"].shape[0], seed=int(g["acc_targets_seed"]))
    np.testing.assert_allclose(scorers.heatmap_accuracy(g["hm"] * mask, tgt * mask), g["acc"], rtol=1e-12)


def test_simplepose_restatement_matches_reference(golden_simplepose):
    g = golden_simplepose
    torch.manual_seed(0)
    m = nets.SimplePoseRef(50)
    keys = list(m.state_dict().keys())
    assert keys == list(g["keys"])                                               # drop-in: same keys, same order
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == list(g["shapes"])
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m.eval()
    x = torch.from_numpy(synth.crops(int(g["batch"])))
    with torch.no_grad():
        hm = m(x).numpy()
        emb = m.get_embedding(x).numpy()
    scale = np.abs(g["heatmaps"]).max()
    assert np.abs(hm - g["heatmaps"]).max() <= 1e-5 * scale
    np.testing.assert_allclose(emb, g["embedding"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(hm.reshape(2, 17, -1).argmax(2), g["heatmaps"].reshape(2, 17, -1).argmax(2))


def test_masked_mse_and_adamw_against_torch():
    r = np.random.RandomState(0)
    out = r.standard_normal((3, 17, 64, 48)).astype(np.float32)
    tgt, mask = synth.gaussian_targets(3, seed=2)
    loss, grad = scorers.masked_mse(out, tgt, mask)
    o = torch.from_numpy(out).requires_grad_()
    l = 0.5 * torch.nn.MSELoss()(o.mul(torch.from_numpy(mask)), torch.from_numpy(tgt).mul(torch.from_numpy(mask)))
    l.backward()
    np.testing.assert_allclose(loss, l.item(), rtol=1e-5)
    np.testing.assert_allclose(grad, o.grad.numpy(), rtol=1e-5, atol=1e-12)

    p = torch.from_numpy(r.standard_normal(1000).astype(np.float32)).requires_grad_()
    opt = torch.optim.AdamW([p], lr=2.5e-3, weight_decay=0.7)
    pn, m, v = p.detach().numpy().copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    for step in range(1, 4):
        gnp = r.standard_normal(1000).astype(np.float32)
        p.grad = torch.from_numpy(gnp.copy())
        opt.step()
        pn, m, v = scorers.adamw_step(pn, gnp, m, v, step, 2.5e-3, 0.7)
        np.testing.assert_allclose(pn, p.detach().numpy(), rtol=2e-5, atol=1e-6)


def test_adam_and_sgd_against_torch():
    r = np.random.RandomState(3)
    p0 = r.standard_normal(1000).astype(np.float32)
    pa = torch.from_numpy(p0.copy()).requires_grad_(); ps = torch.from_numpy(p0.copy()).requires_grad_()
    oa = torch.optim.Adam([pa], lr=2.5e-4)
    os_ = torch.optim.SGD([ps], lr=2.5e-4, momentum=0.9, weight_decay=0.0005)
    a, m, v = p0.copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    b, buf = p0.copy(), np.zeros(1000, np.float32)
    for step in range(1, 5):
        g = r.standard_normal(1000).astype(np.float32)
        pa.grad = torch.from_numpy(g.copy()); oa.step()
        ps.grad = torch.from_numpy(g.copy()); os_.step()
        a, m, v = scorers.adam_step(a, g, m, v, step, 2.5e-4)
        b, buf = scorers.sgd_step(b, g, buf, step, 2.5e-4, 0.9, 0.0005)
        np.testing.assert_allclose(a, pa.detach().numpy(), rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(b, ps.detach().numpy(), rtol=2e-6, atol=1e-7)


@pytest.fixture(scope="module")
def golden_nets2():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_hrnet.npz"))


@pytest.mark.parametrize("name,ctor", [("fastpose", nets.FastPoseRef), ("hrnet", nets.HRNetRef)])
def test_fastpose_hrnet_restatements_match_reference(golden_nets2, name, ctor):
    g = golden_nets2
    m = ctor()
    assert list(m.state_dict().keys()) == list(g[f"{name}_keys"])
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == list(g[f"{name}_shapes"])
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m.eval()
    x = torch.from_numpy(synth.crops(int(g["batch"])))
    with torch.no_grad():
        hm = m(x).numpy()
    ref = g[f"{name}_heatmaps"]
    assert np.abs(hm - ref).max() <= 1e-5 * np.abs(ref).max()
    assert np.array_equal(hm.reshape(2, 17, -1).argmax(2), ref.reshape(2, 17, -1).argmax(2))
    if name == "fastpose":
        with torch.no_grad():
            np.testing.assert_allclose(m.get_embedding(x).numpy(), g["fastpose_embedding"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,ctor", [("simplepose_r50", nets.SimplePoseRef), ("fastpose_r50", nets.FastPoseRef), ("hrnet_w32", nets.HRNetRef)])
def test_restatements_reproduce_the_wide_index_pin(name, ctor):
    """tests/golden/widepin.npz (64 crops per network from the reference, tools/make_golden.py::gen_widepin): the oracle graphs give the
    reference's arg-max indices, peak values and (through oracle.scorers.decode_heatmaps) its decoded key-points on the first 8 crops, and
    the kept heat-maps to 1e-5."""
    import os
    from oracle import scorers
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "widepin.npz"))
    n = 8
    assert int(g[f"{name}_n"]) >= 64 and g[f"{name}_idx"].shape == (64, 17)
    m = ctor()
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m.eval()
    x = torch.from_numpy(synth.crops(64, seed=int(g["seed"]))[:n])
    bb = synth.bboxes(64, seed=int(g["seed"]))
    with torch.no_grad():
        hm = m(x).numpy()
    keep = g[f"{name}_heatmaps"]
    assert np.abs(hm[:keep.shape[0]] - keep).max() <= 1e-5 * np.abs(keep).max()
    assert np.array_equal(hm.reshape(n, 17, -1).argmax(2), g[f"{name}_idx"][:n])
    np.testing.assert_allclose(hm.reshape(n, 17, -1).max(2), g[f"{name}_maxval"][:n], rtol=0, atol=1e-5 * float(g[f"{name}_absmax"]))
    for i in range(keep.shape[0]):                                 # the reference's decode of ITS maps == the oracle's decode of the same maps
        d = scorers.decode_heatmaps(keep[i], bb[i])
        assert np.array_equal(d["idx"], g[f"{name}_idx"][i])
        np.testing.assert_allclose(d["coords"], g[f"{name}_keypoints"][i], rtol=1e-6, atol=1e-4)
    assert np.array_equal(g[f"{name}_idx_f64"], g[f"{name}_idx"])   # on this fixture the reference's fp32 and float64 runs pick the same pixels
    assert float(g[f"{name}_gap"].min()) > 0.0


def test_fastpose_r152_384_restatement_matches_reference():
    """BASELINE.json config 5: FastPose-R152 at 384x288 (96x72 heat-maps)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_r152_384.npz"))
    m = nets.FastPoseRef(152)
    assert list(m.state_dict().keys()) == list(g["keys"])
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m.eval()
    x = torch.from_numpy(synth.crops(1, hw=(384, 288)))
    with torch.no_grad():
        hm = m(x).numpy()
        emb = m.get_embedding(x).numpy()
    ref = g["heatmaps"]
    assert hm.shape == (1, 17, 96, 72)
    assert np.abs(hm - ref).max() <= 1e-5 * np.abs(ref).max()
    np.testing.assert_allclose(emb, g["embedding"], rtol=1e-4, atol=1e-4)


def test_peak_local_max_pinned_on_real_scikit_image():
    """Row a13: `peak_local_max_5`, `mpe_item`, `margin_item` against tests/golden/peaks.npz = scikit-image 0.18.3's own
    `peak_local_max(min_distance=5, num_peaks=5)` and the reference's compute_mpe / compute_margin (ActiveLearning.py:
    762-788) run on top of it.  Exact positions / order / counts wherever the candidates' values are pairwise distinct;
    the order among exactly equal maxima depends on the scikit-image version (oracle/scorers.py) and is only counted."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peaks.npz"))
    assert str(g["skimage_version"]) == "0.18.3"
    hm = synth.peak_items(24, seed=int(g["items_seed"]))
    tied_differ = 0
    for n in range(hm.shape[0]):
        for j in range(17):
            loc = scorers.peak_local_max_5(hm[n, j])
            k = int(g["cnt"][n, j])
            same = len(loc) == k and np.array_equal(loc, g["loc"][n, j, :k])
            if g["tied"][n, j]:
                tied_differ += not same
            else:
                assert same, (n, j, loc.tolist(), g["loc"][n, j, :k].tolist())
    # 16 of the 408 planes have exactly equal candidate maxima; on 4 of them scikit-image 0.18.3's unstable argsort orders the equal values
    # differently from the stable order the oracle (and the reference's 0.24 pin) uses — counted, documented in tests/golden/README.md
    assert int((~g["tied"]).sum()) == 392 and int(g["tied"].sum()) == 16 and tied_differ == 4
    free = ~g["item_tied"]
    assert free.sum() >= 20
    mpe = np.array([scorers.mpe_item(h) for h in hm]); mar = np.array([scorers.margin_item(h) for h in hm])
    np.testing.assert_allclose(mpe[free], g["mpe"][free], rtol=1e-6)
    np.testing.assert_allclose(mar[free], g["margin"][free], rtol=1e-6)
    for name in g["case_names"]:
        name = str(name)
        if not bool(g["tied_" + name]):
            assert scorers.peak_local_max_5(g["case_" + name]).tolist() == g["loc_" + name].tolist(), name
    # the cases that decide the border and spacing rules, spelled out
    assert g["loc_border_rows_cols"].tolist() == [[5, 20], [58, 30]]
    assert g["loc_border_shadow"].tolist() == [[40, 5]]
    assert len(g["loc_distinct_5_apart"]) == 2 and len(g["loc_distinct_6_apart"]) == 4
    assert len(g["loc_nine_distinct_peaks"]) == 5


def test_multi_peak_criteria_restatement():
    """MPE / Margin / Entropy (ActiveLearning.py:762-796) on cases whose answer follows from the documented algorithm
    (the tie cases use the stable order of scikit-image >= 0.19; the pin on a real scikit-image is the test above);
    softmax / entropy are scipy's own."""
    from scipy.special import softmax
    from scipy.stats import entropy
    h = np.zeros((64, 48), np.float32)
    h[10, 10], h[30, 30], h[50, 20] = 3.0, 2.0, 1.0
    h[10, 14] = 2.5                                            # inside the 11x11 window of (10,10): not a local maximum
    h[4, 30] = 9.0                                             # in the excluded border
    loc = scorers.peak_local_max_5(h)
    assert loc.tolist() == [[10, 10], [30, 30], [50, 20]]
    hm = np.stack([h, np.full_like(h, 0.5)])
    peaks = np.array([3.0, 2.0, 1.0], np.float32)
    np.testing.assert_allclose(scorers.mpe_item(hm), entropy(softmax(peaks)), rtol=1e-6)
    np.testing.assert_allclose(scorers.margin_item(hm), 1.0)
    p = h + 1.0
    np.testing.assert_allclose(scorers.entropy_item(p[None]), entropy(p.flatten()), rtol=1e-6)
    assert scorers.entropy_item(-p[None]) == entropy(-p.flatten())          # scipy normalises by the (negative) sum
    g = h.copy(); g[0, 0] = -1.0
    assert scorers.entropy_item(g[None]) == -np.inf
    # plateau: equal maxima 4 apart -> the later one (row-major) is rejected; exactly 5 apart -> both stay
    q = np.zeros((64, 48), np.float32); q[20, 20] = q[20, 24] = 1.0
    assert scorers.peak_local_max_5(q).tolist() == [[20, 20]]
    q = np.zeros((64, 48), np.float32); q[20, 20] = q[20, 25] = 1.0
    assert scorers.peak_local_max_5(q).tolist() == [[20, 20], [20, 25]]


@pytest.mark.parametrize("norm", ["softmax", "sigmoid", "divide_sum"])
def test_l1_joint_regression_restatement_matches_reference(norm):
    """LOSS.TYPE L1JointRegression: loss, predicted joints and heat-map gradients against the reference module."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "l1_joint_regression.npz"))
    hm, gt, vis = synth.l1_inputs(norm)
    h = torch.from_numpy(hm).requires_grad_()
    loss, jts = nets.l1_joint_regression(h, torch.from_numpy(gt), torch.from_numpy(vis), norm)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g[f"{norm}_loss"]), rtol=1e-6)
    np.testing.assert_allclose(jts.detach().numpy(), g[f"{norm}_jts"], rtol=1e-5, atol=1e-6)
    got = h.grad.reshape(-1)[torch.from_numpy(g[f"{norm}_grad_idx"])].numpy()
    np.testing.assert_allclose(got, g[f"{norm}_grad_val"], rtol=1e-4, atol=1e-6 * float(g[f"{norm}_grad_absmax"]))


@pytest.mark.parametrize("tag,hm_hw,in_hw,sigma", [("a", (64, 48), (256, 192), 2), ("b", (96, 72), (384, 288), 1.5)])
def test_target_generator_restatement_matches_reference(tag, hm_hw, in_hw, sigma):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    joints, vis = synth.target_joints(6, hm_hw, in_hw, seed=41)
    for n in range(6):
        t, w = scorers.target_generator(joints[n], vis[n], hm_hw, in_hw, sigma)
        assert np.array_equal(t, g[f"{tag}_target"][n]) and np.array_equal(w, g[f"{tag}_weight"][n])


@pytest.mark.parametrize("tag,in_hw", [("a", (256, 192)), ("b", (384, 288))])
def test_crop_geometry_restatement_matches_reference(tag, in_hw):
    """Centre/scale, the 2x3 crop matrix (with rotation), the corrected box and a transformed point against the reference's
    own _box_to_center_scale / get_affine_transform / _center_scale_to_box / affine_transform (tests/golden/crop.npz)."""
    import os
    from oracle import crop
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "crop.npz"))
    box, rot = synth.crop_cases(24)
    for i, ((xmin, ymin, xmax, ymax), r) in enumerate(zip(box.tolist(), rot.tolist())):
        c, s = crop.box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, float(in_hw[1]) / in_hw[0])
        assert c.dtype == np.float32 and np.array_equal(c, g[f"{tag}_center"][i])
        assert s.dtype == np.float32 and np.array_equal(s, g[f"{tag}_scale"][i])
        t = crop.affine_transform_matrix(c, s, r, [in_hw[1], in_hw[0]])
        np.testing.assert_allclose(t, g[f"{tag}_trans"][i], rtol=1e-13, atol=1e-11)       # float64 solve, same points
        # the fixture was produced under numpy 2 (float32 scalar arithmetic); the pinned numpy 1.23.5 computes in float64
        np.testing.assert_allclose(np.float32(crop.center_scale_to_box(c, s)), np.float32(g[f"{tag}_box"][i]), rtol=2e-7)
        np.testing.assert_allclose(crop.transform_point(np.float32([xmin, ymax]), t), g[f"{tag}_pt"][i], rtol=1e-12, atol=1e-9)


def test_crop_tensor_conversion_matches_reference():
    """im_to_torch incl. its 'only divide when max > 1' rule, then the mean shift."""
    import os
    from oracle import crop
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "crop.npz"))
    frame = synth.u8_frame(40, 56)
    dark = (frame > 250).astype(np.uint8)
    mean = np.float32(crop.MEAN).reshape(3, 1, 1)
    assert dark.max() == 1
    assert np.array_equal(crop.image_to_tensor(frame), g["tensor_bright"] + (-mean))
    assert np.array_equal(crop.image_to_tensor(dark), g["tensor_dark"] + (-mean))
    assert g["tensor_dark"].max() == 1.0


def test_warp_restatement_properties():
    """cv2 is absent (parity unpinned, oracle/crop.py): the fixed-point warp is checked through properties —
    identity and integer shifts are exact copies, the weight table sums to 2^15, the result stays within the 1/32-pixel
    quantisation bound of an independent float64 bilinear warp, and the border is constant 0."""
    from oracle import crop
    tab = crop.bilinear_weight_table().astype(np.int64)
    assert (tab.sum(1) == 1 << 15).all() and tuple(tab[0]) == (32767, 0, 0, 1) and tuple(tab[33]) == (31 * 31 * 32, 31 * 32, 31 * 32, 32)
    f = synth.u8_frame(60, 80)
    assert np.array_equal(crop.warp_affine_u8(f, [[1, 0, 0], [0, 1, 0]], (80, 60)), f)
    sh = crop.warp_affine_u8(f, [[1, 0, 7], [0, 1, -3]], (80, 60))
    assert np.array_equal(sh[:57, 7:], f[3:, :73]) and (sh[57:] == 0).all() and (sh[:, :7] == 0).all()
    box, rot = synth.crop_cases(8)
    big = synth.u8_frame(480, 640)
    smooth = np.stack([big[..., 0], big[..., 1], big[..., 0] // 2 + big[..., 1] // 2], 2)     # gradients: |d/dpx| <= ~21 with noise
    for (xmin, ymin, xmax, ymax), r in zip(box.tolist(), rot.tolist()):
        c, s = crop.box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, 0.75)
        t = crop.affine_transform_matrix(c, s, r, [192, 256])
        got = crop.warp_affine_u8(smooth, t, (192, 256)).astype(np.float64)
        ref = crop.float_bilinear_reference(smooth, t, (192, 256))
        # coordinates are rounded to 1/32 px (<= 1/64 px error per axis): bound = max local gradient * 2/64 + rounding 0.5
        grad = max(np.abs(np.diff(smooth.astype(np.float64), axis=0)).max(), np.abs(np.diff(smooth.astype(np.float64), axis=1)).max())
        mi = crop.invert_affine(t)
        gx, gy = np.meshgrid(np.arange(192.0), np.arange(256.0))
        fx, fy = mi[0, 0] * gx + mi[0, 1] * gy + mi[0, 2], mi[1, 0] * gx + mi[1, 1] * gy + mi[1, 2]
        inside = (fx > 1) & (fx < 638) & (fy > 1) & (fy < 478)                               # taps that never see the border
        assert np.abs(got - ref)[inside].max() <= grad * (2.0 / 64) * 2 + 0.5 + 1e-9
        assert np.abs(got - ref).max() <= 255 * (2.0 / 64) * 2 + 0.5                         # frame edge: step to the border value
        outside = (fx < -1.1) | (fx > 640.1) | (fy < -1.1) | (fy > 480.1)
        assert (got[outside] == 0).all()
