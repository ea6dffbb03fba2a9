"""The reference-named Python surface (alphapose.utils.*, active_learning.*) on MI355X
vs the golden vectors / oracle: same signatures, same results."""
import numpy as np
import pytest
import torch

from oracle import scorers, synth
from tests.gpu_util import dev, record, rel_err, to_dev

pytestmark = pytest.mark.gpu


def test_heatmap_to_coord_simple_per_item(golden_scorers):
    from alphapose.utils.config import edict
    from alphapose.utils.transforms import get_func_heatmap_to_coord, get_max_pred
    g = golden_scorers
    f = get_func_heatmap_to_coord(edict({"DATA_PRESET": {"TYPE": "simple"}, "LOSS": {"TYPE": "MSELoss"}}))
    for i in range(g["hm"].shape[0]):
        for src in (g["hm"][i], torch.from_numpy(g["hm"][i]), to_dev(g["hm"][i])):        # ndarray / cpu tensor / device tensor
            c, m = f(src, g["bbox"][i].tolist(), hm_shape=(64, 48), norm_type=None)
            assert c.shape == (17, 2) and m.shape == (17, 1) and c.dtype == np.float32
            assert np.array_equal(c, g["coords"][i]) and np.array_equal(m, g["maxvals"][i])
    p, m = get_max_pred(g["hm"][5])
    want = scorers.argmax_peaks(g["hm"][5])
    assert np.array_equal(p, want[1]) and np.array_equal(m, want[2])


@pytest.mark.parametrize("nt", ["softmax", "sigmoid", "divide_sum"])
def test_softargmax_decode(golden_scorers, nt):
    from alphapose.utils.transforms import heatmap_to_coord_simple_regress
    g = golden_scorers
    for i in range(5):
        src = g["hm"][i] if nt != "divide_sum" else np.abs(g["hm"][i]) + 1e-3
        c, s = heatmap_to_coord_simple_regress(torch.from_numpy(src), g["bbox"][i].tolist(), (64, 48), nt)
        record("softargmax_" + nt, max_abs=float(np.abs(c - g[f"soft_{nt}_coords"][i]).max()))
        np.testing.assert_allclose(c, g[f"soft_{nt}_coords"][i], rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(s, g[f"soft_{nt}_scores"][i], rtol=1e-5)


def test_tpc_stream(golden_scorers):
    import vatl_hip as vh
    g = golden_scorers
    hm, bb = to_dev(g["hm"]), to_dev(g["bbox"])
    n = hm.shape[0]
    coords, _, _ = vh.decode(hm, bb)
    is_prev = np.array([0, 1, 1, 0, 1, 0, 1, 1], np.uint8)
    is_next = np.array([1, 1, 0, 1, 0, 1, 1, 0], np.uint8)
    got = vh.tpc_stream(hm, bb, coords, to_dev(is_prev, torch.uint8), to_dev(is_next, torch.uint8)).cpu().numpy()
    for i in range(n):
        want = scorers.tpc_item(g["hm"][i], g["hm"][i - 1] if i else None, g["hm"][i + 1] if i + 1 < n else None,
                                g["bbox"][i].tolist(), is_prev[i], is_next[i])
        assert got[i] == want, (i, got[i], want)                                           # integer counts: exact
    # the reference's own pair counts (golden): item i vs i+1 with item i's box
    only_next = vh.tpc_stream(hm, bb, coords, to_dev(np.zeros(n, np.uint8), torch.uint8), to_dev(np.r_[np.ones(n - 1), 0].astype(np.uint8), torch.uint8))
    assert np.array_equal(only_next.cpu().numpy()[:-1], 2 * g["tpc"])


def test_local_peak_api(golden_scorers):
    from active_learning.local_peak import localpeak_mean, localpeak_values
    g = golden_scorers
    assert np.array_equal(localpeak_values(g["toy"].astype(np.float32)), g["toy_vals"])   # [4 3]
    assert localpeak_mean(np.stack([g["toy"]] * 3).astype(np.float32)) == 3.5
    for i in (0, 5, 7):
        np.testing.assert_allclose(localpeak_mean(g["hm"][i]), g["lp_mean"][i], rtol=1e-5)
        for j in (0, 6, 7, 9):
            assert np.array_equal(localpeak_values(g["hm"][i, j]), scorers.localpeak_values(g["hm"][i, j]))
    assert np.isnan(localpeak_mean(g["hm"][6]))


def test_hybrid_feature_and_autoencoder_api(golden_scorers):
    from active_learning.Whole_body_AE.AutoEncoder import WholeBodyAE
    from active_learning.Whole_body_AE.hybrid_feature import compute_hybrid
    g = golden_scorers
    f = compute_hybrid(g["lit_bbox"].tolist(), g["lit_kp"].tolist())                        # the reference's literal fixture
    assert f.shape == (42,) and f.dtype == np.float64
    np.testing.assert_allclose(f, g["lit_feat"], rtol=1e-12, atol=1e-13)
    with pytest.raises(AssertionError):
        compute_hybrid([0, 0, 10, 0], g["lit_kp"].tolist())
    with pytest.raises(AssertionError):
        compute_hybrid([0, 0, 10, 10], [0.0] * 51)
    for d, pre in ((42, "ae42."), (38, "ae38.")):
        ae = WholeBodyAE(z_dim=4, input_dim=d)
        sd = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
        assert list(ae.state_dict().keys()) == list(sd.keys())
        ae.load_state_dict(sd, strict=True)
        ae = ae.to(dev()).eval()
        i = int(np.flatnonzero(g["kp_ok"])[0])
        x = g["hybrid"][i].astype(np.float32)
        x = x if d == 42 else x[np.r_[0:3, 5:20, 22:42]]
        y = ae(to_dev(x)).cpu().numpy()
        np.testing.assert_allclose(y, scorers.ae_forward(x, {k: v.numpy() for k, v in sd.items()}), rtol=1e-5, atol=1e-6)
    assert WholeBodyAE(z_dim=2).input_dim == 38 and WholeBodyAE(z_dim=2, kp_direct=True).input_dim == 51


def test_calc_accuracy(golden_scorers):
    from alphapose.utils.metrics import DataLogger, calc_accuracy
    g = golden_scorers
    tgt, mask = synth.gaussian_targets(g["hm"].shape[0], seed=int(g["acc_targets_seed"]))
    acc = calc_accuracy(to_dev(g["hm"] * mask), to_dev(tgt * mask))
    np.testing.assert_allclose(acc, g["acc"], rtol=1e-12)
    lg = DataLogger(); lg.update(2.0, 3); lg.update(4.0, 1)
    assert lg.avg == 2.5 and lg.cnt == 4


def test_score_batch_all_scores(golden_scorers):
    import vatl_hip as vh
    from active_learning.scoring import score_batch
    g = golden_scorers
    sd42 = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ae42.")}
    n = g["hm"].shape[0]
    is_prev, is_next = synth.video_flags(n, 2)
    s = score_batch(to_dev(g["hm"]), to_dev(g["bbox"]), to_dev(is_prev, torch.uint8), to_dev(is_next, torch.uint8),
                    ae_flat=vh.pack_ae(sd42, dev()), ae_dims=(42, 4))
    kp = s.keypoints.cpu().numpy().reshape(n, 51)
    assert np.array_equal(kp, g["kp"].astype(np.float32))
    np.testing.assert_allclose(s.hp.cpu().numpy(), -g["maxvals"].sum(axis=(1, 2)), rtol=1e-5)
    for i in range(n):
        np.testing.assert_allclose(float(s.pose_score[i]), scorers.pose_score(g["maxvals"][i]), rtol=1e-5)
        want = scorers.thc_item(g["hm"][i], g["hm"][i - 1] if i else None, g["hm"][i + 1] if i + 1 < n else None, is_prev[i], is_next[i])
        np.testing.assert_allclose(float(s.thc[i]), want, rtol=1e-5)
    ok = g["kp_ok"]
    np.testing.assert_allclose(s.wpu.cpu().numpy()[ok], g["wpu42"][ok], rtol=1e-4)


def test_mpe_margin_entropy_criteria():
    """MPE / Margin (<= 5 local peaks per plane: positions and counts exact) and Entropy
    (scipy semantics incl. -inf on signed maps) against the oracle; the hand-made tie cases follow the oracle's stable
    order among equal maxima (scikit-image >= 0.19) — the pin on a real scikit-image is the next test."""
    import vatl_hip as vh
    from active_learning.scoring import multi_peak_scores
    from oracle import scorers, synth
    from tests.gpu_util import record, rel_err, to_dev
    hm = synth.blob_heatmaps(6, seed=21)                      # signed (noise) maps with 1-3 bumps per joint
    r = np.random.RandomState(4)
    hm[1, 3] = 0.25                                           # constant plane: no peak
    hm[2, 5] = 0.0; hm[2, 5, 20, 20] = 1.0; hm[2, 5, 20, 24] = 1.0; hm[2, 5, 20, 25] = 1.0   # plateau ties inside / at the spacing
    hm[3, 0] = 0.0; hm[3, 0, 4, 10] = 2.0; hm[3, 0, 5, 10] = 1.0; hm[3, 0, 40, 5] = 0.5         # a border maximum is dropped but still shadows (5,10); (40,5) is the first non-border column
    hm[4] = np.abs(hm[4]) + 1e-3                              # non-negative item: finite entropy
    hm[5, 2] = 0.0                                            # zero plane: entropy nan, no peaks
    hm[0, 1] = 0.0; hm[0, 1, 30, 20] = -1.0                   # a plateau of thousands of equal candidates (list overflow path)
    d = to_dev(hm)
    val, idx, cnt, mpe, mar = vh.peaks5(d, 5)
    val, idx, cnt = val.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    for n in range(hm.shape[0]):
        for j in range(hm.shape[1]):
            loc = scorers.peak_local_max_5(hm[n, j])
            assert cnt[n, j] == len(loc), (n, j)
            assert np.array_equal(idx[n, j, :len(loc)], loc[:, 0] * hm.shape[3] + loc[:, 1]), (n, j)
            assert np.array_equal(val[n, j, :len(loc)], hm[n, j][loc[:, 0], loc[:, 1]])
            assert (idx[n, j, len(loc):] == -1).all()
    assert cnt[1, 3] == 0 and cnt[5, 2] == 0 and cnt[2, 5] == 2 and list(idx[3, 0]) == [40 * hm.shape[3] + 5, -1, -1, -1, -1]
    want_mpe = np.array([scorers.mpe_item(h) for h in hm]); want_mar = np.array([scorers.margin_item(h) for h in hm])
    got_mpe, got_mar = multi_peak_scores(d, "MPE").cpu().numpy(), multi_peak_scores(d, "Margin").cpu().numpy()
    record("mpe_margin", mpe_rel=rel_err(got_mpe, want_mpe), margin_rel=rel_err(got_mar, want_mar))
    np.testing.assert_allclose(got_mpe, want_mpe, rtol=1e-5)
    np.testing.assert_allclose(got_mar, want_mar, rtol=1e-6)
    # the wave-per-plane kernel (64 x 48, min_distance 5, 16-byte aligned) against the block kernel (any size; reached here through a
    # 4-byte-offset view) on a larger random set: same peaks, same order, same criteria, bit for bit
    big = synth.blob_heatmaps(48, seed=33)
    big[5] = np.round(big[5] * 4) / 4                          # coarse values: many exact ties
    flat = torch.empty(big.size + 1, device=d.device, dtype=torch.float32)
    flat[1:] = to_dev(big).reshape(-1)
    unaligned = flat[1:].view(big.shape)
    assert unaligned.data_ptr() % 16 != 0
    fast, slow = vh.peaks5(to_dev(big), 5), vh.peaks5(unaligned, 5)
    for a_, b_ in zip(fast, slow):
        assert torch.equal(a_, b_)
    want_ent = np.array([scorers.entropy_item(h) for h in hm])
    got_ent = multi_peak_scores(d, "Entropy").cpu().numpy()
    fin = np.isfinite(want_ent)
    assert fin[4] and not fin[0] and np.isnan(want_ent[5])
    assert np.array_equal(np.isnan(got_ent), np.isnan(want_ent)) and np.array_equal(np.isneginf(got_ent), np.isneginf(want_ent))
    np.testing.assert_allclose(got_ent[fin], want_ent[fin], rtol=1e-5)


def test_mpe_margin_pinned_on_real_scikit_image():
    """Row a13 against tests/golden/peaks.npz (scikit-image 0.18.3's `peak_local_max` + the reference's compute_mpe /
    compute_margin): `vatl_peaks5` positions, order and counts exact on every plane without exactly equal candidate
    maxima (their order is scikit-image-version dependent, oracle/scorers.py), MPE / Margin within 1e-5."""
    import os
    import vatl_hip as vh
    from active_learning.scoring import multi_peak_scores
    from oracle import synth
    from tests.gpu_util import record, rel_err, to_dev
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "peaks.npz"))
    hm = synth.peak_items(24, seed=int(g["items_seed"]))
    d = to_dev(hm)
    W = hm.shape[3]
    val, idx, cnt, _, _ = vh.peaks5(d, 5)
    val, idx, cnt = val.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    free = ~g["tied"]
    assert free.sum() >= 390
    assert np.array_equal(cnt[free], g["cnt"][free])
    want_idx = np.where(g["loc"][..., 0] >= 0, g["loc"][..., 0] * W + g["loc"][..., 1], -1)
    assert np.array_equal(idx[free], want_idx[free])
    n_, j_, k_ = np.nonzero((want_idx >= 0) & free[..., None])
    assert np.array_equal(val[n_, j_, k_], hm[n_, j_, g["loc"][n_, j_, k_, 0], g["loc"][n_, j_, k_, 1]])
    ok = ~g["item_tied"]
    got_mpe, got_mar = multi_peak_scores(d, "MPE").cpu().numpy(), multi_peak_scores(d, "Margin").cpu().numpy()
    record("mpe_margin_skimage", mpe_rel=rel_err(got_mpe[ok], g["mpe"][ok]), margin_rel=rel_err(got_mar[ok], g["margin"][ok]),
           planes_exact=int(free.sum()), planes_tied=int((~free).sum()))
    np.testing.assert_allclose(got_mpe[ok], g["mpe"][ok], rtol=1e-5)
    np.testing.assert_allclose(got_mar[ok], g["margin"][ok], rtol=1e-5, atol=1e-6)
    for name in g["case_names"]:                                # border / spacing / truncation / degenerate planes, other sizes
        name = str(name)
        if bool(g["tied_" + name]):
            continue
        plane = g["case_" + name]
        want = g["loc_" + name]
        if min(plane.shape) <= 10:                              # no pixel outside the excluded border: scikit-image returns nothing,
            assert len(want) == 0                               # the C ABI refuses the plane loudly (VATL_EINVAL "plane too small")
            with pytest.raises(vh.VatlError):
                vh.peaks5(to_dev(plane[None, None]), 5)
            continue
        _, ci, cc, _, _ = vh.peaks5(to_dev(plane[None, None]), 5)
        assert int(cc[0, 0]) == len(want), name
        assert ci[0, 0, :len(want)].cpu().tolist() == (want[:, 0] * plane.shape[1] + want[:, 1]).tolist(), name
        assert (ci[0, 0, len(want):] == -1).all(), name


def test_oks_kernel_matches_al_metric():
    """§8f rank 1: compute_OKS on the device vs the numpy restatement (visible / partly visible / nothing visible)."""
    import vatl_hip as vh
    from oracle import scorers
    from tests.gpu_util import to_dev
    r = np.random.RandomState(3)
    n = 9
    pred = (r.uniform(0, 300, (n, 17, 3))).astype(np.float32)
    gt = pred.reshape(n, 51).astype(np.float64) + r.standard_normal((n, 51)) * 4
    gt[:, 2::3] = (r.random_sample((n, 17)) > 0.3).astype(np.float64)
    gt[0, 2::3] = 0.0                                        # nothing visible: distance from the doubled box
    gt[1, 2::3] = 1.0
    box = np.stack([r.uniform(0, 100, n), r.uniform(0, 100, n), r.uniform(40, 200, n), r.uniform(60, 260, n)], 1)
    got = vh.oks(to_dev(pred), torch.from_numpy(gt).cuda(), torch.from_numpy(box).cuda()).cpu().numpy()
    want = np.array([scorers.oks(box[i], pred[i].reshape(-1), gt[i]) for i in range(n)])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_config1_validate_harness_batch4():
    """BASELINE.json configs[0] (scripts/poseestimator_eval.py:47-79): SimpleBaseline-R50, batch of 4 crops, `m.eval()`
    forward, per-crop `heatmap_to_coord` with the crop box, record score = mean + 1.25 max of the joint scores — the
    reference's own per-item calls on our modules against the CPU restatement (pinned to the reference's golden outputs)."""
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from alphapose.utils.transforms import get_func_heatmap_to_coord
    from oracle import nets, scorers, synth
    from tests.gpu_util import dev, record, rel_err, to_dev
    cfg = edict({"MODEL": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
                 "DATA_PRESET": {"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]},
                 "LOSS": {"TYPE": "MSELoss"}})
    m = builder.build_sppe(cfg.MODEL, preset_cfg=cfg.DATA_PRESET)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).eval()
    x = synth.crops(4, seed=23)
    bb = synth.bboxes(4, seed=23)
    heatmap_to_coord = get_func_heatmap_to_coord(cfg)
    with torch.no_grad():
        output = m(to_dev(x))
    assert output.dim() == 4 and output.shape == (4, 17, 64, 48)
    ref = nets.SimplePoseRef(50)
    ref.load_state_dict(synth.state_dict_for(ref), strict=True)
    ref.eval()
    with torch.no_grad():
        want_hm = ref(torch.from_numpy(x)).numpy()
    record("config1_heatmaps", rel=rel_err(output.cpu().numpy(), want_hm))
    assert rel_err(output.cpu().numpy(), want_hm) < 1e-4
    eval_joints = list(range(17))
    for j in range(4):
        pose_coords, pose_scores = heatmap_to_coord(output[j][eval_joints], bb[j].tolist(), hm_shape=cfg.DATA_PRESET.HEATMAP_SIZE, norm_type=None)
        d = scorers.decode_heatmaps(want_hm[j], bb[j])
        assert pose_coords.shape == (17, 2) and pose_scores.shape == (17, 1) and pose_coords.dtype == np.float32
        assert np.array_equal(np.argmax(output[j].cpu().numpy().reshape(17, -1), 1), d["idx"])
        np.testing.assert_allclose(pose_coords, d["coords"], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(pose_scores, d["maxvals"], rtol=1e-4)
        score = float(np.mean(pose_scores) + 1.25 * np.max(pose_scores))
        np.testing.assert_allclose(score, scorers.pose_score(d["maxvals"]), rtol=1e-4)
