"""HIP scorers (through the C ABI) vs the golden vectors and the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import scorers, synth
from tests.gpu_util import dev, record, rel_err, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vh():
    import vatl_hip
    vatl_hip.lib()
    return vatl_hip


def test_library_loaded(vh):
    assert vh.lib().vatl_version() == 100


def test_decode_golden(vh, golden_scorers):
    g = golden_scorers
    coords, maxv, idx = vh.decode(to_dev(g["hm"]), to_dev(g["bbox"]))
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), g["idx"].astype(np.int32))                   # bit-exact integers
    assert np.array_equal(maxv.cpu().numpy(), g["maxvals"][..., 0])                      # copied values: bit-exact
    c = coords.cpu().numpy()
    exact = float((c == g["coords"]).mean())
    record("decode_golden", coords_bit_exact_fraction=exact, max_abs=float(np.abs(c - g["coords"]).max()))
    np.testing.assert_allclose(c, g["coords"], rtol=1e-4, atol=1e-4)                      # north_star: key-points 1e-4 rel
    assert exact == 1.0                                                                   # and in fact bit-identical here


def test_decode_pose_rows_and_item_scores(vh, golden_scorers):
    """vatl_decode_pose: the interleaved (x, y, score) rows equal the separate decode bit for bit; HP = -np.sum(scores) is BIT-identical to
    NumPy's float32 pairwise sum (17 joints, and 5 / 8 / 136 / 300 joints for the other branches of the summation order); the json score
    = mean + 1.25 max BIT-equal to what the reference's pinned numpy 1.23.5 computes (float32 mean, float64 product and sum; ActiveLearning.py:304-314, 329-330)."""
    g = golden_scorers
    hm, bb = to_dev(g["hm"]), to_dev(g["bbox"])
    coords, maxv, idx = vh.decode(hm, bb)
    kpts, idx2, hp, ps = vh.decode_pose(hm, bb)
    assert torch.equal(kpts[..., :2], coords) and torch.equal(kpts[..., 2], maxv) and torch.equal(idx2, idx)
    assert np.array_equal(kpts.cpu().numpy().reshape(len(bb), -1), g["kp"].astype(np.float32))
    mv = g["maxvals"][..., 0].astype(np.float32)
    want_hp = np.array([-np.sum(mv[i]) for i in range(mv.shape[0])], np.float32)
    assert np.array_equal(hp.cpu().numpy(), want_hp)
    np123 = lambda v: float(np.float64(np.mean(v.astype(np.float32))) + 1.25 * np.float64(np.max(v.astype(np.float32))))   # numpy 1.23.5: float32 mean, float64 product + sum
    got_ps = ps.cpu().numpy()
    assert got_ps.dtype == np.float64
    for i in range(mv.shape[0]):
        assert got_ps[i] == np123(mv[i]) == scorers.pose_score(g["maxvals"][i])                      # bit-equal doubles
    r = np.random.RandomState(9)
    for j in (5, 8, 136, 300):
        h = r.standard_normal((3, j, 8, 12)).astype(np.float32) * 3
        k, _, hp, ps = vh.decode_pose(to_dev(h), to_dev(synth.bboxes(3, seed=2)))
        sc = h.reshape(3, j, -1).max(2)
        assert np.array_equal(k[..., 2].cpu().numpy(), sc)
        assert np.array_equal(hp.cpu().numpy(), np.array([-np.sum(sc[i]) for i in range(3)], np.float32)), j
        assert np.array_equal(ps.cpu().numpy(), np.array([np123(sc[i]) for i in range(3)], np.float64)), j


def test_decode_random_vs_oracle(vh):
    r = np.random.RandomState(1)
    hm = r.standard_normal((32, 17, 64, 48)).astype(np.float32)
    hm[3] = np.round(hm[3] * 2) / 2                                                       # many exact ties
    hm[4] = -np.abs(hm[4])                                                                # all non-positive
    bb = synth.bboxes(32, seed=4)
    coords, maxv, idx = vh.decode(to_dev(hm), to_dev(bb))
    for i in range(32):
        d = scorers.decode_heatmaps(hm[i], bb[i])
        assert np.array_equal(idx[i].cpu().numpy(), d["idx"]), i
        assert np.array_equal(maxv[i].cpu().numpy(), d["maxvals"][:, 0])
        np.testing.assert_allclose(coords[i].cpu().numpy(), d["coords"], rtol=1e-4, atol=1e-4)


def test_decode_other_heatmap_sizes(vh):
    r = np.random.RandomState(2)
    for (h, w) in ((96, 72), (5, 7), (1, 1)):
        hm = r.standard_normal((3, 17, h, w)).astype(np.float32)
        bb = synth.bboxes(3, seed=h)
        coords, maxv, idx = vh.decode(to_dev(hm), to_dev(bb))
        for i in range(3):
            d = scorers.decode_heatmaps(hm[i], bb[i])
            assert np.array_equal(idx[i].cpu().numpy(), d["idx"])
            np.testing.assert_allclose(coords[i].cpu().numpy(), d["coords"], rtol=1e-4, atol=1e-4)
    c, m, i = vh.decode(torch.empty((0, 17, 64, 48), device=dev()), torch.empty((0, 4), device=dev()))   # empty batch
    assert c.shape == (0, 17, 2)


def test_thc_golden_and_stream_rule(vh, golden_scorers):
    g = golden_scorers
    hm = to_dev(g["hm"])
    for norm, key in (("L1", "thc_l1"), ("L2", "thc_l2")):
        got = vh.thc_pairs(hm[:-1], hm[1:], norm).cpu().numpy()
        record("thc_" + norm, rel=rel_err(got, g[key]))
        np.testing.assert_allclose(got, g[key], rtol=1e-5)
    n = hm.shape[0]
    is_prev = np.array([0, 1, 1, 0, 1, 0, 0, 1], np.uint8)
    is_next = np.array([1, 1, 0, 1, 0, 0, 1, 0], np.uint8)
    got = vh.thc_stream(hm, to_dev(is_prev, torch.uint8), to_dev(is_next, torch.uint8)).cpu().numpy()
    for i in range(n):
        want = scorers.thc_item(g["hm"][i], g["hm"][i - 1] if i else None, g["hm"][i + 1] if i + 1 < n else None,
                                is_prev[i], is_next[i])
        np.testing.assert_allclose(got[i], want, rtol=1e-5)
    one = vh.thc_stream(hm[:1], to_dev([0], torch.uint8), to_dev([0], torch.uint8))       # single item: no pairs
    assert float(one[0]) == 0.0


def test_localpeak_golden(vh, golden_scorers):
    g = golden_scorers
    mean, cnt = vh.localpeak_mean(to_dev(g["hm"]))
    assert np.array_equal(cnt.cpu().numpy(), g["lp_cnt"].astype(np.int32))               # integer peak counts: bit-exact
    m = mean.cpu().numpy()
    assert np.isnan(m[6]) and np.isnan(g["lp_mean"][6])
    ok = ~np.isnan(g["lp_mean"])
    record("localpeak", rel=rel_err(m[ok], g["lp_mean"][ok]))
    np.testing.assert_allclose(m[ok], g["lp_mean"][ok], rtol=1e-5)
    toy = np.zeros((1, 3, 4, 10), np.float32) + g["toy"].astype(np.float32)
    mean, cnt = vh.localpeak_mean(to_dev(toy))
    assert float(mean[0]) == 3.5 and cnt.cpu().tolist() == [[2, 2, 2]]


def test_localpeak_random_vs_oracle(vh):
    hm = synth.blob_heatmaps(24, seed=21)
    hm[5] = np.round(hm[5] * 8) / 8                                                       # plateaus
    mean, cnt = vh.localpeak_mean(to_dev(hm))
    for i in range(24):
        c, _, m = scorers.localpeak_stats(hm[i])
        assert np.array_equal(cnt[i].cpu().numpy(), c), i
        np.testing.assert_allclose(float(mean[i]), m, rtol=1e-5)


@pytest.mark.parametrize("hw", [(96, 72), (64, 48), (16, 8), (5, 4), (130, 96), (33, 256), (24, 18), (7, 5)])
def test_localpeak_plane_sizes(vh, hw):
    """Every kernel variant behind vatl_localpeak_mean (register tiles: 1, 2 and many waves per plane; LDS tile when a row
    is wider than a wave or the plane is taller than 16 waves cover; scalar path for widths that are not a multiple of 4):
    integer peak counts exact, means to fp32 rounding, borders / plateaus / all-negative planes included."""
    H, W = hw
    r = np.random.RandomState(H * 1000 + W)
    hm = r.random_sample((3, 5, H, W)).astype(np.float32)
    hm[0, 0] = np.round(hm[0, 0] * 4) / 4                       # plateaus (ties)
    hm[0, 1] = -hm[0, 1] - 0.1                                   # all negative: the zero border beats every pixel
    hm[0, 2, 0, :] = 2.0; hm[0, 2, :, -1] = 3.0                  # maxima on the borders
    hm[1, 3] = 0.0
    mean, cnt = vh.localpeak_mean(to_dev(hm))
    for i in range(3):
        c, _, m = scorers.localpeak_stats(hm[i])
        assert np.array_equal(cnt[i].cpu().numpy(), c), (hw, i)
        if np.isnan(m):
            assert np.isnan(float(mean[i]))
        else:
            np.testing.assert_allclose(float(mean[i]), m, rtol=1e-5)


def test_wpu_golden(vh, golden_scorers):
    g = golden_scorers
    kp = to_dev(g["kp"].reshape(-1, 17, 3))
    bb = to_dev(g["bbox"])
    sd42 = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ae42.")}
    sd38 = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ae38.")}
    ok = g["kp_ok"]
    for flat, d, only38, key in ((vh.pack_ae(sd42, dev()), 42, False, "wpu42"), (vh.pack_ae(sd42, dev()), 42, True, "wpu38"),
                                 (vh.pack_ae(sd38, dev()), 38, False, "wpu38cls")):
        wpu, status = vh.hybrid_ae_wpu(kp, bb, flat, d, 4, only38)
        w, s = wpu.cpu().numpy(), status.cpu().numpy()
        assert np.array_equal(s == 0, ok)
        assert (s[~ok] == 2).all() and np.isnan(w[~ok]).all()                              # "at least one visible keypoint"
        record(key, rel=rel_err(w[ok], g[key][ok]))
        np.testing.assert_allclose(w[ok], g[key][ok], rtol=1e-4)
    bad = bb.clone(); bad[:, 3] = bad[:, 1] - 5                                           # height <= 0 -> status 1
    _, status = vh.hybrid_ae_wpu(kp, bad, vh.pack_ae(sd42, dev()), 42, 4)
    assert (status.cpu().numpy()[ok] == 1).all()


def test_masked_mse_and_adamw(vh):
    r = np.random.RandomState(0)
    out = r.standard_normal((5, 17, 64, 48)).astype(np.float32)
    tgt, mask = synth.gaussian_targets(5, seed=2)
    loss, grad = vh.masked_mse_fwd_bwd(to_dev(out), to_dev(tgt), to_dev(mask))
    want_l, want_g = scorers.masked_mse(out, tgt, mask)
    np.testing.assert_allclose(float(loss), want_l, rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), want_g, rtol=1e-5, atol=1e-12)

    n = 100003                                                                             # odd tail
    p = r.standard_normal(n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    tp = torch.from_numpy(p.copy()).requires_grad_()
    opt = torch.optim.AdamW([tp], lr=2.5e-3, weight_decay=0.7)
    dp, dm, dv = to_dev(p), to_dev(m), to_dev(v)
    for step in range(1, 4):
        g_ = r.standard_normal(n).astype(np.float32)
        tp.grad = torch.from_numpy(g_.copy()); opt.step()
        vh.adamw_step(dp, to_dev(g_), dm, dv, step, 2.5e-3, 0.7)
        p, m, v = scorers.adamw_step(p, g_, m, v, step, 2.5e-3, 0.7)
    np.testing.assert_allclose(dp.cpu().numpy(), tp.detach().numpy(), rtol=2e-5, atol=1e-6)   # vs torch.optim.AdamW
    np.testing.assert_allclose(dp.cpu().numpy(), p, rtol=2e-5, atol=1e-6)                      # vs the numpy oracle


def test_adam_and_sgd_steps(vh):
    """ActiveLearning.py:220-223: the two other optimisers the reference can be configured with."""
    from active_learning.optim import SGD, Adam
    r = np.random.RandomState(5)
    n = 70001
    p0 = r.standard_normal(n).astype(np.float32)
    a, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    b, buf = p0.copy(), np.zeros(n, np.float32)
    da = torch.nn.Parameter(to_dev(p0)); ds = torch.nn.Parameter(to_dev(p0))
    oa, os_ = Adam([da], lr=2.5e-4), SGD([ds], lr=2.5e-4, momentum=0.9, weight_decay=0.0005)
    for step in range(1, 5):
        g = r.standard_normal(n).astype(np.float32)
        da.grad = to_dev(g); ds.grad = to_dev(g)
        oa.step(); os_.step()
        a, m, v = scorers.adam_step(a, g, m, v, step, 2.5e-4)
        b, buf = scorers.sgd_step(b, g, buf, step, 2.5e-4, 0.9, 0.0005)
    record("adam_step", rel=rel_err(da.detach().cpu().numpy(), a)); record("sgd_step", rel=rel_err(ds.detach().cpu().numpy(), b))
    np.testing.assert_allclose(da.detach().cpu().numpy(), a, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ds.detach().cpu().numpy(), b, rtol=2e-6, atol=1e-7)
    assert da._version > 0 and ds._version > 0           # plan caches key on the version counter


def test_empty_and_ragged_batches(vh):
    """Edge cases: empty streams return empty results, a 1-item stream has no neighbours, odd sizes work."""
    d = dev()
    e = torch.empty((0, 17, 64, 48), device=d)
    assert vh.thc_stream(e, torch.empty(0, dtype=torch.uint8, device=d), torch.empty(0, dtype=torch.uint8, device=d)).shape == (0,)
    m, c = vh.localpeak_mean(e)
    assert m.shape == (0,) and c.shape == (0, 17)
    w, s = vh.hybrid_ae_wpu(torch.empty((0, 17, 3), device=d), torch.empty((0, 4), device=d), torch.zeros(4000, device=d), 42, 4)
    assert w.shape == (0,)
    hm = synth.blob_heatmaps(3, seed=8)
    for n in (1, 3):                                  # every prefix length agrees with the oracle item by item
        flags = np.ones(n, np.uint8)
        got = vh.thc_stream(to_dev(hm[:n]), to_dev(flags, torch.uint8), to_dev(flags, torch.uint8)).cpu().numpy()
        for i in range(n):
            want = scorers.thc_item(hm[i], hm[i - 1] if i else None, hm[i + 1] if i + 1 < n else None, i > 0, i + 1 < n)
            np.testing.assert_allclose(got[i], want, rtol=1e-5)


@pytest.mark.parametrize("norm", ["softmax", "sigmoid", "divide_sum"])
def test_l1_joint_regression_loss(vh, norm):
    """alphapose.models.criterion.L1JointRegression on MI355X (fused forward + backward) against the reference's
    golden loss / joints / gradients and, densely, against the torch restatement."""
    import os
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from oracle import nets
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "l1_joint_regression.npz"))
    hm, gt, vis = synth.l1_inputs(norm)
    crit = builder.build_loss(edict({"TYPE": "L1JointRegression", "NORM_TYPE": norm}))
    h = to_dev(hm).requires_grad_()
    loss = crit(h, to_dev(gt), to_dev(vis))
    (2.0 * loss).backward()                                               # the upstream gradient is honoured
    np.testing.assert_allclose(float(loss.detach()), float(g[f"{norm}_loss"]), rtol=1e-5)
    _, _, jts = vh.l1_joint_regression_fwd_bwd(to_dev(hm), to_dev(gt), to_dev(vis), norm)
    np.testing.assert_allclose(jts.cpu().numpy(), g[f"{norm}_jts"], rtol=1e-4, atol=1e-5)
    grad = h.grad.cpu().numpy() / 2.0
    amax = float(g[f"{norm}_grad_absmax"])
    np.testing.assert_allclose(grad.reshape(-1)[g[f"{norm}_grad_idx"]], g[f"{norm}_grad_val"], rtol=1e-3, atol=2e-5 * amax)
    ht = torch.from_numpy(hm).requires_grad_()
    lt, _ = nets.l1_joint_regression(ht, torch.from_numpy(gt), torch.from_numpy(vis), norm)
    lt.backward()
    e = rel_err(grad, ht.grad.numpy())
    record("l1_joint_regression_" + norm, grad_rel=e)
    assert e < 1e-4
    with pytest.raises(vh.VatlError):
        crit(torch.from_numpy(hm), torch.from_numpy(gt), torch.from_numpy(vis))


def test_autoencoder_finetune_steps_match_torch_adam(vh):
    """retrain_AE (ActiveLearning.py:905-925) on the device: three Adam steps on mini-batches of 10 against the torch
    restatement (oracle.nets.WholeBodyAERef + torch.optim.Adam) — loss per step and every parameter."""
    from active_learning.Whole_body_AE.AutoEncoder import WholeBodyAE, fit_autoencoder
    from oracle import nets
    r = np.random.RandomState(8)
    for d, z in ((42, 4), (38, 2)):
        ref = nets.WholeBodyAERef(z_dim=z, input_dim=d)
        sd = {k: torch.from_numpy(synth.tensor_for("ae." + k, v.shape)) for k, v in ref.state_dict().items()}
        ref.load_state_dict(sd)
        ae = WholeBodyAE(z_dim=z, input_dim=d)
        ae.load_state_dict(sd)
        ae = ae.to(dev())
        flat = vh.pack_ae(ae.state_dict(), dev()).clone()
        m, v = torch.zeros_like(flat), torch.zeros_like(flat)
        opt = torch.optim.Adam(ref.parameters(), lr=1e-2)
        for step in range(1, 4):
            x = r.uniform(0, 1, (10 if step < 3 else 7, d)).astype(np.float32)
            xt = torch.from_numpy(x)
            loss_t = torch.nn.MSELoss()(ref(xt), xt)
            opt.zero_grad(); loss_t.backward(); opt.step()
            loss = vh.ae_train_step(flat, m, v, to_dev(x), d, z, step, 1e-2)
            np.testing.assert_allclose(float(loss), loss_t.item(), rtol=1e-5)
        vh.unpack_ae(flat, ae)
        worst = 0.0
        for (k, p), (_, q) in zip(ae.state_dict().items(), ref.state_dict().items()):
            worst = max(worst, float((p.cpu() - q).abs().max() / (q.abs().max() + 1e-12)))
            # Adam moves a parameter by ~lr * g/|g|: where the gradient is ~0 its rounding noise decides the direction,
            # so single entries may differ by a fraction of lr = 1e-2 (observed 1.3e-5); everything else agrees to 1e-6
            np.testing.assert_allclose(p.cpu().numpy(), q.numpy(), rtol=2e-4, atol=1e-4)
            assert np.median(np.abs(p.cpu().numpy() - q.numpy())) < 1e-6
        record(f"ae_train_d{d}", worst_param_rel=worst)
    # the epoch driver: loss goes down on a fixed feature set and the module's parameters are the trained ones
    feats = to_dev(r.uniform(0.2, 0.8, (33, 42)).astype(np.float32))
    ae = WholeBodyAE(z_dim=4, input_dim=42).to(dev())
    with torch.no_grad():
        before = float(((ae(feats) - feats) ** 2).mean())
    g = torch.Generator(); g.manual_seed(0)
    mean_loss = fit_autoencoder(ae, feats, epochs=30, lr=1e-2, generator=g)
    with torch.no_grad():
        after = float(((ae(feats) - feats) ** 2).mean())
    assert after < before and np.isfinite(mean_loss)
    with pytest.raises(vh.VatlError):
        vh.ae_train_step(flat, m, v, to_dev(np.zeros((13, 38), np.float32)), 38, 2, 1, 1e-2)        # batch above the kernel's 12


def test_decode_with_nan_and_inf_entries(vh):
    """Null handling like numpy's: a NaN is the arg-max (the first one), its score is NaN and `maxval > 0` is False, so
    the joint decodes to the (0, 0) heat-map corner; +inf is an ordinary maximum."""
    hm = synth.blob_heatmaps(2, seed=13)
    hm[0, 2, 10, 7] = np.nan; hm[0, 2, 40, 30] = np.nan          # two NaNs: the first (row-major) wins
    hm[0, 5, 33, 21] = np.inf
    hm[1, 0, :, :] = np.nan                                      # whole plane NaN -> index 0
    hm[1, 4, 63, 47] = np.nan                                    # NaN in the last element, every thread has seen real values
    bb = synth.bboxes(2)
    c, m, i = vh.decode(to_dev(hm), to_dev(bb))
    c, m, i = c.cpu().numpy(), m.cpu().numpy(), i.cpu().numpy()
    for n in range(2):
        with np.errstate(all="ignore"):
            d = scorers.decode_heatmaps(hm[n], bb[n])
        assert np.array_equal(i[n], d["idx"].astype(np.int32))
        assert np.array_equal(np.isnan(m[n]), np.isnan(d["maxvals"][:, 0]))
        ok = ~np.isnan(d["maxvals"][:, 0])
        assert np.array_equal(m[n][ok], d["maxvals"][ok, 0])
        np.testing.assert_allclose(c[n], d["coords"], rtol=1e-6, atol=1e-4)
    assert i[0, 2] == 10 * 48 + 7 and i[0, 5] == 33 * 48 + 21 and i[1, 0] == 0 and i[1, 4] == 63 * 48 + 47


def test_multi_tensor_adamw_equals_per_tensor(vh):
    """One launch per parameter group (`vatl_adamw_step_multi`) against the per-tensor kernel: same bits, odd sizes and
    unaligned tails included; the optimizer class uses it and bumps every parameter's version counter."""
    from active_learning.optim import AdamW
    r = np.random.RandomState(17)
    sizes = [(64,), (17,), (256, 64, 3, 3), (1001,), (5, 7), (2048, 512, 1, 1)]
    ps = [torch.nn.Parameter(to_dev(r.standard_normal(s).astype(np.float32))) for s in sizes]
    qs = [p.detach().clone() for p in ps]
    ms, vs = [torch.zeros_like(q) for q in qs], [torch.zeros_like(q) for q in qs]
    opt = AdamW([{"params": ps[:2], "lr": 2.5e-3}, {"params": ps[2:], "lr": 2.5e-4}], weight_decay=0.7)
    for step in range(1, 4):
        gs = [to_dev(r.standard_normal(s).astype(np.float32)) for s in sizes]
        for p, g in zip(ps, gs):
            p.grad = g
        opt.step()
        for k, (q, g, m, v) in enumerate(zip(qs, gs, ms, vs)):
            vh.adamw_step(q, g, m, v, step, 2.5e-3 if k < 2 else 2.5e-4, 0.7)
    for p, q in zip(ps, qs):
        assert torch.equal(p.detach(), q) and p._version > 0


@pytest.mark.parametrize("hw", [(64, 48), (96, 72), (32, 24), (10, 7)])
@pytest.mark.parametrize("nt", ["softmax", "sigmoid", "divide_sum"])
def test_softargmax_plane_sizes(vh, hw, nt):
    """Both shipped heat-map sizes (wave-per-plane register kernels) and other sizes (LDS-tile kernel) against the oracle."""
    H, W = hw
    r = np.random.RandomState(H + W)
    hm = (r.random_sample((5, 17, H, W)).astype(np.float32) * 6 - 1)
    if nt == "divide_sum":
        hm = np.abs(hm) + 1e-3
    bb = synth.bboxes(5)
    c, s = vh.decode_softargmax(to_dev(hm), to_dev(bb), nt)
    c, s = c.cpu().numpy(), s.cpu().numpy()
    for i in range(5):
        d = scorers.softargmax_decode(hm[i], bb[i], nt)
        np.testing.assert_allclose(c[i], d["coords"], rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(s[i], d["maxvals"][:, 0], rtol=1e-5)


@pytest.mark.parametrize("hw", [(64, 48), (96, 72), (32, 24), (10, 7)])
def test_entropy_plane_sizes(vh, hw):
    H, W = hw
    r = np.random.RandomState(3 * H + W)
    hm = np.abs(r.standard_normal((4, 17, H, W))).astype(np.float32) + 1e-4
    hm[1, 2] = 0.0                                              # zero plane: nan (0 / 0)
    hm[2, 3, 0, 0] = -0.5                                       # a negative entry: -inf
    hm[3, 4, 1:, :] = 0.0                                       # zeros contribute 0
    got = vh.plane_entropy(to_dev(hm)).cpu().numpy()
    from scipy.stats import entropy
    with np.errstate(all="ignore"):
        want = np.array([[entropy(hm[i, j].flatten()) for j in range(17)] for i in range(4)])
    assert np.isnan(got[1, 2]) and np.isnan(want[1, 2])
    assert got[2, 3] == -np.inf and want[2, 3] == -np.inf
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("hw", [(64, 48), (96, 72), (32, 24), (7, 5)])
def test_decode_plane_sizes(vh, hw):
    """Arg-max decode on both shipped heat-map sizes (wave-per-plane register kernel) and others (block kernel): indices and
    maxima bit-exact incl. ties (first wins), NaN (first NaN wins), non-positive maxima (coords zeroed) and border peaks."""
    H, W = hw
    r = np.random.RandomState(7 * H + W)
    hm = r.standard_normal((4, 17, H, W)).astype(np.float32)
    hm[0, 0] = np.round(hm[0, 0])                               # many ties
    hm[0, 1] = -np.abs(hm[0, 1]) - 1                            # all negative
    hm[0, 2, H - 1, W - 1] = 50.0                               # border peak: no quarter-pixel shift
    hm[0, 3, H // 2, W // 2] = np.nan; hm[0, 3, H - 1, 0] = np.nan
    hm[1, 4] = 0.0
    hm[1, 5, 0, 0] = 9.0; hm[1, 5, H - 1, W - 1] = 9.0         # equal maxima far apart: the first wins
    bb = synth.bboxes(4)
    coords, maxv, idx = vh.decode(to_dev(hm), to_dev(bb))
    coords, maxv, idx = coords.cpu().numpy(), maxv.cpu().numpy(), idx.cpu().numpy()
    for i in range(4):
        with np.errstate(invalid="ignore"):
            d = scorers.decode_heatmaps(hm[i], bb[i])
        assert np.array_equal(idx[i], d["idx"]), (hw, i)
        assert np.array_equal(maxv[i], d["maxvals"][:, 0], equal_nan=True)
        ok = ~np.isnan(d["maxvals"][:, 0])
        assert np.array_equal(coords[i][ok], d["coords"][ok])
