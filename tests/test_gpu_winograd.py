"""Winograd F(2x2, 3x3) route of the 3x3 / stride-1 layers (csrc/conv_winograd.hip) against a float64 convolution, against the
implicit GEMM, and for the properties the evaluation path relies on (bits independent of the batch position)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import dev, record, rel_err, to_dev

pytestmark = pytest.mark.gpu

TOL = 2e-5      # max|err| / max|ref| per layer: the bar of the implicit GEMM (tests/test_gpu_conv.py)


@pytest.fixture(scope="module")
def vh():
    import vatl_hip
    vatl_hip.lib()
    return vatl_hip


def _nhwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


CASES = [
    # name, N, H, W, Cin, Cout, relu, residual, fold (scale / bias)
    ("r50_l1", 2, 64, 48, 64, 64, True, False, True),
    ("r50_l4_tiny", 3, 8, 6, 512, 512, True, False, True),
    ("hr_b32_res", 2, 64, 48, 32, 32, True, True, True),
    ("hr_b128_res", 1, 16, 12, 128, 128, True, True, True),
    ("odd_hw", 3, 13, 11, 32, 40, True, True, True),           # odd H and W, Cout not a multiple of the 64-channel tile
    ("odd_w_r152_l4", 2, 12, 9, 64, 96, False, False, True),
    ("one_pixel_rows", 2, 1, 7, 16, 8, False, False, False),      # a single tile row, Cin = 16 (one stage), no epilogue
    ("h2_w2", 5, 2, 2, 48, 36, True, False, True),               # one tile per image: every column is a border
    ("cout4", 1, 6, 10, 16, 4, False, True, False),
    ("tail_tiles", 7, 10, 6, 32, 64, True, False, True),          # 7 * 15 = 105 tiles: the second block is partly empty
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_winograd_matches_float64_and_the_implicit_gemm(vh, case):
    name, n, h, w, cin, cout, relu, use_res, fold = case
    r = np.random.RandomState(zlib.crc32(name.encode()) % 2 ** 31)
    x = r.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (r.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, 1, 1)
    scale = bias = None
    if fold:
        gamma, beta = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32) * 0.1
        mean, var = r.standard_normal(cout).astype(np.float32) * 0.1, r.uniform(0.5, 1.5, cout).astype(np.float32)
        ref = F.batch_norm(ref, torch.from_numpy(mean).double(), torch.from_numpy(var).double(), torch.from_numpy(gamma).double(),
                           torch.from_numpy(beta).double(), False, 0.0, 1e-5)
        scale, bias = vh.bn_fold(to_dev(gamma), to_dev(beta), to_dev(mean), to_dev(var), 1e-5)
    res = None
    if use_res:
        res = r.standard_normal(tuple(ref.shape)).astype(np.float32)
        ref = ref + torch.from_numpy(res).double()
    if relu:
        ref = ref.relu()
    ref = ref.numpy()
    xd = to_dev(_nhwc(x))
    resd = to_dev(_nhwc(res)) if use_res else None
    u = vh.pack_winograd_weight(to_dev(wt))
    y = vh.conv3x3_winograd_fwd(xd, u, scale, bias, cout, relu, residual=resd)
    got = np.transpose(y.cpu().numpy(), (0, 3, 1, 2))
    e = rel_err(got, ref)
    record("winograd_" + name, rel=e)
    assert e < TOL, (name, e)
    if cin % 32 == 0:                                           # the implicit GEMM needs whole 32-channel k-tiles
        yd = vh.conv2d_fwd(xd, vh.pack_conv_weight(to_dev(wt)), scale, bias, cout, 3, 3, 1, 1, relu, residual=resd)
        assert rel_err(y.cpu().numpy(), yd.cpu().numpy()) < 2 * TOL


def test_winograd_output_buffer_and_rejections(vh):
    x = torch.randn((2, 8, 6, 32), device=dev())
    wt = torch.randn((16, 32, 3, 3), device=dev()) * 0.1
    u = vh.pack_winograd_weight(wt)
    out = torch.full((2, 8, 6, 16), 7.0, device=dev())
    y = vh.conv3x3_winograd_fwd(x, u, None, None, 16, False, out=out)
    assert y.data_ptr() == out.data_ptr() and not (out == 7.0).any()
    with pytest.raises(vh.VatlError):
        vh.pack_winograd_weight(torch.randn((16, 24, 3, 3), device=dev()))          # Cin % 16
    with pytest.raises(vh.VatlError):
        vh.pack_winograd_weight(torch.randn((16, 32, 1, 1), device=dev()))
    with pytest.raises(vh.VatlError):
        vh.conv3x3_winograd_fwd(x, u, None, None, 18, False)                         # Cout % 4


def test_winograd_bits_do_not_depend_on_the_batch_position(vh):
    """A crop's output is the same bit pattern alone, in the middle of a large batch, and at a different offset inside a
    64-tile block (the THC de-duplication compares heat-maps of the same crop computed in different batches)."""
    g = torch.Generator(device="cpu").manual_seed(3)
    for h, w, cin, cout in ((16, 12, 64, 64), (13, 9, 32, 32), (8, 6, 128, 256)):
        x = torch.randn((37, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(dev())
        sc, bi = (torch.rand(cout, generator=g) + 0.5).to(dev()), torch.randn(cout, generator=g).to(dev())
        u = vh.pack_winograd_weight(wt)
        full = vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, True)
        for i in (0, 5, 36):
            alone = vh.conv3x3_winograd_fwd(x[i:i + 1].contiguous(), u, sc, bi, cout, True)
            assert torch.equal(alone[0], full[i]), (h, w, i)
        shifted = vh.conv3x3_winograd_fwd(x[3:].contiguous(), u, sc, bi, cout, True)
        assert torch.equal(shifted, full[3:])
        again = vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, True)
        assert torch.equal(again, full)


def test_winograd_statistics_epilogue_matches_the_stored_tensor(vh):
    """Training forward: y = x * w and the (sum, sum of squares) row-block partials that BatchNorm's finalize reduces."""
    import ctypes as C
    g = torch.Generator(device="cpu").manual_seed(11)
    for n, h, w, cin, cout in ((5, 16, 12, 64, 64), (3, 13, 9, 32, 40), (2, 8, 6, 128, 256)):
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(dev())
        u = vh.pack_winograd_weight(wt)
        y, stats, blocks = vh.conv3x3_winograd_fwd_stats(x, u, cout)
        plain = vh.conv3x3_winograd_fwd(x, u, None, None, cout, False)
        assert torch.equal(y, plain)
        s = stats[:blocks * cout * 2].view(blocks, cout, 2).sum(0)
        y64 = y.double().reshape(-1, cout)
        assert torch.allclose(s[:, 0], y64.sum(0), rtol=1e-6, atol=1e-4)
        assert torch.allclose(s[:, 1], (y64 * y64).sum(0), rtol=1e-6, atol=1e-4)
        record(f"winograd_stats_{cin}_{cout}", blocks=int(blocks))


def test_winograd_data_gradient_filter(vh):
    """dX = conv(dY, rot180(w)^T) through the same kernel with the data-gradient packing, against autograd in float64."""
    g = torch.Generator(device="cpu").manual_seed(17)
    for n, h, w, cin, cout in ((2, 16, 12, 64, 128), (3, 13, 9, 32, 48), (1, 8, 6, 256, 64)):
        x = torch.randn((n, cin, h, w), generator=g, dtype=torch.float64, requires_grad=True)
        wt = torch.randn((cout, cin, 3, 3), generator=g, dtype=torch.float64) * 0.05
        dy = torch.randn((n, cout, h, w), generator=g, dtype=torch.float64)
        F.conv2d(x, wt, None, 1, 1).backward(dy)
        ud = vh.pack_winograd_weight(wt.float().to(dev()), data_gradient=True)
        dx = vh.conv3x3_winograd_fwd(dy.float().permute(0, 2, 3, 1).contiguous().to(dev()), ud, None, None, cin, False)
        e = rel_err(dx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.numpy())
        record(f"winograd_dgrad_{cin}_{cout}", rel=e)
        assert e < TOL, e


def test_inference_plans_route_3x3_layers_through_winograd(vh, monkeypatch):
    """SimplePose-R50: the 13 stride-1 3x3 layers of the trunk take the Winograd routes — F(4x4,3x3) for the 11 on grids of whole 4x4 tiles (stages 1 - 3),
    F(2x2,3x3) for the two at 8x6; with WINO_F4 off all 13 take F(2x2) and the heat-maps agree to fp32 rounding; both agree with the all-implicit-GEMM plan
    to the layer tolerance."""
    from alphapose.models import hip_engine
    from oracle import synth
    from tests.test_gpu_conv import _build_simplepose
    m = _build_simplepose()
    x = to_dev(synth.crops(3))
    calls, calls4 = [], []
    orig, orig4 = vh.conv3x3_winograd_fwd, vh.conv3x3_winograd_f4_fwd
    monkeypatch.setattr(vh, "conv3x3_winograd_fwd", lambda *a, **k: (calls.append(a[0].shape), orig(*a, **k))[1])
    monkeypatch.setattr(vh, "conv3x3_winograd_f4_fwd", lambda *a, **k: (calls4.append(a[0].shape), orig4(*a, **k))[1])
    out = torch.empty((3, 17, 64, 48), device=dev())
    with torch.no_grad():
        hip_engine.forward_into(m, x, out)
    assert (len(calls), len(calls4)) == (2, 11) and all(s[1] % 4 == 0 and s[2] % 4 == 0 for s in calls4) and all(tuple(s[1:3]) == (8, 6) for s in calls)
    monkeypatch.setattr(hip_engine, "WINO_F4", False)
    m.__dict__.pop("_vatl_plan", None)
    f2 = torch.empty_like(out)
    with torch.no_grad():
        hip_engine.forward_into(m, x, f2)
    assert (len(calls), len(calls4)) == (15, 11)
    e42 = rel_err(out.cpu().numpy(), f2.cpu().numpy())
    record("winograd_f4_vs_f2_simplepose_r50", rel=e42)
    assert e42 < 2e-5 and torch.equal(out.flatten(2).argmax(-1), f2.flatten(2).argmax(-1))
    del calls[:]
    del calls4[:]
    monkeypatch.setattr(hip_engine, "WINO_F4", True)
    m.__dict__.pop("_vatl_plan", None)
    with torch.no_grad():
        hip_engine.forward_into(m, x, out)
    monkeypatch.setattr(hip_engine, "WINOGRAD", False)
    m.__dict__.pop("_vatl_plan", None)
    direct = torch.empty_like(out)
    with torch.no_grad():
        hip_engine.forward_into(m, x, direct)
    m.__dict__.pop("_vatl_plan", None)
    assert (len(calls), len(calls4)) == (2, 11)            # the all-implicit-GEMM plan made no Winograd launch
    e = rel_err(out.cpu().numpy(), direct.cpu().numpy())
    record("winograd_vs_direct_simplepose_r50", rel=e)
    assert e < 1e-4


def test_pack_plan_refreshes_winograd_filters_with_the_same_bits(vh):
    """The one-launch re-pack of a fine-tune step (vatl_pack_weights_multi, kinds 3 / 4) writes what vatl_pack_winograd_weight writes."""
    g = torch.Generator(device="cpu").manual_seed(23)
    ws = [(torch.randn(s, generator=g) * 0.1).to(dev()) for s in ((64, 64, 3, 3), (32, 32, 3, 3), (48, 80, 3, 3), (128, 256, 3, 3))]
    wdc = [(torch.randn(s, generator=g) * 0.1).to(dev()) for s in ((64, 48, 4, 4), (32, 256, 4, 4))]
    want_dc = [vh.pack_winograd_deconv_weight(w) for w in wdc] + [vh.pack_winograd_deconv_dgrad_weight(w) for w in wdc]
    want = [(vh.pack_winograd_weight(w), vh.pack_winograd_weight(w, data_gradient=True)) for w in ws]
    plan = vh.PackPlan()
    prev = vh.set_pack_plan(plan)
    try:
        plan.begin()
        first = [(vh.pack_winograd_weight(w), vh.pack_winograd_weight(w, data_gradient=True)) for w in ws]
        other = vh.pack_conv_weight(ws[0])                  # a job of another kind in the same table
        first_dc = [vh.pack_winograd_deconv_weight(w) for w in wdc] + [vh.pack_winograd_deconv_dgrad_weight(w) for w in wdc]
        plan.seal()
        for t in first_dc:
            t.zero_()
        for a, b in first:
            a.zero_(); b.zero_()
        other_want = other.clone(); other.zero_()
        plan.begin()                                        # one launch refreshes every recorded destination
        again = [(vh.pack_winograd_weight(w), vh.pack_winograd_weight(w, data_gradient=True)) for w in ws]
    finally:
        vh.set_pack_plan(prev)
    for (a, b), (fa, fb), (wa, wb) in zip(again, first, want):
        assert a.data_ptr() == fa.data_ptr() and b.data_ptr() == fb.data_ptr()
        assert torch.equal(a, wa) and torch.equal(b, wb)
    assert torch.equal(other, other_want)
    for t, w in zip(first_dc, want_dc):
        assert torch.equal(t, w)


DECONV_CASES = [("2048_256", 2, 8, 6, 2048, 256), ("256_256", 3, 16, 12, 256, 256), ("odd", 2, 5, 3, 64, 48), ("256_256_big", 1, 32, 24, 256, 256),
                ("w_multiple_of_3", 5, 9, 6, 32, 36), ("one_row", 3, 1, 4, 16, 8), ("ragged", 7, 7, 10, 48, 100)]


@pytest.mark.parametrize("case", DECONV_CASES, ids=[c[0] for c in DECONV_CASES])
def test_deconv_winograd_matches_float64_and_the_implicit_gemm(vh, case):
    """ConvTranspose2d(4,2,1) + folded BN + ReLU as F(3x3,2x2) on the four phases (the shapes of tests/test_gpu_conv.py + border cases:
    widths that are multiples of the tile step, a single input row, tiles that straddle images)."""
    name, n, h, w, cin, cout = case
    r = np.random.RandomState(len(name) + cin)
    x = r.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (r.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    gamma, beta = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32) * 0.1
    mean, var = r.standard_normal(cout).astype(np.float32) * 0.1, r.uniform(0.5, 1.5, cout).astype(np.float32)
    ref = F.conv_transpose2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, 2, 1)
    ref = F.batch_norm(ref, torch.from_numpy(mean).double(), torch.from_numpy(var).double(), torch.from_numpy(gamma).double(),
                       torch.from_numpy(beta).double(), False, 0.0, 1e-5).relu().numpy()
    scale, bias = vh.bn_fold(to_dev(gamma), to_dev(beta), to_dev(mean), to_dev(var), 1e-5)
    xd = to_dev(_nhwc(x))
    y = vh.deconv4x4s2_winograd_fwd(xd, vh.pack_winograd_deconv_weight(to_dev(wt)), scale, bias, cout, True)
    e = rel_err(np.transpose(y.cpu().numpy(), (0, 3, 1, 2)), ref)
    record("deconv_winograd_" + name, rel=e)
    assert e < TOL, (name, e)
    if cin % 32 == 0:
        yd = vh.deconv4x4s2_fwd(xd, vh.pack_deconv_weight(to_dev(wt)), scale, bias, cout, True)
        assert rel_err(y.cpu().numpy(), yd.cpu().numpy()) < 2 * TOL


def test_deconv_winograd_statistics_and_batch_position(vh):
    g = torch.Generator(device="cpu").manual_seed(31)
    for n, h, w, cin, cout in ((5, 8, 6, 64, 64), (3, 5, 3, 32, 40)):
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cin, cout, 4, 4), generator=g) * 0.05).to(dev())
        u = vh.pack_winograd_deconv_weight(wt)
        gamma, beta = torch.ones(cout, device=dev()), torch.zeros(cout, device=dev())
        rm, rv = torch.zeros(cout, device=dev()), torch.ones(cout, device=dev())
        z, mean, invstd, scale, bias = vh.deconv4x4s2_winograd_fwd_bnstats(x, u, cout, gamma, beta, rm, rv, 0.1, 1e-5)
        plain = vh.deconv4x4s2_winograd_fwd(x, u, None, None, cout, False)
        assert torch.equal(z, plain)
        z64 = z.double().reshape(-1, cout)
        assert torch.allclose(mean.double(), z64.mean(0), rtol=1e-5, atol=1e-6)
        assert torch.allclose(invstd.double(), 1.0 / torch.sqrt(z64.var(0, unbiased=False) + 1e-5), rtol=1e-5)
        for i in (0, n - 1):
            alone = vh.deconv4x4s2_winograd_fwd(x[i:i + 1].contiguous(), u, None, None, cout, False)
            assert torch.equal(alone[0], plain[i])


WGRAD_CASES = [(2, 16, 12, 64, 128), (3, 13, 9, 32, 48), (1, 8, 6, 256, 64), (7, 8, 6, 64, 64), (5, 2, 2, 16, 8), (2, 64, 48, 32, 32), (40, 16, 12, 36, 20)]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=["x".join(map(str, c)) for c in WGRAD_CASES])
def test_winograd_weight_gradient_of_the_3x3_layers(vh, case):
    """dW of a 3x3 / stride 1 / pad 1 conv through the transform domain against autograd in float64 and against the implicit GEMM."""
    n, h, w, cin, cout = case
    g = torch.Generator(device="cpu").manual_seed(41 + n)
    x = torch.randn((n, cin, h, w), generator=g, dtype=torch.float64)
    wt = (torch.randn((cout, cin, 3, 3), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    dy = torch.randn((n, cout, h, w), generator=g, dtype=torch.float64)
    F.conv2d(x, wt, None, 1, 1).backward(dy)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev())
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev())
    dw = vh.conv3x3_winograd_wgrad(xd, dyd)
    e = rel_err(dw.cpu().numpy(), wt.grad.numpy())
    record(f"winograd_wgrad_{n}x{h}x{w}_{cin}_{cout}", rel=e)
    assert e < TOL, e
    again = vh.conv3x3_winograd_wgrad(xd, dyd)
    assert torch.equal(again, dw)                               # fixed reduction order
    if cin % 4 == 0 and cout % 4 == 0:
        direct = vh.conv2d_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
        assert rel_err(dw.cpu().numpy(), direct.cpu().numpy()) < 2 * TOL


@pytest.mark.parametrize("case", [(2, 8, 6, 64, 48), (3, 5, 3, 32, 40), (1, 16, 12, 128, 64), (9, 9, 6, 16, 16), (4, 1, 4, 16, 8)], ids=str)
def test_winograd_weight_gradient_of_the_transposed_conv(vh, case):
    n, h, w, cin, cout = case
    g = torch.Generator(device="cpu").manual_seed(43 + n)
    x = torch.randn((n, cin, h, w), generator=g, dtype=torch.float64)
    wt = (torch.randn((cin, cout, 4, 4), generator=g, dtype=torch.float64) * 0.05).requires_grad_(True)
    dy = torch.randn((n, cout, 2 * h, 2 * w), generator=g, dtype=torch.float64)
    F.conv_transpose2d(x, wt, None, 2, 1).backward(dy)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev())
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev())
    dw = vh.deconv4x4s2_winograd_wgrad(xd, dyd)
    e = rel_err(dw.cpu().numpy(), wt.grad.numpy())
    record(f"winograd_deconv_wgrad_{n}x{h}x{w}_{cin}_{cout}", rel=e)
    assert e < TOL, e
    direct = vh.deconv4x4s2_wgrad(xd, dyd)
    assert rel_err(dw.cpu().numpy(), direct.cpu().numpy()) < 2 * TOL


def test_small_batch_module_calls_stay_on_the_implicit_gemm(vh, monkeypatch):
    """`model(x)` with <= 16 crops is the latency path (split-K for this thread, hip_engine.run_module_nchw): it keeps its own,
    batch-size-independent bits on the implicit GEMM — no Winograd launch — while the stream entry point takes the Winograd route."""
    from alphapose.models import hip_engine
    from oracle import synth
    from tests.test_gpu_conv import _build_simplepose
    m = _build_simplepose()
    x = to_dev(synth.crops(4))
    calls = []
    for name in ("conv3x3_winograd_fwd", "conv3x3_winograd_f4_fwd", "deconv4x4s2_winograd_fwd"):
        orig = getattr(vh, name)
        monkeypatch.setattr(vh, name, (lambda o: lambda *a, **k: (calls.append(1), o(*a, **k))[1])(orig))
    with torch.no_grad():
        small = m(x)
    assert not calls and not vh.latency_mode()
    with torch.no_grad():
        one = m(x[1:2])
        stream = hip_engine.forward_into(m, x, torch.empty_like(small))
    assert torch.equal(one[0], small[1])                       # the module-call bits do not depend on the batch size
    assert len(calls) == 16                                     # 13 3x3 layers + 3 transposed convs
    assert rel_err(stream.cpu().numpy(), small.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("case", [(2, 8, 6, 64, 48), (3, 5, 3, 32, 48), (1, 16, 12, 128, 64), (9, 9, 6, 16, 32), (4, 1, 4, 8, 16), (120, 8, 6, 32, 16)], ids=str)
def test_winograd_data_gradient_of_the_transposed_conv(vh, case):
    """dx of ConvTranspose2d(4,2,1) = a 4x4 / stride 2 conv over dz, as F(3x3,2x2) over the four pixel phases of dz; plain, with a
    residual, and with the BatchNorm-backward epilogue against the implicit-GEMM launch it replaces."""
    n, h, w, cin, cout = case
    g = torch.Generator(device="cpu").manual_seed(47 + n)
    x = torch.randn((n, cin, h, w), generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn((cin, cout, 4, 4), generator=g, dtype=torch.float64) * 0.05
    dy = torch.randn((n, cout, 2 * h, 2 * w), generator=g, dtype=torch.float64)
    F.conv_transpose2d(x, wt, None, 2, 1).backward(dy)
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().to(dev())
    u = vh.pack_winograd_deconv_dgrad_weight(wt.float().to(dev()))
    dx = vh.deconv4x4s2_winograd_dgrad(dyd, u, cin)
    e = rel_err(dx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.numpy())
    record(f"winograd_deconv_dgrad_{n}x{h}x{w}_{cin}_{cout}", rel=e)
    assert e < TOL, e
    r = torch.randn((n, h, w, cin), generator=g).to(dev())
    assert torch.allclose(vh.deconv4x4s2_winograd_dgrad(dyd, u, cin, residual=r), dx + r, rtol=0, atol=1e-5)
    if cin % 4 == 0 and cout % 32 == 0:
        # the fused BatchNorm-backward epilogue: same masked gradient and the same statistics as the implicit-GEMM launch
        z = torch.randn((n, h, w, cin), generator=g).to(dev())
        mean, invstd = z.reshape(-1, cin).mean(0), 1.0 / (z.reshape(-1, cin).var(0, unbiased=False) + 1e-5).sqrt()
        sc, bi = (torch.rand(cin, generator=g) + 0.5).to(dev()), (torch.randn(cin, generator=g) * 0.3).to(dev())
        s1, s2 = vh.BnBwdSpec(z, mean, invstd, scale=sc, bias=bi), vh.BnBwdSpec(z, mean, invstd, scale=sc, bias=bi)
        g1 = vh.deconv4x4s2_winograd_dgrad(dyd, u, cin, spec=s1)
        g2 = vh.conv2d_fwd_ex_bnbwd(dyd, vh.pack_conv_weight(wt.float().to(dev())), cin, 4, 4, 2, 1, 1, h, w, h, w, 1, 1, 0, 0, s2)
        assert rel_err(g1.cpu().numpy(), g2.cpu().numpy()) < 2 * TOL
        t1 = s1.stats[:s1.blocks * cin * 2].view(s1.blocks, cin, 2).sum(0)
        t2 = s2.stats[:s2.blocks * cin * 2].view(s2.blocks, cin, 2).sum(0)
        assert torch.allclose(t1, t2, rtol=1e-4, atol=1e-3 * float(t2.abs().max()))


def test_winograd_full_batch_properties(vh):
    """BASELINE.json's batch (1024 crops per launch sequence): the properties that do not need a reference at that size — a crop's bits
    equal its single-crop launch, linearity in the input, and a float64 spot check of eight crops — for a 3x3 layer and a transposed conv."""
    g = torch.Generator(device="cpu").manual_seed(53)
    n, h, w, cin, cout = 1024, 16, 12, 256, 256
    x = torch.randn((n, h, w, cin), generator=g).to(dev())
    x2 = torch.randn((n, h, w, cin), generator=g).to(dev())
    wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.03).to(dev())
    u = vh.pack_winograd_weight(wt)
    y = vh.conv3x3_winograd_fwd(x, u, None, None, cout, False)
    for i in (0, 517, 1023):
        assert torch.equal(vh.conv3x3_winograd_fwd(x[i:i + 1].contiguous(), u, None, None, cout, False)[0], y[i])
    lin = vh.conv3x3_winograd_fwd(2.0 * x - 0.5 * x2, u, None, None, cout, False)
    y2 = vh.conv3x3_winograd_fwd(x2, u, None, None, cout, False)
    assert rel_err(lin.cpu().numpy(), (2.0 * y - 0.5 * y2).cpu().numpy()) < 1e-5
    pick = [3, 100, 511, 512, 640, 900, 1000, 1023]
    ref = F.conv2d(x[pick].permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), None, 1, 1).permute(0, 2, 3, 1)
    e = rel_err(y[pick].cpu().numpy(), ref.numpy())
    record("winograd_full_batch_3x3", rel=e)
    assert e < TOL
    # transposed conv at the batch of the headline (deconv1's grid, narrower channels to keep the float64 reference cheap)
    n, h, w, cin, cout = 1024, 8, 6, 128, 64
    x = torch.randn((n, h, w, cin), generator=g).to(dev())
    wd = (torch.randn((cin, cout, 4, 4), generator=g) * 0.05).to(dev())
    ud = vh.pack_winograd_deconv_weight(wd)
    y = vh.deconv4x4s2_winograd_fwd(x, ud, None, None, cout, False)
    for i in (0, 700, 1023):
        assert torch.equal(vh.deconv4x4s2_winograd_fwd(x[i:i + 1].contiguous(), ud, None, None, cout, False)[0], y[i])
    ref = F.conv_transpose2d(x[pick].permute(0, 3, 1, 2).double().cpu(), wd.double().cpu(), None, 2, 1).permute(0, 2, 3, 1)
    e = rel_err(y[pick].cpu().numpy(), ref.numpy())
    record("winograd_full_batch_deconv", rel=e)
    assert e < TOL


def test_two_half_blocks_give_the_bits_of_the_small_shape(vh):
    """vatl_tune_set(21, v): 32 tiles x 64 channels per block (two filter halves, taken by default only where a launch keeps >= 400
    blocks) against 32 x 32 — same products, same summation order per output: every epilogue must give identical bits.  Forced on
    (v = 3) for shapes far below the default floor, including Cout = 96 / 192 (odd / even half counts) and tail tiles."""
    g = torch.Generator(device="cpu").manual_seed(77)

    def both(fn):
        outs = []
        for v in (1, 3):
            vh.tune_set(21, v)
            try:
                outs.append(fn())
            finally:
                vh.tune_set(21, 2)
        return outs

    for n, h, w, cin, cout in ((3, 16, 12, 64, 64), (2, 13, 9, 32, 128), (5, 8, 6, 128, 192), (2, 10, 6, 48, 96), (1, 7, 5, 16, 256)):
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(dev())
        sc, bi = (torch.rand(cout, generator=g) + 0.5).to(dev()), torch.randn(cout, generator=g).to(dev())
        r = torch.randn((n, h, w, cout), generator=g).to(dev())
        u = vh.pack_winograd_weight(wt)
        a, b = both(lambda: vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, True, residual=r))
        assert torch.equal(a, b)
        (ya, sa, ba), (yb, sb, bb) = both(lambda: vh.conv3x3_winograd_fwd_stats(x, u, cout))
        assert torch.equal(ya, yb) and ba == bb and torch.equal(sa[:ba * cout * 2], sb[:bb * cout * 2])
        # data gradient with the BatchNorm-backward epilogue (u of the rotated / transposed filter: cin <-> cout)
        if cin % 4 == 0:
            ud = vh.pack_winograd_weight(wt, data_gradient=True)
            dz = torch.randn((n, h, w, cout), generator=g).to(dev())
            z = torch.randn((n, h, w, cin), generator=g).to(dev())
            mean, invstd = z.reshape(-1, cin).mean(0), 1.0 / (z.reshape(-1, cin).var(0, unbiased=False) + 1e-5).sqrt()
            s2, b2 = (torch.rand(cin, generator=g) + 0.5).to(dev()), (torch.randn(cin, generator=g) * 0.3).to(dev())

            def bnb():
                spec = vh.BnBwdSpec(z, mean, invstd, scale=s2, bias=b2)
                gq = vh.conv3x3_winograd_fwd_bnbwd(dz, ud, cin, spec)
                return gq, spec.stats[:spec.blocks * cin * 2].clone()
            (ga, ta), (gb, tb) = both(bnb)
            assert torch.equal(ga, gb) and torch.equal(ta, tb)
    for n, h, w, cin, cout in ((2, 8, 6, 64, 64), (3, 5, 3, 32, 128), (1, 16, 12, 128, 192)):
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cin, cout, 4, 4), generator=g) * 0.05).to(dev())
        sc, bi = (torch.rand(cout, generator=g) + 0.5).to(dev()), torch.randn(cout, generator=g).to(dev())
        u = vh.pack_winograd_deconv_weight(wt)
        a, b = both(lambda: vh.deconv4x4s2_winograd_fwd(x, u, sc, bi, cout, True))
        assert torch.equal(a, b)
        # its data gradient (gather mode): reduction over cout, output channels = cin
        ug = vh.pack_winograd_deconv_dgrad_weight(wt)
        dy = torch.randn((n, 2 * h, 2 * w, cout), generator=g).to(dev())
        a, b = both(lambda: vh.deconv4x4s2_winograd_dgrad(dy, ug, cin))
        assert torch.equal(a, b)
    record("winograd_two_half_blocks_bit_identical", ok=True)


def test_two_half_blocks_random_shapes(vh):
    """Seeded sweep over odd sizes for the block-shape identity: 3x3 forward (+ residual / ReLU / folded affine at random) and the
    transposed conv's forward and data gradient, both shapes forced (knob 21 = 1 / 3)."""
    rng = np.random.RandomState(2024)
    g = torch.Generator(device="cpu").manual_seed(2024)
    for k in range(16):
        n, h, w = int(rng.randint(1, 7)), int(rng.randint(1, 20)), int(rng.randint(1, 20))
        cin, cout = 16 * int(rng.randint(1, 7)), 64 * int(rng.randint(1, 4))
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        outs = []
        if k % 2 == 0:
            wt = (torch.randn((cout, cin, 3, 3), generator=g) * 0.05).to(dev())
            u = vh.pack_winograd_weight(wt)
            sc = (torch.rand(cout, generator=g) + 0.5).to(dev()) if rng.rand() < 0.5 else None
            bi = torch.randn(cout, generator=g).to(dev()) if sc is not None else None
            r = torch.randn((n, h, w, cout), generator=g).to(dev()) if rng.rand() < 0.5 else None
            relu = bool(rng.rand() < 0.5)
            fn = lambda: (vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, relu, residual=r),)      # noqa: E731
        else:
            wt = (torch.randn((cin, cout, 4, 4), generator=g) * 0.05).to(dev())
            u, ug = vh.pack_winograd_deconv_weight(wt), vh.pack_winograd_deconv_dgrad_weight(wt)
            dy = torch.randn((n, 2 * h, 2 * w, cout), generator=g).to(dev())
            fn = lambda: (vh.deconv4x4s2_winograd_fwd(x, u, None, None, cout, False), vh.deconv4x4s2_winograd_dgrad(dy, ug, cin))   # noqa: E731
        for v in (1, 3):
            vh.tune_set(21, v)
            try:
                outs.append(fn())
            finally:
                vh.tune_set(21, 2)
        for a, b in zip(*outs):
            assert torch.equal(a, b), (k, n, h, w, cin, cout)
    record("winograd_two_half_blocks_random", shapes=16)


def test_persistent_route_gives_the_bits_of_the_plain_kernel(vh):
    """winograd_persist_kernel (short blocks: <= 128 input channels): a block keeps one 32-tile group of the period lcm(tiles per image, 32)
    and walks the periods by base address; images that do not fill a period go through the plain kernel.  Same per-tile arithmetic and
    summation order: bit-identical to the plain kernel (vatl_tune_set(22, 0)), with scale / bias / residual / ReLU, one and two filter
    halves per block, periods of 1 / 2 / 8 / 16 images, partial tiles at the image border, and a remainder."""
    lib = vh.lib()
    g = torch.Generator(device="cpu").manual_seed(31)
    #        n,  h,  w, cin, cout, residual, expected route (1 persistent, 2 persistent + tail launch, 0 plain, None: whatever the block-count rule picks)
    cases = [(64, 64, 48, 32, 32, True, 1), (1024, 32, 24, 64, 64, True, 1), (4101, 8, 6, 32, 32, True, 2), (1025, 16, 12, 32, 32, False, 2),
             (8199, 5, 4, 32, 32, True, 2), (2041, 16, 12, 64, 64, True, 2), (517, 16, 12, 128, 128, False, None), (40, 13, 9, 32, 48, True, 0),
             (24, 16, 12, 256, 256, False, 0)]
    took = []
    for n, h, w, cin, cout, res, want in cases:
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
        sc = (torch.rand(cout, generator=g) + 0.5).to(dev()); bi = torch.randn(cout, generator=g).to(dev())
        r = torch.randn((n, h, w, cout), generator=g).to(dev()) if res else None
        u = vh.pack_winograd_weight(wt)
        y = vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, True, residual=r)
        route = lib.vatl_winograd_last_route()
        y2 = vh.conv3x3_winograd_fwd(x, u, None, None, cout, False)
        vh.tune_set(22, 0)
        try:
            p = vh.conv3x3_winograd_fwd(x, u, sc, bi, cout, True, residual=r)
            assert lib.vatl_winograd_last_route() == 0
            p2 = vh.conv3x3_winograd_fwd(x, u, None, None, cout, False)
        finally:
            vh.tune_set(22, 8)
        assert want is None or route == want, (n, h, w, cin, cout, route)
        assert torch.equal(y, p) and torch.equal(y2, p2), (n, h, w, cin, cout, route)
        took.append(route)
        k = min(n, 3)                                       # and against float64 (last images of the batch: the tail launch / the last period)
        ref = F.conv2d(x[-k:].permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1)
        assert rel_err(y2[-k:].cpu().numpy(), ref.numpy()) < TOL
    record("winograd_persistent_route", routes=took)
    assert took.count(1) >= 2 and took.count(2) >= 4, took       # whole periods only, and periods + a plain launch for the remainder


def test_wgrad_address_tables_give_the_bits_of_the_in_kernel_arithmetic(vh):
    """The Winograd weight-gradient kernels read their staging addresses from tables written by winograd_wgrad_table_kernel right before them (vatl_tune_set(25, 1),
    default) instead of forming them per stage (two divisions and ~25 vector instructions per request): same addresses, so the same bits — 3x3 layers with one and
    two gradient halves per block, a transposed conv (four x phases), image / tile counts that leave partial stages and splits."""
    g = torch.Generator(device="cpu").manual_seed(15)
    cases = [("conv", 2, 8, 6, 32, 32), ("conv", 5, 16, 12, 64, 128), ("conv", 3, 10, 14, 256, 256), ("conv", 120, 16, 12, 128, 128),
             ("deconv", 2, 8, 6, 64, 32), ("deconv", 7, 16, 12, 128, 64), ("deconv", 3, 5, 7, 32, 32)]
    try:
        for kind, b, h, w, cin, cout in cases:
            x = torch.randn((b, h, w, cin), generator=g).to(dev())
            dy = torch.randn((b, 2 * h, 2 * w, cout) if kind == "deconv" else (b, h, w, cout), generator=g).to(dev())
            f = (lambda: vh.deconv4x4s2_winograd_wgrad(x, dy)) if kind == "deconv" else (lambda: vh.conv3x3_winograd_wgrad(x, dy))
            vh.tune_set(25, 0)
            a = f().clone()
            vh.tune_set(25, 1)
            c = f().clone()
            assert torch.equal(a, c), (kind, b, h, w, cin, cout)
    finally:
        vh.tune_set(25, 1)


def test_winograd_f4_vs_float64_and_the_f2_route(vh):
    """vatl_conv3x3_winograd_f4_fwd (csrc/winograd_f4.hip): F(4x4,3x3) on 16x16x4 MFMAs for the layers whose grid is whole 4x4 tiles.  Against float64 convolutions on
    shapes with whole and partial 16-tile blocks, one to eight 64-channel slices, short and long reductions, with / without folded BN, skip connection and ReLU; held to
    the bound every Winograd test of this file uses (2e-5 of the largest output; measured 5e-6 .. 1.1e-5 on unit-variance data, the F(2x2) route 2e-7 .. 7e-7: the larger
    transform constants cost ~4 bits per layer — on the reference's golden crops the heat-maps do not move, tests/test_gpu_conv.py); a crop's bits independent of its
    batch position; unsupported shapes refused."""
    r = np.random.RandomState(63)
    assert vh.conv3x3_winograd_f4_supported(1024, 16, 12, 256, 256) and vh.conv3x3_winograd_f4_supported(1024, 32, 24, 128, 128)
    assert not vh.conv3x3_winograd_f4_supported(4, 8, 6, 512, 512) and not vh.conv3x3_winograd_f4_supported(4, 16, 12, 32, 32)
    assert not vh.conv3x3_winograd_f4_supported(4, 16, 12, 256, 96)
    for n, h, wd, cin, cout in ((3, 16, 12, 256, 256), (2, 32, 24, 128, 128), (5, 8, 8, 64, 64), (1, 4, 4, 64, 128), (7, 12, 20, 80, 192), (2, 16, 12, 128, 512), (33, 4, 8, 64, 64)):
        w = (r.standard_normal((cout, cin, 3, 3)) * (2.0 / (9 * cin)) ** 0.5).astype(np.float32)
        sc, bi = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32)
        u4, u2 = vh.pack_winograd_f4_weight(to_dev(w)), vh.pack_winograd_weight(to_dev(w))
        x = r.standard_normal((n, h, wd, cin)).astype(np.float32)
        res = r.standard_normal((n, h, wd, cout)).astype(np.float32)
        conv = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w).double(), padding=1).permute(0, 2, 3, 1).numpy()
        xd, rd = to_dev(x), to_dev(res)
        for relu, use_res, use_sb in ((True, True, True), (False, False, False), (True, False, True)):
            want = conv * (sc if use_sb else 1.0) + (bi if use_sb else 0.0) + (res if use_res else 0.0)
            want = np.maximum(want, 0) if relu else want
            s_, b_ = (to_dev(sc), to_dev(bi)) if use_sb else (None, None)
            got = vh.conv3x3_winograd_f4_fwd(xd, u4, s_, b_, cout, relu, residual=rd if use_res else None)
            f2 = vh.conv3x3_winograd_fwd(xd, u2, s_, b_, cout, relu, residual=rd if use_res else None)
            e4, e2 = rel_err(got.cpu().numpy(), want), rel_err(f2.cpu().numpy(), want)
            assert e4 < 2e-5 and e2 < 2e-5, (n, h, wd, cin, cout, relu, use_res, use_sb, e4, e2)
        record(f"winograd_f4_{n}x{h}x{wd}x{cin}x{cout}", vs_fp64=e4, f2_route_vs_fp64=e2)
        if n > 1:
            full = vh.conv3x3_winograd_f4_fwd(xd, u4, to_dev(sc), to_dev(bi), cout, True, residual=rd)
            solo = vh.conv3x3_winograd_f4_fwd(xd[1:2].contiguous(), u4, to_dev(sc), to_dev(bi), cout, True, residual=rd[1:2].contiguous())
            assert torch.equal(solo, full[1:2])
            assert torch.equal(vh.conv3x3_winograd_f4_fwd(xd, u4, to_dev(sc), to_dev(bi), cout, True, residual=rd), full)      # same bits again
    with pytest.raises(vh.VatlError):
        vh.conv3x3_winograd_f4_fwd(to_dev(r.standard_normal((1, 6, 8, 64)).astype(np.float32)), u4, None, None, 64, False)


def test_winograd_f4_training_epilogues(vh):
    """The F(4x4,3x3) kernel under `model.train()`: (1) statistics epilogue: z bit-identical to the plain launch, the (sum, sum^2) row-block partials equal to the
    sums of the stored tensor, and the finalized mean / inverse deviation equal to float64 statistics of z; (2) the data-gradient packing against autograd in
    float64; (3) the BatchNorm-backward epilogue (mask recomputed from z / taken from a saved output / none; with and without a residual): g = (conv + residual) *
    mask with the plain launch's bits, (sum g, sum g * xhat) against float64 sums of the stored g, bit-reproducible; (4) the step's one re-pack launch refreshes
    the F(4x4) filters with the bits of the pack kernel."""
    g = torch.Generator(device="cpu").manual_seed(29)
    for n, h, w, cin, cout in ((5, 16, 12, 64, 64), (3, 8, 12, 128, 256), (17, 4, 4, 64, 128)):
        x = torch.randn((n, h, w, cin), generator=g).to(dev())
        wt = (torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev())
        u = vh.pack_winograd_f4_weight(wt)
        plain = vh.conv3x3_winograd_f4_fwd(x, u, None, None, cout, False)
        gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(dev()), torch.randn(cout, generator=g).to(dev())
        rm, rv = torch.zeros(cout, device=dev()), torch.ones(cout, device=dev())
        z, mean, invstd, scale, bias = vh.conv3x3_winograd_f4_fwd_bnstats(x, u, cout, gamma, beta, rm, rv, 0.1, 1e-5)
        assert torch.equal(z, plain)
        z64 = z.double().reshape(-1, cout)
        assert torch.allclose(mean.double(), z64.mean(0), rtol=1e-5, atol=1e-6)
        assert torch.allclose(invstd.double(), 1.0 / (z64.var(0, unbiased=False) + 1e-5).sqrt(), rtol=1e-5)
        assert torch.allclose(rm.double(), 0.1 * z64.mean(0), rtol=1e-5, atol=1e-6)
        # (2) data gradient: a 3x3 conv of dz with cout input and cin output channels (cin % 64 == 0 in every case here)
        dy = torch.randn((n, h, w, cout), generator=g).to(dev())
        xr = x.double().permute(0, 3, 1, 2).cpu().requires_grad_(True)
        F.conv2d(xr, wt.double().cpu(), None, 1, 1).backward(dy.double().permute(0, 3, 1, 2).cpu())
        ud = vh.pack_winograd_f4_weight(wt, data_gradient=True)
        dx = vh.conv3x3_winograd_f4_fwd(dy, ud, None, None, cin, False)
        e = rel_err(dx.permute(0, 3, 1, 2).cpu().numpy(), xr.grad.numpy())
        record(f"winograd_f4_dgrad_{cin}_{cout}", rel=e)
        assert e < 2e-5, e
        # (3) BatchNorm-backward epilogue of that data gradient; the consumer layer's conv output zc has the data gradient's shape (n, h, w, cin)
        zc = torch.randn((n, h, w, cin), generator=g).to(dev())
        yc = torch.randn((n, h, w, cin), generator=g).to(dev())
        mu, isd = zc.reshape(-1, cin).mean(0), 1.0 / (zc.reshape(-1, cin).var(0, unbiased=False) + 1e-5).sqrt()
        s2, b2 = (torch.rand(cin, generator=g) + 0.5).to(dev()), (torch.randn(cin, generator=g) * 0.3).to(dev())
        res = torch.randn((n, h, w, cin), generator=g).to(dev())
        for kind, use_res in (("recompute", False), ("saved", True), ("none", True)):
            def run():
                spec = vh.BnBwdSpec(zc, mu, isd, scale=s2, bias=b2) if kind == "recompute" else (vh.BnBwdSpec(zc, mu, isd, mask_y=yc) if kind == "saved" else vh.BnBwdSpec(zc, mu, isd))
                gq = vh.conv3x3_winograd_f4_fwd_bnbwd(dy, ud, cin, spec, residual=res if use_res else None)
                return gq, spec.stats[:spec.blocks * cin * 2].clone(), spec.blocks
            (ga, ta, blocks), (gb, tb, _) = run(), run()
            assert torch.equal(ga, gb) and torch.equal(ta, tb)
            d = vh.conv3x3_winograd_f4_fwd(dy, ud, None, None, cin, False, residual=res if use_res else None)
            mask = (zc * s2 + b2 > 0) if kind == "recompute" else ((yc > 0) if kind == "saved" else torch.ones_like(zc, dtype=torch.bool))
            assert torch.equal(ga, torch.where(mask, d, torch.zeros_like(d))), kind
            sums = ta.view(blocks, cin, 2).sum(0)
            g64, xhat = ga.double().reshape(-1, cin), ((zc - mu) * isd).double().reshape(-1, cin)
            assert torch.allclose(sums[:, 0], g64.sum(0), rtol=1e-5, atol=1e-3) and torch.allclose(sums[:, 1], (g64 * xhat).sum(0), rtol=1e-5, atol=1e-3), kind
        # (4) the pack plan
        plan = vh.PackPlan()
        prev = vh.set_pack_plan(plan)
        try:
            plan.begin()
            first = (vh.pack_winograd_f4_weight(wt), vh.pack_winograd_f4_weight(wt, data_gradient=True), vh.pack_conv_weight(wt))
            plan.seal()
            want = [t.clone() for t in first]
            for t in first:
                t.zero_()
            plan.begin()                                           # the one launch re-packs all three
            again = (vh.pack_winograd_f4_weight(wt), vh.pack_winograd_f4_weight(wt, data_gradient=True), vh.pack_conv_weight(wt))
            assert all(a.data_ptr() == b.data_ptr() for a, b in zip(first, again)) and all(torch.equal(a, b) for a, b in zip(again, want))
        finally:
            vh.set_pack_plan(prev)
        assert torch.equal(want[0], u) and torch.equal(want[1], ud)
