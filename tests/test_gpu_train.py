"""Training-mode kernels (BN batch stats, dgrad, wgrad, pool backward) vs torch-CPU autograd in float64,
and one whole SimplePose-R50 fine-tune step vs the reference-generated golden step."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import synth
from tests.gpu_util import dev, record, rel_err, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vh():
    import vatl_hip
    vatl_hip.lib()
    return vatl_hip


def _nhwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


def _nchw(x):
    return np.transpose(x, (0, 3, 1, 2))


def test_bn_train_forward_backward(vh):
    r = np.random.RandomState(0)
    for (n, c, h, w, relu, skip) in ((3, 64, 9, 7, True, True), (2, 256, 8, 6, True, False), (2, 128, 5, 5, False, False)):
        z = (r.standard_normal((n, c, h, w)) * 2 + 0.5).astype(np.float32)
        res = r.standard_normal((n, c, h, w)).astype(np.float32) if skip else None
        gamma, beta = r.uniform(0.5, 1.5, c).astype(np.float32), (0.1 * r.standard_normal(c)).astype(np.float32)
        rm, rv = (0.1 * r.standard_normal(c)).astype(np.float32), r.uniform(0.5, 1.5, c).astype(np.float32)
        dy = r.standard_normal((n, c, h, w)).astype(np.float32)
        # reference (float64 autograd)
        zt = torch.from_numpy(z).double().requires_grad_()
        gt, bt = torch.from_numpy(gamma).double().requires_grad_(), torch.from_numpy(beta).double().requires_grad_()
        rmt, rvt = torch.from_numpy(rm).double(), torch.from_numpy(rv).double()
        rt = torch.from_numpy(res).double().requires_grad_() if skip else None
        yt = F.batch_norm(zt, rmt, rvt, gt, bt, True, 0.1, 1e-5)
        if skip:
            yt = yt + rt
        if relu:
            yt = yt.relu()
        yt.backward(torch.from_numpy(dy).double())
        # HIP
        zd = to_dev(_nhwc(z)); drm, drv = to_dev(rm), to_dev(rv)
        mean, invstd, scale, bias = vh.bn_train_fwd_stats(zd, to_dev(gamma), to_dev(beta), drm, drv, 0.1, 1e-5)
        y = vh.scale_bias_act(zd, scale, bias, to_dev(_nhwc(res)) if skip else None, relu)
        assert rel_err(_nchw(y.cpu().numpy()), yt.detach().numpy()) < 1e-5
        np.testing.assert_allclose(drm.cpu().numpy(), rmt.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(drv.cpu().numpy(), rvt.numpy(), rtol=1e-5, atol=1e-6)
        dz, g, dgamma, dbeta = vh.bn_train_bwd(to_dev(_nhwc(dy)), y if relu else None, zd, to_dev(gamma), mean, invstd, want_g=skip)
        assert rel_err(_nchw(dz.cpu().numpy()), zt.grad.numpy()) < 2e-5
        assert rel_err(dgamma.cpu().numpy(), gt.grad.numpy()) < 2e-5 and rel_err(dbeta.cpu().numpy(), bt.grad.numpy()) < 2e-5
        if skip:
            assert rel_err(_nchw(g.cpu().numpy()), rt.grad.numpy()) < 1e-6


GRAD_CASES = [  # name, N, H, W, Cin, Cout, k, stride, pad
    ("1x1_256_64", 2, 16, 12, 256, 64, 1, 1, 0),
    ("1x1_64_256", 2, 16, 12, 64, 256, 1, 1, 0),
    ("3x3_64_64", 2, 16, 12, 64, 64, 3, 1, 1),
    ("3x3_128_128", 3, 8, 6, 128, 128, 3, 1, 1),
    ("3x3s2_128_128", 2, 16, 12, 128, 128, 3, 2, 1),
    ("1x1s2_256_512", 2, 16, 12, 256, 512, 1, 2, 0),
    ("3x3_512_512_oddM", 1, 8, 6, 512, 512, 3, 1, 1),
]


@pytest.mark.parametrize("case", GRAD_CASES, ids=[c[0] for c in GRAD_CASES])
def test_conv_dgrad_wgrad(vh, case):
    from alphapose.models import hip_train
    name, n, h, w, cin, cout, k, stride, pad = case
    import zlib
    r = np.random.RandomState(zlib.crc32(name.encode()) % 2 ** 31)
    x = r.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    xt = torch.from_numpy(x).double().requires_grad_()
    wtt = torch.from_numpy(wt).double().requires_grad_()
    out = F.conv2d(xt, wtt, None, stride, pad)
    dz = r.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(dz).double())
    conv = torch.nn.Conv2d(cin, cout, k, stride, pad, bias=False).to(dev())
    conv.weight.data.copy_(torch.from_numpy(wt))
    layer = hip_train._ConvBN(conv, torch.nn.BatchNorm2d(cout), False)
    dzd = to_dev(_nhwc(dz))
    dw = vh.conv2d_wgrad(vh.nchw_to_nhwc(to_dev(x)), dzd, cout, cin, k, k, stride, pad)
    dx = layer._dgrad(dzd, (n, h, w, cin), None)
    ew, ex = rel_err(dw.cpu().numpy(), wtt.grad.numpy()), rel_err(_nchw(dx.cpu().numpy()), xt.grad.numpy())
    record("grad_" + name, wgrad_rel=ew, dgrad_rel=ex)
    assert ew < 5e-5 and ex < 2e-5, (name, ew, ex)


def test_stem_and_head_wgrad(vh):
    r = np.random.RandomState(3)
    x = r.standard_normal((2, 3, 32, 24)).astype(np.float32)
    wt = (r.standard_normal((64, 3, 7, 7)) / 12).astype(np.float32)
    xt, wtt = torch.from_numpy(x).double(), torch.from_numpy(wt).double().requires_grad_()
    out = F.conv2d(xt, wtt, None, 2, 3)
    dz = r.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(dz).double())
    dw = vh.conv2d_wgrad(vh.nchw_to_nhwc(to_dev(x), 4), to_dev(_nhwc(dz)), 64, 3, 7, 7, 2, 3)
    assert rel_err(dw.cpu().numpy(), wtt.grad.numpy()) < 5e-5
    # head: 17 output channels carried in a 32-channel NHWC gradient
    x = r.standard_normal((2, 256, 16, 12)).astype(np.float32)
    wt = (r.standard_normal((17, 256, 1, 1)) / 16).astype(np.float32)
    xt, wtt = torch.from_numpy(x).double().requires_grad_(), torch.from_numpy(wt).double().requires_grad_()
    bt = torch.zeros(17, dtype=torch.float64, requires_grad=True)
    out = F.conv2d(xt, wtt, bt)
    dy = r.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(dy).double())
    dyd = vh.nchw_to_nhwc(to_dev(dy), 32)
    dw = vh.conv2d_wgrad(vh.nchw_to_nhwc(to_dev(x)), dyd, 17, 256, 1, 1, 1, 0)
    assert rel_err(dw.cpu().numpy(), wtt.grad.numpy()) < 5e-5
    assert rel_err(vh.col_sum(dyd).cpu().numpy()[:17], bt.grad.numpy()) < 1e-5
    wd = vh.pack_dgrad_weight(to_dev(wt), [(0, 0)], cout_k=32)
    dx = vh.conv2d_fwd_ex(dyd, wd, 256, 1, 1, 1, 0, 0, 16, 12, 16, 12, 1, 1, 0, 0)
    assert rel_err(_nchw(dx.cpu().numpy()), xt.grad.numpy()) < 2e-5


def test_deconv_backward(vh):
    r = np.random.RandomState(4)
    for (n, h, w, cin, cout) in ((2, 8, 6, 256, 128), (1, 5, 3, 64, 64)):
        x = r.standard_normal((n, cin, h, w)).astype(np.float32)
        wt = (r.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
        xt, wtt = torch.from_numpy(x).double().requires_grad_(), torch.from_numpy(wt).double().requires_grad_()
        out = F.conv_transpose2d(xt, wtt, None, 2, 1)
        dy = r.standard_normal(tuple(out.shape)).astype(np.float32)
        out.backward(torch.from_numpy(dy).double())
        dyd = to_dev(_nhwc(dy))
        dw = vh.deconv4x4s2_wgrad(vh.nchw_to_nhwc(to_dev(x)), dyd)
        dx = vh.conv2d_fwd(dyd, vh.pack_conv_weight(to_dev(wt)), None, None, cin, 4, 4, 2, 1, False)
        assert rel_err(dw.cpu().numpy(), wtt.grad.numpy()) < 5e-5
        assert rel_err(_nchw(dx.cpu().numpy()), xt.grad.numpy()) < 2e-5


def test_maxpool_backward(vh):
    r = np.random.RandomState(5)
    for (n, c, h, w) in ((2, 64, 32, 24), (1, 8, 7, 5)):
        x = np.round(r.standard_normal((n, c, h, w)) * 2).astype(np.float32) / 2          # ties inside windows
        xt = torch.from_numpy(x).requires_grad_()
        out = F.max_pool2d(xt, 3, 2, 1)
        dy = r.standard_normal(tuple(out.shape)).astype(np.float32)
        out.backward(torch.from_numpy(dy))
        dx = vh.maxpool3x3s2_bwd(vh.nchw_to_nhwc(to_dev(x)), to_dev(_nhwc(dy)))
        np.testing.assert_allclose(_nchw(dx.cpu().numpy()), xt.grad.numpy(), rtol=1e-6, atol=1e-6)
        yp, idx = vh.maxpool3x3s2_fwd_idx(vh.nchw_to_nhwc(to_dev(x)))                       # training path: saved arg-max taps
        assert np.array_equal(_nchw(yp.cpu().numpy()), out.detach().numpy())
        dx2 = vh.maxpool3x3s2_bwd_idx(to_dev(_nhwc(dy)), idx, (h, w))
        np.testing.assert_allclose(_nchw(dx2.cpu().numpy()), xt.grad.numpy(), rtol=1e-6, atol=1e-6)


def test_stem_tail_fused_pool_matches_the_separate_passes_and_float64(vh):
    """bn1 -> relu -> maxpool (Resnet.py:171-172) in training mode without the full-resolution activation / gradient:
    `maxpool3x3s2_fwd_idx_affine` == scale_bias_act + maxpool3x3s2_fwd_idx (values and winners, bit for bit, ties and odd sizes
    included), `bn_train_bwd_relu_pool` == maxpool3x3s2_bwd_idx + bn_train_bwd_relu, and both against float64 autograd."""
    r = np.random.RandomState(17)
    for (n, c, h, w) in ((3, 64, 32, 24), (2, 8, 7, 5), (1, 128, 9, 12)):
        z = (np.round(r.standard_normal((n, h, w, c)) * 4) / 4).astype(np.float32)          # quantised: ties inside windows
        gamma, beta = r.uniform(-1.5, 1.5, c).astype(np.float32), r.standard_normal(c).astype(np.float32)   # negative scales too
        zd, gd, bd = to_dev(z), to_dev(gamma), to_dev(beta)
        rm, rv = torch.zeros(c, device=zd.device), torch.ones(c, device=zd.device)
        mean, invstd, scale, bias = vh.bn_train_fwd_stats(zd, gd, bd, rm, rv, 0.1, 1e-5)
        y_full = vh.scale_bias_act(zd, scale, bias, None, True)
        yp0, idx0 = vh.maxpool3x3s2_fwd_idx(y_full)
        yp1, idx1 = vh.maxpool3x3s2_fwd_idx_affine(zd, scale, bias)
        assert torch.equal(yp0, yp1) and torch.equal(idx0, idx1)
        dpool = to_dev(r.standard_normal(tuple(yp0.shape)).astype(np.float32))
        dy_full = vh.maxpool3x3s2_bwd_idx(dpool, idx0, (h, w))
        dz0, dg0, db0 = vh.bn_train_bwd_relu(dy_full, scale, bias, zd, gd, mean, invstd)
        dz1, dg1, db1 = vh.bn_train_bwd_relu_pool(dpool, idx1, scale, bias, zd, gd, mean, invstd)
        assert torch.equal(dz0, dz1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
        # float64 autograd of the same graph
        zt = torch.from_numpy(_nchw(z).astype(np.float64)).requires_grad_()
        gt, bt = torch.from_numpy(gamma.astype(np.float64)).requires_grad_(), torch.from_numpy(beta.astype(np.float64)).requires_grad_()
        out = F.max_pool2d(F.relu(F.batch_norm(zt, None, None, gt, bt, True, 0.1, 1e-5)), 3, 2, 1)
        np.testing.assert_allclose(_nchw(yp1.cpu().numpy()), out.detach().numpy(), rtol=1e-5, atol=1e-5)
        out.backward(torch.from_numpy(_nchw(dpool.cpu().numpy()).astype(np.float64)))
        e = rel_err(_nchw(dz1.cpu().numpy()), zt.grad.numpy())
        record("stem_pool_fused", shape=[n, c, h, w], dz=e, dgamma=rel_err(dg1.cpu().numpy(), gt.grad.numpy()), dbeta=rel_err(db1.cpu().numpy(), bt.grad.numpy()))
        assert e < 5e-5 and rel_err(dg1.cpu().numpy(), gt.grad.numpy()) < 5e-5 and rel_err(db1.cpu().numpy(), bt.grad.numpy()) < 5e-5


def test_simplepose_finetune_step_vs_reference_golden(vh, golden_simplepose):
    """retrain_model's step (ActiveLearning.py:662-673): forward in train mode, 0.5*masked MSE, backward, AdamW with
    the three reference param groups — against the step the reference itself took (tools/make_golden.py)."""
    from active_learning.optim import AdamW
    from alphapose.models import builder
    from alphapose.utils.config import edict
    g = golden_simplepose
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    m = builder.build_sppe(cfg, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).train()
    lr, wd = 2.5e-4, 0.7
    opt = AdamW(params=[{"params": m.final_layer.parameters(), "lr": lr * 10}, {"params": m.preact.parameters(), "lr": lr},
                        {"params": m.deconv_layers.parameters(), "lr": lr * 5}], weight_decay=wd)
    x = to_dev(synth.crops(2))
    labels, masks = synth.gaussian_targets(2, seed=11)
    labels, masks = to_dev(labels), to_dev(masks)
    out = m(x.requires_grad_())
    loss = 0.5 * torch.nn.MSELoss()(out.mul(masks), labels.mul(masks))                     # the reference's own expression
    opt.zero_grad(); loss.backward(); opt.step()
    torch.cuda.synchronize()
    record("train_step", loss=float(loss), loss_ref=float(g["train_loss"]))
    np.testing.assert_allclose(float(loss), float(g["train_loss"]), rtol=1e-4)
    # fused loss kernel agrees with the torch expression
    lk, gk = vh.masked_mse_fwd_bwd(out.detach().contiguous(), labels, masks)
    np.testing.assert_allclose(float(lk), float(loss), rtol=1e-5)
    named = dict(m.named_parameters())
    sd = m.state_dict()
    # A 2-crop step through 53 batch-norm layers is ill-conditioned (96 samples per channel in layer4) and
    # chaotic at the 1e-3 level: the reference graph in fp32 on the CPU lands 2.5e-3 .. 8e-3 (max over a whole
    # tensor, relative to its largest entry) away from float64 depending only on thread count / memory format
    # (measured with oracle/nets.py, max-norm: 8 threads 7.9e-3, 1 thread 7.8e-3, channels_last 2.5e-3 on
    # deconv_layers.6; up to 4.4e-2 on layer4.0.downsample; L2-norm 1e-4 .. 5.3e-3) — ReLU masks flip on values
    # near zero.  Yardstick: the oracle graph in float64; we must sit inside that same band: L2-relative error
    # < 2e-2 and max-relative < 5e-2 per tensor (every per-op kernel is checked to <= 5e-5 against float64
    # autograd in the tests above; a real defect shows up here as an O(0.1..1) error).
    from oracle import nets
    ref64 = nets.SimplePoseRef(50)
    ref64.load_state_dict(synth.state_dict_for(ref64), strict=True)
    ref64 = ref64.double().train()
    o64 = ref64(torch.from_numpy(synth.crops(2)).double())
    l64 = 0.5 * torch.nn.MSELoss()(o64 * masks.cpu().double(), labels.cpu().double() * masks.cpu().double())
    l64.backward()
    exact = {k: p.grad.numpy() for k, p in ref64.named_parameters()}
    worst = 0.0
    for key in [k[10:] for k in g.files if k.startswith("grad_idx::")]:
        idx = g[f"grad_idx::{key}"]
        got = named[key].grad.reshape(-1)[torch.from_numpy(idx).to(dev())].cpu().numpy()
        ref = g[f"grad_val::{key}"]
        ex = exact[key].reshape(-1)[idx]
        scale = max(np.abs(ex).max(), 1e-30)
        e_ours, e_ref = float(np.abs(got - ex).max() / scale), float(np.abs(ref - ex).max() / scale)
        worst = max(worst, e_ours)
        record("train_grad", key=key, ours_vs_fp64=e_ours, reference_fp32_vs_fp64=e_ref, ours_vs_reference=float(np.abs(got - ref).max() / scale))
        l2_ours = float(np.linalg.norm(got - ex) / max(np.linalg.norm(ex), 1e-30))
        assert l2_ours < 2e-2 and e_ours < 5e-2, (key, l2_ours, e_ours, e_ref)
        new = sd[key].reshape(-1)[torch.from_numpy(idx).to(dev())].cpu().numpy()
        # AdamW's first step moves every weight by ~lr*sign(g): compare the updated values where the gradient's sign is certain
        sure = np.abs(ex) > 100 * np.abs(ref - ex).max()
        np.testing.assert_allclose(new[sure], g[f"new_val::{key}"][sure], rtol=1e-3, atol=1e-5)
    for key in [k[8:] for k in g.files if k.startswith("bnstat::")]:
        np.testing.assert_allclose(sd[key].cpu().numpy(), g[f"bnstat::{key}"], rtol=1e-4, atol=1e-5)
    assert int(sd["preact.bn1.num_batches_tracked"]) == int(g["bn_tracked"])
    record("train_step_worst_grad_rel", rel=worst)


def test_fastpose_training_kernels(vh):
    """PixelShuffle backward, SE gate backward, Linear backward vs float64 autograd."""
    r = np.random.RandomState(9)
    x = r.standard_normal((2, 64, 5, 3)).astype(np.float32)
    y = F.pixel_shuffle(torch.from_numpy(x), 2).numpy()
    back = vh.pixelunshuffle2(vh.nchw_to_nhwc(to_dev(y)))
    assert np.array_equal(_nchw(back.cpu().numpy()), x)
    # SE-gated residual
    n, c, h, w = 3, 64, 4, 6
    u = r.standard_normal((n, c, h, w)).astype(np.float32); sc = r.standard_normal((n, c, h, w)).astype(np.float32)
    g = r.standard_normal((n, c)).astype(np.float32); dy = r.standard_normal((n, c, h, w)).astype(np.float32)
    ut, st, gt = (torch.from_numpy(a).double().requires_grad_() for a in (u, sc, g))
    yt = (ut * torch.sigmoid(gt)[:, :, None, None] + st).relu()
    yt.backward(torch.from_numpy(dy).double())
    ud, dyd = to_dev(_nhwc(u)), to_dev(_nhwc(dy))
    yd = vh.se_scale_add_relu(ud, to_dev(g), to_dev(_nhwc(sc)))
    dgate = vh.se_bwd_gate(dyd, yd, ud, to_dev(g))
    assert rel_err(dgate.cpu().numpy(), gt.grad.numpy()) < 2e-5
    du, gm = vh.se_bwd_apply(dyd, yd, to_dev(g), torch.zeros((n, c), device=dev()))
    assert rel_err(_nchw(du.cpu().numpy()), ut.grad.numpy()) < 2e-5 and rel_err(_nchw(gm.cpu().numpy()), st.grad.numpy()) < 1e-6
    # Linear + ReLU
    from alphapose.models import hip_train
    lin = torch.nn.Linear(256, 128).to(dev())
    xt = torch.from_numpy(r.standard_normal((5, 256)).astype(np.float32))
    ref_lin = torch.nn.Linear(256, 128).double()
    ref_lin.load_state_dict({k: v.detach().cpu().double() for k, v in lin.state_dict().items()})
    xr = xt.double().requires_grad_()
    out = ref_lin(xr).relu()
    dyl = torch.from_numpy(r.standard_normal((5, 128)).astype(np.float32))
    out.backward(dyl.double())
    lt = hip_train._LinearT(lin, True)
    grads = {}
    yl = lt.forward(xt.to(dev()))
    dxl = lt.backward(dyl.to(dev()), grads)
    assert rel_err(yl.cpu().numpy(), out.detach().numpy()) < 1e-5 and rel_err(dxl.cpu().numpy(), xr.grad.numpy()) < 2e-5
    assert rel_err(grads[lin.weight].cpu().numpy(), ref_lin.weight.grad.numpy()) < 5e-5
    assert rel_err(grads[lin.bias].cpu().numpy(), ref_lin.bias.grad.numpy()) < 1e-5


def test_fastpose_finetune_step_vs_reference_golden(vh):
    """FastPose-R50 (SE blocks, DUC head) in training mode: loss and gradients against the reference's own step,
    judged with the float64 oracle as for SimplePose."""
    import os
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from oracle import nets
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_hrnet.npz"))
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    m = builder.build_sppe(edict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}), preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).train()
    x = to_dev(synth.crops(2))
    labels, masks = synth.gaussian_targets(2, seed=11)
    labels, masks = to_dev(labels), to_dev(masks)
    out = m(x.requires_grad_())
    loss = 0.5 * torch.nn.MSELoss()(out.mul(masks), labels.mul(masks))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["fastpose_train_loss"]), rtol=1e-4)
    ref64 = nets.FastPoseRef()
    ref64.load_state_dict(synth.state_dict_for(ref64), strict=True)
    ref64 = ref64.double().train()
    o64 = ref64(torch.from_numpy(synth.crops(2)).double())
    (0.5 * torch.nn.MSELoss()(o64 * masks.cpu().double(), labels.cpu().double() * masks.cpu().double())).backward()
    exact = {k: p.grad.numpy() for k, p in ref64.named_parameters()}
    named = dict(m.named_parameters())
    assert all(p.grad is not None for p in m.parameters())
    for key in [k[19:] for k in g.files if k.startswith("fastpose_grad_idx::")]:
        idx = g[f"fastpose_grad_idx::{key}"]
        got = named[key].grad.reshape(-1)[torch.from_numpy(idx).to(dev())].cpu().numpy()
        ex = exact[key].reshape(-1)[idx]
        ref = g[f"fastpose_grad_val::{key}"]
        scale = max(np.abs(ex).max(), 1e-30)
        l2 = float(np.linalg.norm(got - ex) / max(np.linalg.norm(ex), 1e-30))
        record("fastpose_train_grad", key=key, ours_l2_vs_fp64=l2, reference_fp32_l2_vs_fp64=float(np.linalg.norm(ref - ex) / max(np.linalg.norm(ex), 1e-30)),
               ours_max_vs_fp64=float(np.abs(got - ex).max() / scale))
        assert l2 < 2e-2 and np.abs(got - ex).max() / scale < 5e-2, (key, l2)


def test_upsample_and_gap_backward_kernels(vh):
    r = np.random.RandomState(9)
    for shift in (1, 2, 3):
        dy = r.standard_normal((2, 16, 24, 32)).astype(np.float32)
        y = r.standard_normal((2, 16, 24, 32)).astype(np.float32)
        f = 1 << shift
        for mask in (None, y):
            g = dy if mask is None else dy * (mask > 0)
            want = g.reshape(2, 16 // f, f, 24 // f, f, 32).astype(np.float64).sum((2, 4))
            got = vh.upsample_nearest_bwd(to_dev(dy), None if mask is None else to_dev(mask), shift).cpu().numpy()
            assert rel_err(got, want) < 1e-6
    dy = r.standard_normal((3, 64)).astype(np.float32)
    got = vh.gap_bwd(to_dev(dy), 48).cpu().numpy()
    np.testing.assert_allclose(got, np.broadcast_to(dy[:, None, :] / np.float32(48), (3, 48, 64)), rtol=1e-6)


def _hrnet_cfg():
    from alphapose.utils.config import edict
    return edict({"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1,
                  "PRETRAINED_LAYERS": ["*"],
                  "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                  "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                  "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}})


def test_hrnet_finetune_step_vs_reference_golden(vh):
    """HRNet-W32 in training mode (basic blocks, transitions, multi-resolution fusion backward): loss, BN running
    statistics and gradients against the reference's own step, judged with the float64 oracle."""
    import os
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from oracle import nets
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_hrnet.npz"))
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    m = builder.build_sppe(_hrnet_cfg(), preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).train()
    x = to_dev(synth.crops(2))
    labels, masks = synth.gaussian_targets(2, seed=11)
    labels, masks = to_dev(labels), to_dev(masks)
    out = m(x.requires_grad_())
    loss = 0.5 * torch.nn.MSELoss()(out.mul(masks), labels.mul(masks))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["hrnet_train_loss"]), rtol=1e-4)
    np.testing.assert_allclose(m.bn1.running_mean.cpu().numpy(), g["hrnet_bn1_running_mean"], rtol=1e-4, atol=1e-6)
    ref64 = nets.HRNetRef()
    ref64.load_state_dict(synth.state_dict_for(ref64), strict=True)
    ref64 = ref64.double().train()
    o64 = ref64(torch.from_numpy(synth.crops(2)).double())
    (0.5 * torch.nn.MSELoss()(o64 * masks.cpu().double(), labels.cpu().double() * masks.cpu().double())).backward()
    exact = {k: p.grad.numpy() for k, p in ref64.named_parameters()}
    named = dict(m.named_parameters())
    assert all(p.grad is not None for p in m.parameters())
    # every one of the 900+ parameter tensors against float64.  The B = 2 step is ill-conditioned (ReLU / BN sign
    # flips): the reference's own fp32 CPU step differs from float64 by 1.7e-2 at worst (BN biases of stage 4),
    # 1.6e-3 in the median (tools: oracle.nets.HRNetRef fp32 vs fp64) — the band below is that yardstick x3
    l2s = {}
    for key, p in named.items():
        ex = exact[key]
        l2s[key] = float(np.linalg.norm(p.grad.cpu().numpy() - ex) / max(np.linalg.norm(ex), 1e-30))
    worst = max(l2s, key=l2s.get)
    record("hrnet_train_grad_all", tensors=len(named), worst_key=worst, worst_l2_vs_fp64=l2s[worst], median_l2_vs_fp64=float(np.median(list(l2s.values()))))
    assert l2s[worst] < 5e-2, (worst, l2s[worst])
    assert np.median(list(l2s.values())) < 5e-3
    for key in [k[16:] for k in g.files if k.startswith("hrnet_grad_idx::")]:
        idx = g[f"hrnet_grad_idx::{key}"]
        got = named[key].grad.reshape(-1)[torch.from_numpy(idx).to(dev())].cpu().numpy()
        ex = exact[key].reshape(-1)[idx]
        ref = g[f"hrnet_grad_val::{key}"]
        scale = max(np.abs(ex).max(), 1e-30)
        l2 = float(np.linalg.norm(got - ex) / max(np.linalg.norm(ex), 1e-30))
        record("hrnet_train_grad", key=key, ours_l2_vs_fp64=l2, reference_fp32_l2_vs_fp64=float(np.linalg.norm(ref - ex) / max(np.linalg.norm(ex), 1e-30)),
               ours_max_vs_fp64=float(np.abs(got - ex).max() / scale))
        assert l2 < 2e-2 and np.abs(got - ex).max() / scale < 5e-2, (key, l2)


def test_fused_bn_statistics_and_mask_recompute(vh):
    """The training forward takes the BN batch statistics in the conv epilogue, the backward of a skip-less
    Conv+BN+ReLU recomputes the ReLU mask from z: both must agree with the stand-alone kernels."""
    r = np.random.RandomState(12)
    for (n, h, w, cin, cout, k, stride, pad) in ((3, 9, 7, 64, 64, 3, 1, 1), (2, 16, 12, 128, 256, 1, 1, 0), (5, 8, 6, 32, 32, 3, 2, 1)):
        x = to_dev(r.standard_normal((n, h, w, cin)).astype(np.float32))
        wt = to_dev((r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
        gamma, beta = to_dev(r.uniform(0.5, 1.5, cout).astype(np.float32)), to_dev((0.1 * r.standard_normal(cout)).astype(np.float32))
        rm1, rv1 = to_dev(np.zeros(cout, np.float32)), to_dev(np.ones(cout, np.float32))
        rm2, rv2 = rm1.clone(), rv1.clone()
        wp = vh.pack_conv_weight(wt)
        z1 = vh.conv2d_fwd(x, wp, None, None, cout, k, k, stride, pad, False)
        mean1, inv1, sc1, bi1 = vh.bn_train_fwd_stats(z1, gamma, beta, rm1, rv1, 0.1, 1e-5)
        z2, mean2, inv2, sc2, bi2 = vh.conv2d_fwd_bnstats(x, wp, cout, k, k, stride, pad, gamma, beta, rm2, rv2, 0.1, 1e-5)
        assert torch.equal(z1, z2)
        for a, b in ((mean1, mean2), (inv1, inv2), (sc1, sc2), (bi1, bi2), (rm1, rm2), (rv1, rv2)):
            np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol=1e-7)
        y = vh.scale_bias_act(z1, sc1, bi1, None, True)
        dy = to_dev(r.standard_normal(tuple(z1.shape)).astype(np.float32))
        dz1, _, dg1, db1 = vh.bn_train_bwd(dy, y, z1, gamma, mean1, inv1)
        dz2, dg2, db2 = vh.bn_train_bwd_relu(dy, sc1, bi1, z1, gamma, mean1, inv1)
        assert torch.equal(dz1, dz2) and torch.equal(dg1, dg2) and torch.equal(db1, db2)
    # 3-channel stem (its own tile configuration): statistics must cover every row block the chosen tile writes
    x = vh.nchw_to_nhwc(to_dev(r.standard_normal((3, 3, 64, 48)).astype(np.float32)), 4)
    wt = to_dev((r.standard_normal((64, 3, 7, 7)) / 12).astype(np.float32))
    gamma, beta = to_dev(np.ones(64, np.float32)), to_dev(np.zeros(64, np.float32))
    wp = vh.pack_conv_weight(wt)
    z1 = vh.conv2d_fwd(x, wp, None, None, 64, 7, 7, 2, 3, False)
    s1 = vh.bn_train_fwd_stats(z1, gamma, beta, None, None, 0.1, 1e-5)
    out = vh.conv2d_fwd_bnstats(x, wp, 64, 7, 7, 2, 3, gamma, beta, None, None, 0.1, 1e-5)
    assert torch.equal(out[0], z1)
    for a, b in zip(s1, out[1:]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol=1e-7)
    # transposed conv: four phases contribute to the same channel statistics
    x = to_dev(r.standard_normal((2, 5, 3, 64)).astype(np.float32))
    wt = to_dev((r.standard_normal((64, 128, 4, 4)) / 16).astype(np.float32))
    gamma, beta = to_dev(np.ones(128, np.float32)), to_dev(np.zeros(128, np.float32))
    wp = vh.pack_deconv_weight(wt)
    z1 = vh.deconv4x4s2_fwd(x, wp, None, None, 128, False)
    s1 = vh.bn_train_fwd_stats(z1, gamma, beta, None, None, 0.1, 1e-5)
    out = vh.deconv4x4s2_fwd_bnstats(x, wp, 128, gamma, beta, None, None, 0.1, 1e-5)
    assert torch.equal(out[0], z1)
    for a, b in zip(s1, out[1:]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("tag,hm_hw,in_hw,sigma", [("a", (64, 48), (256, 192), 2.0), ("b", (96, 72), (384, 288), 1.5)])
def test_gaussian_targets_vs_reference_golden(vh, tag, hm_hw, in_hw, sigma):
    """SimpleTransform._target_generator on the device: patch positions, clipping and weights exact (the support of
    every map is identical), values within float32 exp rounding."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "targets.npz"))
    joints, vis = synth.target_joints(6, hm_hw, in_hw, seed=41)
    t, w = vh.gaussian_targets(to_dev(joints), to_dev(vis), hm_hw, in_hw, sigma)
    t, w = t.cpu().numpy(), w.cpu().numpy()
    want_t, want_w = g[f"{tag}_target"], g[f"{tag}_weight"]
    assert np.array_equal(w.reshape(6, 17), want_w.reshape(6, 17))
    assert np.array_equal(t != 0, want_t != 0)
    np.testing.assert_allclose(t, want_t, rtol=2e-6, atol=1e-12)


def test_simplepose_step_well_conditioned_batch(vh):
    """The same fine-tune step on a larger batch (12 crops of 128x96: 576 samples per channel in the last stage instead
    of 96), every one of the 170 parameter tensors against float64 autograd, with torch's own fp32 run of the same graph
    as the yardstick: measured on MI355X ours / torch-fp32 = 4.6e-3 / 4.5e-3 (median L2) and 1.48e-2 / 1.50e-2 (worst) —
    the distance to float64 is the fp32 conditioning of a 53-BatchNorm network with synthetic weights, and the HIP path
    sits exactly where fp32 PyTorch sits."""
    from alphapose.models import builder
    from alphapose.utils.config import edict
    from oracle import nets
    hw = (128, 96)
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": list(hw), "HEATMAP_SIZE": [hw[0] // 4, hw[1] // 4]})
    m = builder.build_sppe(cfg, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).train()
    n = 12
    x = synth.crops(n, hw=hw)
    labels, masks = synth.gaussian_targets(n, seed=5, hw=(hw[0] // 4, hw[1] // 4))
    out = m(to_dev(x))
    (0.5 * torch.nn.MSELoss()(out.mul(to_dev(masks)), to_dev(labels).mul(to_dev(masks)))).backward()
    grads = {}
    for dt in (torch.float64, torch.float32):
        ref = nets.SimplePoseRef(50)
        ref.load_state_dict(synth.state_dict_for(ref), strict=True)
        ref = ref.to(dt).train()
        o = ref(torch.from_numpy(x).to(dt))
        (0.5 * torch.nn.MSELoss()(o * torch.from_numpy(masks).to(dt), torch.from_numpy(labels).to(dt) * torch.from_numpy(masks).to(dt))).backward()
        grads[dt] = {k: p.grad.double().numpy() for k, p in ref.named_parameters()}
    ours, ref32 = [], []
    for k, p in m.named_parameters():
        ex = grads[torch.float64][k]
        den = max(np.linalg.norm(ex), 1e-30)
        ours.append(float(np.linalg.norm(p.grad.cpu().double().numpy() - ex) / den))
        ref32.append(float(np.linalg.norm(grads[torch.float32][k] - ex) / den))
    record("train_step_well_conditioned", tensors=len(ours), ours_worst_l2=max(ours), ours_median_l2=float(np.median(ours)),
           torch_fp32_worst_l2=max(ref32), torch_fp32_median_l2=float(np.median(ref32)))
    assert np.median(ours) < 1.5 * np.median(ref32) + 1e-4 and max(ours) < 2 * max(ref32) + 1e-3


@pytest.mark.parametrize("name", ["simplepose", "fastpose"])
def test_one_launch_weight_repack_plan_equals_per_tensor_packs(vh, name, monkeypatch):
    """vatl_pack_weights_multi behind vatl_hip.PackPlan: three fine-tune steps with AdamW (weights change every step, so every
    step re-packs) end with the same bits whether the packed copies come from the plan's single launch (recorded in step 1,
    replayed in steps 2-3) or from one launch per tensor; a weight written behind the plan's back is noticed, not used stale."""
    from active_learning.optim import AdamW
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    cfgs = {"simplepose": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
            "fastpose": {"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}}
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    x = to_dev(synth.crops(4, seed=3))
    labels, masks = synth.gaussian_targets(4, seed=4)
    labels, masks = to_dev(labels), to_dev(masks)

    def run(planned):
        monkeypatch.setattr(hip_train, "_USE_PACK_PLAN", planned)
        torch.manual_seed(21)
        m = builder.build_sppe(edict(cfgs[name]), preset_cfg=preset).to(dev()).train()
        opt = AdamW(params=[{"params": m.parameters(), "lr": 1e-3}], weight_decay=0.1)
        tr = hip_train.trainer_for(m)
        losses = []
        for step in range(3):
            with torch.no_grad():
                out = tr.forward(x)
                loss, dout = vh.masked_mse_fwd_bwd(out, labels, masks)
                for p_, g_ in tr.backward(dout).items():
                    p_.grad = g_
            opt.step()
            losses.append(float(loss))
        return m, tr, losses

    m1, tr1, l1 = run(True)
    plan = tr1.__dict__["_pack_plan"]
    assert plan.ready and not plan.stale and len(plan.jobs) >= 50 and plan.total_blocks > 0
    m0, _, l0 = run(False)
    assert l0 == l1 and l0[2] != l0[0]
    for (k, a), (_, b) in zip(m1.state_dict().items(), m0.state_dict().items()):
        assert torch.equal(a, b), k
    # a write the plan did not see (version counter bumped after begin()): the lookup refuses the kept copy
    monkeypatch.setattr(hip_train, "_USE_PACK_PLAN", True)
    with torch.no_grad():
        out = tr1.forward(x)
        w = next(p_ for n_, p_ in m1.named_parameters() if p_.dim() == 4 and p_.shape[2] == 3)
        w.mul_(1.0)                                              # in-place: same values, new version
        _, dout = vh.masked_mse_fwd_bwd(out, labels, masks)
        tr1.backward(dout)
    assert plan.stale


def test_fine_tune_step_is_bitwise_reproducible(vh):
    """No atomics anywhere in the step (weight gradients are reduced over pixel splits in a fixed order, BN statistics
    and the loss through ordered partials): two runs of forward + backward give identical bits for every gradient."""
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    m = builder.build_sppe(cfg, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).train()
    x = to_dev(synth.crops(5))
    labels, masks = synth.gaussian_targets(5, seed=11)
    labels, masks = to_dev(labels), to_dev(masks)
    runs = []
    for _ in range(2):
        tr = hip_train.trainer_for(m)
        with torch.no_grad():
            out = tr.forward(x)
            loss, dout = vh.masked_mse_fwd_bwd(out, labels, masks)
            grads = tr.backward(dout)
        runs.append((out.clone(), float(loss), {k: v.clone() for k, v in grads.items()}))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    for p in runs[0][2]:
        assert torch.equal(runs[0][2][p], runs[1][2][p])


@pytest.mark.parametrize("shape", [(3, 8, 12, 64), (32, 24, 18, 256), (2, 7, 10, 96), (5, 6, 8, 2048), (4, 96, 72, 64), (1, 9, 9, 32)])
def test_per_item_pixel_reductions_small_batches(shape):
    """Global average pool and the SE gate gradient through both kernel families (block per (item, channel group) for small
    batches with many pixels; thread per (item, channel) otherwise) against float64."""
    import vatl_hip as vh
    n, h, w, c = shape
    g = torch.Generator(device="cuda").manual_seed(n * 1000 + c)
    x = torch.randn(shape, device="cuda", generator=g)
    got = vh.gap_fwd(x).cpu().double()
    want = x.double().mean(dim=(1, 2)).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
    dy, y, u = (torch.randn(shape, device="cuda", generator=g) for _ in range(3))
    gate = torch.randn((n, c), device="cuda", generator=g)
    dg = vh.se_bwd_gate(dy, y, u, gate).cpu().double()
    sg = torch.sigmoid(gate.double())
    want = ((dy.double() * (y > 0) * u.double()).sum(dim=(1, 2)) * sg * (1 - sg)).cpu()
    np.testing.assert_allclose(dg.numpy(), want.numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(vh.gap_fwd(x), vh.gap_fwd(x))                       # fixed summation order


@pytest.mark.parametrize("name", ["simplepose", "fastpose"])
def test_b16_default_init_step_vs_reference_and_float64(vh, name):
    """tests/golden/wellcond_step.npz: the reference's own fine-tune step (ActiveLearning.py:662-673) at B = 16 with default
    initialisation and untouched BatchNorm statistics, plus the same step in float64.  Loss, output and running statistics
    must equal the reference's to fp32 rounding.  Gradients: the reference's OWN fp32 step is 1e-6 (head) .. 2.4e-2 (trunk)
    away from the float64 gradients on this fixture — the first BatchNorm layers behind the freshly initialised head
    amplify rounding noise 1000x, whatever computes them — so every sampled tensor (every 10th parameter + the head) must be
    as close to float64 as the reference is: ours <= 1.5 x reference + 1e-4 (L2, relative).  In the head / last deconv that is
    a 1e-4 .. 3e-3 bound; the block-by-block check below is what discriminates in the trunk."""
    import os
    from tests.conftest import GOLDEN
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    g = np.load(os.path.join(GOLDEN, "wellcond_step.npz"))
    cfgs = {"simplepose": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
            "fastpose": {"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}}
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(int(g["seed"]))
    m = builder.build_sppe(edict(cfgs[name]), preset_cfg=preset)                 # default init under the fixture's seed = the fixture's weights
    params = list(m.named_parameters())
    np.testing.assert_allclose(sum(float(p.detach().double().abs().sum()) for _, p in params), float(g[f"{name}_wsum"]), rtol=1e-12)
    assert np.array_equal(params[0][1].detach().reshape(-1)[:8].numpy(), g[f"{name}_w0"])
    m = m.to(dev()).train()
    B = int(g["batch"])
    x = to_dev(synth.crops(B, seed=91))
    labels, masks = synth.gaussian_targets(B, seed=92)
    labels, masks = to_dev(labels), to_dev(masks)
    tr, arena = hip_train.trainer_for(m), hip_train.arena_for(m)
    with torch.no_grad():
        out = tr.forward(x)
        loss, dout = vh.masked_mse_fwd_bwd(out, labels, masks)
        arena.begin()
        tr.backward(dout, arena=arena)
        arena.finish()
        arena.attach()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss), float(g[f"{name}_loss"]), rtol=1e-5)
    np.testing.assert_allclose(float(out.abs().mean()), float(g[f"{name}_out_absmean"]), rtol=1e-5)
    np.testing.assert_allclose(m.preact.bn1.running_mean.cpu().numpy(), g[f"{name}_bn1_running_mean"], rtol=1e-4, atol=1e-6)
    named = dict(m.named_parameters())
    keys = [k.split("::", 1)[1] for k in g.files if k.startswith(f"{name}_grad_idx::")]
    assert len(keys) >= 18
    bad, tight = [], 0
    for key in keys:
        idx = torch.from_numpy(g[f"{name}_grad_idx::{key}"]).to(dev())
        got = named[key].grad.reshape(-1)[idx].cpu().numpy().astype(np.float64)
        ref, exact = g[f"{name}_grad_val::{key}"].astype(np.float64), g[f"{name}_grad_f64::{key}"]
        n = max(np.linalg.norm(exact), 1e-30)
        ours, theirs = float(np.linalg.norm(got - exact) / n), float(np.linalg.norm(ref - exact) / n)
        nrm = float(named[key].grad.double().norm()) / max(float(g[f"{name}_grad_norm::{key}"]), 1e-30)
        record("b16_grad", model=name, key=key, ours_vs_f64=ours, reference_vs_f64=theirs, norm_ratio=nrm)
        tight += theirs < 1e-4
        # the norm may differ from the reference's by 0.5 % plus half of the reference's own distance from float64 on that tensor (two
        # fp32 evaluations of a tensor that is 2 % away from float64 do not share their norm to 0.5 %)
        if not (ours <= 1.5 * theirs + 1e-4 and abs(nrm - 1) < 5e-3 + 0.5 * theirs):
            bad.append((key, ours, theirs, nrm))
    assert not bad, bad
    assert tight >= 2                                        # the head tensors are pinned at the 1e-4 level


def test_every_block_gradient_vs_float64_on_real_activations(vh):
    """The discriminating whole-network check.  The float64 oracle graph (bit-identical to the reference in fp32,
    tests/test_oracle_golden.py) runs one fine-tune step of SimplePose-R50 at B = 4 and records, for every bottleneck block and
    every deconv layer, its input and the gradient arriving at its output: real activations, real gradient statistics.
    Every block of the HIP trainer is then run ALONE on those tensors (train-mode forward, backward), so rounding noise cannot
    accumulate across blocks, and compared with the same block in float64 — evaluated with the ReLU decisions the HIP forward
    actually took.  (About one pre-activation per million lands on the other side of zero in fp32 than in float64, in torch's
    fp32 run as often as in ours; such a flip is a legitimate fp32 outcome but moves that block's gradients by 1e-4 .. 5e-3, the
    size of the defects this test is after.  With the masks pinned the comparison is rounding-only.)  Bounds, L2 relative to
    float64: block output 2e-6, input gradient and EVERY parameter gradient 2e-5 — a 1 % error in any layer's data or weight
    gradient (wrong tap, wrong parity launch, a dropped split, a mis-scaled BatchNorm term) is 500x over the bound.  The fused
    path (data gradient carrying the consumer's ReLU mask + BatchNorm-backward partial sums, weight gradients on the side stream)
    is held to the same float64 tensors."""
    import copy
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    from oracle import nets
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(77)
    m = builder.build_sppe(cfg, preset_cfg=preset)
    with torch.no_grad():                                    # non-trivial BatchNorm affine parameters
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.2)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    B = 4
    x = torch.from_numpy(synth.crops(B, seed=93))
    labels, masks = synth.gaussian_targets(B, seed=94)
    ref = nets.SimplePoseRef(50)
    ref.load_state_dict(sd, strict=True)
    ref = ref.double().train()
    taps = {}

    def tap_in(name):
        def fn(_m, inp, out):
            taps[name + "_in"] = inp[0]
        return fn

    def tap_out(name):
        def fn(_m, inp, out):
            out.retain_grad()
            taps[name + "_out"] = out
        return fn
    blocks64 = [(f"layer{s}.{i}", b) for s in (1, 2, 3, 4) for i, b in enumerate(getattr(ref.preact, f"layer{s}"))]
    hooks = []
    for name, b in blocks64:
        hooks += [b.register_forward_hook(tap_in(name)), b.register_forward_hook(tap_out(name))]
    d64 = ref.deconv_layers
    deconv64 = [("deconv0", 0), ("deconv1", 3), ("deconv2", 6)]
    for name, k in deconv64:
        hooks += [d64[k].register_forward_hook(tap_in(name)), d64[k + 2].register_forward_hook(tap_out(name))]
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    o64 = ref(x.double())
    l64 = 0.5 * torch.nn.MSELoss()(o64 * torch.from_numpy(masks).double(), torch.from_numpy(labels).double() * torch.from_numpy(masks).double())
    l64.backward()
    for h in hooks:
        h.remove()

    m = m.to(dev()).train()
    tr = hip_train.trainer_for(m)
    named = {p: k for k, p in m.named_parameters()}

    def nhwc(t):
        return to_dev(t.detach().permute(0, 2, 3, 1).contiguous().float().numpy())

    def nchw64(t_nhwc):
        return t_nhwc.detach().permute(0, 3, 1, 2).double().cpu()

    def l2(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

    def bneck64(b64, xin, dy, m1, m2, m3):
        """The oracle block in float64 with the ReLU decisions (m1, m2, m3) pinned; -> y, dx, {parameter name: gradient}."""
        blk = copy.deepcopy(b64)
        blk.zero_grad()
        xi = xin.detach().clone().requires_grad_()
        with torch.enable_grad():
            y = blk.bn1(blk.conv1(xi)) * m1
            y = blk.bn2(blk.conv2(y)) * m2
            y = blk.bn3(blk.conv3(y))
            s_ = xi if blk.downsample is None else blk.downsample(xi)
            out = (y + s_) * m3
            out.backward(dy)
        return out.detach(), xi.grad, {k: p.grad for k, p in blk.named_parameters()}

    def deconv64_ref(mods, xin, dy, mk):
        mods = [copy.deepcopy(mm) for mm in mods]
        for mm in mods:
            mm.zero_grad()
        xi = xin.detach().clone().requires_grad_()
        with torch.enable_grad():
            out = mods[1](mods[0](xi)) * mk
            out.backward(dy)
        g = {f"{i}.{k}": p.grad for i, mm in enumerate(mods) for k, p in mm.named_parameters()}
        return out.detach(), xi.grad, g

    failures, worst = [], {"y": 0.0, "dx": 0.0, "dw": 0.0}

    def compare(tag, prefix, y, y64, dx, dx64, grads, g64):
        ey, ed = l2(nchw64(y), y64), l2(nchw64(dx), dx64)
        worst["y"], worst["dx"] = max(worst["y"], ey), max(worst["dx"], ed)
        record("block_check", block=tag, y=ey, dx=ed)
        if not (ey < 2e-6 and ed < 2e-5):
            failures.append((tag, "y / dx", ey, ed))
        assert len(grads) == len(g64), (tag, len(grads), len(g64))
        for p, gr in grads.items():
            k = named[p]
            e = l2(gr.cpu().numpy(), g64[k[len(prefix):]].numpy())
            worst["dw"] = max(worst["dw"], e)
            record("block_grad", block=tag, key=k, l2=e)
            if not e < 2e-5:
                failures.append((tag, k, e))

    def run_block(blk, xin_dev, dy_dev, **kw):
        y = blk.forward(xin_dev)
        mk = [nchw64(blk.c2.saved[0] > 0), nchw64(blk.c3.saved[0] > 0), nchw64(y > 0)]
        grads = hip_train._Grads()
        dx = blk.backward(dy_dev, grads, **kw)
        hip_train._side.join()
        hip_train._flush_batch_counters()
        return y, mk, dx, grads

    with torch.no_grad():
        for (name, b64), blk in zip(blocks64, tr.blocks):
            xin, dy = taps[name + "_in"].detach(), taps[name + "_out"].grad
            y, mk, dx, grads = run_block(blk, nhwc(xin), nhwc(dy))
            y64, dx64, g64 = bneck64(b64, xin, dy, *mk)
            compare(name, f"preact.{name}.", y, y64, dx, dx64, grads, g64)
        for (name, k), dc in zip(deconv64, tr.deconvs):
            xin, dy = taps[name + "_in"].detach(), taps[name + "_out"].grad
            y = dc.forward(nhwc(xin))
            mk = nchw64(y > 0)
            grads = hip_train._Grads()
            dx = dc.backward(nhwc(dy), grads)
            hip_train._side.join()
            hip_train._flush_batch_counters()
            y64, dx64, g64 = deconv64_ref([d64[k], d64[k + 1]], xin, dy, mk)
            compare(name, "deconv_layers.", y, y64, dx, dx64, grads, {f"{k + int(a.split('.')[0])}.{a.split('.', 1)[1]}": v for a, v in g64.items()})
        # fused path: block k's input gradient arrives masked by block k-1's output ReLU, with the BatchNorm-backward partial sums of
        # block k-1's last layer taken in the same epilogue; block k-1 then starts from those sums
        fused = 0
        for k in range(len(tr.blocks) - 1, 0, -1):
            (nb, b64b), (na, b64a) = blocks64[k], blocks64[k - 1]
            below, blk = tr.blocks[k - 1], tr.blocks[k]
            xa, dya = taps[na + "_in"].detach(), taps[na + "_out"].grad
            xb, dyb = taps[nb + "_in"].detach(), taps[nb + "_out"].grad
            ya = below.forward(nhwc(xa))
            mka = [nchw64(below.c2.saved[0] > 0), nchw64(below.c3.saved[0] > 0), nchw64(ya > 0)]
            spec = below.out_spec()
            if spec is None:
                below.backward(nhwc(dya), hip_train._Grads())
                hip_train._side.join()
                continue
            fused += 1
            # block k forward on the float64 run's input (== block k-1's float64 output up to fp32 rounding): its masked input
            # gradient must be the float64 dL/dx_k times block k-1's OWN output mask
            yb = blk.forward(nhwc(xb))
            mkb = [nchw64(blk.c2.saved[0] > 0), nchw64(blk.c3.saved[0] > 0), nchw64(yb > 0)]
            gb, ga = hip_train._Grads(), hip_train._Grads()
            g_masked = blk.backward(nhwc(dyb), gb, consumer=spec)
            dxa = below.backward(g_masked, ga, pre=spec)
            hip_train._side.join()
            hip_train._flush_batch_counters()
            _, dxb64, _ = bneck64(b64b, xb, dyb, *mkb)
            e = l2(nchw64(g_masked), dxb64 * mka[2])
            if not e < 2e-5:
                failures.append(("fused g", nb, e))
            # block k-1 driven by that gradient: the float64 block gets the unmasked float64 dL/dx_k as its upstream gradient
            _, dxa64, ga64 = bneck64(b64a, xa, dxb64, *mka)
            e = l2(nchw64(dxa), dxa64)
            worst["dx"] = max(worst["dx"], e)
            if not e < 2e-5:
                failures.append(("fused dx", na, e))
            for p, gr in ga.items():
                kk = named[p]
                e = l2(gr.cpu().numpy(), ga64[kk[len(f"preact.{na}."):]].numpy())
                worst["dw"] = max(worst["dw"], e)
                record("block_grad_fused", block=na, key=kk, l2=e)
                if not e < 2e-5:
                    failures.append(("fused", na, kk, e))
        assert fused >= 8
    record("block_check_summary", **worst)
    assert not failures, failures


def _unit_vs_float64(tag, prefix, unit, ref64, xs64, dys64, named, hip_train, failures, worst, bound_y=2e-6, bound_g=2e-5):
    """One trainer unit (an SE bottleneck, an HRNet fusion module ...) run ALONE on the float64 oracle's real inputs and upstream
    gradients, against the same oracle module in float64 evaluated with the ReLU decisions the fp32 forward took (the trainers
    report every ReLU mask in execution order through hip_train._relu_tap; the oracle replays them in ITS execution order — the
    shapes must agree at every step, which also checks that both sides run the same graph).  xs64 / dys64: lists of NCHW tensors."""
    import copy
    from unittest import mock

    def nhwc(t):
        return to_dev(t.detach().permute(0, 2, 3, 1).contiguous().float().numpy())

    def nchw64(t):
        return t.detach().permute(0, 3, 1, 2).double().cpu() if t.dim() == 4 else t.detach().double().cpu()

    def l2(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

    multi = isinstance(xs64, (list, tuple))
    xin = [nhwc(t) for t in xs64] if multi else nhwc(xs64)
    hip_train._relu_tap = []
    try:
        ys = unit.forward(xin)
    finally:
        masks, hip_train._relu_tap = hip_train._relu_tap, None
    masks = [nchw64(mk) for mk in masks]
    grads = hip_train._Grads()
    dxs = unit.backward([nhwc(t) for t in dys64] if multi else nhwc(dys64), grads)
    hip_train._side.join()
    hip_train._flush_batch_counters()
    blk = copy.deepcopy(ref64)
    blk.zero_grad()
    xi = [t.detach().clone().requires_grad_() for t in xs64] if multi else xs64.detach().clone().requires_grad_()
    replay = list(masks)

    def pinned_relu(t, inplace=False):
        mk = replay.pop(0)
        assert tuple(mk.shape) == tuple(t.shape), (tag, tuple(mk.shape), tuple(t.shape))
        return t * mk
    with torch.enable_grad(), mock.patch.object(torch.nn.functional, "relu", pinned_relu):
        out = blk(xi)
        if multi:
            torch.autograd.backward(list(out), [d.detach() for d in dys64])
        else:
            out.backward(dys64.detach())
    assert not replay, (tag, len(replay))
    outs, ys_l, dx_l, xi_l = (list(out), list(ys), list(dxs), xi) if multi else ([out], [ys], [dxs], [xi])
    for k, (y, y64, dx, x64) in enumerate(zip(ys_l, outs, dx_l, xi_l)):
        ey, ed = l2(nchw64(y), y64.detach()), l2(nchw64(dx), x64.grad)
        worst["y"], worst["dx"] = max(worst["y"], ey), max(worst["dx"], ed)
        record("unit_check", unit=tag, branch=k, y=ey, dx=ed)
        if not (ey < bound_y and ed < bound_g):
            failures.append((tag, k, "y / dx", ey, ed))
    g64 = {k: p.grad for k, p in blk.named_parameters()}
    assert len(grads) == len(g64), (tag, len(grads), len(g64))
    for p, gr in grads.items():
        k = named[p]
        e = l2(gr.cpu().numpy(), g64[k[len(prefix):]].numpy())
        worst["dw"] = max(worst["dw"], e)
        record("unit_grad", unit=tag, key=k, l2=e)
        if not e < bound_g:
            failures.append((tag, k, e))


def _randomise_bn(m):
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5); mod.bias.normal_(0, 0.2)


def test_fastpose_se_blocks_vs_float64_on_real_activations(vh):
    """SE_Resnet.py:110-137 / SE_module.py:20-24 backward (gate, squeeze, fc pair, projection) — the path the whole-network
    5e-2 band alone covered: each of the four SE bottlenecks of FastPose-R50 run alone on the float64 oracle's real activations and
    gradients, block output 2e-6, input gradient and every parameter gradient (conv, BatchNorm, both fc layers) 2e-5."""
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    from oracle import nets
    cfg = edict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(78)
    m = builder.build_sppe(cfg, preset_cfg=preset)
    _randomise_bn(m)
    B = 2
    x = torch.from_numpy(synth.crops(B, seed=95))
    labels, masks = synth.gaussian_targets(B, seed=96)
    ref = nets.FastPoseRef(50)
    ref.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, strict=True)
    ref = ref.double().train()
    taps, hooks = {}, []
    se64 = [(f"layer{s}.0", getattr(ref.preact, f"layer{s}")[0]) for s in (1, 2, 3, 4)]
    def se_hook(name):
        def fn(_m, inp, out):
            out.retain_grad()
            taps[name] = (inp[0], out)
        return fn
    for name, b in se64:
        hooks.append(b.register_forward_hook(se_hook(name)))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    o64 = ref(x.double())
    mk = torch.from_numpy(masks).double()
    (0.5 * torch.nn.MSELoss()(o64 * mk, torch.from_numpy(labels).double() * mk)).backward()
    for h in hooks:
        h.remove()
    m = m.to(dev()).train()
    tr = hip_train.trainer_for(m)
    named = {p: k for k, p in m.named_parameters()}
    se_units = [b for b in tr.blocks if isinstance(b, hip_train._SEBottleneckT)]
    assert len(se_units) == 4
    failures, worst = [], {"y": 0.0, "dx": 0.0, "dw": 0.0}

    class _One:                                             # single-tensor units behind the list-free calling convention
        def __init__(self, u): self.u = u
        def forward(self, x_): return self.u.forward(x_)
        def backward(self, dy, grads): return self.u.backward(dy, grads)
    with torch.no_grad():
        for (name, b64), unit in zip(se64, se_units):
            xin, out = taps[name]
            _unit_vs_float64(name, f"preact.{name}.", _One(unit), b64, xin.detach(), out.grad, named, hip_train, failures, worst)
    record("se_block_check_summary", **worst)
    assert not failures, failures


def test_hrnet_fusion_modules_vs_float64_on_real_activations(vh):
    """hrnet.py:242-260 (multi-resolution fusion: strided 3x3 chains down, 1x1 + nearest up-sampling, sum, ReLU) and the basic
    blocks of every branch, backward included: the first HighResolutionModule of stages 2, 3 and 4 of HRNet-W32 (2, 3 and 4
    branches) run alone on the float64 oracle's real activations and gradients; every output, every input gradient and every
    parameter gradient of the module within 2e-6 / 2e-5 of float64."""
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    from oracle import nets
    from tests.test_gpu_conv import HRNET_CFG
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(79)
    m = builder.build_sppe(edict(HRNET_CFG), preset_cfg=preset)
    _randomise_bn(m)
    B = 2
    x = torch.from_numpy(synth.crops(B, seed=97))
    labels, masks = synth.gaussian_targets(B, seed=98)
    ref = nets.HRNetRef()
    ref.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, strict=True)
    ref = ref.double().train()
    taps, hooks = {}, []
    mods64 = [(f"stage{s}.0", getattr(ref, f"stage{s}")[0]) for s in (2, 3, 4)]

    def hook(name):
        def fn(_m, inp, out):
            for o in out:
                o.retain_grad()
            taps[name] = (list(inp[0]), list(out))
        return fn
    for name, b in mods64:
        hooks.append(b.register_forward_hook(hook(name)))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    o64 = ref(x.double())
    mk = torch.from_numpy(masks).double()
    (0.5 * torch.nn.MSELoss()(o64 * mk, torch.from_numpy(labels).double() * mk)).backward()
    for h in hooks:
        h.remove()
    m = m.to(dev()).train()
    tr = hip_train.trainer_for(m)
    named = {p: k for k, p in m.named_parameters()}
    failures, worst = [], {"y": 0.0, "dx": 0.0, "dw": 0.0}
    with torch.no_grad():
        for (name, b64), (_, mods) in zip(mods64, tr.stages):
            xs, outs = taps[name]
            assert len(outs) == len(xs)                      # multi-scale output: every branch is fused
            _unit_vs_float64(name, name + ".", mods[0], b64, [t.detach() for t in xs], [o.grad for o in outs], named, hip_train, failures, worst)
    record("hrnet_module_check_summary", **worst)
    assert not failures, failures
