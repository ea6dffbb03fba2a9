"""Implicit-GEMM conv / deconv / pooling kernels vs a float64 torch-CPU reference,
and the whole SimplePose-R50 forward vs the reference-generated golden heat-maps."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import synth
from tests.gpu_util import dev, record, rel_err, to_dev

pytestmark = pytest.mark.gpu

TOL = 2e-5      # max|err| / max|ref| per layer (fp32 accumulate, K <= 4608); network-level bar is 1e-4


@pytest.fixture(scope="module")
def vh():
    import vatl_hip
    vatl_hip.lib()
    return vatl_hip


def _nhwc(x):
    return np.ascontiguousarray(np.transpose(x, (0, 2, 3, 1)))


def test_layout_roundtrip(vh):
    r = np.random.RandomState(0)
    x = r.standard_normal((3, 3, 10, 7)).astype(np.float32)
    y = vh.nchw_to_nhwc(to_dev(x), 4).cpu().numpy()
    assert np.array_equal(y[..., :3], _nhwc(x)) and (y[..., 3] == 0).all()
    x = r.standard_normal((2, 8, 5, 6)).astype(np.float32)
    y = vh.nchw_to_nhwc(to_dev(x))
    assert np.array_equal(y.cpu().numpy(), _nhwc(x))
    assert np.array_equal(vh.nhwc_to_nchw(y).cpu().numpy(), x)


CONV_CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad, relu, residual, bias, nchw
    ("1x1_64_256_res", 2, 64, 48, 64, 256, 1, 1, 0, True, True, False, False),
    ("1x1_256_64", 1, 64, 48, 256, 64, 1, 1, 0, True, False, False, False),
    ("3x3_64_64", 2, 64, 48, 64, 64, 3, 1, 1, True, False, False, False),
    ("3x3s2_128_128", 2, 64, 48, 128, 128, 3, 2, 1, True, False, False, False),
    ("1x1s2_256_512", 2, 64, 48, 256, 512, 1, 2, 0, False, False, False, False),
    ("3x3_512_512_tiny", 3, 8, 6, 512, 512, 3, 1, 1, True, False, False, False),
    ("1x1_2048_512_oddM", 1, 8, 6, 2048, 512, 1, 1, 0, True, False, False, False),
    ("1x1_cout96", 1, 9, 7, 64, 96, 1, 1, 0, False, True, True, False),
    ("3x3_oddHW", 1, 13, 11, 32, 40, 3, 1, 1, True, True, False, False),
    ("3x3s2_oddHW", 2, 13, 11, 32, 64, 3, 2, 1, False, False, True, False),
    ("head_256_17_nchw", 2, 64, 48, 256, 17, 1, 1, 0, False, False, True, True),
    ("stem_7x7", 2, 256, 192, 3, 64, 7, 2, 3, True, False, False, False),
    ("stem_7x7_odd", 1, 37, 29, 3, 64, 7, 2, 3, True, False, False, False),
    ("3x3_32_17_nchw", 1, 16, 12, 128, 17, 3, 1, 1, False, False, True, True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_fwd(vh, case):
    name, n, h, w, cin, cout, k, stride, pad, relu, use_res, use_bias, nchw = case
    import zlib; r = np.random.RandomState(zlib.crc32(name.encode()) % 2 ** 31)
    x = r.standard_normal((n, cin, h, w)).astype(np.float32)
    wt = (r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    gamma, beta = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32) * 0.1
    mean, var = r.standard_normal(cout).astype(np.float32) * 0.1, r.uniform(0.5, 1.5, cout).astype(np.float32)
    cb = r.standard_normal(cout).astype(np.float32) if use_bias else None
    ref = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None if cb is None else torch.from_numpy(cb).double(),
                   stride, pad)
    ref = F.batch_norm(ref, torch.from_numpy(mean).double(), torch.from_numpy(var).double(), torch.from_numpy(gamma).double(),
                       torch.from_numpy(beta).double(), False, 0.0, 1e-5)
    res = None
    if use_res:
        res = r.standard_normal(tuple(ref.shape)).astype(np.float32)
        ref = ref + torch.from_numpy(res).double()
    if relu:
        ref = ref.relu()
    ref = ref.numpy()

    xd = vh.nchw_to_nhwc(to_dev(x), 4 if cin == 3 else cin)
    wp = vh.pack_conv_weight(to_dev(wt))
    scale, bias = vh.bn_fold(to_dev(gamma), to_dev(beta), to_dev(mean), to_dev(var), 1e-5, None if cb is None else to_dev(cb))
    resd = None
    if use_res:
        resd = to_dev(res) if nchw else to_dev(_nhwc(res))
    y = vh.conv2d_fwd(xd, wp, scale, bias, cout, k, k, stride, pad, relu, residual=resd, out_nchw=nchw)
    torch.cuda.synchronize()
    got = y.cpu().numpy() if nchw else np.transpose(y.cpu().numpy(), (0, 3, 1, 2))
    assert got.shape == ref.shape
    e = rel_err(got, ref)
    record("conv2d_" + name, rel=e)
    assert e < TOL, (name, e)


DECONV_CASES = [("2048_256", 2, 8, 6, 2048, 256), ("256_256", 1, 16, 12, 256, 256), ("odd", 1, 5, 3, 64, 48), ("256_256_big", 1, 32, 24, 256, 256)]


@pytest.mark.parametrize("case", DECONV_CASES, ids=[c[0] for c in DECONV_CASES])
def test_deconv4x4s2_fwd(vh, case):
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
    y = vh.deconv4x4s2_fwd(vh.nchw_to_nhwc(to_dev(x)), vh.pack_deconv_weight(to_dev(wt)), scale, bias, cout, True)
    got = np.transpose(y.cpu().numpy(), (0, 3, 1, 2))
    e = rel_err(got, ref)
    record("deconv_" + name, rel=e)
    assert got.shape == ref.shape and e < TOL, (name, e)


def test_maxpool_and_gap(vh):
    r = np.random.RandomState(3)
    for (n, c, h, w) in ((2, 64, 128, 96), (1, 8, 7, 5)):
        x = r.standard_normal((n, c, h, w)).astype(np.float32)
        ref = F.max_pool2d(torch.from_numpy(x), 3, 2, 1).numpy()
        got = np.transpose(vh.maxpool3x3s2_fwd(vh.nchw_to_nhwc(to_dev(x))).cpu().numpy(), (0, 3, 1, 2))
        assert np.array_equal(got, ref)                                                   # max is exact
    x = r.standard_normal((3, 2048, 8, 6)).astype(np.float32)
    got = vh.gap_fwd(vh.nchw_to_nhwc(to_dev(x))).cpu().numpy()
    np.testing.assert_allclose(got, x.reshape(3, 2048, -1).mean(2), rtol=1e-5, atol=1e-6)


def _build_simplepose():
    from alphapose.models import builder
    from alphapose.utils.config import edict
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    m = builder.build_sppe(cfg, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    return m.to(dev()).eval()


def test_simplepose_forward_vs_reference_golden(vh, golden_simplepose):
    g = golden_simplepose
    m = _build_simplepose()
    x = to_dev(synth.crops(int(g["batch"])))
    with torch.no_grad():
        hm = m(x)
        emb = m.get_embedding(x)
    torch.cuda.synchronize()
    hm, emb = hm.cpu().numpy(), emb.cpu().numpy()
    e = rel_err(hm, g["heatmaps"])
    record("simplepose_r50_heatmaps", rel=e, emb_rel=rel_err(emb, g["embedding"]))
    assert hm.shape == (2, 17, 64, 48)
    assert e < 1e-4                                                                       # north_star: heat-maps within 1e-4 rel fp32
    assert np.array_equal(hm.reshape(2, 17, -1).argmax(2), g["heatmaps"].reshape(2, 17, -1).argmax(2))   # integer peaks bit-exact
    assert rel_err(emb, g["embedding"]) < 1e-4
    # staged taps (debug aid: which stage diverges first)
    from alphapose.models import hip_engine
    plan = hip_engine._plan_for(m, dev())
    with torch.no_grad():
        feat = plan.features(x)
    v = np.transpose(feat.cpu().numpy(), (0, 3, 1, 2)).reshape(-1)[g["tap_layer4_idx"]]
    assert rel_err(v, g["tap_layer4_val"]) < 1e-4


def test_simplepose_batch_invariance(vh):
    """A crop's heat-map must not depend on its batch position (THC de-duplication)."""
    m = _build_simplepose()
    x = to_dev(synth.crops(3, seed=5))
    with torch.no_grad():
        full = m(x)
        solo = torch.cat([m(x[i:i + 1]) for i in range(3)], 0)
    assert torch.equal(full, solo)


def test_cpu_input_fails_loudly(vh):
    m = _build_simplepose()
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 256, 192))


HRNET_CFG = {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1, "PRETRAINED_LAYERS": ["*"],
             "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
             "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
             "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}}


def _build(cfg):
    from alphapose.models import builder
    from alphapose.utils.config import edict
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    m = builder.build_sppe(edict(cfg), preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    return m.to(dev()).eval()


def test_pixelshuffle_se_fuse_kernels(vh):
    r = np.random.RandomState(7)
    x = r.standard_normal((2, 64, 5, 3)).astype(np.float32)
    got = np.transpose(vh.pixelshuffle2_fwd(vh.nchw_to_nhwc(to_dev(x))).cpu().numpy(), (0, 3, 1, 2))
    assert np.array_equal(got, F.pixel_shuffle(torch.from_numpy(x), 2).numpy())
    x = r.standard_normal((2, 32, 4, 6)).astype(np.float32); res = r.standard_normal((2, 32, 4, 6)).astype(np.float32)
    g = r.standard_normal((2, 32)).astype(np.float32)
    got = np.transpose(vh.se_scale_add_relu(vh.nchw_to_nhwc(to_dev(x)), to_dev(g), vh.nchw_to_nhwc(to_dev(res))).cpu().numpy(), (0, 3, 1, 2))
    want = np.maximum(x * (1 / (1 + np.exp(-g)))[:, :, None, None] + res, 0)
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    base = r.standard_normal((2, 8, 8, 16)).astype(np.float32)
    z1 = r.standard_normal((2, 8, 4, 8)).astype(np.float32); z2 = r.standard_normal((2, 8, 2, 4)).astype(np.float32)
    got = np.transpose(vh.fuse_upsample_add(vh.nchw_to_nhwc(to_dev(base)), [(vh.nchw_to_nhwc(to_dev(z1)), 1), (vh.nchw_to_nhwc(to_dev(z2)), 2)], True).cpu().numpy(), (0, 3, 1, 2))
    want = np.maximum(base + np.repeat(np.repeat(z1, 2, 2), 2, 3) + np.repeat(np.repeat(z2, 4, 2), 4, 3), 0)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", ["fastpose", "hrnet"])
def test_fastpose_hrnet_forward_vs_reference_golden(vh, name):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_hrnet.npz"))
    m = _build({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50} if name == "fastpose" else HRNET_CFG)
    x = to_dev(synth.crops(int(g["batch"])))
    with torch.no_grad():
        hm = m(x).cpu().numpy()
    ref = g[f"{name}_heatmaps"]
    e = rel_err(hm, ref)
    record(f"{name}_heatmaps", rel=e)
    assert hm.shape == ref.shape and e < 1e-4
    assert np.array_equal(hm.reshape(2, 17, -1).argmax(2), ref.reshape(2, 17, -1).argmax(2))
    if name == "fastpose":
        with torch.no_grad():
            emb = m.get_embedding(x).cpu().numpy()
        assert rel_err(emb, g["fastpose_embedding"]) < 1e-4


def test_fastpose_r152_384_forward_and_step_vs_reference_golden(vh):
    """BASELINE.json config 5 (FastPose-R152, 384x288 crops, 96x72 heat-maps): forward against the reference's golden
    output; one B = 2 fine-tune step against the reference's loss and sampled gradients."""
    import os
    from alphapose.models import builder
    from alphapose.utils.config import edict
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fastpose_r152_384.npz"))
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [384, 288], "HEATMAP_SIZE": [96, 72]})
    m = builder.build_sppe(edict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152}), preset_cfg=preset)
    assert list(m.state_dict().keys()) == list(g["keys"])
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).eval()
    x = to_dev(synth.crops(1, hw=(384, 288)))
    with torch.no_grad():
        hm = m(x).cpu().numpy()
        emb = m.get_embedding(x).cpu().numpy()
    e = rel_err(hm, g["heatmaps"])
    record("fastpose_r152_384_heatmaps", rel=e, emb_rel=rel_err(emb, g["embedding"]))
    assert hm.shape == (1, 17, 96, 72) and e < 1e-4
    assert np.array_equal(hm.reshape(1, 17, -1).argmax(2), g["heatmaps"].reshape(1, 17, -1).argmax(2))
    assert rel_err(emb, g["embedding"]) < 1e-4
    m.train()
    labels, masks = synth.gaussian_targets(2, seed=11, hw=(96, 72))
    labels, masks = to_dev(labels), to_dev(masks)
    out = m(to_dev(synth.crops(2, hw=(384, 288))))
    loss = 0.5 * torch.nn.MSELoss()(out.mul(masks), labels.mul(masks))
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(g["train_loss"]), rtol=1e-4)
    named = dict(m.named_parameters())
    for key in [k[10:] for k in g.files if k.startswith("grad_idx::")]:
        idx = g[f"grad_idx::{key}"]
        got = named[key].grad.reshape(-1)[torch.from_numpy(idx).to(dev())].cpu().numpy()
        ref = g[f"grad_val::{key}"]
        l2 = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
        record("fastpose_r152_384_grad", key=key, l2_vs_reference_fp32=l2)
        # two fp32 evaluations of an ill-conditioned B = 2 step (152 layers): same band as the R50 step tests
        assert l2 < (1e-4 if key.startswith("conv_out") else 5e-2), (key, l2)


def test_chunked_batches_are_bit_identical(vh):
    """Batches above hip_engine.MAX_CHUNK are processed in chunks (32-bit buffer offsets): same bits as one launch
    sequence, through forward(), forward_into() and get_embedding(), including a ragged last chunk."""
    from alphapose.models import hip_engine
    m = _build_simplepose()
    x = to_dev(synth.crops(20))                               # > 16 crops: module calls this large never use the small-batch split-K
    with torch.no_grad():
        whole, emb = m(x), m.get_embedding(x)
    old = hip_engine.MAX_CHUNK
    try:
        hip_engine.MAX_CHUNK = 6                               # 6 + 6 + 6 + 2
        with torch.no_grad():
            parts, emb_parts = m(x), m.get_embedding(x)
            buf = torch.empty_like(whole)
            hip_engine.forward_into(m, x, buf)
    finally:
        hip_engine.MAX_CHUNK = old
    assert torch.equal(parts, whole) and torch.equal(buf, whole) and torch.equal(emb_parts, emb)


def test_forward_with_embedding_is_bit_identical(vh):
    """One trunk pass serving heat-maps and get_embedding (the reference runs the trunk twice)."""
    from alphapose.models import hip_engine
    m = _build_simplepose()
    x = to_dev(synth.crops(3))
    with torch.no_grad():                                      # (the stream entry points: `m(x)` with <= 16 crops runs with split-K)
        hm, emb = hip_engine.forward_into(m, x, torch.empty((3, 17, 64, 48), device=x.device)), m.get_embedding(x)
    out, e = torch.empty_like(hm), torch.empty_like(emb)
    hip_engine.forward_with_embedding(m, x, out, e)
    assert torch.equal(out, hm) and torch.equal(e, emb)
    mf = _build({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50})
    with torch.no_grad():
        hm, emb = hip_engine.forward_into(mf, x, torch.empty((3, 17, 64, 48), device=x.device)), mf.get_embedding(x)
    out, e = torch.empty_like(hm), torch.empty_like(emb)
    hip_engine.forward_with_embedding(mf, x, out, e)
    assert torch.equal(out, hm) and torch.equal(e, emb)
    with pytest.raises(vh.VatlError):
        hip_engine.forward_with_embedding(_build(HRNET_CFG), x, out, e)


@pytest.mark.parametrize("stride", [1, 2])
def test_conv1x1_dual_matches_two_launches(vh, stride):
    """conv3 + projection shortcut as one dual-source GEMM vs the two-launch form and vs float64."""
    r = np.random.RandomState(30 + stride)
    n, h2, w2, c1, c2, cout = 3, 12, 10, 64, 96, 256
    ho, wo = (h2 - 1) // stride + 1, (w2 - 1) // stride + 1
    a = r.standard_normal((n, ho, wo, c1)).astype(np.float32)
    x = r.standard_normal((n, h2, w2, c2)).astype(np.float32)
    w1 = (r.standard_normal((cout, c1, 1, 1)) / 8).astype(np.float32); w2_ = (r.standard_normal((cout, c2, 1, 1)) / 10).astype(np.float32)
    s1, b1 = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32)
    s2, b2 = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32)
    wp, bias = vh.pack_conv1x1_dual_weight(to_dev(w1), to_dev(s1), to_dev(b1), to_dev(w2_), to_dev(s2), to_dev(b2))
    got = vh.conv1x1_dual_fwd(to_dev(a), to_dev(x), wp, bias, cout, stride, True).cpu().numpy()
    skip = vh.conv2d_fwd(to_dev(x), vh.pack_conv_weight(to_dev(w2_)), to_dev(s2), to_dev(b2), cout, 1, 1, stride, 0, False)
    two = vh.conv2d_fwd(to_dev(a), vh.pack_conv_weight(to_dev(w1)), to_dev(s1), to_dev(b1), cout, 1, 1, 1, 0, True, residual=skip).cpu().numpy()
    xs = x[:, ::stride, ::stride].astype(np.float64)
    want = np.maximum((a.astype(np.float64) @ w1[:, :, 0, 0].T.astype(np.float64)) * s1 + b1 + (xs @ w2_[:, :, 0, 0].T.astype(np.float64)) * s2 + b2, 0)
    e_two, e_64 = rel_err(got, two), rel_err(got, want)
    record(f"conv1x1_dual_s{stride}", vs_two_launches=e_two, vs_fp64=e_64, two_launches_vs_fp64=rel_err(two, want))
    assert got.shape == (n, ho, wo, cout) and e_two < 2e-6 and e_64 < 2e-6


def test_bottleneck_chain_vs_two_launches_and_float64(vh):
    """vatl_bottleneck_chain_fwd: conv3 + bn3 + skip + relu of one bottleneck and conv1 + bn1 + relu of the next in one launch (csrc/bottleneck_chain.hip).
    t must have the bits of the tiled implicit GEMM (it IS the network's activation and the next skip connection); y1 sums K in four fixed pieces: fp32
    rounding of the two-launch value, checked against float64 too.  Ragged pixel counts (partial 32-row tiles, a single pixel), the first GEMM alone,
    no skip, and a crop's bits independent of the batch it sits in."""
    r = np.random.RandomState(77)
    w3 = (r.standard_normal((256, 64, 1, 1)) / 8).astype(np.float32); w1 = (r.standard_normal((64, 256, 1, 1)) / 16).astype(np.float32)
    s3, b3 = r.uniform(0.5, 1.5, 256).astype(np.float32), r.standard_normal(256).astype(np.float32)
    s1, b1 = r.uniform(0.5, 1.5, 64).astype(np.float32), r.standard_normal(64).astype(np.float32)
    w3p, w1p = vh.pack_conv_weight(to_dev(w3)), vh.pack_conv_weight(to_dev(w1))
    s3d, b3d, s1d, b1d = to_dev(s3), to_dev(b3), to_dev(s1), to_dev(b1)
    assert vh.bottleneck_chain_supported(64, 256, 64, 1024 * 64 * 48) and vh.bottleneck_chain_supported(64, 256, 0, 100)
    assert not vh.bottleneck_chain_supported(128, 512, 128, 100) and not vh.bottleneck_chain_supported(64, 256, 128, 100)
    assert not vh.bottleneck_chain_supported(64, 256, 64, 1 << 22)
    for n, h, w in ((2, 64, 48), (1, 5, 7), (1, 1, 1), (3, 16, 12), (70, 8, 6), (1, 1, 33)):
        a = r.standard_normal((n, h, w, 64)).astype(np.float32)
        skip = r.standard_normal((n, h, w, 256)).astype(np.float32)
        ad, sd = to_dev(a), to_dev(skip)
        t_two = vh.conv2d_fwd(ad, w3p, s3d, b3d, 256, 1, 1, 1, 0, True, residual=sd)
        y_two = vh.conv2d_fwd(t_two, w1p, s1d, b1d, 64, 1, 1, 1, 0, True)
        t, y1 = vh.bottleneck_chain_fwd(ad, w3p, s3d, b3d, sd, w1p, s1d, b1d)
        assert torch.equal(t, t_two), (n, h, w)
        t64 = np.maximum((a.astype(np.float64) @ w3[:, :, 0, 0].T.astype(np.float64)) * s3 + b3 + skip, 0)
        # the float64 value of the second conv ON the fp32 tensor t the launch stored (its own input), so the bound is one GEMM's rounding
        y64 = np.maximum((t.cpu().numpy().astype(np.float64) @ w1[:, :, 0, 0].T.astype(np.float64)) * s1 + b1, 0)
        e_two, e_64, e_t = rel_err(y1.cpu().numpy(), y_two.cpu().numpy()), rel_err(y1.cpu().numpy(), y64), rel_err(t.cpu().numpy(), t64)
        record(f"bottleneck_chain_{n}x{h}x{w}", y1_vs_two_launches=e_two, y1_vs_fp64=e_64, two_launches_vs_fp64=rel_err(y_two.cpu().numpy(), y64), t_vs_fp64=e_t)
        assert e_two < 2e-6 and e_64 < 2e-6 and e_t < 2e-6
        t1, none = vh.bottleneck_chain_fwd(ad, w3p, s3d, b3d, sd)                       # first GEMM alone
        assert none is None and torch.equal(t1, t_two)
        t0, y0 = vh.bottleneck_chain_fwd(ad, w3p, None, None, None, w1p, None, None)   # no folded BN, no skip
        t0_two = vh.conv2d_fwd(ad, w3p, None, None, 256, 1, 1, 1, 0, True)
        assert torch.equal(t0, t0_two) and rel_err(y0.cpu().numpy(), vh.conv2d_fwd(t0_two, w1p, None, None, 64, 1, 1, 1, 0, True).cpu().numpy()) < 2e-6
        if n > 1:                                                                       # bits of a crop do not depend on its batch (32-row tiles straddle crops)
            ts, ys = vh.bottleneck_chain_fwd(ad[1:2].contiguous(), w3p, s3d, b3d, sd[1:2].contiguous(), w1p, s1d, b1d)
            assert torch.equal(ts, t[1:2]) and torch.equal(ys, y1[1:2])
    with pytest.raises(vh.VatlError):
        vh.bottleneck_chain_fwd(to_dev(r.standard_normal((1, 2, 2, 32)).astype(np.float32)), w3p[:, :, :, :32].contiguous(), s3d, b3d, None)


def test_conv1x1_rows_has_the_bits_of_the_tiled_kernels(vh):
    """vatl_conv1x1_rows_fwd (csrc/conv1x1_rows.hip): K = 128 (and K = 256) 1x1 layers as a row-streaming GEMM — same k order as the tiled implicit GEMM, so the output must be
    BIT-IDENTICAL to vatl_conv2d_fwd (with / without scale, bias, residual, ReLU; N = 128 .. 1024; ragged pixel counts, a single pixel, more tiles than one walk) and,
    in the two-source form, to vatl_conv1x1_dual_fwd; plus a float64 check and the batch independence of a crop's bits."""
    r = np.random.RandomState(91)
    assert vh.conv1x1_rows_supported(128, 0, 512, 1024 * 32 * 24) and vh.conv1x1_rows_supported(64, 64, 256, 1024 * 64 * 48)
    assert vh.conv1x1_rows_supported(256, 0, 1024, 1024 * 16 * 12) and not vh.conv1x1_rows_supported(512, 0, 512, 100)
    assert not vh.conv1x1_rows_supported(128, 0, 192, 100) and not vh.conv1x1_rows_supported(128, 0, 512, 1 << 21) and not vh.conv1x1_rows_supported(256, 64, 512, 100)
    for cout, kin in ((128, 128), (512, 128), (1024, 128), (1024, 256), (256, 256), (512, 256)):
        w = (r.standard_normal((cout, kin, 1, 1)) / 11).astype(np.float32)
        sc, bi = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32)
        wp, scd, bid = vh.pack_conv_weight(to_dev(w)), to_dev(sc), to_dev(bi)
        for n, h, wd in ((3, 32, 24), (1, 5, 7), (1, 1, 1), (70, 8, 6), (1, 1, 33), (40, 32, 24)):
            a = r.standard_normal((n, h, wd, kin)).astype(np.float32)
            res = r.standard_normal((n, h, wd, cout)).astype(np.float32)
            ad, rd = to_dev(a), to_dev(res)
            for relu, use_res, use_sb in ((True, True, True), (False, False, True), (True, False, False)):
                s_, b_ = (scd, bid) if use_sb else (None, None)
                want = vh.conv2d_fwd(ad, wp, s_, b_, cout, 1, 1, 1, 0, relu, residual=rd if use_res else None)
                got = vh.conv1x1_rows_fwd(ad, wp, s_, b_, cout, relu, residual=rd if use_res else None)
                assert torch.equal(got, want), (cout, kin, n, h, wd, relu, use_res, use_sb)
            if cout in (512, 1024) and n == 3:
                ref = np.maximum((a.astype(np.float64) @ w[:, :, 0, 0].T.astype(np.float64)) * sc + bi + res, 0)
                e = rel_err(vh.conv1x1_rows_fwd(ad, wp, scd, bid, cout, True, residual=rd).cpu().numpy(), ref)
                record(f"conv1x1_rows_k{kin}_n{cout}", vs_fp64=e)
                assert e < 2e-6
                solo = vh.conv1x1_rows_fwd(ad[1:2].contiguous(), wp, scd, bid, cout, True, residual=rd[1:2].contiguous())
                assert torch.equal(solo, vh.conv1x1_rows_fwd(ad, wp, scd, bid, cout, True, residual=rd)[1:2])
    # two sources: conv3 + projection shortcut of a stage's first block
    cout = 256
    w1 = (r.standard_normal((cout, 64, 1, 1)) / 8).astype(np.float32); w2 = (r.standard_normal((cout, 64, 1, 1)) / 8).astype(np.float32)
    s1, b1 = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32)
    s2, b2 = r.uniform(0.5, 1.5, cout).astype(np.float32), r.standard_normal(cout).astype(np.float32)
    wp, bias = vh.pack_conv1x1_dual_weight(to_dev(w1), to_dev(s1), to_dev(b1), to_dev(w2), to_dev(s2), to_dev(b2))
    for n, h, wd in ((2, 64, 48), (1, 3, 5), (9, 16, 12)):
        a, x = to_dev(r.standard_normal((n, h, wd, 64)).astype(np.float32)), to_dev(r.standard_normal((n, h, wd, 64)).astype(np.float32))
        want = vh.conv1x1_dual_fwd(a, x, wp, bias, cout, 1, True)
        got = vh.conv1x1_rows_fwd(a, wp, None, bias, cout, True, x2=x)
        assert torch.equal(got, want), (n, h, wd)
    with pytest.raises(vh.VatlError):
        vh.conv1x1_rows_fwd(to_dev(r.standard_normal((1, 2, 2, 64)).astype(np.float32)), wp, None, bias, cout, True)


def test_winograd_c32_vs_float64_and_the_general_route(vh):
    """vatl_conv3x3_winograd_c32_fwd (csrc/winograd_c32.hip): 32 -> 32 channel 3x3 / stride 1 / pad 1 layers with wave-private tiles on 16x16x4 MFMAs.  Against
    float64 (same bound as the general Winograd kernel: the two differ only in the grouping of the channel sums) on sizes with whole and partial 8 x 2 tile
    patches, with / without folded BN, skip connection and ReLU; a crop's bits independent of its batch and position; the plan takes it for HRNet's first branch."""
    r = np.random.RandomState(57)
    w = (r.standard_normal((32, 32, 3, 3)) * (2.0 / 288) ** 0.5).astype(np.float32)
    sc, bi = r.uniform(0.5, 1.5, 32).astype(np.float32), r.standard_normal(32).astype(np.float32)
    u32, u = vh.pack_winograd_c32_weight(to_dev(w)), vh.pack_winograd_weight(to_dev(w))
    assert vh.conv3x3_winograd_c32_supported(1024, 64, 48, 32, 32) and not vh.conv3x3_winograd_c32_supported(4, 7, 8, 32, 32)
    assert not vh.conv3x3_winograd_c32_supported(4, 8, 8, 64, 64)
    wt = torch.from_numpy(w).double()
    for n, h, wd in ((2, 64, 48), (3, 10, 22), (1, 2, 2), (5, 4, 34), (1, 96, 72), (9, 8, 6)):
        x = r.standard_normal((n, h, wd, 32)).astype(np.float32)
        res = r.standard_normal((n, h, wd, 32)).astype(np.float32)
        conv = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), wt, padding=1).permute(0, 2, 3, 1).numpy()
        xd, rd = to_dev(x), to_dev(res)
        for relu, use_res, use_sb in ((True, True, True), (False, False, False), (True, False, True)):
            want = conv * (sc if use_sb else 1.0) + (bi if use_sb else 0.0) + (res if use_res else 0.0)
            want = np.maximum(want, 0) if relu else want
            s_, b_ = (to_dev(sc), to_dev(bi)) if use_sb else (None, None)
            got = vh.conv3x3_winograd_c32_fwd(xd, u32, s_, b_, relu, residual=rd if use_res else None)
            gen = vh.conv3x3_winograd_fwd(xd, u, s_, b_, 32, relu, residual=rd if use_res else None)
            e, eg = rel_err(got.cpu().numpy(), want), rel_err(gen.cpu().numpy(), want)
            assert e < 2e-6 and rel_err(got.cpu().numpy(), gen.cpu().numpy()) < 2e-6, (n, h, wd, relu, use_res, use_sb, e, eg)
        record(f"winograd_c32_{n}x{h}x{wd}", vs_fp64=e, general_route_vs_fp64=eg)
        if n > 1:
            full = vh.conv3x3_winograd_c32_fwd(xd, u32, to_dev(sc), to_dev(bi), True, residual=rd)
            solo = vh.conv3x3_winograd_c32_fwd(xd[1:2].contiguous(), u32, to_dev(sc), to_dev(bi), True, residual=rd[1:2].contiguous())
            assert torch.equal(solo, full[1:2])
            assert torch.equal(vh.conv3x3_winograd_c32_fwd(xd, u32, to_dev(sc), to_dev(bi), True, residual=rd), full)      # same bits again
    with pytest.raises(vh.VatlError):
        vh.conv3x3_winograd_c32_fwd(to_dev(r.standard_normal((1, 7, 8, 32)).astype(np.float32)), u32, None, None, False)


def test_chained_bottlenecks_in_the_plans(vh, monkeypatch):
    """The stream route of SimplePose-R50 / HRNet-W32 takes the chained launch where an identity-shortcut bottleneck is followed by a 256 -> 64 conv1
    (R50 stage 1: one linked + one first-GEMM-only launch; HRNet layer1: two linked + one alone), and the pass equals the separate launches to fp32 rounding."""
    from alphapose.models import hip_engine
    torch.manual_seed(5)
    calls, real = [], vh.bottleneck_chain_fwd
    monkeypatch.setattr(vh, "bottleneck_chain_fwd", lambda *a, **k: (calls.append(len(a) > 5 and a[5] is not None), real(*a, **k))[1])
    for cfg, calls_want in (({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "NUM_DECONV_FILTERS": [256, 256, 256]}, (1, 1)), (HRNET_CFG, (2, 1))):
        m = _build(cfg)
        x = torch.randn((20, 3, 256, 192), device=dev())
        calls.clear()
        with torch.no_grad():
            on = hip_engine.forward_into(m, x, torch.empty((20, 17, 64, 48), device=dev())).clone()
            n_on = (sum(calls), len(calls) - sum(calls))
            monkeypatch.setattr(hip_engine, "FUSE_CHAIN", False)
            off = hip_engine.forward_into(m, x, torch.empty((20, 17, 64, 48), device=dev())).clone()
            monkeypatch.setattr(hip_engine, "FUSE_CHAIN", True)
        assert n_on == calls_want and len(calls) == sum(calls_want), (cfg["TYPE"], n_on)
        e = rel_err(on.cpu().numpy(), off.cpu().numpy())
        record(f"chained_bottlenecks_{cfg['TYPE']}", on_vs_off=e)
        assert e < 1e-5 and torch.equal(on.flatten(2).argmax(-1), off.flatten(2).argmax(-1))


def test_ablation_knobs_are_refused(vh):
    import os
    assert os.environ.get("VATL_ALLOW_ABLATION") != "1"
    for knob, value in ((0, 10), (0, 13), (4, 1)):
        with pytest.raises(vh.VatlError):
            vh.tune_set(knob, value)
    vh.tune_set(0, 4); vh.tune_set(5, 0)


def test_launches_from_several_host_threads(vh):
    """SURVEY.md §8b threading: DataParallel drives one host thread per replica — the library keeps no per-call
    global state, so concurrent launches on different streams give the single-thread results bit for bit."""
    import threading
    m = _build_simplepose()
    xs = [to_dev(synth.crops(2, seed=100 + i)) for i in range(4)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
    torch.cuda.synchronize()
    got, errs = [None] * 4, []

    def work(i):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s), torch.no_grad():
                for _ in range(3):
                    got[i] = m(xs[i])
            s.synchronize()
        except Exception as e:                                       # pragma: no cover
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_every_tile_and_schedule_gives_identical_bits(vh):
    """The tuning knobs change the block tile (64 / 128 rows), the k-loop schedule (two-phase, interleaved, distance-2
    prefetch, LDS-DMA) and the tile order — never the per-output reduction order: results are bit-identical."""
    r = np.random.RandomState(77)
    for (n, h, w, cin, cout, k, stride) in ((3, 16, 12, 128, 128, 3, 1), (5, 8, 6, 256, 512, 1, 1), (2, 16, 12, 64, 64, 3, 2)):
        x = to_dev(r.standard_normal((n, h, w, cin)).astype(np.float32))
        wt = to_dev((r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
        sc, bi = to_dev(r.uniform(0.5, 1.5, cout).astype(np.float32)), to_dev(r.standard_normal(cout).astype(np.float32))
        ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
        res = to_dev(r.standard_normal((n, ho, wo, cout)).astype(np.float32))
        wp = vh.pack_conv_weight(wt)
        outs = []
        try:
            for var in (4, 0, 2, 5):
                for bm in (0, 64, 128):
                    for order in (0, 1):
                        vh.tune_set(0, var); vh.tune_set(5, bm); vh.tune_set(1, order)
                        outs.append(vh.conv2d_fwd(x, wp, sc, bi, cout, k, k, stride, k // 2, True, residual=res).clone())
        finally:
            vh.tune_set(0, 4); vh.tune_set(5, 0); vh.tune_set(1, 0)
        for o in outs[1:]:
            assert torch.equal(o, outs[0])


def test_persistent_1x1_kernel_is_bit_identical(vh):
    """The persistent GEMM kernel for short-K 1x1 layers (several tiles per block, next tile prefetched during the
    epilogue) against the one-tile-per-block kernel: identical bits, with residual / ReLU / tails / several runs."""
    r = np.random.RandomState(91)
    for (n, h, w, cin, cout, res) in ((40, 16, 12, 64, 256, True), (9, 16, 12, 128, 512, True), (33, 8, 6, 256, 1024, False), (7, 13, 11, 512, 128, False)):
        x = to_dev(r.standard_normal((n, h, w, cin)).astype(np.float32))
        wp = vh.pack_conv_weight(to_dev((r.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)))
        sc, bi = to_dev(r.uniform(0.5, 1.5, cout).astype(np.float32)), to_dev(r.standard_normal(cout).astype(np.float32))
        rs = to_dev(r.standard_normal((n, h, w, cout)).astype(np.float32)) if res else None
        try:
            vh.tune_set(5, 128)                                        # whole 128-row tiles on both paths
            vh.tune_set(7, 0); a = vh.conv2d_fwd(x, wp, sc, bi, cout, 1, 1, 1, 0, True, residual=rs).clone()
            vh.tune_set(7, 4); b = vh.conv2d_fwd(x, wp, sc, bi, cout, 1, 1, 1, 0, True, residual=rs).clone()
        finally:
            vh.tune_set(5, 0); vh.tune_set(7, 1)
        assert torch.equal(a, b), (n, cin, cout)


def test_large_384x288_batch_is_chunked_below_the_offset_limit(vh):
    """610 crops of 384x288 would put 2^30+ elements into the stem output: the engine cuts the batch (2 x 305) and the
    results equal those of the same crops in a small batch."""
    from alphapose.models import builder, hip_engine
    from alphapose.utils.config import edict
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [384, 288], "HEATMAP_SIZE": [96, 72]})
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    m = builder.build_sppe(cfg, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m = m.to(dev()).eval()
    n = 610
    assert hip_engine._chunk_limit((384, 288)) < n
    g = torch.Generator(device=dev()); g.manual_seed(3)
    x = torch.rand((n, 3, 384, 288), device=dev(), generator=g) - 0.45
    with torch.no_grad():
        hm = m(x)
        pick = torch.tensor([0, 304, 305, 609], device=dev())
        small = hip_engine.forward_into(m, x[pick], torch.empty((4, 17, 96, 72), device=dev()))   # (the stream entry point: `m(4 crops)` runs with split-K)
    assert hm.shape == (n, 17, 96, 72) and torch.equal(hm[pick], small)


def test_optin_splitk_matches_unsplit_kernel(vh):
    """Opt-in split-K (small batches): same results as the unsplit kernel to fp32 rounding, through plain / residual /
    ReLU / NCHW-head / transposed-conv launches and a whole batch-2 forward; off again afterwards."""
    r = np.random.RandomState(55)
    try:
        cases = []
        for (n, h, w, cin, cout, k, res, nchw) in ((2, 8, 6, 2048, 512, 1, False, False), (1, 8, 6, 512, 512, 3, True, False),
                                                   (2, 16, 12, 1024, 256, 1, True, False), (1, 16, 12, 512, 17, 1, False, True)):
            x = to_dev(r.standard_normal((n, h, w, cin)).astype(np.float32))
            wp = vh.pack_conv_weight(to_dev((r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)))
            sc, bi = to_dev(r.uniform(0.5, 1.5, cout).astype(np.float32)), to_dev(r.standard_normal(cout).astype(np.float32))
            rs = to_dev(r.standard_normal((n, h, w, cout)).astype(np.float32)) if res else None
            cases.append(lambda x=x, wp=wp, sc=sc, bi=bi, cout=cout, k=k, rs=rs, nchw=nchw: vh.conv2d_fwd(x, wp, sc, bi, cout, k, k, 1, k // 2, not nchw, residual=rs, out_nchw=nchw))
        xd = to_dev(r.standard_normal((2, 8, 6, 2048)).astype(np.float32))
        wd = vh.pack_deconv_weight(to_dev((r.standard_normal((2048, 256, 4, 4)) / 90).astype(np.float32)))
        sd, bd = to_dev(np.ones(256, np.float32)), to_dev(np.zeros(256, np.float32))
        cases.append(lambda: vh.deconv4x4s2_fwd(xd, wd, sd, bd, 256, True))
        m = _build_simplepose()
        xm = to_dev(synth.crops(2))
        cases.append(lambda: m(xm))
        with torch.no_grad():
            base = [c().clone() for c in cases]
            vh.enable_splitk(32)
            split = [c().clone() for c in cases]
        for a, b in zip(base, split):
            assert a.shape == b.shape and rel_err(b.cpu().numpy(), a.cpu().numpy()) < 1e-5      # fp32 rounding of a different summation order (K up to 8192)
        assert not torch.equal(base[0], split[0])                 # the split path really ran (different summation order)
    finally:
        vh.enable_splitk(0)
    with torch.no_grad():
        again = cases[0]()
    assert torch.equal(again, base[0])


@pytest.mark.parametrize("n,h,w", [(3, 64, 48), (2, 16, 24), (5, 8, 16), (1, 32, 8), (7, 32, 24)])
def test_halo_tile_kernel_is_bit_identical_to_the_implicit_gemm(vh, n, h, w):
    """csrc/conv3x3_halo.hip serves the 32-channel 3x3 layers (HRNet's high-resolution branch) from a persistent block with the
    filter and a halo tile in LDS; its reduction order is the implicit GEMM's, so the two kernels must agree bit for bit — with
    and without BatchNorm affine, residual and ReLU, on both patch shapes (8x16 and 16x8) and on image borders."""
    rng = np.random.RandomState(n * 1000 + h)
    x = to_dev(rng.randn(n, h, w, 32).astype(np.float32))
    wt = to_dev((rng.randn(32, 32, 3, 3) * 0.1).astype(np.float32))
    wp = vh.pack_conv_weight(wt)
    sc, bi = to_dev(rng.rand(32).astype(np.float32) + 0.5), to_dev(rng.randn(32).astype(np.float32))
    res = to_dev(rng.randn(n, h, w, 32).astype(np.float32))
    for scale, bias, residual, relu in ((sc, bi, res, True), (None, None, None, False), (sc, bi, None, True), (None, bi, res, False)):
        vh.tune_set(8, 1)
        a = vh.conv2d_fwd(x, wp, scale, bias, 32, 3, 3, 1, 1, relu, residual=residual)
        vh.tune_set(8, 0)
        try:
            b = vh.conv2d_fwd(x, wp, scale, bias, 32, 3, 3, 1, 1, relu, residual=residual)
        finally:
            vh.tune_set(8, 1)
        assert torch.equal(a, b)
    # and against float64 (the generic kernel's own bound)
    want = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1).numpy()
    got = vh.conv2d_fwd(x, wp, None, None, 32, 3, 3, 1, 1, False).cpu().numpy()
    assert rel_err(got, want) < 2e-6


def test_small_module_calls_use_splitk_and_the_stream_path_never_does(vh):
    """`model(x)` with <= 16 crops (BASELINE.json configs[0] shape, scripts/poseestimator_eval.py) runs with split-K for the duration
    of the call: same heat-maps to fp32 rounding (1e-5), arg-max identical.  The evaluation stream entry point keeps
    batch-size-independent bits before and after such a call (the workspace is registered for the call only)."""
    from alphapose.models import hip_engine
    m = _build_simplepose()
    x = to_dev(synth.crops(24, seed=8))
    with torch.no_grad():
        big = torch.empty((24, 17, 64, 48), device=x.device)
        hip_engine.forward_into(m, x, big)
        small = m(x[:4])                                           # split-K inside
        solo, sixteen = m(x[1:2]), m(x[:16])                       # ... with a cut that depends on the layer geometry only:
        after = torch.empty((4, 17, 64, 48), device=x.device)
        hip_engine.forward_into(m, x[:4], after)
        plain = m(x)                                               # 24 crops: unsplit
    assert torch.equal(after, big[:4]) and torch.equal(plain, big)
    assert not torch.equal(small, big[:4])                         # a different summation order really ran
    assert torch.equal(solo[0], small[1]) and torch.equal(sixteen[:4], small)   # a crop's bits do not depend on its (small) batch
    assert rel_err(small.cpu().numpy(), big[:4].cpu().numpy()) < 1e-5
    assert torch.equal(small.flatten(2).argmax(2), big[:4].flatten(2).argmax(2))
    record("auto_splitk", rel=rel_err(small.cpu().numpy(), big[:4].cpu().numpy()))


def test_streamk_route_matches_the_whole_tile_kernels(vh):
    """vatl_set_streamk_workspace_thread: launches that would leave most block slots idle run as 768 persistent blocks sharing the
    (tile, k-tile) units; tiles split between blocks are completed through workspace slabs in a fixed order.  Same results as
    the whole-tile kernels to fp32 rounding (different summation order over K), bitwise reproducible run to run, ragged M,
    residual / ReLU / BatchNorm-statistics epilogues included; nothing changes once the scope is closed."""
    r = np.random.RandomState(77)
    dev_ = dev()
    cases = []
    for (n, h, w, cin, cout, k, res) in ((120, 8, 6, 512, 512, 3, True), (24, 16, 12, 256, 256, 3, False), (120, 8, 6, 2048, 512, 1, False),
                                         (31, 12, 9, 512, 512, 3, True), (32, 24, 18, 256, 1024, 1, True)):
        x = to_dev(r.standard_normal((n, h, w, cin)).astype(np.float32))
        wp = vh.pack_conv_weight(to_dev((r.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)))
        sc, bi = to_dev(r.uniform(0.5, 1.5, cout).astype(np.float32)), to_dev(r.standard_normal(cout).astype(np.float32))
        rs = to_dev(r.standard_normal((n, h, w, cout)).astype(np.float32)) if res else None
        cases.append((x, wp, sc, bi, cout, k, rs))
    with torch.no_grad():
        base = [vh.conv2d_fwd(x, wp, sc, bi, cout, k, k, 1, k // 2, True, residual=rs) for (x, wp, sc, bi, cout, k, rs) in cases]
        with vh.streamk_scope(dev_, force=True):
            sk1 = [vh.conv2d_fwd(x, wp, sc, bi, cout, k, k, 1, k // 2, True, residual=rs) for (x, wp, sc, bi, cout, k, rs) in cases]
            sk2 = [vh.conv2d_fwd(x, wp, sc, bi, cout, k, k, 1, k // 2, True, residual=rs) for (x, wp, sc, bi, cout, k, rs) in cases]
            # training forward: z + BatchNorm batch statistics from the epilogue of the completing blocks
            x, wp, _, _, cout, k, _ = cases[0]
            g1, b1 = torch.ones(cout, device=dev_), torch.zeros(cout, device=dev_)
            zs = vh.conv2d_fwd_bnstats(x, wp, cout, k, k, 1, k // 2, g1, b1, torch.zeros(cout, device=dev_), torch.ones(cout, device=dev_), 0.1, 1e-5)
        zb = vh.conv2d_fwd_bnstats(x, wp, cout, k, k, 1, k // 2, g1, b1, torch.zeros(cout, device=dev_), torch.ones(cout, device=dev_), 0.1, 1e-5)
        after = vh.conv2d_fwd(*cases[0][:4], cases[0][4], cases[0][5], cases[0][5], 1, cases[0][5] // 2, True, residual=cases[0][6])
    took = 0
    for a, b, c in zip(base, sk1, sk2):
        assert torch.equal(b, c)                                  # reproducible
        e = rel_err(b.cpu().numpy(), a.cpu().numpy())
        assert e < 1e-5, e
        took += not torch.equal(a, b)
    assert took >= 4                                              # the route really ran (another summation order)
    assert torch.equal(after, base[0])
    for a, b in zip(zs, zb):                                      # z, mean, invstd, scale, bias
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    record("streamk", worst=max(rel_err(b.cpu().numpy(), a.cpu().numpy()) for a, b in zip(base, sk1)))


# ---------------------------------------------------------------------------------------------------------------------
# The route the headline times (hip_engine.forward_into: Winograd for every eligible 3x3 / transposed conv, no split-K),
# pinned DIRECTLY to the reference's golden heat-maps and arg-max (simplepose.py:82-86, fastpose.py:52-59, hrnet.py:421-456)
# ---------------------------------------------------------------------------------------------------------------------

def _count_winograd(vh, monkeypatch):
    calls = {"conv": 0, "deconv": 0, "splitk": 0, "f4": 0}
    oc, od = vh.conv3x3_winograd_fwd, vh.deconv4x4s2_winograd_fwd
    monkeypatch.setattr(vh, "conv3x3_winograd_fwd", lambda *a, **k: (calls.__setitem__("conv", calls["conv"] + 1), oc(*a, **k))[1])
    o32 = vh.conv3x3_winograd_c32_fwd                      # (the 32 -> 32 channel layers' own Winograd kernel counts as a Winograd launch)
    monkeypatch.setattr(vh, "conv3x3_winograd_c32_fwd", lambda *a, **k: (calls.__setitem__("conv", calls["conv"] + 1), o32(*a, **k))[1])
    monkeypatch.setattr(vh, "deconv4x4s2_winograd_fwd", lambda *a, **k: (calls.__setitem__("deconv", calls["deconv"] + 1), od(*a, **k))[1])
    o4 = vh.conv3x3_winograd_f4_fwd                        # (F(4x4,3x3) launches: Winograd launches too, counted on their own as well)

    def f4(*a, **k):
        calls["conv"] += 1; calls["f4"] += 1
        return o4(*a, **k)
    monkeypatch.setattr(vh, "conv3x3_winograd_f4_fwd", f4)
    return calls


def _eligible(m, head):
    """(3x3 / stride 1 / pad 1 convs, transposed 4x4 / stride 2 convs) the plan must send through the Winograd kernels: every one
    except the layer that writes the NCHW heat-maps."""
    c = sum(1 for mod in m.modules() if isinstance(mod, torch.nn.Conv2d) and mod is not head and mod.kernel_size == (3, 3)
            and mod.stride == (1, 1) and mod.padding == (1, 1) and mod.in_channels % 16 == 0 and mod.out_channels % 4 == 0)
    d = sum(1 for mod in m.modules() if isinstance(mod, torch.nn.ConvTranspose2d))
    return c, d


# (3x3 Winograd launches, transposed-conv Winograd launches, how many of the former are F(4x4,3x3): R50 = stages 1 - 3 (3 + 3 + 5; stage 4 is 8x6); FastPose-R50 adds the
#  two DUC convs; HRNet-W32 = the 64- and 128-channel branch convs; R152 at 384x288 = stage 1 (3) + stage 2 (7: 48x36) + duc2 (48x36) — 24x18 and 12x9 are not whole tiles)
STREAM_CASES = [("simplepose_r50", (13, 3, 11)), ("fastpose_r50", (15, 0, 13)), ("hrnet_w32", (213, 0, None)), ("fastpose_r152_384", (49, 0, None))]


@pytest.mark.parametrize("case", STREAM_CASES, ids=[c[0] for c in STREAM_CASES])
def test_stream_route_vs_reference_golden(vh, monkeypatch, case):
    """The bench / ActiveLearning.eval_and_query route: forward_into (+ score_batch) on the golden inputs.  Heat-maps within 1e-4 of
    the REFERENCE's output, arg-max identical, decode / local-peak / THC of the HIP heat-maps equal to the oracle's scorers run on the
    reference's heat-maps (indices exact), and the expected number of Winograd launches really happened (no split-K, no implicit-GEMM
    stand-in for an eligible layer)."""
    import os
    from active_learning.scoring import score_batch
    from alphapose.models import builder, hip_engine
    from alphapose.utils.config import edict
    from oracle import scorers
    name, (want_conv, want_deconv, want_f4) = case
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    hw = (256, 192)
    if name == "simplepose_r50":
        g = np.load(os.path.join(gdir, "simplepose_r50.npz")); ref = g["heatmaps"]; m = _build_simplepose(); head = m.final_layer
    elif name == "fastpose_r50":
        g = np.load(os.path.join(gdir, "fastpose_hrnet.npz")); ref = g["fastpose_heatmaps"]
        m = _build({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}); head = m.conv_out
    elif name == "hrnet_w32":
        g = np.load(os.path.join(gdir, "fastpose_hrnet.npz")); ref = g["hrnet_heatmaps"]; m = _build(HRNET_CFG); head = m.final_layer
    else:
        g = np.load(os.path.join(gdir, "fastpose_r152_384.npz")); ref = g["heatmaps"]; hw = (384, 288)
        preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [384, 288], "HEATMAP_SIZE": [96, 72]})
        m = builder.build_sppe(edict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152}), preset_cfg=preset)
        m.load_state_dict(synth.state_dict_for(m), strict=True)
        m = m.to(dev()).eval(); head = m.conv_out
    n = ref.shape[0]
    assert _eligible(m, head) == (want_conv, want_deconv)
    x = to_dev(synth.crops(n, hw=hw) if hw != (256, 192) else synth.crops(n))
    bb = synth.bboxes(n)
    calls = _count_winograd(vh, monkeypatch)
    out = torch.empty(ref.shape, device=dev())
    ip = torch.tensor([0] + [1] * (n - 1), device=dev(), dtype=torch.uint8)
    inx = torch.tensor([1] * (n - 1) + [0], device=dev(), dtype=torch.uint8)
    with torch.no_grad():
        assert not vh.latency_mode()
        hip_engine.forward_into(m, x, out)
        s = score_batch(out, to_dev(bb), ip, inx, thc_norm="L1")
    torch.cuda.synchronize()
    assert (calls["conv"], calls["deconv"]) == (want_conv, want_deconv), calls
    assert calls["f4"] == want_f4 if want_f4 is not None else calls["f4"] >= 8, calls
    hm = out.cpu().numpy()
    e = rel_err(hm, ref)
    record(f"stream_route_{name}", rel=e, winograd_conv=calls["conv"], winograd_deconv=calls["deconv"], winograd_f4=calls["f4"])
    assert e < 1e-4                                                                   # north_star: heat-maps within 1e-4 rel fp32
    ref_idx = ref.reshape(n, 17, -1).argmax(2)
    assert np.array_equal(hm.reshape(n, 17, -1).argmax(2), ref_idx)                   # integer peak indices bit-exact
    assert np.array_equal(s.argmax.cpu().numpy(), ref_idx)                            # ... and through the decode kernel
    import warnings
    for i in range(n):                                                                # scorers of the HIP stream vs the oracle on the REFERENCE's maps
        d = scorers.decode_heatmaps(ref[i], bb[i])
        assert np.array_equal(s.argmax[i].cpu().numpy(), d["idx"])
        np.testing.assert_allclose(s.keypoints[i, :, :2].cpu().numpy(), d["coords"], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(s.keypoints[i, :, 2].cpu().numpy(), d["maxvals"].reshape(-1), rtol=1e-4, atol=1e-6 + 1e-4 * np.abs(ref).max())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lp = scorers.localpeak_mean(ref[i])
            lp_own = scorers.localpeak_mean(hm[i])
        if np.isfinite(lp):
            np.testing.assert_allclose(float(s.localpeak[i]), lp_own, rtol=1e-5)       # the kernel on its own maps: exact peak set
            np.testing.assert_allclose(float(s.localpeak[i]), lp, rtol=1e-2)           # vs the reference's maps: a borderline peak may enter / leave the set
    if n > 1:
        want = scorers.thc_item(ref[0], None, ref[1], False, True, "L1")
        np.testing.assert_allclose(float(s.thc[0]), want, rtol=1e-3)


def test_full_video_stream_spot_checks(vh):
    """bench.py's step on a 1024-frame video: the heat-maps of spot frames are BIT-identical to their solo forward_into (a frame's bits
    do not depend on the launch it shares), and the stream's decode / THC agree with the oracle run on those spot heat-maps."""
    from active_learning.scoring import score_batch
    from alphapose.models import hip_engine
    from oracle import scorers
    m = _build_simplepose()
    n = 1024
    g = torch.Generator(device=dev()); g.manual_seed(5)
    x = torch.rand((n, 3, 256, 192), device=dev(), generator=g) - 0.45
    base = to_dev(synth.crops(8, seed=3))
    spots = [0, 1, 255, 256, 511, 777, 1022, 1023]
    for k, i in enumerate(spots):
        x[i] = base[k]
    w = 60 + 180 * torch.rand(n, device=dev(), generator=g)
    bbox = torch.stack([torch.full_like(w, 100.0), torch.full_like(w, 50.0), 100 + w, 50 + w * 4 / 3], 1).contiguous()
    pos = torch.arange(n, device=dev()) % 256
    ip, inx = (pos != 0).to(torch.uint8), (pos != 255).to(torch.uint8)
    hm = torch.empty((n, 17, 64, 48), device=dev())
    with torch.no_grad():
        hip_engine.forward_into(m, x, hm)
        s = score_batch(hm, bbox, ip, inx, thc_norm="L1")
        solo = torch.empty((1, 17, 64, 48), device=dev())
        for i in spots:
            hip_engine.forward_into(m, x[i:i + 1], solo)
            assert torch.equal(solo[0], hm[i]), i
    hmc, bbc = hm.cpu().numpy(), bbox.cpu().numpy()
    # ... and against the ORACLE graph (round 6: the spot frames inside the 1024-frame launch, not only their solo forwards): heat-maps within 1e-4, every arg-max index identical
    from oracle import nets
    ref = nets.SimplePoseRef(50)
    ref.load_state_dict(synth.state_dict_for(ref), strict=True)
    ref.eval()
    with torch.no_grad():
        want = ref(base.cpu()).numpy()
    got = hmc[spots]
    e = rel_err(got, want)
    record("full_video_spot_frames_vs_oracle", rel=e)
    assert e < 1e-4 and np.array_equal(got.reshape(8, 17, -1).argmax(2), want.reshape(8, 17, -1).argmax(2))
    for i in spots:
        d = scorers.decode_heatmaps(hmc[i], bbc[i])
        assert np.array_equal(s.argmax[i].cpu().numpy(), d["idx"])
        np.testing.assert_allclose(s.keypoints[i, :, :2].cpu().numpy(), d["coords"], rtol=1e-5, atol=1e-4)
        prev = hmc[i - 1] if ip[i] else None
        nxt = hmc[i + 1] if inx[i] else None
        np.testing.assert_allclose(float(s.thc[i]), scorers.thc_item(hmc[i], prev, nxt, bool(ip[i]), bool(inx[i]), "L1"), rtol=1e-5)


def test_fused_stem_pool_kernel_vs_float64_and_the_three_launch_path(vh):
    """vatl_stem7x7s2_pool_fwd: NCHW crops -> conv7x7/2 + folded BN + ReLU + maxpool3x3/2 -> NHWC in one launch (csrc/stem_pool.hip), against float64
    (conv + affine + relu + max_pool2d of torch on the CPU) and against the three-launch path it replaces (same values to fp32 rounding: K is summed in
    another order), for the three served widths, batch sizes that take 1 .. 16 row bands per image, and with a crop's bits independent of its batch."""
    g = torch.Generator(device="cpu").manual_seed(41)
    w = (torch.randn((64, 3, 7, 7), generator=g) * (2.0 / 147) ** 0.5)
    sc = torch.rand(64, generator=g) + 0.5
    bi = torch.randn(64, generator=g) * 0.5
    wd, scd, bid = w.to(dev()), sc.to(dev()), bi.to(dev())
    pw = vh.pack_stem_pool_weight(wd)
    wp = vh.pack_conv_weight(torch.cat([wd, torch.zeros((64, 1, 7, 7), device=dev())], 1)) if False else None
    for n, h, wdt in ((3, 256, 192), (40, 64, 128), (1, 32, 64), (600, 16, 192), (5, 256, 192)):
        assert vh.stem_pool_supported(h, wdt)
        x = torch.rand((n, 3, h, wdt), generator=g) - 0.45
        got = vh.stem_pool_fwd(x.to(dev()), pw, scd, bid)
        assert got.shape == (n, h // 4, wdt // 4, 64)
        k = min(n, 3)
        ref = F.conv2d(x[:k].double(), w.double(), None, 2, 3) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)
        ref = F.max_pool2d(ref.clamp_min(0), 3, 2, 1).permute(0, 2, 3, 1).numpy()
        e = rel_err(got[:k].cpu().numpy(), ref)
        record(f"stem_pool_{n}x{h}x{wdt}", rel=e)
        assert e < TOL, (n, h, wdt, e)
        solo = vh.stem_pool_fwd(x[n - 1:n].contiguous().to(dev()), pw, scd, bid)       # another band cut, same bits
        assert torch.equal(solo[0], got[n - 1])
    # HRNet's conv1 (3x3 / stride 2 / pad 1) + bn1 + relu through the same sliding block, no pooling: (N, H/2, W/2, 64)
    w3 = (torch.randn((64, 3, 3, 3), generator=g) * (2.0 / 27) ** 0.5)
    pw3 = vh.pack_stem3_weight(w3.to(dev()))
    for n, h, wdt in ((3, 256, 192), (33, 64, 128), (2, 32, 64), (700, 8, 64)):
        x = torch.rand((n, 3, h, wdt), generator=g) - 0.45
        got = vh.stem3_fwd(x.to(dev()), pw3, scd, bid)
        assert got.shape == (n, h // 2, wdt // 2, 64)
        k = min(n, 3)
        ref = (F.conv2d(x[-k:].double(), w3.double(), None, 2, 1) * sc.double().view(1, -1, 1, 1) + bi.double().view(1, -1, 1, 1)).clamp_min(0)
        e = rel_err(got[-k:].cpu().numpy(), ref.permute(0, 2, 3, 1).numpy())
        record(f"stem3_{n}x{h}x{wdt}", rel=e)
        assert e < TOL, (n, h, wdt, e)
        solo = vh.stem3_fwd(x[:1].contiguous().to(dev()), pw3, scd, bid)
        assert torch.equal(solo[0], got[0])
    assert not vh.stem_pool_supported(384, 288) and not vh.stem_pool_supported(256, 200) and not vh.stem_pool_supported(258, 192)
    with pytest.raises(vh.VatlError):
        vh.stem_pool_fwd(torch.zeros((1, 3, 384, 288), device=dev()), pw, scd, bid)
    # the plan: fused stem on and off give the same heat-maps to rounding, arg-max identical
    from alphapose.models import hip_engine
    m = _build_simplepose()
    xs = to_dev(synth.crops(20))
    out_f, out_3 = torch.empty((20, 17, 64, 48), device=dev()), torch.empty((20, 17, 64, 48), device=dev())
    with torch.no_grad():
        hip_engine.forward_into(m, xs, out_f)
        plan = hip_engine._plan_for(m, dev())
        keep, plan.trunk.stem_pw = plan.trunk.stem_pw, None
        assert keep is not None
        try:
            hip_engine.forward_into(m, xs, out_3)
        finally:
            plan.trunk.stem_pw = keep
    assert not torch.equal(out_f, out_3)                    # (another summation order really ran)
    e = rel_err(out_f.cpu().numpy(), out_3.cpu().numpy())
    record("stem_pool_plan_vs_three_launches", rel=e)
    assert e < 1e-5 and torch.equal(out_f.flatten(2).argmax(2), out_3.flatten(2).argmax(2))
