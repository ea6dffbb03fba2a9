"""Wide pin of the TIMED route's integer results against the reference (round 6).

``tests/golden/widepin.npz`` (tools/make_golden.py::gen_widepin, the reference imported in the build container) holds, for 64 seeded crops
per 256x192 network and 8 crops of FastPose-R152 at 384x288: the arg-max index of every joint plane as the reference's ``get_max_pred``
finds it (alphapose/utils/transforms.py:710-727), its value, the gap to the plane's second largest value, the key-points the reference's
``heatmap_to_coord_simple`` decodes (transforms.py:550-583), the same indices from the reference modules run in float64, and full heat-maps
of a few crops.  The stream route (`hip_engine.forward_into` + `score_batch`: Winograd F(4x4,3x3) / F(2x2,3x3) / F(3x3,2x2), no split-K)
must reproduce EVERY index — 3 400 planes instead of the 119 of the two-crop fixtures — and the decoded key-points.

A second test draws 64 FRESH crops per network, runs the oracle graph on the host in fp32 and in float64, and requires the HIP route to
disagree with exact arithmetic on no more planes than plain fp32 does.
"""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from tests.gpu_util import dev, record, rel_err, to_dev
from tests.test_gpu_conv import HRNET_CFG

pytestmark = pytest.mark.gpu

WIDE_SEED = 2024
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "widepin.npz")

CASES = {   # name -> (MODEL cfg, input size, heat-map size)
    "simplepose_r50": ({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50}, (256, 192), (64, 48)),
    "fastpose_r50": ({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}, (256, 192), (64, 48)),
    "hrnet_w32": (HRNET_CFG, (256, 192), (64, 48)),
    "fastpose_r152_384": ({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152}, (384, 288), (96, 72)),
}


@pytest.fixture(scope="module")
def vh():
    import vatl_hip
    vatl_hip.lib()
    return vatl_hip


def _build(name):
    from alphapose.models import builder
    from alphapose.utils.config import edict
    cfg, in_hw, hm_hw = CASES[name]
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": list(in_hw), "HEATMAP_SIZE": list(hm_hw)})
    m = builder.build_sppe(edict(cfg), preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    return m.to(dev()).eval()


def _stream(m, x, bb, hm_hw):
    """The entry points bench.py and ActiveLearning.eval_and_query use."""
    from active_learning.scoring import score_batch
    from alphapose.models import hip_engine
    n = x.shape[0]
    out = torch.empty((n, 17) + tuple(hm_hw), device=dev())
    ip = torch.tensor([0] + [1] * (n - 1), device=dev(), dtype=torch.uint8)
    inx = torch.tensor([1] * (n - 1) + [0], device=dev(), dtype=torch.uint8)
    with torch.no_grad():
        hip_engine.forward_into(m, to_dev(x), out)
        s = score_batch(out, to_dev(bb), ip, inx, thc_norm="L1")
    torch.cuda.synchronize()
    return out.cpu().numpy(), s


@pytest.mark.parametrize("name", list(CASES))
def test_stream_route_reproduces_every_reference_index(vh, name):
    g = np.load(GOLDEN)
    cfg, in_hw, hm_hw = CASES[name]
    n = int(g[f"{name}_n"])
    assert n >= (64 if in_hw == (256, 192) else 8)
    x = synth.crops(n, seed=WIDE_SEED, hw=in_hw)
    bb = synth.bboxes(n, seed=WIDE_SEED)
    m = _build(name)
    with vh.flop_meter() as fm:
        hm, s = _stream(m, x, bb, hm_hw)
    routes = {k: v for k, v in fm.routes.items() if v}
    assert routes.get("winograd_f4", 0) >= (8 if name != "simplepose_r50" else 11), routes          # the F(4x4,3x3) route really ran
    ref_idx = g[f"{name}_idx"].astype(np.int64)
    got_idx = hm.reshape(n, 17, -1).argmax(2)
    gap = g[f"{name}_gap"]
    absmax = float(g[f"{name}_absmax"])
    moved = np.argwhere(got_idx != ref_idx)
    # how close the nearest plane came to moving: the smallest top-2 gap met, against the error this route has on the kept maps
    keep = g[f"{name}_heatmaps"]
    e_net = rel_err(hm[:keep.shape[0]], keep)
    record(f"widepin_{name}", planes=int(ref_idx.size), moved=int(len(moved)), min_gap=float(gap.min()), min_gap_rel=float(gap.min()) / absmax,
           heatmap_rel=e_net, routes=str(routes))
    assert e_net < 1e-4                                                               # north_star: heat-maps within 1e-4 rel fp32 (normwise)
    assert len(moved) == 0, f"{name}: {len(moved)} of {ref_idx.size} arg-max indices moved, first {moved[:4].tolist()}, gaps {[float(gap[a, b]) for a, b in moved[:4]]}"
    assert np.array_equal(s.argmax.cpu().numpy(), ref_idx)                            # ... and through vatl_decode_pose
    mv = s.keypoints[:, :, 2].cpu().numpy()
    np.testing.assert_allclose(mv, g[f"{name}_maxval"], rtol=0, atol=1e-4 * absmax)   # peak values within 1e-4 of the map's range
    # decoded key-points: the reference's own decode of ITS maps.  The quarter-pixel shift is the sign of a difference of two neighbours
    # (transforms.py:563-568): a plane whose two neighbours differ by less than this route's error may shift the other way — counted, bounded
    # by what that error allows, and required to be a quarter-pixel event (never an index event).
    kp = s.keypoints[:, :, :2].cpu().numpy()
    ref_kp = g[f"{name}_keypoints"]
    scale = ((bb[:, 2] - bb[:, 0]) / hm_hw[1]).reshape(n, 1, 1)                        # image pixels per heat-map pixel
    d = np.abs(kp - ref_kp) / scale                                                   # in heat-map pixels
    exact = d < 1e-3
    half = np.abs(d - 0.25) < 1e-3                                                    # sign(…) went 0 <-> ±1
    full = np.abs(d - 0.5) < 1e-3                                                     # sign(…) went -1 <-> +1
    assert (exact | half | full).all(), float(d.max())
    flips = int((~exact).sum())
    record(f"widepin_{name}_quarter_pixel", coords=int(d.size), flips=flips)
    if flips:                                                                          # every flip must be explained by a neighbour difference inside the error band
        tol = 4 * e_net * absmax
        for i, j, c in np.argwhere(~exact):
            if i >= keep.shape[0]:
                continue                                                              # (no reference map kept for this crop: counted above)
            px, py = int(ref_idx[i, j] % hm_hw[1]), int(ref_idx[i, j] // hm_hw[1])
            pl = keep[i, j]
            diff = pl[py, px + 1] - pl[py, px - 1] if c == 0 else pl[py + 1, px] - pl[py - 1, px]
            assert abs(float(diff)) <= tol, (name, i, j, c, float(diff), tol)
    assert flips <= max(2, d.size // 500), flips


@pytest.mark.parametrize("name", ["simplepose_r50", "fastpose_r50", "hrnet_w32"])
def test_fresh_crops_no_further_from_float64_than_fp32(vh, name):
    """64 crops no fixture has seen: planes whose arg-max differs from the float64 oracle graph's — the HIP route may have no more of them
    than the oracle graph in plain fp32 (torch on the host) has."""
    from oracle import nets
    cfg, in_hw, hm_hw = CASES[name]
    n = 64
    x = synth.crops(n, seed=977, hw=in_hw)
    bb = synth.bboxes(n, seed=977)
    m = _build(name)
    hm, s = _stream(m, x, bb, hm_hw)
    sd = synth.state_dict_for(m)
    ref = {"simplepose_r50": lambda: nets.SimplePoseRef(50), "fastpose_r50": lambda: nets.FastPoseRef(50), "hrnet_w32": lambda: nets.HRNetRef()}[name]()
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    xt = torch.from_numpy(x)
    with torch.no_grad():
        h32 = torch.cat([ref(xt[i:i + 16]) for i in range(0, n, 16)], 0).numpy()
        ref = ref.double()
        h64 = torch.cat([ref(xt[i:i + 8].double()) for i in range(0, n, 8)], 0).numpy()
    i64 = h64.reshape(n, 17, -1).argmax(2)
    mis_hip = int((hm.reshape(n, 17, -1).argmax(2) != i64).sum())
    mis_f32 = int((h32.reshape(n, 17, -1).argmax(2) != i64).sum())
    part = np.partition(h64.reshape(n, 17, -1), -2, axis=2)
    record(f"widepin_fresh_{name}", planes=int(i64.size), hip_vs_f64=mis_hip, fp32_vs_f64=mis_f32, hip_rel=rel_err(hm, h64), fp32_rel=rel_err(h32, h64),
           min_gap_rel=float((part[..., -1] - part[..., -2]).min() / np.abs(h64).max()))
    assert rel_err(hm, h64) < 1e-4
    assert mis_hip <= mis_f32, (mis_hip, mis_f32)
    assert np.array_equal(s.argmax.cpu().numpy(), hm.reshape(n, 17, -1).argmax(2))
