"""ctypes binding of libvatl_hip.so (include/vatl_hip.h) + thin torch-tensor wrappers.

PyTorch is plumbing only: it owns device memory and the HIP stream; every
wrapper passes ``tensor.data_ptr()`` and the current stream to the C ABI.
There is NO fallback: a missing library or a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VATL_HIP_LIB: an explicit library path — the profiling variant (build.py --ablation) for tools/conv_bench.py --ablate
LIB_PATH = os.environ.get("VATL_HIP_LIB") or os.path.join(_HERE, "libvatl_hip.so")

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_i64 = C.c_int64

# name -> (restype, argtypes); mirrors include/vatl_hip.h one to one
# (tests/test_cabi.py parses the header and checks this table against it)
SIGNATURES = {
    "vatl_version": (_i, []),
    "vatl_last_error": (C.c_char_p, []),
    "vatl_flop_meter_begin": (_i, []),
    "vatl_flop_meter_end": (_i, [_p, _p, _p, _p]),
    "vatl_flop_meter_routes": (_i, [_p, _i]),
    "vatl_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vatl_pack_conv_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_pack_deconv4x4s2_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "vatl_pack_conv1x1_dual_weight": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_conv1x1_dual_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_winograd_c32_weight_floats": (_i64, []),
    "vatl_pack_winograd_c32_weight": (_i, [_p, _p, _p]),
    "vatl_conv3x3_winograd_c32_supported": (_i, [_i, _i, _i, _i, _i]),
    "vatl_conv3x3_winograd_c32_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_winograd_f4_weight_floats": (_i64, [_i, _i]),
    "vatl_pack_winograd_f4_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "vatl_winograd_f4_stats_row_blocks": (_i64, [_i64, _i, _i]),
    "vatl_conv3x3_winograd_f4_fwd_stats": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_conv3x3_winograd_f4_fwd_bnbwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "vatl_conv3x3_winograd_f4_supported": (_i, [_i, _i, _i, _i, _i]),
    "vatl_conv3x3_winograd_f4_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_conv1x1_rows_supported": (_i, [_i, _i, _i, _i64]),
    "vatl_conv1x1_rows_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "vatl_bottleneck_chain_supported": (_i, [_i, _i, _i, _i64]),
    "vatl_bottleneck_chain_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "vatl_stem_pool_weight_floats": (_i64, []),
    "vatl_pack_stem_pool_weight": (_i, [_p, _p, _p]),
    "vatl_stem_pool_supported": (_i, [_i, _i]),
    "vatl_stem7x7s2_pool_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vatl_stem3_weight_floats": (_i64, []),
    "vatl_pack_stem3_weight": (_i, [_p, _p, _p]),
    "vatl_stem3x3s2_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vatl_bn_fold": (_i, [_p, _p, _p, _p, _p, _f, _p, _p, _i, _p]),
    "vatl_tune_set": (_i, [_i, _i]),
    "vatl_set_splitk_workspace": (_i, [_p, _i64]),
    "vatl_set_splitk_workspace_thread": (_i, [_p, _i64]),
    "vatl_streamk_workspace_bytes": (_i64, []),
    "vatl_set_streamk_workspace_thread": (_i, [_p, _i64]),
    "vatl_conv_cout_pad": (_i, [_i]),
    "vatl_conv2d_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_deconv4x4s2_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_maxpool3x3s2_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vatl_gap_fwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "vatl_pixelshuffle2_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vatl_se_scale_add_relu": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "vatl_fuse_upsample_add": (_i, [_p, _p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_upsample_nearest_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_gap_bwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "vatl_decode_argmax_affine": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_decode_pose": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_thc_pairs": (_i, [_p, _p, _i64, _i64, _p, _i, _i, _i, _i, _p]),
    "vatl_thc_combine": (_i, [_p, _p, _p, _p, _i, _p]),
    "vatl_localpeak_mean": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "vatl_hybrid_ae_wpu": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "vatl_tpc_stream": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "vatl_decode_softargmax": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_ae_forward": (_i, [_p, _p, _i, _i, _p, _p, _i, _p]),
    "vatl_hybrid_feature_f64": (_i, [_p, _p, _p, _p, _i, _p]),
    "vatl_localpeak_mask": (_i, [_p, _p, _i, _i, _i, _f, _p]),
    "vatl_peaks5": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_plane_entropy": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vatl_conv2d_fwd_ex": (_i, [_p] * 6 + [_i] * 20 + [_p]),
    "vatl_conv2d_fwd_ex_bnbwd": (_i, [_p] * 4 + [_i] * 19 + [_p] * 9),
    "vatl_bn_bwd_from_stats": (_i, [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _p]),
    "vatl_pack_dgrad_weight": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "vatl_conv2d_wgrad_workspace_floats": (_i64, [_i, _i, _i, _i, _i64]),
    "vatl_conv2d_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_deconv4x4s2_wgrad_workspace_floats": (_i64, [_i, _i, _i64]),
    "vatl_deconv4x4s2_wgrad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_col_reduce_workspace_doubles": (_i64, [_i64, _i]),
    "vatl_bn_train_fwd_stats": (_i, [_p, _i64, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p]),
    "vatl_scale_bias_act": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _p]),
    "vatl_bn_train_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _p, _p]),
    "vatl_bn_train_bwd_relu": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _p, _p]),
    "vatl_conv_stats_row_blocks": (_i64, [_i64, _i]),
    "vatl_winograd_cout_pad": (_i, [_i]),
    "vatl_winograd_weight_floats": (_i64, [_i, _i]),
    "vatl_pack_winograd_weight": (_i, [_p, _p, _i, _i, _i, _p]),
    "vatl_conv3x3_winograd_fwd": (_i, [_p] * 6 + [_i] * 6 + [_p]),
    "vatl_winograd_last_route": (_i, []),
    "vatl_winograd_stats_row_blocks": (_i64, [_i64, _i, _i]),
    "vatl_conv3x3_winograd_fwd_stats": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_conv3x3_winograd_fwd_bnbwd": (_i, [_p] * 4 + [_i] * 5 + [_p] * 9),
    "vatl_conv3x3_winograd_wgrad_workspace_floats": (_i64, [_i, _i, _i64, _i, _i]),
    "vatl_conv3x3_winograd_wgrad": (_i, [_p] * 4 + [_i] * 5 + [_p]),
    "vatl_deconv4x4s2_winograd_wgrad_workspace_floats": (_i64, [_i, _i, _i64, _i, _i]),
    "vatl_deconv4x4s2_winograd_wgrad": (_i, [_p] * 4 + [_i] * 5 + [_p]),
    "vatl_winograd_deconv_dgrad_weight_floats": (_i64, [_i, _i]),
    "vatl_pack_winograd_deconv_dgrad_weight": (_i, [_p, _p, _i, _i, _p]),
    "vatl_deconv4x4s2_winograd_dgrad": (_i, [_p] * 4 + [_i] * 5 + [_p]),
    "vatl_deconv4x4s2_winograd_dgrad_bnbwd": (_i, [_p] * 4 + [_i] * 5 + [_p] * 9),
    "vatl_winograd_deconv_weight_floats": (_i64, [_i, _i]),
    "vatl_pack_winograd_deconv_weight": (_i, [_p, _p, _i, _i, _p]),
    "vatl_deconv4x4s2_winograd_fwd": (_i, [_p] * 5 + [_i] * 6 + [_p]),
    "vatl_winograd_deconv_stats_row_blocks": (_i64, [_i64, _i, _i]),
    "vatl_deconv4x4s2_winograd_fwd_stats": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "vatl_conv2d_fwd_stats": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_deconv4x4s2_fwd_stats": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_bn_train_finalize": (_i, [_p, _i64, _i64, _i, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p]),
    "vatl_maxpool3x3s2_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_maxpool3x3s2_fwd_idx": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_pack_weights_multi": (_i, [_p, _i, _i64, _p]),
    "vatl_adamw_multi_block_elems": (_i64, []),
    "vatl_checksum_block_words": (_i64, []),
    "vatl_checksum_multi": (_i, [_p, _i, _i64, _p, _p]),
    "vatl_maxpool3x3s2_fwd_idx_affine": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_bn_train_bwd_relu_pool": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "vatl_maxpool3x3s2_bwd_idx": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "vatl_pixelunshuffle2": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "vatl_se_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vatl_relu_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "vatl_col_sum": (_i, [_p, _i64, _i, _p, _p, _p]),
    "vatl_masked_mse_workspace_floats": (_i64, [_i64]),
    "vatl_masked_mse_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vatl_l1_joint_regression_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vatl_gaussian_targets": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "vatl_crop_warp_affine": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _p]),
    "vatl_ae_train_step": (_i, [_p, _p, _p, _p, _i, _i, _i, _d, _d, _d, _d, _i, _p, _p]),
    "vatl_adamw_step": (_i, [_p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _i, _p]),
    "vatl_oks": (_i, [_p, _p, _p, _p, _i, _p]),
    "vatl_cosine_rowsum": (_i, [_p, _i64, _i, _p, _p, _p]),
    "vatl_kcenter_update": (_i, [_p, _i64, _i, _p, _i, _p, _i, _p]),
    "vatl_kcenter_pick": (_i, [_p, _p, _d, _d, _p, _i, _i64, _p]),
    "vatl_adamw_step_multi": (_i, [_p, _i, _i64, _d, _d, _d, _d, _d, _i, _p]),
    "vatl_adam_step": (_i, [_p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _i, _p]),
    "vatl_sgd_step": (_i, [_p, _p, _p, _i64, _d, _d, _d, _i, _p]),
}

_lib = None


class VatlError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VatlError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(or vatl4pose-wacv2024_amd/build.py). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise VatlError(f"{what} failed ({rc}): {lib().vatl_last_error().decode()}")


def upload(host, device, dtype=None, owned: bool = False):
    """Host array / CPU tensor -> device tensor WITHOUT draining the stream: staged through PyTorch's pinned-memory cache and copied
    non-blocking.  (A plain `.to(device)` from pageable memory blocks the host until every launch queued before it has finished — a few
    of those per loader batch serialise the host's batch preparation with the device's forward pass.)

    The source is gathered ONCE, straight into a pinned staging buffer of this call (strided views of a loader batch included: no
    pageable intermediate), so the caller may overwrite ``host`` as soon as the call returns — also when ``host`` itself is pinned,
    unless it passes ``owned=True`` ("the buffer is mine and stays untouched until the copy has run": then it is read in place)."""
    t = torch.as_tensor(host)
    device = torch.device(device)
    if t.is_cuda:                                           # already on a device: nothing to stage
        return t.to(device=device, dtype=dtype if dtype is not None else t.dtype)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if device.type != "cuda" or t.numel() == 0:
        return t.to(device)
    if owned and t.is_pinned() and t.is_contiguous():
        return t.to(device, non_blocking=True)
    stage = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)         # caching host allocator: reused once the copy below has completed
    stage.copy_(t)
    return stage.to(device, non_blocking=True)


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise VatlError("vatl_hip needs device (HIP) tensors; there is no CPU path")
    if t.dtype != dtype:
        raise VatlError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise VatlError("tensor must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------
# layout / parameter packing
# ----------------------------------------------------------------------------

def nchw_to_nhwc(x: torch.Tensor, cpad: int | None = None) -> torch.Tensor:
    n, c, h, w = x.shape
    cpad = cpad or c
    y = torch.empty((n, h, w, cpad), device=x.device, dtype=torch.float32)
    _check(lib().vatl_nchw_to_nhwc(_ptr(x), _ptr(y), n, c, h, w, cpad, _stream()), "vatl_nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    _check(lib().vatl_nhwc_to_nchw(_ptr(x), _ptr(y), n, c, h, w, _stream()), "vatl_nhwc_to_nchw")
    return y


ROUTE_NAMES = ("igemm", "igemm_bnbwd", "igemm_dma", "persistent_1x1", "streamk", "rows_1x1", "bottleneck_chain", "stem_pool", "halo_3x3", "winograd",
               "winograd_2h", "winograd_bnbwd", "winograd_persist", "winograd_c32", "wgrad", "winograd_wgrad", "winograd_wgrad_2h", "winograd_wgrad_table", "winograd_f4", "winograd_f4_bnbwd")


class flop_meter:
    """`with vh.flop_meter() as fm: ...` — executed MFMA FLOPs (tile padding included) of the matrix-core launches the calling host
    thread makes inside the block: fm.direct (implicit GEMM / weight gradients), fm.winograd (transform-domain GEMMs), fm.total, and the
    launch counts.  Thread-local in the library (vatl_flop_meter_begin / _end); used by bench.py for roofline.frac."""

    def __enter__(self):
        _check(lib().vatl_flop_meter_begin(), "flop_meter_begin")
        self.direct = self.winograd = self.total = 0.0
        self.direct_launches = self.winograd_launches = 0
        return self

    @staticmethod
    def launches_now() -> int:
        """Matrix-core launches metered since __enter__ (every metered launch also bumps exactly one route counter)."""
        counts = (C.c_int64 * len(ROUTE_NAMES))()
        _check(min(0, lib().vatl_flop_meter_routes(counts, len(ROUTE_NAMES))), "flop_meter_routes")
        return int(sum(counts))

    def __exit__(self, *exc):
        counts = (C.c_int64 * len(ROUTE_NAMES))()
        _check(min(0, lib().vatl_flop_meter_routes(counts, len(ROUTE_NAMES))), "flop_meter_routes")
        self.routes = {n: int(c) for n, c in zip(ROUTE_NAMES, counts)}          # launches per kernel family (include/vatl_hip.h VATL_ROUTE_NAMES)
        d, w, nd, nw = C.c_double(0), C.c_double(0), C.c_int64(0), C.c_int64(0)
        _check(lib().vatl_flop_meter_end(C.byref(d), C.byref(w), C.byref(nd), C.byref(nw)), "flop_meter_end")
        self.direct, self.winograd, self.total = d.value, w.value, d.value + w.value
        self.direct_launches, self.winograd_launches = nd.value, nw.value
        return False


def tune_set(knob: int, value: int):
    _check(lib().vatl_tune_set(knob, value), "vatl_tune_set")


_splitk_buf = {}                                     # device index -> workspace tensor kept alive while registered


def enable_splitk(megabytes: int = 64, device=None):
    """Opt-in split-K for small-batch latency on ``device`` (default: the current one; see vatl_set_splitk_workspace);
    ``megabytes = 0`` switches it off for that device."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):                                     # the library files the workspace under the current device
        if megabytes <= 0:
            _check(lib().vatl_set_splitk_workspace(None, 0), "vatl_set_splitk_workspace")
            _splitk_buf.pop(idx, None)
            return
        buf = torch.empty(megabytes * (1 << 18), device=torch.device("cuda", idx), dtype=torch.float32)
        _check(lib().vatl_set_splitk_workspace(_ptr(buf), buf.numel()), "vatl_set_splitk_workspace")
        _splitk_buf[idx] = buf


_splitk_scratch = {}                                 # device index -> free list of workspaces of the automatic small-batch mode
_splitk_scratch_lock = __import__("threading").Lock()
# the batch-invariant cut sizes a layer's split from what a 16-crop batch needs (conv_igemm.hip: launch): the largest is SimplePose's last
# transposed conv, 3 slices x 16 crops x 64 x 48 x 256 floats = 151 MB; a smaller workspace only makes that layer run unsplit
SPLITK_SCRATCH_MB = 256


class splitk_scope:
    """Split-K for the launches of THIS host thread for the duration of one call (the module-call path of small batches,
    alphapose/models/hip_engine.py: run_module_nchw), with the batch-invariant cut, and removed afterwards: the evaluation stream
    — whose crops must have batch-size-independent bits — never sees it, other host threads (DataParallel replicas) are not
    affected and share no buffer with this one.  A device on which split-K was switched on explicitly (``enable_splitk``) is
    left alone.  Workspaces live in a per-DEVICE free list: a scope takes one (allocating only when every buffer of the device
    is in use by another thread's open scope) and puts it back on exit, so a thread-per-call caller does not accumulate buffers."""

    def __init__(self, device, megabytes: int | None = None):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.mb = megabytes or SPLITK_SCRATCH_MB
        self.active = False
        self.buf = None

    def __enter__(self):
        if self.idx in _splitk_buf:
            return self
        want = self.mb * (1 << 18)
        with _splitk_scratch_lock:
            free = _splitk_scratch.setdefault(self.idx, [])
            hit = next((e for e in free if e[0].numel() >= want), None)
            if hit is not None:
                free.remove(hit)
        if hit is not None:
            self.buf = hit[0]
            torch.cuda.current_stream(self.idx).wait_event(hit[1])      # the previous user's launches (possibly another thread's stream) are done with it
        else:
            self.buf = torch.empty(want, device=torch.device("cuda", self.idx), dtype=torch.float32)
        _check(lib().vatl_set_splitk_workspace_thread(_ptr(self.buf), self.buf.numel()), "vatl_set_splitk_workspace_thread")
        self.active = True
        _tls.latency_mode = getattr(_tls, "latency_mode", 0) + 1
        return self

    def __exit__(self, *exc):
        if self.active:
            _tls.latency_mode -= 1
            _check(lib().vatl_set_splitk_workspace_thread(None, 0), "vatl_set_splitk_workspace_thread")
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.idx))
            with _splitk_scratch_lock:
                _splitk_scratch[self.idx].append((self.buf, ev))
            self.buf = None
            self.active = False
        return False


def latency_mode() -> bool:
    """True inside a ``splitk_scope`` of this host thread: the module-call path of small batches (<= 16 crops).  The inference plans
    then keep every conv on the implicit GEMM, whose split-K fills the chip from a handful of tiles; the Winograd kernels have no
    reduction split (one crop: deconv1 = 32 blocks x 128 serial stages) and are 27 - 40 % slower end to end below ~8 crops."""
    return getattr(_tls, "latency_mode", 0) > 0

def conv_cout_pad(cout: int) -> int:
    return lib().vatl_conv_cout_pad(cout)


class _PackJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("first_block", C.c_int64), ("kind", C.c_int32), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("R", C.c_int32), ("S", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32), ("tap_r", C.c_int32 * 16), ("tap_s", C.c_int32 * 16)]


class PackPlan:
    """The weight re-packs of one fine-tune step as ONE launch (vatl_pack_weights_multi).

    The trainers (alphapose/models/hip_train.py) ask for packed copies through ``pack_conv_weight`` / ``pack_deconv_weight`` /
    ``pack_dgrad_weight`` where they need them.  The first step under a plan runs those calls as usual and RECORDS them (source
    tensor, layout, destination buffer, which is then kept); ``seal()`` uploads the descriptor table.  From the next step on,
    ``begin()`` refreshes every recorded destination with one launch and the pack calls return the kept buffers.  A lookup is
    honoured only if the source tensor is the recorded storage at the version it had at ``begin()`` (optimizers that write
    through the C ABI bump the version counters): anything else is packed on the spot, and makes the plan re-record."""

    def __init__(self):
        self.jobs = {}            # key -> [src tensor, dst tensor, descriptor fields, version at begin()]
        self.table = None
        self.total_blocks = 0
        self.ready = False
        self.stale = False

    def begin(self):
        if self.ready and not self.stale:
            for j in self.jobs.values():
                j[3] = j[0]._version
            _check(lib().vatl_pack_weights_multi(_ptr(self.table, torch.uint8), len(self.jobs), self.total_blocks, _stream()), "vatl_pack_weights_multi")
        else:
            self.jobs, self.table, self.ready, self.stale = {}, None, False, False

    def lookup(self, key, w):
        if not self.ready:
            return None
        j = self.jobs.get(key)
        if j is None or j[0].data_ptr() != w.data_ptr() or j[3] != w._version:
            self.stale = True                      # a layout the recording did not see, or weights that changed since begin()
            return None
        return j[1]

    def record(self, key, w, dst, fields):
        if not self.ready and key not in self.jobs:
            self.jobs[key] = [w, dst, fields, w._version]

    def seal(self):
        if self.ready or not self.jobs:
            return
        arr = (_PackJob * len(self.jobs))()
        blocks = 0
        for k, (src, dst, f, _) in enumerate(self.jobs.values()):
            jb = arr[k]
            jb.src, jb.dst, jb.first_block = src.data_ptr(), dst.data_ptr(), blocks
            jb.kind, jb.Cout, jb.Cin, jb.R, jb.S, jb.a, jb.b, jb.c = f[:8]
            for t, (tr, ts) in enumerate(f[8]):
                jb.tap_r[t], jb.tap_s[t] = tr, ts
            if f[0] == 1:                                    # data-gradient layout [a = CinPad][c = taps][b = CoutK]: 32 x 32 tiles of one tap
                blocks += ((f[5] + 31) // 32) * f[7] * ((f[6] + 31) // 32)
            else:                                            # F(2x2) / F(3x3,2x2) Winograd filters (kinds 3 .. 6): 4096 elements per block; everything else 1024
                blocks += (dst.numel() + 1023) // 1024 if (f[0] < 3 or f[0] >= 7) else dst.numel() // 4096
        dev = next(iter(self.jobs.values()))[1].device
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = host.to(dev)
        self.total_blocks = blocks
        self.ready = True


STREAMK_IN_TRAINING = False                          # module constant, not an environment switch (see streamk_scope)
_streamk_ws = {}                                     # (host thread, device index) -> zero-initialised stream-K workspace


class streamk_scope:
    """Stream-K for this host thread's conv launches while the scope is open (the trainers' forward / backward passes): launches
    that would leave much of the chip idle share their (tile, k-tile) units evenly over 768 persistent blocks
    (vatl_set_streamk_workspace_thread).  Results are deterministic but not the unsplit kernels' bits, so nothing outside
    training opens it — and the trainers open it only when STREAMK_IN_TRAINING is set (off): measured on MI355X (profiles/r03_notes.md) the route
    gains 8-23 % on the launches it takes when they run ALONE (R50 stage 4 at B = 120, FastPose-R152 stages 3-4 at B = 32), but
    in the real step the block slots those launches leave idle are where the side stream's weight-gradient kernels run, and 768
    persistent blocks take that overlap away: fine-tune step 44.8 -> 44.7 ms (R50), 67.5 -> 70.4 ms (R152).  ``force`` opens it
    regardless (benchmarks, tests)."""

    def __init__(self, device, force: bool = False):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.active = False
        self.force = force

    def __enter__(self):
        if not self.force and not STREAMK_IN_TRAINING:
            return self
        import threading
        key = (threading.get_ident(), self.idx)
        buf = _streamk_ws.get(key)
        if buf is None:
            buf = _streamk_ws[key] = torch.zeros(int(lib().vatl_streamk_workspace_bytes()), device=torch.device("cuda", self.idx), dtype=torch.uint8)
        _check(lib().vatl_set_streamk_workspace_thread(_ptr(buf, torch.uint8), buf.numel()), "vatl_set_streamk_workspace_thread")
        self.active = True
        return self

    def __exit__(self, *exc):
        if self.active:
            _check(lib().vatl_set_streamk_workspace_thread(None, 0), "vatl_set_streamk_workspace_thread")
        return False


import threading as _threading

_tls = _threading.local()                            # per host thread: the plan its pack_* calls consult


def set_pack_plan(plan):
    """The plan the pack_* calls of THIS host thread consult (None: every call packs on the spot).  Returns the previous one."""
    prev = getattr(_tls, "pack_plan", None)
    _tls.pack_plan = plan
    return prev


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """(Cout,Cin,R,S) -> [CoutPad][R][Spad][CinPad]; the 3-channel stem is padded to 4 channels x 8 taps."""
    cout, cin, r, s = w.shape
    cpad = conv_cout_pad(cout)
    if r == 1 and s == 1 and cpad == cout and cin != 3 and w.is_contiguous() and w.dtype == torch.float32:
        return w.detach().view(cout, 1, 1, cin)              # (Cout,Cin,1,1) already is the packed [Cout][1][1][Cin]: no copy
    spad, cinpad = (8, 4) if cin == 3 else (s, cin)
    plan, key = getattr(_tls, "pack_plan", None), ("conv", w.data_ptr(), tuple(w.shape))
    if plan is not None and w.is_contiguous():
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    out = torch.empty((cpad, r, spad, cinpad), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_conv_weight(_ptr(w.contiguous()), _ptr(out), cout, cin, r, s, cpad, spad, cinpad, _stream()), "vatl_pack_conv_weight")
    if plan is not None and w.is_contiguous():
        plan.record(key, w, out, (0, cout, cin, r, s, cpad, spad, cinpad, ()))
    return out


def pack_conv1x1_dual_weight(w1, scale1, bias1, w2, scale2, bias2):
    """Two (Cout,C,1,1) weights + folded BN scale/bias -> ([CoutPad][C1+C2] scaled rows, bias1 + bias2)."""
    cout, c1 = w1.shape[:2]
    c2 = w2.shape[1]
    cpad = conv_cout_pad(cout)
    out = torch.empty((cpad, c1 + c2), device=w1.device, dtype=torch.float32)
    bias = torch.empty(cout, device=w1.device, dtype=torch.float32)
    _check(lib().vatl_pack_conv1x1_dual_weight(_ptr(w1.contiguous()), _ptr(scale1), _ptr(bias1), _ptr(w2.contiguous()), _ptr(scale2), _ptr(bias2),
                                               _ptr(out), _ptr(bias), cout, c1, c2, cpad, _stream()), "vatl_pack_conv1x1_dual_weight")
    return out, bias


def conv1x1_dual_fwd(a, x, w_packed, bias, cout: int, stride2: int, relu: bool, out=None):
    """relu?(W1' a + W2' x[::s, ::s] + bias): a (N,Ho,Wo,C1) and x (N,H2,W2,C2) NHWC -> (N,Ho,Wo,Cout)."""
    n, ho, wo, c1 = a.shape
    _, h2, w2, c2 = x.shape
    y = out if out is not None else torch.empty((n, ho, wo, cout), device=a.device, dtype=torch.float32)
    _check(lib().vatl_conv1x1_dual_fwd(_ptr(a), _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(y), n, ho, wo, c1, h2, w2, c2, stride2, cout,
                                       w_packed.shape[0], int(relu), _stream()), "vatl_conv1x1_dual_fwd")
    return y


def pack_winograd_c32_weight(w: torch.Tensor) -> torch.Tensor:
    """(32,32,3,3) -> G g G^T in the LDS order of csrc/winograd_c32.hip."""
    if tuple(w.shape) != (32, 32, 3, 3):
        raise VatlError("pack_winograd_c32_weight: a (32, 32, 3, 3) filter")
    out = torch.empty(int(lib().vatl_winograd_c32_weight_floats()), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_winograd_c32_weight(_ptr(w.contiguous()), _ptr(out), _stream()), "vatl_pack_winograd_c32_weight")
    return out


def conv3x3_winograd_c32_supported(n: int, h: int, w: int, cin: int, cout: int) -> bool:
    return bool(lib().vatl_conv3x3_winograd_c32_supported(n, h, w, cin, cout))


def conv3x3_winograd_c32_fwd(x, u, scale, bias, relu: bool, residual=None, out=None):
    """3x3 / stride 1 / pad 1, 32 -> 32 channels, NHWC, Winograd F(2x2,3x3) with wave-private tiles (csrc/winograd_c32.hip)."""
    n, h, w, c = x.shape
    y = out if out is not None else torch.empty_like(x)
    assert x.is_contiguous() and y.is_contiguous() and (residual is None or (residual.is_contiguous() and residual.shape == x.shape))
    _check(lib().vatl_conv3x3_winograd_c32_fwd(_ptr(x), _ptr(u), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), n, h, w, int(relu), _stream()),
           "vatl_conv3x3_winograd_c32_fwd")
    return y



def pack_winograd_f4_weight(w: torch.Tensor, data_gradient: bool = False) -> torch.Tensor:
    """(Cout,Cin,3,3) -> U = G g G^T of F(4x4,3x3) in the MFMA fragment order of csrc/winograd_f4.hip.  With ``data_gradient`` the result is the filter of
    dX = conv(dY, rot180(w)^T): its output channels are the forward Cin.  Inside a PackPlan (the trainers' forward / backward) the buffer is kept and refreshed by
    the step's one re-pack launch."""
    co, ci = w.shape[:2]
    cout, cin = (ci, co) if data_gradient else (co, ci)
    plan, key = getattr(_tls, "pack_plan", None), ("winograd_f4", w.data_ptr(), tuple(w.shape), bool(data_gradient))
    if plan is not None and w.is_contiguous():
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    u = torch.empty(int(lib().vatl_winograd_f4_weight_floats(cout, cin)), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_winograd_f4_weight(_ptr(w.contiguous()), _ptr(u), cout, cin, int(data_gradient), _stream()), "vatl_pack_winograd_f4_weight")
    if plan is not None and w.is_contiguous():
        plan.record(key, w, u, (8 if data_gradient else 7, cout, cin, 3, 3, 0, 0, ci, ()))
    return u


def conv3x3_winograd_f4_supported(n: int, h: int, w: int, cin: int, cout: int) -> bool:
    return bool(lib().vatl_conv3x3_winograd_f4_supported(n, h, w, cin, cout))


def conv3x3_winograd_f4_fwd(x, u, scale, bias, cout: int, relu: bool, residual=None, out=None):
    """3x3 / stride 1 / pad 1 conv as Winograd F(4x4,3x3): x NHWC (N,H,W,Cin), H and W multiples of 4 -> (N,H,W,Cout)."""
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    _check(lib().vatl_conv3x3_winograd_f4_fwd(_ptr(x), _ptr(u), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), n, h, w, cin, cout, int(relu), _stream()),
           "vatl_conv3x3_winograd_f4_fwd")
    return y


def conv3x3_winograd_f4_fwd_bnstats(x, u, cout: int, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """conv3x3_winograd_fwd_bnstats on the F(4x4,3x3) route: -> z, save_mean, save_invstd, scale, bias; running stats updated in place."""
    n, h, w, cin = x.shape
    z = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    stats = torch.empty(int(lib().vatl_winograd_f4_stats_row_blocks(n, h, w)) * cout * 2, device=x.device, dtype=torch.float64)
    used = C.c_int64(0)
    _check(lib().vatl_conv3x3_winograd_f4_fwd_stats(_ptr(x), _ptr(u), _ptr(z), _ptr(stats, torch.float64), C.addressof(used), n, h, w, cin, cout, _stream()),
           "vatl_conv3x3_winograd_f4_fwd_stats")
    return [z] + _bn_finalize(stats, used.value, z.numel() // cout, cout, (gamma, beta, running_mean, running_var, momentum, eps), x.device)


def conv3x3_winograd_f4_fwd_bnbwd(x, u, cout: int, spec: "BnBwdSpec", out=None, residual=None):
    """conv3x3_winograd_fwd_bnbwd on the F(4x4,3x3) route (u: data-gradient packing)."""
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    if y.shape != spec.z.shape:
        raise VatlError("conv3x3_winograd_f4_fwd_bnbwd: the BatchNorm tensors must have the layout of the output")
    used = C.c_int64(0)
    c = spec.z.shape[-1]
    need = int(lib().vatl_winograd_f4_stats_row_blocks(n, h, w))
    if (spec.blocks + need) * c * 2 > spec.stats.numel():
        raise VatlError("conv3x3_winograd_f4_fwd_bnbwd: statistics buffer too small")
    stats_ptr = spec.stats.data_ptr() + spec.blocks * c * 2 * 8
    _check(lib().vatl_conv3x3_winograd_f4_fwd_bnbwd(_ptr(x), _ptr(u), _ptr(residual), _ptr(y), n, h, w, cin, cout, _ptr(spec.z), _ptr(spec.mask_y), _ptr(spec.scale),
                                                    _ptr(spec.bias), _ptr(spec.mean), _ptr(spec.invstd), stats_ptr, C.addressof(used), _stream()),
           "vatl_conv3x3_winograd_f4_fwd_bnbwd")
    spec.blocks += used.value
    return y


def conv1x1_rows_supported(k1: int, k2: int, n: int, m: int) -> bool:
    return bool(lib().vatl_conv1x1_rows_supported(k1, k2, n, m))


def conv1x1_rows_fwd(a, w, scale, bias, cout: int, relu: bool, residual=None, x2=None, out=None):
    """relu?(scale * (A W^T) + bias + residual) for K = 128 input channels (a (N,H,W,128), or a (N,H,W,64) + x2 (N,H,W,64)); w [cout][128] packed."""
    n, h, w_, k1 = a.shape
    k2 = 0 if x2 is None else x2.shape[-1]
    assert w.numel() >= cout * (k1 + k2) and (x2 is None or tuple(x2.shape[:3]) == (n, h, w_)) and (residual is None or tuple(residual.shape) == (n, h, w_, cout))
    y = out if out is not None else torch.empty((n, h, w_, cout), device=a.device, dtype=torch.float32)
    assert a.is_contiguous() and y.is_contiguous() and (x2 is None or x2.is_contiguous()) and (residual is None or residual.is_contiguous())
    _check(lib().vatl_conv1x1_rows_fwd(_ptr(a), _ptr(x2), _ptr(w), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), n * h * w_, k1, k2, cout, int(relu),
                                       _stream()), "vatl_conv1x1_rows_fwd")
    return y


def bottleneck_chain_supported(cmid: int, cout: int, cnext: int, m: int) -> bool:
    return bool(lib().vatl_bottleneck_chain_supported(cmid, cout, cnext, m))


def bottleneck_chain_fwd(a, w3, scale3, bias3, skip, w1=None, scale1=None, bias1=None, out=None, y1_out=None):
    """t = relu(bn3(conv3(a)) + skip) and, when w1 is given, y1 = relu(bn1(conv1_next(t))) in one launch: a (N,H,W,Cmid), skip (N,H,W,Cout) NHWC,
    w3 / w1 = packed 1x1 filters (pack_conv_weight).  Returns (t, y1) (y1 None without w1)."""
    n, h, w_, cmid = a.shape
    cout = w3.shape[0]
    cnext = 0 if w1 is None else w1.shape[0]
    assert w3.numel() == cout * cmid and (w1 is None or w1.numel() == cnext * cout) and (skip is None or tuple(skip.shape) == (n, h, w_, cout))
    t = out if out is not None else torch.empty((n, h, w_, cout), device=a.device, dtype=torch.float32)
    assert t.is_contiguous() and a.is_contiguous() and (skip is None or skip.is_contiguous())
    y1 = None if w1 is None else (y1_out if y1_out is not None else torch.empty((n, h, w_, cnext), device=a.device, dtype=torch.float32))
    _check(lib().vatl_bottleneck_chain_fwd(_ptr(a), _ptr(w3), _ptr(scale3), _ptr(bias3), _ptr(skip), _ptr(t), _ptr(w1), _ptr(scale1), _ptr(bias1), _ptr(y1),
                                           n * h * w_, cmid, cout, cnext, _stream()), "vatl_bottleneck_chain_fwd")
    return t, y1


def pack_deconv_weight(w: torch.Tensor) -> torch.Tensor:
    cin, cout, kh, kw = w.shape
    if (kh, kw) != (4, 4):
        raise VatlError("only ConvTranspose2d(4, 2, 1) is supported")
    cpad = conv_cout_pad(cout)
    plan, key = getattr(_tls, "pack_plan", None), ("deconv", w.data_ptr(), tuple(w.shape))
    if plan is not None and w.is_contiguous():
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    out = torch.empty((4, cpad, 2, 2, cin), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_deconv4x4s2_weight(_ptr(w.contiguous()), _ptr(out), cin, cout, cpad, _stream()), "vatl_pack_deconv4x4s2_weight")
    if plan is not None and w.is_contiguous():
        plan.record(key, w, out, (2, cout, cin, 4, 4, cpad, 0, 0, ()))
    return out


def bn_fold(gamma, beta, mean, var, eps: float, conv_bias=None, channels: int | None = None):
    c = channels if channels is not None else (var.numel() if var is not None else conv_bias.numel())
    dev = (var if var is not None else conv_bias).device
    scale = torch.empty(c, device=dev, dtype=torch.float32)
    bias = torch.empty(c, device=dev, dtype=torch.float32)
    _check(lib().vatl_bn_fold(_ptr(gamma), _ptr(beta), _ptr(mean), _ptr(var), _ptr(conv_bias), eps, _ptr(scale), _ptr(bias), c, _stream()), "vatl_bn_fold")
    return scale, bias


# ----------------------------------------------------------------------------
# backbone ops (NHWC)
# ----------------------------------------------------------------------------

def conv2d_fwd(x, w_packed, scale, bias, cout: int, r: int, s: int, stride: int, pad: int, relu: bool,
               residual=None, out_nchw: bool = False, out=None):
    n, h, w, cin = x.shape
    ho = (h + 2 * pad - r) // stride + 1
    wo = (w + 2 * pad - s) // stride + 1
    shape = (n, cout, ho, wo) if out_nchw else (n, ho, wo, cout)
    y = out if out is not None else torch.empty(shape, device=x.device, dtype=torch.float32)
    _check(lib().vatl_conv2d_fwd(_ptr(x), _ptr(w_packed), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), n, h, w, cin, cout,
                                 w_packed.shape[0], r, s, stride, pad, int(relu), int(out_nchw), _stream()), "vatl_conv2d_fwd")
    return y


def pack_stem_pool_weight(w: torch.Tensor) -> torch.Tensor:
    """(64,3,7,7) OIHW stem filter -> the fragment order of vatl_stem7x7s2_pool_fwd."""
    if tuple(w.shape) != (64, 3, 7, 7):
        raise VatlError(f"pack_stem_pool_weight: expected a (64,3,7,7) filter, got {tuple(w.shape)}")
    w = w.detach().float().contiguous()
    out = torch.empty(int(lib().vatl_stem_pool_weight_floats()), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_stem_pool_weight(_ptr(w), _ptr(out), _stream()), "vatl_pack_stem_pool_weight")
    return out


def pack_stem3_weight(w: torch.Tensor) -> torch.Tensor:
    """(64,3,3,3) OIHW filter of HRNet's conv1 -> the fragment order of vatl_stem3x3s2_fwd."""
    if tuple(w.shape) != (64, 3, 3, 3):
        raise VatlError(f"pack_stem3_weight: expected a (64,3,3,3) filter, got {tuple(w.shape)}")
    w = w.detach().float().contiguous()
    out = torch.empty(int(lib().vatl_stem3_weight_floats()), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_stem3_weight(_ptr(w), _ptr(out), _stream()), "vatl_pack_stem3_weight")
    return out


def stem3_fwd(x_nchw: torch.Tensor, w_packed: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """NCHW crops (N,3,H,W) -> conv3x3/2 + folded BN + ReLU -> NHWC (N,H/2,W/2,64) in one launch (hrnet.py:109-110, 426-428)."""
    n, c, h, w = x_nchw.shape
    if c != 3:
        raise VatlError(f"stem3_fwd: expected 3 input channels, got {c}")
    y = torch.empty((n, h // 2, w // 2, 64), device=x_nchw.device, dtype=torch.float32)
    _check(lib().vatl_stem3x3s2_fwd(_ptr(x_nchw), _ptr(w_packed), _ptr(scale), _ptr(bias), _ptr(y), n, h, w, _stream()), "vatl_stem3x3s2_fwd")
    return y


def stem_pool_supported(h: int, w: int) -> bool:
    return bool(lib().vatl_stem_pool_supported(int(h), int(w)))


def stem_pool_fwd(x_nchw: torch.Tensor, w_packed: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, out=None) -> torch.Tensor:
    """NCHW crops (N,3,H,W) -> conv7x7/2 + folded BN + ReLU + maxpool3x3/2 -> NHWC (N,H/4,W/4,64) in one launch (Resnet.py:155-158, 171-172)."""
    n, c, h, w = x_nchw.shape
    if c != 3:
        raise VatlError(f"stem_pool_fwd: expected 3 input channels, got {c}")
    y = out if out is not None else torch.empty((n, h // 4, w // 4, 64), device=x_nchw.device, dtype=torch.float32)
    _check(lib().vatl_stem7x7s2_pool_fwd(_ptr(x_nchw), _ptr(w_packed), _ptr(scale), _ptr(bias), _ptr(y), n, h, w, _stream()), "vatl_stem7x7s2_pool_fwd")
    return y


def pack_winograd_weight(w: torch.Tensor, data_gradient: bool = False) -> torch.Tensor:
    """(Cout,Cin,3,3) -> the F(2x2,3x3) filter transform G g G^T in MFMA fragment order (csrc/conv_winograd.hip).  With
    ``data_gradient`` the result is the filter of dX = conv(dY, rot180(w)^T): its output channels are the forward Cin."""
    co, ci = w.shape[:2]
    if tuple(w.shape[2:]) != (3, 3):
        raise VatlError("pack_winograd_weight: 3x3 filters only")
    cout, cin = (ci, co) if data_gradient else (co, ci)
    plan, key = getattr(_tls, "pack_plan", None), ("winograd", w.data_ptr(), tuple(w.shape), bool(data_gradient))
    if plan is not None and w.is_contiguous():
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    out = torch.empty(int(lib().vatl_winograd_weight_floats(cout, cin)), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_winograd_weight(_ptr(w.contiguous()), _ptr(out), cout, cin, int(data_gradient), _stream()), "vatl_pack_winograd_weight")
    if plan is not None and w.is_contiguous():
        pad = int(lib().vatl_winograd_cout_pad(cout))
        plan.record(key, w, out, (4 if data_gradient else 3, cout, cin, 3, 3, pad, 1 if pad <= 32 else 2, ci, ()))
    return out


def pack_winograd_deconv_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(4,2,1) weight (Cin,Cout,4,4) -> the four sub-pixel phase filters in the F(3x3,2x2) transform domain,
    MFMA fragment order (csrc/conv_winograd.hip)."""
    cin, cout = w.shape[:2]
    if tuple(w.shape[2:]) != (4, 4):
        raise VatlError("pack_winograd_deconv_weight: 4x4 filters only")
    plan, key = getattr(_tls, "pack_plan", None), ("winograd_deconv", w.data_ptr(), tuple(w.shape))
    if plan is not None and w.is_contiguous():
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    out = torch.empty(int(lib().vatl_winograd_deconv_weight_floats(cout, cin)), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_winograd_deconv_weight(_ptr(w.contiguous()), _ptr(out), cout, cin, _stream()), "vatl_pack_winograd_deconv_weight")
    if plan is not None and w.is_contiguous():
        pad = int(lib().vatl_winograd_cout_pad(cout))
        plan.record(key, w, out, (5, cout, cin, 4, 4, pad, 1 if pad <= 32 else 2, cout, ()))
    return out


def pack_winograd_deconv_dgrad_weight(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(4,2,1) weight (Cin,Cout,4,4) -> the filters of its DATA gradient (a 4x4 / stride 2 conv over dz = four pixel
    phases x 2x2 convolutions) in the F(3x3,2x2) transform domain."""
    cin, cout = w.shape[:2]
    if tuple(w.shape[2:]) != (4, 4):
        raise VatlError("pack_winograd_deconv_dgrad_weight: 4x4 filters only")
    plan, key = getattr(_tls, "pack_plan", None), ("winograd_deconv_dgrad", w.data_ptr(), tuple(w.shape))
    if plan is not None and w.is_contiguous():
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    out = torch.empty(int(lib().vatl_winograd_deconv_dgrad_weight_floats(cin, cout)), device=w.device, dtype=torch.float32)
    _check(lib().vatl_pack_winograd_deconv_dgrad_weight(_ptr(w.contiguous()), _ptr(out), cin, cout, _stream()), "vatl_pack_winograd_deconv_dgrad_weight")
    if plan is not None and w.is_contiguous():
        pad = int(lib().vatl_winograd_cout_pad(cin))
        plan.record(key, w, out, (6, cin, cout, 4, 4, pad, 1 if pad <= 32 else 2, cout, ()))
    return out


def deconv4x4s2_winograd_dgrad(dz, u_packed, cin: int, spec: "BnBwdSpec" = None, residual=None, out=None):
    """dx (N,H,W,Cin) of ConvTranspose2d(4,2,1) from dz (N,2H,2W,Cout); with ``spec`` the epilogue masks the result with the consumer
    layer's ReLU and appends the (sum g, sum g*xhat) row-block partials (conv2d_fwd_ex_bnbwd semantics)."""
    n, h2, w2, cout = dz.shape
    h, w = h2 // 2, w2 // 2
    dx = out if out is not None else torch.empty((n, h, w, cin), device=dz.device, dtype=torch.float32)
    if spec is None:
        _check(lib().vatl_deconv4x4s2_winograd_dgrad(_ptr(dz), _ptr(u_packed), _ptr(residual), _ptr(dx), n, h, w, cin, cout, _stream()),
               "vatl_deconv4x4s2_winograd_dgrad")
        return dx
    if dx.shape != spec.z.shape:
        raise VatlError("deconv4x4s2_winograd_dgrad: the BatchNorm tensors must have the layout of the output")
    used = C.c_int64(0)
    c = spec.z.shape[-1]
    need = (n * ((h + 2) // 3) * ((w + 2) // 3) + 31) // 32
    if (spec.blocks + need) * c * 2 > spec.stats.numel():
        raise VatlError("deconv4x4s2_winograd_dgrad: statistics buffer too small")
    stats_ptr = spec.stats.data_ptr() + spec.blocks * c * 2 * 8
    _check(lib().vatl_deconv4x4s2_winograd_dgrad_bnbwd(_ptr(dz), _ptr(u_packed), _ptr(residual), _ptr(dx), n, h, w, cin, cout, _ptr(spec.z),
                                                       _ptr(spec.mask_y), _ptr(spec.scale), _ptr(spec.bias), _ptr(spec.mean), _ptr(spec.invstd),
                                                       stats_ptr, C.addressof(used), _stream()), "vatl_deconv4x4s2_winograd_dgrad_bnbwd")
    spec.blocks += used.value
    return dx


def deconv4x4s2_winograd_fwd(x, u_packed, scale, bias, cout: int, relu: bool, out=None):
    """ConvTranspose2d(4,2,1) of an NHWC tensor through Winograd F(3x3, 2x2) on its four sub-pixel phases."""
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, 2 * h, 2 * w, cout), device=x.device, dtype=torch.float32)
    _check(lib().vatl_deconv4x4s2_winograd_fwd(_ptr(x), _ptr(u_packed), _ptr(scale), _ptr(bias), _ptr(y), n, h, w, cin, cout, int(relu), _stream()),
           "vatl_deconv4x4s2_winograd_fwd")
    return y


def conv3x3_winograd_fwd(x, u_packed, scale, bias, cout: int, relu: bool, residual=None, out=None):
    """3x3 / stride 1 / pad 1 convolution of an NHWC tensor through Winograd F(2x2, 3x3)."""
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    _check(lib().vatl_conv3x3_winograd_fwd(_ptr(x), _ptr(u_packed), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), n, h, w, cin, cout,
                                           int(relu), _stream()), "vatl_conv3x3_winograd_fwd")
    return y


def deconv4x4s2_fwd(x, w_packed, scale, bias, cout: int, relu: bool):
    n, h, w, cin = x.shape
    y = torch.empty((n, 2 * h, 2 * w, cout), device=x.device, dtype=torch.float32)
    _check(lib().vatl_deconv4x4s2_fwd(_ptr(x), _ptr(w_packed), _ptr(scale), _ptr(bias), _ptr(y), n, h, w, cin, cout,
                                      w_packed.shape[1], int(relu), _stream()), "vatl_deconv4x4s2_fwd")
    return y


def maxpool3x3s2_fwd(x):
    n, h, w, c = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=x.device, dtype=torch.float32)
    _check(lib().vatl_maxpool3x3s2_fwd(_ptr(x), _ptr(y), n, h, w, c, _stream()), "vatl_maxpool3x3s2_fwd")
    return y


def gap_fwd(x):
    n, h, w, c = x.shape
    y = torch.empty((n, c), device=x.device, dtype=torch.float32)
    _check(lib().vatl_gap_fwd(_ptr(x), _ptr(y), n, h * w, c, _stream()), "vatl_gap_fwd")
    return y


def pixelshuffle2_fwd(x):
    n, h, w, c = x.shape
    y = torch.empty((n, 2 * h, 2 * w, c // 4), device=x.device, dtype=torch.float32)
    _check(lib().vatl_pixelshuffle2_fwd(_ptr(x), _ptr(y), n, h, w, c, _stream()), "vatl_pixelshuffle2_fwd")
    return y


def se_scale_add_relu(x, gate, residual):
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    _check(lib().vatl_se_scale_add_relu(_ptr(x), _ptr(gate), _ptr(residual), _ptr(y), n, h * w, c, _stream()), "vatl_se_scale_add_relu")
    return y


def fuse_upsample_add(base, ups, relu: bool):
    """base (N,H,W,C); ups = [(z, shift)] with z (N,H>>shift,W>>shift,C), at most 3."""
    n, h, w, c = base.shape
    y = torch.empty_like(base)
    a = list(ups) + [(None, 0)] * (3 - len(ups))
    _check(lib().vatl_fuse_upsample_add(_ptr(base), _ptr(a[0][0]), a[0][1], _ptr(a[1][0]), a[1][1], _ptr(a[2][0]), a[2][1], _ptr(y),
                                        n, h, w, c, int(relu), _stream()), "vatl_fuse_upsample_add")
    return y


def peaks5(hm: torch.Tensor, min_distance: int = 5):
    """(N,J,H,W) -> peak values (N,J,5), flat indices (N,J,5) int32 (-1 = none), counts (N,J) int32, MPE terms (N,J),
    Margin terms (N,J)   [skimage.feature.peak_local_max(min_distance, num_peaks=5) per plane]."""
    n, j, h, w = hm.shape
    val = torch.empty((n, j, 5), device=hm.device, dtype=torch.float32)
    idx = torch.empty((n, j, 5), device=hm.device, dtype=torch.int32)
    cnt = torch.empty((n, j), device=hm.device, dtype=torch.int32)
    mpe = torch.empty((n, j), device=hm.device, dtype=torch.float32)
    mar = torch.empty((n, j), device=hm.device, dtype=torch.float32)
    _check(lib().vatl_peaks5(_ptr(hm), _ptr(val), _ptr(idx, torch.int32), _ptr(cnt, torch.int32), _ptr(mpe), _ptr(mar), n, j, h, w, min_distance,
                             _stream()), "vatl_peaks5")
    return val, idx, cnt, mpe, mar


def plane_entropy(hm: torch.Tensor) -> torch.Tensor:
    n, j, h, w = hm.shape
    out = torch.empty((n, j), device=hm.device, dtype=torch.float32)
    _check(lib().vatl_plane_entropy(_ptr(hm), _ptr(out), n, j, h, w, _stream()), "vatl_plane_entropy")
    return out


def oks(pred_kpts: torch.Tensor, gt_kpts: torch.Tensor, bbox_xywh: torch.Tensor) -> torch.Tensor:
    """pred (N,17,3) fp32, gt (N,51) float64, boxes (N,4) xywh float64 -> (N,) float64 OKS."""
    n = pred_kpts.shape[0]
    out = torch.empty(n, device=pred_kpts.device, dtype=torch.float64)
    _check(lib().vatl_oks(_ptr(pred_kpts), _ptr(gt_kpts, torch.float64), _ptr(bbox_xywh, torch.float64), _ptr(out, torch.float64), n, _stream()),
           "vatl_oks")
    return out


def cosine_rowsum(emb: torch.Tensor) -> torch.Tensor:
    """(n, D) fp32 -> (n,) float64 sums of cosine distances to every row (incl. itself: 0)."""
    n, d = emb.shape
    out = torch.empty(n, device=emb.device, dtype=torch.float64)
    ws = torch.empty(n + d, device=emb.device, dtype=torch.float64)
    _check(lib().vatl_cosine_rowsum(_ptr(emb), n, d, _ptr(out, torch.float64), _ptr(ws, torch.float64), _stream()), "vatl_cosine_rowsum")
    return out


def kcenter_update(emb, centers: torch.Tensor, min_dist: torch.Tensor, first: bool):
    _check(lib().vatl_kcenter_update(_ptr(emb), emb.shape[0], emb.shape[1], _ptr(centers, torch.int32), centers.numel(), _ptr(min_dist, torch.float64),
                                     int(first), _stream()), "vatl_kcenter_update")


def kcenter_pick(min_dist, unc, a: float, b: float, selected: torch.Tensor, step: int, n: int):
    _check(lib().vatl_kcenter_pick(_ptr(min_dist, torch.float64), _ptr(unc, torch.float64), a, b, _ptr(selected, torch.int32), step, n, _stream()),
           "vatl_kcenter_pick")


def upsample_nearest_bwd(dy, yact, shift: int):
    """dy (N,H,W,C) [masked by yact > 0] -> block sums (N,H>>shift,W>>shift,C)."""
    n, h, w, c = dy.shape
    dz = torch.empty((n, h >> shift, w >> shift, c), device=dy.device, dtype=torch.float32)
    _check(lib().vatl_upsample_nearest_bwd(_ptr(dy), _ptr(yact), _ptr(dz), n, h, w, c, shift, _stream()), "vatl_upsample_nearest_bwd")
    return dz


def gap_bwd(dy, hw: int):
    n, c = dy.shape
    dx = torch.empty((n, hw, c), device=dy.device, dtype=torch.float32)
    _check(lib().vatl_gap_bwd(_ptr(dy), _ptr(dx), n, hw, c, _stream()), "vatl_gap_bwd")
    return dx


# ----------------------------------------------------------------------------
# scorers on (N,J,H,W) heat-maps
# ----------------------------------------------------------------------------

def decode(hm: torch.Tensor, bbox: torch.Tensor):
    """-> coords (N,J,2) f32, maxvals (N,J) f32, idx (N,J) i32."""
    n, j, h, w = hm.shape
    coords = torch.empty((n, j, 2), device=hm.device, dtype=torch.float32)
    maxv = torch.empty((n, j), device=hm.device, dtype=torch.float32)
    idx = torch.empty((n, j), device=hm.device, dtype=torch.int32)
    _check(lib().vatl_decode_argmax_affine(_ptr(hm), _ptr(bbox), _ptr(coords), _ptr(maxv), _ptr(idx, torch.int32), n, j, h, w, _stream()),
           "vatl_decode_argmax_affine")
    return coords, maxv, idx


def decode_pose(hm: torch.Tensor, bbox: torch.Tensor):
    """-> kpts (N,J,3) f32 rows (x, y, score), idx (N,J) i32, hp (N) = -sum of scores, pose_score (N) f64 = float32 mean + 1.25 max in float64 (numpy 1.23 promotion): the decode and the
    per-item scores of ActiveLearning.py:304-314, 329-330 in two launches (no cat / neg / sum / max / mean kernels around them)."""
    n, j, h, w = hm.shape
    kpts = torch.empty((n, j, 3), device=hm.device, dtype=torch.float32)
    idx = torch.empty((n, j), device=hm.device, dtype=torch.int32)
    hp = torch.empty((n,), device=hm.device, dtype=torch.float32)
    ps = torch.empty((n,), device=hm.device, dtype=torch.float64)
    _check(lib().vatl_decode_pose(_ptr(hm), _ptr(bbox), _ptr(kpts), _ptr(idx, torch.int32), _ptr(hp), _ptr(ps, torch.float64), n, j, h, w, _stream()), "vatl_decode_pose")
    return kpts, idx, hp, ps


def thc_pairs(a: torch.Tensor, b: torch.Tensor, norm: str = "L1") -> torch.Tensor:
    """a, b (P,J,H,W) (views with a uniform item stride are fine) -> (P,) f32."""
    p, j, h, w = a.shape
    for t in (a, b):
        if not t.is_cuda or t.dtype != torch.float32 or t[0].is_contiguous() is False:
            raise VatlError("thc_pairs: fp32 device tensors with contiguous items required")
    out = torch.empty(p, device=a.device, dtype=torch.float32)
    sa = a.stride(0) if p > 1 else j * h * w
    sb = b.stride(0) if p > 1 else j * h * w
    _check(lib().vatl_thc_pairs(a.data_ptr(), b.data_ptr(), sa, sb, _ptr(out), p, j, h * w, {"L1": 1, "L2": 2}[norm], _stream()), "vatl_thc_pairs")
    return out


def thc_stream(hm: torch.Tensor, is_prev: torch.Tensor, is_next: torch.Tensor, norm: str = "L1") -> torch.Tensor:
    """THC of an id-sorted, de-duplicated stream: neighbours are items i-1 / i+1."""
    n = hm.shape[0]
    thc = torch.empty(n, device=hm.device, dtype=torch.float32)
    pair = thc_pairs(hm[:-1], hm[1:], norm) if n > 1 else None
    _check(lib().vatl_thc_combine(_ptr(pair), _ptr(is_prev, torch.uint8), _ptr(is_next, torch.uint8), _ptr(thc), n, _stream()), "vatl_thc_combine")
    return thc


def localpeak_mean(hm: torch.Tensor, order: float = 0.5):
    """-> mean (N,) f32 (nan when no peak is kept), count (N,J) i32."""
    n, j, h, w = hm.shape
    mean = torch.empty(n, device=hm.device, dtype=torch.float32)
    cnt = torch.empty((n, j), device=hm.device, dtype=torch.int32)
    ws = torch.empty(2 * n * j, device=hm.device, dtype=torch.float64)
    _check(lib().vatl_localpeak_mean(_ptr(hm), _ptr(mean), _ptr(cnt, torch.int32), _ptr(ws, torch.float64), n, j, h, w, order, _stream()),
           "vatl_localpeak_mean")
    return mean, cnt


def pack_ae(state_dict, device) -> torch.Tensor:
    parts = []
    for half in ("encoder", "decoder"):
        for i in (0, 2, 4, 6):
            parts.append(state_dict[f"{half}.{i}.weight"].detach().reshape(-1).float())
            parts.append(state_dict[f"{half}.{i}.bias"].detach().reshape(-1).float())
    return torch.cat([p.to(device) for p in parts]).contiguous()


def hybrid_ae_wpu(kpts: torch.Tensor, bbox: torch.Tensor, ae_flat: torch.Tensor, d: int, z: int, only38: bool = False):
    """kpts (N,17,3), bbox (N,4) crop xyxy -> wpu (N,) f32, status (N,) i32."""
    n = kpts.shape[0]
    wpu = torch.empty(n, device=kpts.device, dtype=torch.float32)
    status = torch.empty(n, device=kpts.device, dtype=torch.int32)
    _check(lib().vatl_hybrid_ae_wpu(_ptr(kpts), _ptr(bbox), _ptr(ae_flat), d, z, int(only38), _ptr(wpu), _ptr(status, torch.int32), n, _stream()),
           "vatl_hybrid_ae_wpu")
    return wpu, status


def tpc_stream(hm: torch.Tensor, bbox: torch.Tensor, cur_coords: torch.Tensor, is_prev: torch.Tensor, is_next: torch.Tensor) -> torch.Tensor:
    """TPC of an id-sorted de-duplicated stream (neighbours = items i-1 / i+1, decoded with item i's box)."""
    n, j = hm.shape[:2]
    adj_prev = torch.zeros((n, j, 2), device=hm.device, dtype=torch.float32)
    adj_next = torch.zeros((n, j, 2), device=hm.device, dtype=torch.float32)
    if n > 1:
        h, w = hm.shape[2:]
        scratch = torch.empty((n - 1, j), device=hm.device, dtype=torch.float32)
        _check(lib().vatl_decode_argmax_affine(hm[:-1].data_ptr(), bbox[1:].data_ptr(), adj_prev[1:].data_ptr(), _ptr(scratch), None,
                                               n - 1, j, h, w, _stream()), "vatl_decode_argmax_affine")
        _check(lib().vatl_decode_argmax_affine(hm[1:].data_ptr(), bbox[:-1].data_ptr(), adj_next[:-1].data_ptr(), _ptr(scratch), None,
                                               n - 1, j, h, w, _stream()), "vatl_decode_argmax_affine")
    out = torch.empty(n, device=hm.device, dtype=torch.float32)
    _check(lib().vatl_tpc_stream(_ptr(cur_coords), _ptr(adj_prev), _ptr(adj_next), _ptr(bbox), _ptr(is_prev, torch.uint8),
                                 _ptr(is_next, torch.uint8), _ptr(out), n, j, _stream()), "vatl_tpc_stream")
    return out


def decode_softargmax(hm: torch.Tensor, bbox: torch.Tensor, norm_type: str = "softmax"):
    n, j, h, w = hm.shape
    coords = torch.empty((n, j, 2), device=hm.device, dtype=torch.float32)
    scores = torch.empty((n, j), device=hm.device, dtype=torch.float32)
    code = {"softmax": 0, "sigmoid": 1, "divide_sum": 2}.get(norm_type)
    if code is None:
        raise NotImplementedError(norm_type)
    _check(lib().vatl_decode_softargmax(_ptr(hm), _ptr(bbox), _ptr(coords), _ptr(scores), n, j, h, w, code, _stream()), "vatl_decode_softargmax")
    return coords, scores


def ae_forward(feat: torch.Tensor, ae_flat: torch.Tensor, d: int, z: int):
    """feat (N,D) -> recon (N,D), mse (N,)."""
    n = feat.shape[0]
    recon = torch.empty((n, d), device=feat.device, dtype=torch.float32)
    mse = torch.empty(n, device=feat.device, dtype=torch.float32)
    _check(lib().vatl_ae_forward(_ptr(feat), _ptr(ae_flat), d, z, _ptr(recon), _ptr(mse), n, _stream()), "vatl_ae_forward")
    return recon, mse


def hybrid_feature_f64(kpts: torch.Tensor, bbox_xywh: torch.Tensor):
    """kpts (N,51) f64, bbox (N,4) f64 xywh -> feat (N,42) f64, status (N,) i32."""
    n = kpts.shape[0]
    feat = torch.empty((n, 42), device=kpts.device, dtype=torch.float64)
    status = torch.empty(n, device=kpts.device, dtype=torch.int32)
    _check(lib().vatl_hybrid_feature_f64(_ptr(kpts, torch.float64), _ptr(bbox_xywh, torch.float64), _ptr(feat, torch.float64),
                                         _ptr(status, torch.int32), n, _stream()), "vatl_hybrid_feature_f64")
    return feat, status


def localpeak_mask(hm: torch.Tensor, order: float = 0.5) -> torch.Tensor:
    """hm (..., H, W) -> uint8 mask of kept local peaks, same shape."""
    h, w = hm.shape[-2:]
    planes = hm.numel() // (h * w)
    mask = torch.empty(hm.shape, device=hm.device, dtype=torch.uint8)
    _check(lib().vatl_localpeak_mask(_ptr(hm), _ptr(mask, torch.uint8), planes, h, w, order, _stream()), "vatl_localpeak_mask")
    return mask


# ----------------------------------------------------------------------------
# training-mode backbone ops (NHWC)
# ----------------------------------------------------------------------------

def conv2d_fwd_ex(x, w_packed, cout, r, s, stride, pad_y, pad_x, ho, wo, oh, ow, osy, osx, ooy, oox, out=None, residual=None,
                  scale=None, bias=None, relu=False):
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, oh, ow, cout), device=x.device, dtype=torch.float32)
    _check(lib().vatl_conv2d_fwd_ex(_ptr(x), _ptr(w_packed), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), n, h, w, cin, cout,
                                    w_packed.shape[0], r, s, stride, pad_y, pad_x, ho, wo, oh, ow, osy, osx, ooy, oox, int(relu), _stream()),
           "vatl_conv2d_fwd_ex")
    return y


class BnBwdSpec:
    """What a data-gradient launch needs to run the reduction pass of the consumer layer's BatchNorm backward in its
    epilogue: z (the consumer's conv output), its ReLU mask source (mask_y, or (scale, bias) to recompute it from z, or
    neither) and the saved batch statistics.  ``stats`` / ``blocks`` collect the partial sums of one or more launches."""

    def __init__(self, z, mean, invstd, mask_y=None, scale=None, bias=None):
        self.z, self.mean, self.invstd, self.mask_y, self.scale, self.bias = z, mean, invstd, mask_y, scale, bias
        c = z.shape[-1]
        cap = int(lib().vatl_conv_stats_row_blocks(z.numel() // c, 1)) + 8     # + a partial tile per parity launch
        self.stats = torch.empty(cap * c * 2, device=z.device, dtype=torch.float64)
        self.blocks = 0


def conv2d_fwd_ex_bnbwd(x, w_packed, cout, r, s, stride, pad_y, pad_x, ho, wo, oh, ow, osy, osx, ooy, oox, spec: BnBwdSpec, out=None, residual=None):
    """conv2d_fwd_ex whose epilogue masks the result with the consumer layer's ReLU and appends the (sum g, sum g*xhat) row-block
    partials to ``spec`` (several launches — the parity launches of a strided conv's data gradient — append one after another)."""
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, oh, ow, cout), device=x.device, dtype=torch.float32)
    if y.shape != spec.z.shape:
        raise VatlError("conv2d_fwd_ex_bnbwd: the BatchNorm tensors must have the layout of the output")
    used = C.c_int64(0)
    c = spec.z.shape[-1]
    stats_ptr = spec.stats.data_ptr() + spec.blocks * c * 2 * 8
    need = int(lib().vatl_conv_stats_row_blocks(n * ho * wo, 1))
    if (spec.blocks + need) * c * 2 > spec.stats.numel():
        raise VatlError("conv2d_fwd_ex_bnbwd: statistics buffer too small")
    _check(lib().vatl_conv2d_fwd_ex_bnbwd(_ptr(x), _ptr(w_packed), _ptr(residual), _ptr(y), n, h, w, cin, cout, w_packed.shape[0], r, s, stride, pad_y,
                                          pad_x, ho, wo, oh, ow, osy, osx, ooy, oox, _ptr(spec.z), _ptr(spec.mask_y), _ptr(spec.scale), _ptr(spec.bias),
                                          _ptr(spec.mean), _ptr(spec.invstd), stats_ptr, C.addressof(used), _stream()), "vatl_conv2d_fwd_ex_bnbwd")
    spec.blocks += used.value
    return y


def bn_bwd_from_stats(spec: BnBwdSpec, g, gamma, dgamma=None, dbeta=None):
    """Finish the BatchNorm backward whose reduction ran in the data-gradient epilogue: -> dz, dgamma, dbeta."""
    z = spec.z
    c = z.shape[-1]
    m = z.numel() // c
    dz = torch.empty_like(z)
    dgamma = dgamma if dgamma is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    dbeta = dbeta if dbeta is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    coef = torch.empty(3 * c, device=z.device, dtype=torch.float32)
    _check(lib().vatl_bn_bwd_from_stats(_ptr(spec.stats, torch.float64), spec.blocks, _ptr(g), _ptr(z), _ptr(gamma), _ptr(spec.mean), _ptr(spec.invstd),
                                        _ptr(dz), _ptr(dgamma), _ptr(dbeta), m, c, _ptr(coef), _stream()), "vatl_bn_bwd_from_stats")
    return dz, dgamma, dbeta


def pack_dgrad_weight(w: torch.Tensor, taps, cout_k: int | None = None) -> torch.Tensor:
    """(Cout,Cin,R,S) -> [CinPad][len(taps)][CoutK] with out[c][t][n] = w[n][c][taps[t]]."""
    cout, cin, r, s = w.shape
    cinpad = conv_cout_pad(cin)
    cout_k = cout_k or cout
    taps = [(int(a), int(b)) for a, b in taps]
    plan, key = getattr(_tls, "pack_plan", None), ("dgrad", w.data_ptr(), tuple(w.shape), tuple(taps), cout_k)
    if plan is not None and w.is_contiguous() and len(taps) <= 16:
        kept = plan.lookup(key, w)
        if kept is not None:
            return kept
    out = torch.empty((cinpad, len(taps), cout_k), device=w.device, dtype=torch.float32)
    if plan is not None and w.is_contiguous() and len(taps) <= 16:
        plan.record(key, w, out, (1, cout, cin, r, s, cinpad, cout_k, len(taps), tuple(taps)))
    tr = (C.c_int * len(taps))(*[t[0] for t in taps])
    ts = (C.c_int * len(taps))(*[t[1] for t in taps])
    _check(lib().vatl_pack_dgrad_weight(_ptr(w.contiguous()), _ptr(out), cout, cin, r, s, cinpad, cout_k, len(taps),
                                        C.cast(tr, C.c_void_p), C.cast(ts, C.c_void_p), _stream()), "vatl_pack_dgrad_weight")
    return out


def conv2d_wgrad(x, dz, cout: int, cin: int, r: int, s: int, stride: int, pad: int, out=None) -> torch.Tensor:
    """x NHWC (N,H,W,Cin or 4 for the stem), dz NHWC (N,Ho,Wo,CoutG) -> dw (Cout,Cin,R,S) (written into ``out`` when
    given: a slice of the flat gradient arena)."""
    n, h, w, _ = x.shape
    dw = out if out is not None else torch.empty((cout, cin, r, s), device=x.device, dtype=torch.float32)
    if dw.numel() != cout * cin * r * s:
        raise VatlError("conv2d_wgrad: out has the wrong size")
    m = n * dz.shape[1] * dz.shape[2]
    ws = torch.empty(int(lib().vatl_conv2d_wgrad_workspace_floats(cout, cin, r, s, m)), device=x.device, dtype=torch.float32)
    _check(lib().vatl_conv2d_wgrad(_ptr(x), _ptr(dz), _ptr(dw), _ptr(ws), n, h, w, cin, cout, dz.shape[3], r, s, stride, pad, _stream()),
           "vatl_conv2d_wgrad")
    return dw


def conv3x3_winograd_wgrad(x, dz, out=None) -> torch.Tensor:
    """Weight gradient of a 3x3 / stride 1 / pad 1 conv on the Winograd route: x (N,H,W,Cin), dz (N,H,W,Cout) -> dw (Cout,Cin,3,3)."""
    n, h, w, cin = x.shape
    cout = dz.shape[3]
    dw = out if out is not None else torch.empty((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    if dw.numel() != cout * cin * 9:
        raise VatlError("conv3x3_winograd_wgrad: out has the wrong size")
    ws = torch.empty(int(lib().vatl_conv3x3_winograd_wgrad_workspace_floats(cout, cin, n, h, w)), device=x.device, dtype=torch.float32)
    _check(lib().vatl_conv3x3_winograd_wgrad(_ptr(x), _ptr(dz), _ptr(dw), _ptr(ws), n, h, w, cin, cout, _stream()), "vatl_conv3x3_winograd_wgrad")
    return dw


def deconv4x4s2_winograd_wgrad(x, dy, out=None) -> torch.Tensor:
    """Weight gradient of ConvTranspose2d(4,2,1) on the Winograd route: x (N,H,W,Cin), dy (N,2H,2W,Cout) -> dw (Cin,Cout,4,4)."""
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    dw = out if out is not None else torch.empty((cin, cout, 4, 4), device=x.device, dtype=torch.float32)
    if dw.numel() != cin * cout * 16:
        raise VatlError("deconv4x4s2_winograd_wgrad: out has the wrong size")
    ws = torch.empty(int(lib().vatl_deconv4x4s2_winograd_wgrad_workspace_floats(cin, cout, n, h, w)), device=x.device, dtype=torch.float32)
    _check(lib().vatl_deconv4x4s2_winograd_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), n, h, w, cin, cout, _stream()), "vatl_deconv4x4s2_winograd_wgrad")
    return dw


def deconv4x4s2_wgrad(x, dy, out=None) -> torch.Tensor:
    n, h, w, cin = x.shape
    cout = dy.shape[3]
    dw = out if out is not None else torch.empty((cin, cout, 4, 4), device=x.device, dtype=torch.float32)
    if dw.numel() != cin * cout * 16:
        raise VatlError("deconv4x4s2_wgrad: out has the wrong size")
    ws = torch.empty(int(lib().vatl_deconv4x4s2_wgrad_workspace_floats(cin, cout, n * h * w)), device=x.device, dtype=torch.float32)
    _check(lib().vatl_deconv4x4s2_wgrad(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), n, h, w, cin, cout, _stream()), "vatl_deconv4x4s2_wgrad")
    return dw


def _col_ws(m: int, c: int, device):
    return torch.empty(int(lib().vatl_col_reduce_workspace_doubles(m, c)), device=device, dtype=torch.float64)


def bn_train_fwd_stats(z, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """z NHWC (..., C) -> save_mean, save_invstd, scale, bias (C,); running stats updated in place."""
    c = z.shape[-1]
    m = z.numel() // c
    outs = [torch.empty(c, device=z.device, dtype=torch.float32) for _ in range(4)]
    _check(lib().vatl_bn_train_fwd_stats(_ptr(z), m, c, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), momentum, eps,
                                         _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), _ptr(_col_ws(m, c, z.device), torch.float64),
                                         _stream()), "vatl_bn_train_fwd_stats")
    return outs


def _bn_finalize(stats, nblk, m, c, bn_args, device):
    gamma, beta, running_mean, running_var, momentum, eps = bn_args
    outs = [torch.empty(c, device=device, dtype=torch.float32) for _ in range(4)]
    _check(lib().vatl_bn_train_finalize(_ptr(stats, torch.float64), nblk, m, c, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                        momentum, eps, _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), _stream()), "vatl_bn_train_finalize")
    return outs


def conv2d_fwd_bnstats(x, w_packed, cout: int, r: int, s: int, stride: int, pad: int, gamma, beta, running_mean, running_var,
                       momentum: float, eps: float):
    """Training forward: z = conv(x) with the BatchNorm batch statistics taken in the conv epilogue.
    -> z, save_mean, save_invstd, scale, bias; running stats updated in place."""
    n, h, w, cin = x.shape
    ho, wo = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1
    z = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    m = n * ho * wo
    stats = torch.empty(int(lib().vatl_conv_stats_row_blocks(m, 1)) * cout * 2, device=x.device, dtype=torch.float64)
    used = C.c_int64(0)
    _check(lib().vatl_conv2d_fwd_stats(_ptr(x), _ptr(w_packed), _ptr(z), _ptr(stats, torch.float64), C.addressof(used), n, h, w, cin, cout,
                                       w_packed.shape[0], r, s, stride, pad, _stream()), "vatl_conv2d_fwd_stats")
    return [z] + _bn_finalize(stats, used.value, m, cout, (gamma, beta, running_mean, running_var, momentum, eps), x.device)


def conv3x3_winograd_fwd_stats(x, u_packed, cout: int):
    """Training forward of a 3x3 / stride-1 layer through Winograd: z = conv(x) and the row-block (sum, sum^2) partials of z.
    -> z, stats (float64), row blocks written."""
    n, h, w, cin = x.shape
    z = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    stats = torch.empty(int(lib().vatl_winograd_stats_row_blocks(n, h, w)) * cout * 2, device=x.device, dtype=torch.float64)
    used = C.c_int64(0)
    _check(lib().vatl_conv3x3_winograd_fwd_stats(_ptr(x), _ptr(u_packed), _ptr(z), _ptr(stats, torch.float64), C.addressof(used), n, h, w, cin,
                                                 cout, _stream()), "vatl_conv3x3_winograd_fwd_stats")
    return z, stats, used.value


def conv3x3_winograd_fwd_bnbwd(x, u_packed, cout: int, spec: "BnBwdSpec", out=None, residual=None):
    """conv2d_fwd_ex_bnbwd for the 3x3 / stride-1 data gradients on the Winograd route (u_packed: data-gradient packing)."""
    n, h, w, cin = x.shape
    y = out if out is not None else torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    if y.shape != spec.z.shape:
        raise VatlError("conv3x3_winograd_fwd_bnbwd: the BatchNorm tensors must have the layout of the output")
    used = C.c_int64(0)
    c = spec.z.shape[-1]
    need = int(lib().vatl_winograd_stats_row_blocks(n, h, w))
    if (spec.blocks + need) * c * 2 > spec.stats.numel():
        raise VatlError("conv3x3_winograd_fwd_bnbwd: statistics buffer too small")
    stats_ptr = spec.stats.data_ptr() + spec.blocks * c * 2 * 8
    _check(lib().vatl_conv3x3_winograd_fwd_bnbwd(_ptr(x), _ptr(u_packed), _ptr(residual), _ptr(y), n, h, w, cin, cout, _ptr(spec.z), _ptr(spec.mask_y),
                                                 _ptr(spec.scale), _ptr(spec.bias), _ptr(spec.mean), _ptr(spec.invstd), stats_ptr, C.addressof(used),
                                                 _stream()), "vatl_conv3x3_winograd_fwd_bnbwd")
    spec.blocks += used.value
    return y


def deconv4x4s2_winograd_fwd_bnstats(x, u_packed, cout: int, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """deconv4x4s2_fwd_bnstats on the Winograd route: -> z, save_mean, save_invstd, scale, bias; running stats updated in place."""
    n, h, w, cin = x.shape
    z = torch.empty((n, 2 * h, 2 * w, cout), device=x.device, dtype=torch.float32)
    stats = torch.empty(int(lib().vatl_winograd_deconv_stats_row_blocks(n, h, w)) * cout * 2, device=x.device, dtype=torch.float64)
    used = C.c_int64(0)
    _check(lib().vatl_deconv4x4s2_winograd_fwd_stats(_ptr(x), _ptr(u_packed), _ptr(z), _ptr(stats, torch.float64), C.addressof(used), n, h, w, cin,
                                                     cout, _stream()), "vatl_deconv4x4s2_winograd_fwd_stats")
    return [z] + _bn_finalize(stats, used.value, z.numel() // cout, cout, (gamma, beta, running_mean, running_var, momentum, eps), x.device)


def conv3x3_winograd_fwd_bnstats(x, u_packed, cout: int, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """conv2d_fwd_bnstats for the Winograd route: -> z, save_mean, save_invstd, scale, bias; running stats updated in place."""
    z, stats, used = conv3x3_winograd_fwd_stats(x, u_packed, cout)
    return [z] + _bn_finalize(stats, used, z.numel() // cout, cout, (gamma, beta, running_mean, running_var, momentum, eps), x.device)


def deconv4x4s2_fwd_bnstats(x, w_packed, cout: int, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    n, h, w, cin = x.shape
    z = torch.empty((n, 2 * h, 2 * w, cout), device=x.device, dtype=torch.float32)
    stats = torch.empty(int(lib().vatl_conv_stats_row_blocks(n * h * w, 4)) * cout * 2, device=x.device, dtype=torch.float64)
    used = C.c_int64(0)
    _check(lib().vatl_deconv4x4s2_fwd_stats(_ptr(x), _ptr(w_packed), _ptr(z), _ptr(stats, torch.float64), C.addressof(used), n, h, w, cin, cout,
                                            w_packed.shape[1], _stream()), "vatl_deconv4x4s2_fwd_stats")
    return [z] + _bn_finalize(stats, used.value, 4 * n * h * w, cout, (gamma, beta, running_mean, running_var, momentum, eps), x.device)


def bn_train_bwd_relu(dy, scale, bias, z, gamma, save_mean, save_invstd, dgamma=None, dbeta=None):
    """Backward of Conv+BN+ReLU without a skip input; the ReLU mask is recomputed from z. -> dz, dgamma, dbeta."""
    c = z.shape[-1]
    m = z.numel() // c
    dz = torch.empty_like(z)
    dgamma = dgamma if dgamma is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    dbeta = dbeta if dbeta is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    coef = torch.empty(3 * c, device=z.device, dtype=torch.float32)
    _check(lib().vatl_bn_train_bwd_relu(_ptr(dy), _ptr(scale), _ptr(bias), _ptr(z), _ptr(gamma), _ptr(save_mean), _ptr(save_invstd), _ptr(dz),
                                        _ptr(dgamma), _ptr(dbeta), m, c, _ptr(coef), _ptr(_col_ws(m, c, z.device), torch.float64), _stream()),
           "vatl_bn_train_bwd_relu")
    return dz, dgamma, dbeta


def scale_bias_act(z, scale, bias, residual=None, relu=True):
    c = z.shape[-1]
    y = torch.empty_like(z)
    _check(lib().vatl_scale_bias_act(_ptr(z), _ptr(scale), _ptr(bias), _ptr(residual), _ptr(y), z.numel() // c, c, int(relu), _stream()),
           "vatl_scale_bias_act")
    return y


def bn_train_bwd(dy, y, z, gamma, save_mean, save_invstd, want_g: bool = False, dgamma=None, dbeta=None):
    """-> dz, g (or None), dgamma, dbeta."""
    c = z.shape[-1]
    m = z.numel() // c
    dz = torch.empty_like(z)
    g = torch.empty_like(z) if want_g else None
    dgamma = dgamma if dgamma is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    dbeta = dbeta if dbeta is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    coef = torch.empty(3 * c, device=z.device, dtype=torch.float32)
    _check(lib().vatl_bn_train_bwd(_ptr(dy), _ptr(y), _ptr(z), _ptr(gamma), _ptr(save_mean), _ptr(save_invstd), _ptr(dz), _ptr(g),
                                   _ptr(dgamma), _ptr(dbeta), m, c, _ptr(coef), _ptr(_col_ws(m, c, z.device), torch.float64), _stream()),
           "vatl_bn_train_bwd")
    return dz, g, dgamma, dbeta


def maxpool3x3s2_bwd(x, dy):
    n, h, w, c = x.shape
    dx = torch.empty_like(x)
    _check(lib().vatl_maxpool3x3s2_bwd(_ptr(x), _ptr(dy), _ptr(dx), n, h, w, c, _stream()), "vatl_maxpool3x3s2_bwd")
    return dx


def maxpool3x3s2_fwd_idx(x):
    n, h, w, c = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=x.device, dtype=torch.float32)
    idx = torch.empty(y.shape, device=x.device, dtype=torch.uint8)
    _check(lib().vatl_maxpool3x3s2_fwd_idx(_ptr(x), _ptr(y), _ptr(idx, torch.uint8), n, h, w, c, _stream()), "vatl_maxpool3x3s2_fwd_idx")
    return y, idx


def maxpool3x3s2_fwd_idx_affine(z, scale, bias):
    """pool(relu(z*scale + bias)) and the winning taps in one pass over the conv output z (the activation is never stored)."""
    n, h, w, c = z.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=z.device, dtype=torch.float32)
    idx = torch.empty(y.shape, device=z.device, dtype=torch.uint8)
    _check(lib().vatl_maxpool3x3s2_fwd_idx_affine(_ptr(z), _ptr(scale), _ptr(bias), _ptr(y), _ptr(idx, torch.uint8), n, h, w, c, _stream()),
           "vatl_maxpool3x3s2_fwd_idx_affine")
    return y, idx


def bn_train_bwd_relu_pool(dpool, idx, scale, bias, z, gamma, save_mean, save_invstd, dgamma=None, dbeta=None):
    """Backward of Conv+BN+ReLU+MaxPool(3,2,1) from the POOLED output's gradient (the full-resolution gradient is gathered on
    the fly, never stored). -> dz, dgamma, dbeta."""
    n, h, w, c = z.shape
    dz = torch.empty_like(z)
    dgamma = dgamma if dgamma is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    dbeta = dbeta if dbeta is not None else torch.empty(c, device=z.device, dtype=torch.float32)
    coef = torch.empty(3 * c, device=z.device, dtype=torch.float32)
    _check(lib().vatl_bn_train_bwd_relu_pool(_ptr(dpool), _ptr(idx, torch.uint8), _ptr(scale), _ptr(bias), _ptr(z), _ptr(gamma), _ptr(save_mean),
                                             _ptr(save_invstd), _ptr(dz), _ptr(dgamma), _ptr(dbeta), n, h, w, c, _ptr(coef),
                                             _ptr(_col_ws(n * h * w, c, z.device), torch.float64), _stream()), "vatl_bn_train_bwd_relu_pool")
    return dz, dgamma, dbeta


def maxpool3x3s2_bwd_idx(dy, idx, in_hw):
    n, _, _, c = dy.shape
    dx = torch.empty((n, in_hw[0], in_hw[1], c), device=dy.device, dtype=torch.float32)
    _check(lib().vatl_maxpool3x3s2_bwd_idx(_ptr(dy), _ptr(idx, torch.uint8), _ptr(dx), n, in_hw[0], in_hw[1], c, _stream()), "vatl_maxpool3x3s2_bwd_idx")
    return dx


def pixelunshuffle2(x):
    n, h2, w2, c4 = x.shape
    y = torch.empty((n, h2 // 2, w2 // 2, c4 * 4), device=x.device, dtype=torch.float32)
    _check(lib().vatl_pixelunshuffle2(_ptr(x), _ptr(y), n, h2 // 2, w2 // 2, c4 * 4, _stream()), "vatl_pixelunshuffle2")
    return y


def se_bwd_gate(dy, y, u, gate):
    n, h, w, c = dy.shape
    dgate = torch.empty((n, c), device=dy.device, dtype=torch.float32)
    _check(lib().vatl_se_bwd(_ptr(dy), _ptr(y), _ptr(u), _ptr(gate), None, _ptr(dgate), None, None, n, h * w, c, _stream()), "vatl_se_bwd")
    return dgate


def se_bwd_apply(dy, y, gate, dpool):
    n, h, w, c = dy.shape
    du, gm = torch.empty_like(dy), torch.empty_like(dy)
    _check(lib().vatl_se_bwd(_ptr(dy), _ptr(y), None, _ptr(gate), _ptr(dpool), None, _ptr(du), _ptr(gm), n, h * w, c, _stream()), "vatl_se_bwd")
    return du, gm


def relu_bwd(dy, y):
    dx = torch.empty_like(dy)
    _check(lib().vatl_relu_bwd(_ptr(dy), _ptr(y), _ptr(dx), dy.numel(), _stream()), "vatl_relu_bwd")
    return dx


def col_sum(x2d):
    c = x2d.shape[-1]
    m = x2d.numel() // c
    out = torch.empty(c, device=x2d.device, dtype=torch.float32)
    _check(lib().vatl_col_sum(_ptr(x2d), m, c, _ptr(out), _ptr(_col_ws(m, c, x2d.device), torch.float64), _stream()), "vatl_col_sum")
    return out


# ----------------------------------------------------------------------------
# fine-tune step pieces
# ----------------------------------------------------------------------------

def masked_mse_fwd_bwd(out: torch.Tensor, target: torch.Tensor, mask: torch.Tensor):
    """-> loss (1,) f32 on device, grad like ``out``."""
    n, j, h, w = out.shape
    grad = torch.empty_like(out)
    loss = torch.empty(1, device=out.device, dtype=torch.float32)
    ws = torch.empty(int(lib().vatl_masked_mse_workspace_floats(out.numel())), device=out.device, dtype=torch.float32)
    m = mask.reshape(n, j).contiguous()
    _check(lib().vatl_masked_mse_fwd_bwd(_ptr(out), _ptr(target), _ptr(m), _ptr(grad), _ptr(loss), _ptr(ws), n, j, h * w, _stream()),
           "vatl_masked_mse_fwd_bwd")
    return loss, grad


NORM_TYPES = {"softmax": 0, "sigmoid": 1, "divide_sum": 2}


def l1_joint_regression_fwd_bwd(hm, gt_joints, gt_joints_vis, norm_type: str = "softmax", size_average: bool = True):
    """-> loss (0-dim), grad (B,J,H,W), pred_jts (B,2J)."""
    b, j, h, w = hm.shape
    grad = torch.empty_like(hm)
    loss = torch.empty((), device=hm.device, dtype=torch.float32)
    jts = torch.empty((b, 2 * j), device=hm.device, dtype=torch.float32)
    partial = torch.empty(b * j, device=hm.device, dtype=torch.float64)
    _check(lib().vatl_l1_joint_regression_fwd_bwd(_ptr(hm), _ptr(gt_joints), _ptr(gt_joints_vis), _ptr(grad), _ptr(loss), _ptr(jts),
                                                  _ptr(partial, torch.float64), b, j, h, w, NORM_TYPES[norm_type], int(size_average), _stream()),
           "vatl_l1_joint_regression_fwd_bwd")
    return loss, grad, jts


def gaussian_targets(joints_xy, vis, hm_hw=(64, 48), in_hw=(256, 192), sigma: float = 2.0):
    """joints (N,J,2) input-pixel coordinates, vis (N,J) -> target (N,J,H,W), weight (N,J,1,1)  [_target_generator]."""
    n, j, _ = joints_xy.shape
    target = torch.empty((n, j, hm_hw[0], hm_hw[1]), device=joints_xy.device, dtype=torch.float32)
    weight = torch.empty((n, j), device=joints_xy.device, dtype=torch.float32)
    _check(lib().vatl_gaussian_targets(_ptr(joints_xy.contiguous()), _ptr(vis.contiguous()), _ptr(target), _ptr(weight), n, j, hm_hw[0], hm_hw[1],
                                       in_hw[0], in_hw[1], sigma, _stream()), "vatl_gaussian_targets")
    return target, weight.reshape(n, j, 1, 1)


PIXEL_MEAN = (0.406, 0.457, 0.480)          # simple_transform.py:93-95


def crop_warp_affine(arena, src_off, src_hwf, minv, out_hw=(256, 192), mean=PIXEL_MEAN, out=None):
    """cv2.warpAffine(INTER_LINEAR) + im_to_torch + mean shift for B crops in one call  [SimpleTransform.test_transform].

    arena: uint8 device tensor holding the packed (h, w, 3) frames; src_off (B,) int64 byte offsets; src_hwf (B,3) int32
    {h, w, mirror}; minv (B,2,3) float64 dst->src maps.  Returns (B,3,out_h,out_w) fp32 and the per-crop u8 maxima."""
    b = int(src_off.shape[0])
    dev = arena.device
    if out is None:
        out = torch.empty((b, 3, out_hw[0], out_hw[1]), device=dev, dtype=torch.float32)
    cmax = torch.empty(b, device=dev, dtype=torch.int32)
    _check(lib().vatl_crop_warp_affine(_ptr(arena, torch.uint8), _ptr(src_off, torch.int64), _ptr(src_hwf.contiguous(), torch.int32),
                                       _ptr(minv.contiguous(), torch.float64), _ptr(out), _ptr(cmax, torch.int32), b, out_hw[0], out_hw[1],
                                       float(mean[0]), float(mean[1]), float(mean[2]), _stream()),
           "vatl_crop_warp_affine")
    return out, cmax


def ae_train_step(ae_flat, m, v, feat, d: int, z: int, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8):
    """One Adam step of the packed WholeBodyAE on the mini-batch feat (B, d); returns the loss (0-dim tensor)."""
    loss = torch.empty((), device=feat.device, dtype=torch.float32)
    _check(lib().vatl_ae_train_step(_ptr(ae_flat), _ptr(m), _ptr(v), _ptr(feat), feat.shape[0], d, z, lr, betas[0], betas[1], eps, step,
                                    _ptr(loss), _stream()), "vatl_ae_train_step")
    return loss


def unpack_ae(flat: torch.Tensor, module) -> None:
    """Inverse of pack_ae: copy the packed parameters back into the encoder / decoder Linear layers."""
    off = 0
    with torch.no_grad():
        for half in (module.encoder, module.decoder):
            for i in (0, 2, 4, 6):
                for t in (half[i].weight, half[i].bias):
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()


class ChecksumTable:
    """Device table for vatl_checksum_multi over a fixed list of tensors (built once: their storage does not move while a plan lives)."""

    def __init__(self, tensors):
        bw = int(lib().vatl_checksum_block_words())
        rows, blocks = [], 0
        for t in tensors:
            if not (t.is_cuda and t.is_contiguous()):
                raise VatlError("checksum: contiguous device tensors only")
            nbytes = t.numel() * t.element_size()
            if nbytes % 4 or t.data_ptr() % 4:
                raise VatlError(f"checksum: a tensor of {nbytes} bytes at {t.data_ptr():#x} is not made of aligned 32-bit words")
            rows.append((t.data_ptr(), nbytes // 4, blocks))
            blocks += max(1, (nbytes // 4 + bw - 1) // bw)
        self.n, self.blocks = len(rows), blocks
        self.device = tensors[0].device if tensors else None
        self.table = upload(torch.tensor(rows, dtype=torch.int64), self.device) if rows else None

    def launch(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """Enqueue the checksums of the tensors' CURRENT contents on the current stream -> (n,) int64 device tensor (the bits of the uint64 sums)."""
        if out is None:
            out = torch.empty((self.n,), dtype=torch.int64, device=self.device)
        if self.n:
            _check(lib().vatl_checksum_multi(_ptr(self.table, torch.int64), self.n, self.blocks, _ptr(out, torch.int64), _stream()), "vatl_checksum_multi")
        return out


_adamw_tables = {}                                   # (device, pointers...) -> (device table, total blocks): the pointers of a group do not change


def adamw_step_multi(params, grads, ms, vs, step: int, lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8):
    """One launch for a list of tensors sharing hyper-parameters and step count."""
    if not params:
        return
    key = [params[0].device.index]
    for p_, g_, m_, v_ in zip(params, grads, ms, vs):
        for t in (p_, g_, m_, v_):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise VatlError("adamw_step_multi needs contiguous fp32 device tensors")
        key += [p_.data_ptr(), g_.data_ptr(), m_.data_ptr(), v_.data_ptr(), p_.numel()]
    key = tuple(key)
    hit = _adamw_tables.get(key)
    if hit is None:                                  # built and uploaded once per (group, storage): no per-step host-to-device copy
        be = int(lib().vatl_adamw_multi_block_elems())
        rows, blocks = [], 0
        for k in range(len(params)):
            p_, g_, m_, v_, n = key[1 + 5 * k:6 + 5 * k]
            rows.append((p_, g_, m_, v_, n, blocks))
            blocks += (n + be - 1) // be
        if len(_adamw_tables) > 64:
            _adamw_tables.clear()
        hit = _adamw_tables[key] = (torch.tensor(rows, dtype=torch.int64).to(params[0].device), blocks)
    table, blocks = hit
    _check(lib().vatl_adamw_step_multi(_ptr(table, torch.int64), len(params), blocks, lr, betas[0], betas[1], eps, weight_decay, step, _stream()),
           "vatl_adamw_step_multi")


def adam_step(p, g, m, v, step: int, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8):
    _check(lib().vatl_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, _stream()),
           "vatl_adam_step")


def sgd_step(p, g, buf, step: int, lr: float, momentum: float = 0.9, weight_decay: float = 0.0):
    _check(lib().vatl_sgd_step(_ptr(p), _ptr(g), _ptr(buf), p.numel(), lr, momentum, weight_decay, step, _stream()), "vatl_sgd_step")


def adamw_step(p, g, m, v, step: int, lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8):
    _check(lib().vatl_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, _stream()),
           "vatl_adamw_step")
