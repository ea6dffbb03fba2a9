#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

Usage:  python tools/make_golden.py [--ref /root/reference] [--out tests/golden]

The reference (Python, CPU) is imported from ``--ref`` with three import shims
injected into ``sys.modules``: ``cv2`` (only getAffineTransform,
solved exactly in float64), ``torchvision.models.resnet*`` (returns an object
with an empty state dict so the ImageNet overwrite is a no-op) and
``easydict``.  Leaf modules of ``active_learning`` are loaded by file path
because the package ``__init__`` needs many absent third-party packages.

Nothing from the reference is copied: the fixtures hold seeded inputs (or the
seed that regenerates them through ``oracle/synth.py``) and the outputs the
reference produced for them.  This script never runs on the GPU box.
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402


# ----------------------------------------------------------------------------
# import shims
# ----------------------------------------------------------------------------

def install_shims():
    cv2 = types.ModuleType("cv2")

    def getAffineTransform(src, dst):
        a = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], axis=1)
        return np.linalg.solve(a, np.asarray(dst, np.float64)).T

    cv2.getAffineTransform = getAffineTransform
    cv2.INTER_LINEAR = 1
    cv2.BORDER_CONSTANT = 0
    sys.modules["cv2"] = cv2

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")

    class _Empty:
        def state_dict(self):
            return {}

    for n in (18, 34, 50, 101, 152):
        setattr(tvm, f"resnet{n}", lambda *a, **k: _Empty())
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm

    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in {**(d or {}), **kw}.items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed
    return EasyDict


def load_by_path(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# ----------------------------------------------------------------------------
# edge-case heat-maps
# ----------------------------------------------------------------------------

def edge_case_item(H=64, W=48) -> np.ndarray:
    m = np.zeros((17, H, W), np.float32)
    m[0, 10, 20] = 1.0; m[0, 40, 5] = 1.0                       # tie -> first in row-major
    m[1] = -1.0                                                 # all negative
    m[2, 1, 1] = 1.0; m[2, 1, 2] = 0.5                          # px == 1 -> no shift
    m[3, 2, 2] = 1.0; m[3, 2, 3] = 0.5; m[3, 3, 2] = 0.2; m[3, 1, 2] = 0.4   # (+.25,-.25)
    m[4, 62, 46] = 1.0; m[4, 62, 45] = 0.5                      # (-.25, 0)
    m[5, 63, 47] = 1.0                                          # corner
    m[6] = 0.3                                                  # constant plateau
    m[7, 0, 0] = 1.0; m[7, 10, 10] = 0.5; m[7, 20, 20] = 0.49; m[7, 63, 47] = 0.7
    m[8] = -1.0; m[8, 5, 5] = 2.0; m[8, 30, 30] = -0.5
    m[9, 30:33, 20:23] = 0.8                                    # 3x3 plateau
    m[10, 31, 23] = 0.9; m[10, 31, 24] = 0.9                    # equal neighbours -> sign 0
    m[11, 0, 30] = 0.6; m[11, 63, 3] = 0.6                      # border peaks
    m[12] = 0.0                                                 # all zero -> maxval 0 -> coords zeroed
    m[13, 32, 24] = 1e-30                                       # tiny positive
    m[14, 5, 46] = 1.0; m[14, 5, 47] = 0.3                      # px == W-2
    m[15, 62, 10] = 1.0; m[15, 63, 10] = 0.3                    # py == H-2
    m[16, 17, 17] = 0.7; m[16, 17, 16] = 0.2; m[16, 16, 17] = 0.6; m[16, 18, 17] = 0.1
    return m


# ----------------------------------------------------------------------------
# scorers
# ----------------------------------------------------------------------------

def gen_scorers(ref: str, out: str):
    from alphapose.utils import transforms as T                # the reference's
    pkg = types.ModuleType("active_learning"); pkg.__path__ = [os.path.join(ref, "active_learning")]
    sub = types.ModuleType("active_learning.Whole_body_AE"); sub.__path__ = [os.path.join(ref, "active_learning", "Whole_body_AE")]
    sys.modules["active_learning"] = pkg
    sys.modules["active_learning.Whole_body_AE"] = sub
    lp = load_by_path("active_learning.local_peak", os.path.join(ref, "active_learning/local_peak.py"))
    hf = load_by_path("active_learning.Whole_body_AE.hybrid_feature", os.path.join(ref, "active_learning/Whole_body_AE/hybrid_feature.py"))
    ae = load_by_path("active_learning.Whole_body_AE.AutoEncoder", os.path.join(ref, "active_learning/Whole_body_AE/AutoEncoder.py"))
    alm = load_by_path("active_learning.al_metric", os.path.join(ref, "active_learning/al_metric.py"))
    from alphapose.utils.bbox import bbox_xyxy_to_xywh
    from alphapose.utils.metrics import calc_accuracy

    N = 8
    hm = np.empty((N, 17, 64, 48), np.float32)
    hm[:5] = synth.blob_heatmaps(5, seed=synth.SEED)
    hm[5] = edge_case_item()
    hm[6] = -np.abs(synth.blob_heatmaps(1, seed=7)[0]) - 0.1     # everything negative
    hm[7] = synth.blob_heatmaps(1, seed=9, noise=0.0)[0]         # noise-free (flat zero background)
    bbox = synth.bboxes(N)
    bbox[5] = [100, 50, 196, 178]                                # Appendix-B box (w 96 -> scale 2)

    f = T.heatmap_to_coord_simple
    coords = np.zeros((N, 17, 2), np.float32)
    maxvals = np.zeros((N, 17, 1), np.float32)
    idx = np.zeros((N, 17), np.int64)
    for i in range(N):
        c, m = f(torch.from_numpy(hm[i]), bbox[i].tolist(), hm_shape=(64, 48), norm_type=None)
        coords[i], maxvals[i] = c, m
        idx[i] = np.argmax(hm[i].reshape(17, -1), 1)             # what get_max_pred does internally

    # soft-arg-max decode (a8')
    soft = {}
    for nt in ("softmax", "sigmoid", "divide_sum"):
        sc = np.zeros((5, 17, 2), np.float32); ss = np.zeros((5, 17, 1), np.float32)
        for i in range(5):
            src = hm[i] if nt != "divide_sum" else np.abs(hm[i]) + 1e-3
            c, s = T.heatmap_to_coord_simple_regress(torch.from_numpy(src), bbox[i].tolist(), hm_shape=(64, 48), norm_type=nt)
            sc[i], ss[i] = c, s
        soft[f"soft_{nt}_coords"] = sc
        soft[f"soft_{nt}_scores"] = ss

    # local peaks
    import warnings
    lp_mean = np.zeros(N, np.float32)
    lp_cnt = np.zeros((N, 17), np.int64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for i in range(N):
            lp_mean[i] = lp.localpeak_mean(hm[i])
            for j in range(17):
                lp_cnt[i, j] = lp.localpeak_values(hm[i, j]).size
    toy = np.array([[0, 0, 0, 0, 0, 0, 0, 4, 0, 0], [0, 0, 0, 1, 1, 0, 0, 0, 0, 0],
                    [0, 0, 0, 0, 3, 2, 0, 0, 0, 0], [0, 0, 0, 0, 2, 2, 0, 0, 0, 0]])   # data literal of local_peak.py:26-29
    toy_vals = lp.localpeak_values(toy)
    toy_mean = lp.localpeak_mean(np.array([toy, toy, toy]))

    # THC / TPC through the reference's own (unbound) methods
    almod_src = os.path.join(ref, "active_learning/ActiveLearning.py")
    ns = _extract_methods(almod_src, ("compute_thc", "compute_tpc"))

    class Dummy:
        eval_joints = list(range(17)); hm_size = (64, 48); norm_type = None
        heatmap_to_coord = staticmethod(f)
    d = Dummy()
    thc_l1 = np.array([ns["compute_thc"](d, hm[i], hm[i + 1], "L1") for i in range(N - 1)], np.float64)
    thc_l2 = np.array([ns["compute_thc"](d, hm[i], hm[i + 1], "L2") for i in range(N - 1)], np.float64)
    tpc = np.zeros(N - 1, np.int64)
    for i in range(N - 1):
        bb = bbox[i].tolist()
        thr = 0.01 * np.sqrt((bb[2] - bb[0]) * (bb[3] - bb[1]))
        tpc[i] = ns["compute_tpc"](d, coords[i], torch.from_numpy(hm[i + 1]), bb, thr)

    # hybrid feature + AE (items with positive score sums only)
    kp = np.concatenate([coords, maxvals], 2).reshape(N, 51)
    ok = np.array([kp[i, 2::3].sum() > 0 for i in range(N)])
    hyb = np.full((N, 42), np.nan)
    for i in range(N):
        if ok[i]:
            hyb[i] = hf.compute_hybrid(bbox_xyxy_to_xywh(bbox[i].tolist()), kp[i].tolist())
    lit_bbox = [10, 20, 30, 40]
    lit_kp = [411.0, 296.0, 0.0, 397.7706599832915, 324.5295867768595, 1.0, 394.74, 280.64, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
              405.0146198830409, 339.11983471074376, 0.0, 377.37593984962405, 339.11983471074376, 1.0,
              403.73708461707747, 377.9917443864447, 0.0, 378.0, 385.0, 1.0, 431.68074658890845, 361.15543076883813, 0.0,
              390.0, 427.0, 1.0, 388.0, 437.5, 0.0, 368.0, 438.0, 1.0, 404.0, 517.5, 1.0, 384.0, 518.0, 1.0,
              396.0, 582.0, 1.0, 372.0, 581.5, 1.0]                # data literal of hybrid_feature.py:65-84
    lit_feat = hf.compute_hybrid(lit_bbox, lit_kp)

    torch.manual_seed(synth.SEED)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        ae38 = ae.WholeBodyAE(z_dim=4)                            # the reference class (D = 38)
    sd38 = {k: synth.tensor_for("ae38." + k, v.shape) for k, v in ae38.state_dict().items()}
    ae38.load_state_dict({k: torch.from_numpy(v) for k, v in sd38.items()})
    ae38.eval()
    ae42 = torch.nn.Sequential()                                  # same topology, D = 42
    from oracle.nets import WholeBodyAERef
    ae42 = WholeBodyAERef(4, 42)
    sd42 = {k: synth.tensor_for("ae42." + k, v.shape) for k, v in ae42.state_dict().items()}
    ae42.load_state_dict({k: torch.from_numpy(v) for k, v in sd42.items()})
    crit = torch.nn.MSELoss()
    keep38 = np.r_[0:3, 5:20, 22:42]
    wpu42 = np.full(N, np.nan); wpu38 = np.full(N, np.nan); wpu38cls = np.full(N, np.nan)
    with torch.no_grad():
        for i in range(N):
            if not ok[i]:
                continue
            x = torch.tensor(hyb[i]).float()
            y = ae42(x)
            wpu42[i] = float(crit(y, x))                          # THC+WPU branch, ActiveLearning.py:364-370
            xi, yo = x.numpy(), y.numpy()
            xi = np.concatenate([xi[:3], xi[5:20], xi[22:]]); yo = np.concatenate([yo[:3], yo[5:20], yo[22:]])
            wpu38[i] = float(crit(torch.tensor(yo).float(), torch.tensor(xi).float()))   # WPU-only branch :371-386
            x38 = torch.tensor(hyb[i][keep38]).float()
            wpu38cls[i] = float(crit(ae38(x38), x38))             # reference class fed its declared width

    # OKS + heat-map accuracy (§8f rank 1)
    gt = kp.copy()
    r = np.random.RandomState(5)
    gt[:, 0::3] += r.normal(0, 3, (N, 17)); gt[:, 1::3] += r.normal(0, 3, (N, 17))
    gt[:, 2::3] = (r.random_sample((N, 17)) > 0.3).astype(np.float64)
    gt[6, 2::3] = 0                                               # no visible GT joint -> box-distance branch
    bb_ann = np.stack([bbox[:, 0], bbox[:, 1], bbox[:, 2] - bbox[:, 0] + 1, bbox[:, 3] - bbox[:, 1] + 1], 1)
    oks = np.array([alm.compute_OKS(bb_ann[i].tolist(), kp[i].tolist(), gt[i].tolist()) for i in range(N)])
    tgt, mask = synth.gaussian_targets(N, seed=3)
    acc = calc_accuracy(torch.from_numpy(hm * mask), torch.from_numpy(tgt * mask))

    np.savez_compressed(
        os.path.join(out, "scorers.npz"),
        hm=hm, bbox=bbox, idx=idx, coords=coords, maxvals=maxvals, **soft,
        lp_mean=lp_mean, lp_cnt=lp_cnt, toy=toy, toy_vals=np.asarray(toy_vals), toy_mean=np.float64(toy_mean),
        thc_l1=thc_l1, thc_l2=thc_l2, tpc=tpc,
        kp=kp, kp_ok=ok, hybrid=hyb, lit_bbox=np.array(lit_bbox, np.float64), lit_kp=np.array(lit_kp), lit_feat=lit_feat,
        wpu42=wpu42, wpu38=wpu38, wpu38cls=wpu38cls,
        **{"ae38." + k: v for k, v in sd38.items()}, **{"ae42." + k: v for k, v in sd42.items()},
        gt=gt, bb_ann=bb_ann, oks=oks, acc_targets_seed=np.int64(3), acc=np.float64(acc))
    print("scorers.npz written")


def _extract_methods(path: str, names):
    """Compile selected ``def``s of class ActiveLearning out of the reference file
    *in memory* (the class itself cannot be imported here: skimage/seaborn/umap/...
    are absent).  Nothing is written to disk."""
    import ast
    src = open(path, encoding="utf-8").read()
    tree = ast.parse(src)
    ns = {"np": np, "torch": torch}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "ActiveLearning":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in names:
                    mod = ast.Module(body=[item], type_ignores=[])
                    exec(compile(mod, path, "exec"), ns)
    return ns


# ----------------------------------------------------------------------------
# SimplePose-R50 forward, embedding, and one fine-tune step
# ----------------------------------------------------------------------------

def _sample_idx(numel: int, key: str, k: int = 256) -> np.ndarray:
    r = np.random.RandomState(abs(hash_str(key)) % (2 ** 31))
    return np.sort(r.randint(0, numel, size=min(k, numel)))


def hash_str(s: str) -> int:
    import zlib
    return zlib.crc32(s.encode())


def gen_simplepose(EasyDict, out: str):
    from alphapose.models import builder                         # the reference's
    cfg = EasyDict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "",
                    "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = EasyDict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(synth.SEED)
    m = builder.build_sppe(cfg, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    keys = list(m.state_dict().keys())
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}

    B = 2
    x = torch.from_numpy(synth.crops(B))
    m.eval()
    stage = {}
    hooks = []

    def tap(name):
        def fn(_m, _i, o):
            stage[name] = o.detach().numpy().copy()
        return fn
    for name, mod in (("stem_pool", m.preact.maxpool), ("layer1", m.preact.layer1), ("layer2", m.preact.layer2),
                      ("layer3", m.preact.layer3), ("layer4", m.preact.layer4), ("deconv1", m.deconv_layers[2]),
                      ("deconv2", m.deconv_layers[5]), ("deconv3", m.deconv_layers[8])):
        hooks.append(mod.register_forward_hook(tap(name)))
    with torch.no_grad():
        hm = m(x).numpy()
        for h in hooks:
            h.remove()
        emb = m.get_embedding(x).numpy()
    taps = {}
    for k, v in stage.items():
        ii = _sample_idx(v.size, k)
        taps[f"tap_{k}_idx"] = ii
        taps[f"tap_{k}_val"] = v.reshape(-1)[ii]
        taps[f"tap_{k}_absmean"] = np.float64(np.abs(v).mean())

    # one fine-tune step exactly as retrain_model does it (ActiveLearning.py:662-677)
    torch.manual_seed(synth.SEED)
    mt = builder.build_sppe(cfg, preset_cfg=preset)
    mt.load_state_dict(synth.state_dict_for(mt), strict=True)
    lr, wd = 2.5e-4, 0.7
    opt = torch.optim.AdamW(params=[{"params": mt.final_layer.parameters(), "lr": lr * 10},
                                    {"params": mt.preact.parameters(), "lr": lr},
                                    {"params": mt.deconv_layers.parameters(), "lr": lr * 5}], weight_decay=wd)
    crit = builder.build_loss(EasyDict({"TYPE": "MSELoss"}))
    labels, masks = synth.gaussian_targets(B, seed=11)
    labels, masks = torch.from_numpy(labels), torch.from_numpy(masks)
    mt.train()
    outp = mt(x.clone().requires_grad_())
    loss = 0.5 * crit(outp.mul(masks), labels.mul(masks))
    opt.zero_grad(); loss.backward(); opt.step()
    train = {"train_loss": np.float64(loss.item()), "train_out_absmean": np.float64(outp.detach().abs().mean().item())}
    sd = mt.state_dict()
    watch = ["final_layer.weight", "final_layer.bias", "deconv_layers.6.weight", "deconv_layers.7.weight", "deconv_layers.0.weight",
             "preact.layer4.2.conv3.weight", "preact.layer4.0.downsample.0.weight", "preact.layer3.0.conv2.weight",
             "preact.layer1.0.conv1.weight", "preact.layer1.0.bn1.weight", "preact.layer1.0.bn1.bias", "preact.conv1.weight", "preact.bn1.weight"]
    named = dict(mt.named_parameters())
    for k in watch:
        ii = _sample_idx(named[k].numel(), "g" + k, 512)
        train[f"grad_idx::{k}"] = ii
        train[f"grad_val::{k}"] = named[k].grad.reshape(-1)[ii].numpy()
        train[f"grad_norm::{k}"] = np.float64(named[k].grad.double().norm().item())
        train[f"new_val::{k}"] = sd[k].reshape(-1)[ii].numpy()
    for k in ("preact.bn1.running_mean", "preact.bn1.running_var", "preact.layer4.2.bn3.running_mean",
              "preact.layer4.2.bn3.running_var", "deconv_layers.7.running_mean", "deconv_layers.7.running_var"):
        train[f"bnstat::{k}"] = sd[k].numpy()
    train["bn_tracked"] = np.int64(sd["preact.bn1.num_batches_tracked"].item())

    np.savez_compressed(os.path.join(out, "simplepose_r50.npz"), batch=np.int64(B), seed=np.int64(synth.SEED),
                        heatmaps=hm, embedding=emb, keys=np.array(keys),
                        shapes=np.array([str(shapes[k]) for k in keys]), **taps, **train)
    print("simplepose_r50.npz written", hm.shape, float(np.abs(hm).mean()))



def gen_fastpose_hrnet(EasyDict, out: str):
    """FastPose-R50 and HRNet-W32 forward on the seeded crops (B = 2)."""
    from alphapose.models import builder                         # the reference's
    preset = EasyDict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    x = torch.from_numpy(synth.crops(2))
    res = {}
    cfgs = {
        "fastpose": {"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50},
        "hrnet": {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1,
                  "PRETRAINED_LAYERS": ["*"],
                  "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                  "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
                  "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}},
    }
    for name, c in cfgs.items():
        torch.manual_seed(synth.SEED)
        m = builder.build_sppe(EasyDict(c), preset_cfg=preset)
        m.load_state_dict(synth.state_dict_for(m), strict=True)
        m.eval()
        with torch.no_grad():
            hm = m(x).numpy()
            res[f"{name}_heatmaps"] = hm
            if hasattr(m, "get_embedding"):
                res[f"{name}_embedding"] = m.get_embedding(x).numpy()
        if name == "fastpose":                                   # one fine-tune step (ActiveLearning.py:662-673), gradients sampled
            torch.manual_seed(synth.SEED)
            mt = builder.build_sppe(EasyDict(c), preset_cfg=preset)
            mt.load_state_dict(synth.state_dict_for(mt), strict=True)
            mt.train()
            labels, masks = synth.gaussian_targets(2, seed=11)
            outp = mt(x.clone().requires_grad_())
            loss = 0.5 * torch.nn.MSELoss()(outp.mul(torch.from_numpy(masks)), torch.from_numpy(labels).mul(torch.from_numpy(masks)))
            loss.backward()
            res["fastpose_train_loss"] = np.float64(loss.item())
            named = dict(mt.named_parameters())
            for k in ("conv_out.weight", "conv_out.bias", "duc2.conv.weight", "duc1.bn.weight", "preact.layer4.0.se.fc.2.weight",
                      "preact.layer4.0.se.fc.0.bias", "preact.layer4.0.conv3.weight", "preact.layer2.0.se.fc.0.weight",
                      "preact.layer1.0.conv1.weight", "preact.conv1.weight"):
                ii = _sample_idx(named[k].numel(), "fg" + k, 512)
                res[f"fastpose_grad_idx::{k}"] = ii
                res[f"fastpose_grad_val::{k}"] = named[k].grad.reshape(-1)[ii].numpy()
        if name == "hrnet":                                      # same step on HRNet-W32 (trained by posetrack_train.py:207-231 / SGD, Adam in the AL loop)
            torch.manual_seed(synth.SEED)
            mt = builder.build_sppe(EasyDict(c), preset_cfg=preset)
            mt.load_state_dict(synth.state_dict_for(mt), strict=True)
            mt.train()
            labels, masks = synth.gaussian_targets(2, seed=11)
            outp = mt(x.clone().requires_grad_())
            loss = 0.5 * torch.nn.MSELoss()(outp.mul(torch.from_numpy(masks)), torch.from_numpy(labels).mul(torch.from_numpy(masks)))
            loss.backward()
            res["hrnet_train_loss"] = np.float64(loss.item())
            named = dict(mt.named_parameters())
            for k in ("final_layer.weight", "final_layer.bias", "stage4.2.fuse_layers.0.3.0.weight", "stage4.0.fuse_layers.3.0.1.0.weight",
                      "stage4.0.fuse_layers.2.3.1.weight", "stage3.1.branches.2.3.conv2.weight", "stage2.0.fuse_layers.1.0.0.0.weight",
                      "transition1.1.0.0.weight", "transition3.3.0.0.weight", "layer1.0.conv1.weight", "conv1.weight"):
                ii = _sample_idx(named[k].numel(), "hg" + k, 512)
                res[f"hrnet_grad_idx::{k}"] = ii
                res[f"hrnet_grad_val::{k}"] = named[k].grad.reshape(-1)[ii].numpy()
            res["hrnet_bn1_running_mean"] = mt.bn1.running_mean.numpy().copy()
        sd = m.state_dict()
        res[f"{name}_keys"] = np.array(list(sd.keys()))
        res[f"{name}_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        print(name, hm.shape, float(np.abs(hm).mean()), len(sd))
    np.savez_compressed(os.path.join(out, "fastpose_hrnet.npz"), batch=np.int64(2), seed=np.int64(synth.SEED), **res)


def gen_fastpose_r152(EasyDict, out: str):
    """BASELINE.json config 5 (synthetic stress config): FastPose-R152 at 384x288 -> 96x72 heat-maps, B = 1 forward,
    B = 2 fine-tune step (loss + sampled gradients)."""
    from alphapose.models import builder                         # the reference's
    preset = EasyDict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [384, 288], "HEATMAP_SIZE": [96, 72]})
    c = EasyDict({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152})
    res = {}
    torch.manual_seed(synth.SEED)
    m = builder.build_sppe(c, preset_cfg=preset)
    m.load_state_dict(synth.state_dict_for(m), strict=True)
    m.eval()
    x = torch.from_numpy(synth.crops(1, hw=(384, 288)))
    with torch.no_grad():
        res["heatmaps"] = m(x).numpy()
        res["embedding"] = m.get_embedding(x).numpy()
    m.train()
    x2 = torch.from_numpy(synth.crops(2, hw=(384, 288)))
    labels, masks = synth.gaussian_targets(2, seed=11, hw=(96, 72))
    outp = m(x2)
    loss = 0.5 * torch.nn.MSELoss()(outp.mul(torch.from_numpy(masks)), torch.from_numpy(labels).mul(torch.from_numpy(masks)))
    loss.backward()
    res["train_loss"] = np.float64(loss.item())
    named = dict(m.named_parameters())
    for k in ("conv_out.weight", "conv_out.bias", "duc2.conv.weight", "preact.layer3.35.conv2.weight", "preact.layer3.17.bn3.weight", "preact.conv1.weight"):
        ii = _sample_idx(named[k].numel(), "r152" + k, 512)
        res[f"grad_idx::{k}"] = ii
        res[f"grad_val::{k}"] = named[k].grad.reshape(-1)[ii].numpy()
    res["keys"] = np.array(list(m.state_dict().keys()))
    print("fastpose-r152", res["heatmaps"].shape, float(np.abs(res["heatmaps"]).mean()), len(res["keys"]), res["train_loss"])
    np.savez_compressed(os.path.join(out, "fastpose_r152_384.npz"), seed=np.int64(synth.SEED), **res)


def gen_l1_loss(out: str):
    """LOSS.TYPE L1JointRegression (criterion.py:46-94): loss value, predicted joints and sampled heat-map gradients
    from the reference module + torch autograd, for the three NORM_TYPEs."""
    from alphapose.models.criterion import L1JointRegression          # the reference's
    from alphapose.utils.transforms import _integral_tensor
    res = {}
    for norm in ("softmax", "sigmoid", "divide_sum"):
        hm, gt, vis = synth.l1_inputs(norm)
        crit = L1JointRegression(NORM_TYPE=norm)
        h = torch.from_numpy(hm).requires_grad_()
        loss = crit(h, torch.from_numpy(gt), torch.from_numpy(vis))
        loss.backward()
        jts, _ = _integral_tensor(torch.from_numpy(hm), 17, False, 48, 64, 1, integral_operation=crit.integral_operation, norm_type=norm)
        ii = _sample_idx(h.grad.numel(), "l1" + norm, 4096)
        res[f"{norm}_loss"] = np.float64(loss.item())
        res[f"{norm}_jts"] = jts.detach().numpy()
        res[f"{norm}_grad_idx"] = ii
        res[f"{norm}_grad_val"] = h.grad.reshape(-1)[ii].numpy()
        res[f"{norm}_grad_absmax"] = np.float64(h.grad.abs().max())
        print("l1", norm, loss.item(), float(h.grad.abs().max()))
    np.savez_compressed(os.path.join(out, "l1_joint_regression.npz"), **res)


def gen_targets(out: str):
    """SimpleTransform._target_generator (simple_transform.py:122-158) called unbound on seeded joints, incl. joints on
    and beyond the borders, invisible joints and a non-integer sigma."""
    from alphapose.utils.presets.simple_transform import SimpleTransform          # the reference's
    res = {}
    for tag, hm_hw, in_hw, sigma in (("a", (64, 48), (256, 192), 2), ("b", (96, 72), (384, 288), 1.5)):
        joints, vis = synth.target_joints(6, hm_hw, in_hw, seed=41)
        me = types.SimpleNamespace(_heatmap_size=np.array(hm_hw), _sigma=sigma,
                                   _feat_stride=np.array([in_hw[0], in_hw[1]]) / np.array(hm_hw))
        # the reference computes _feat_stride = input_size / heatmap_size with both given as (H, W); index 0 is applied to x,
        # index 1 to y (both are 4.0 for every shipped preset)
        tg, tw = [], []
        for n in range(joints.shape[0]):
            j3 = np.zeros((17, 3, 2), np.float32)
            j3[:, 0, 0], j3[:, 1, 0] = joints[n, :, 0], joints[n, :, 1]
            j3[:, 0, 1] = j3[:, 1, 1] = vis[n]
            t, w = SimpleTransform._target_generator(me, j3, 17)
            tg.append(t); tw.append(w)
        res[f"{tag}_target"] = np.stack(tg); res[f"{tag}_weight"] = np.stack(tw)
        print("targets", tag, res[f"{tag}_target"].shape, float(res[f"{tag}_target"].sum()), float(res[f"{tag}_weight"].sum()))
    np.savez_compressed(os.path.join(out, "targets.npz"), **res)


def gen_crop(out: str):
    """The numpy/torch pieces of the crop producer, called on the reference itself: _box_to_center_scale,
    _center_scale_to_box (alphapose/utils/bbox.py:197-226), get_affine_transform / affine_transform
    (alphapose/utils/transforms.py:753-792, cv2.getAffineTransform through the float64 shim) and im_to_torch (:76-91).
    cv2.warpAffine itself cannot be run here (no cv2) — see oracle/crop.py."""
    from alphapose.utils.bbox import _box_to_center_scale, _center_scale_to_box                 # the reference's
    from alphapose.utils.transforms import affine_transform, get_affine_transform, im_to_torch
    box, rot = synth.crop_cases(24)
    res = {}
    for tag, (inp_h, inp_w) in (("a", (256, 192)), ("b", (384, 288))):
        cs, ss, ts, bs, ps = [], [], [], [], []
        for (xmin, ymin, xmax, ymax), r in zip(box.tolist(), rot.tolist()):
            c, s = _box_to_center_scale(xmin, ymin, xmax - xmin, ymax - ymin, float(inp_w) / inp_h)
            s = s * 1.0
            t = get_affine_transform(c, s, r, [inp_w, inp_h])
            cs.append(c); ss.append(s); ts.append(t)
            bs.append(np.array(_center_scale_to_box(c, s), np.float64))
            ps.append(affine_transform(np.array([xmin, ymax], np.float32), t))
        res[f"{tag}_center"], res[f"{tag}_scale"] = np.stack(cs), np.stack(ss)
        res[f"{tag}_trans"], res[f"{tag}_box"], res[f"{tag}_pt"] = np.stack(ts), np.stack(bs), np.stack(ps)
        print("crop", tag, res[f"{tag}_trans"].dtype, res[f"{tag}_center"].dtype, float(np.abs(res[f"{tag}_trans"]).sum()))
    frame = synth.u8_frame(40, 56)
    dark = (frame > 250).astype(np.uint8)                         # max == 1: im_to_torch must NOT divide by 255
    res["tensor_bright"] = im_to_torch(frame.copy()).numpy()
    res["tensor_dark"] = im_to_torch(dark.copy()).numpy()
    np.savez_compressed(os.path.join(out, "crop.npz"), **res)


WELLCOND_SEED = 1234
WELLCOND_MODELS = {
    "simplepose": {"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50},
    "fastpose": {"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50},
}


def gen_wellcond(EasyDict, out: str):
    """A DISCRIMINATING whole-network gradient fixture: one fine-tune step (ActiveLearning.py:662-673) of the reference's
    SimplePose-R50 and FastPose-R50 at B = 16 with default-initialised weights and untouched BatchNorm statistics (running
    mean 0 / variance 1, gamma 1 / beta 0) — well conditioned, unlike the B = 2 steps with randomised statistics above, so a
    1 % defect in any layer's gradient shows.  The weights are torch's default initialisation of the BUILD's modules under
    a fixed seed (dumped by a child process that imports the build's package, loaded here into the reference's modules with
    strict=True: same keys, same shapes), so the GPU box regenerates them without a 136 MB file.  Stored: loss, the gradient
    norm and 512 sampled gradient values of every 10th parameter tensor, two weight checksums."""
    import subprocess
    import tempfile
    from alphapose.models import builder                         # the reference's
    preset = EasyDict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    B = 16
    x = torch.from_numpy(synth.crops(B, seed=91))
    labels, masks = synth.gaussian_targets(B, seed=92)
    labels, masks = torch.from_numpy(labels), torch.from_numpy(masks)
    res = {"batch": np.int64(B), "seed": np.int64(WELLCOND_SEED)}
    for name, c in WELLCOND_MODELS.items():
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "sd.pt")
            child = (f"import sys, torch; sys.path[:0] = [{ROOT!r}, {os.path.join(ROOT, 'vatl4pose-wacv2024_amd')!r}]\n"
                     "from alphapose.models import builder; from alphapose.utils.config import edict\n"
                     f"torch.manual_seed({WELLCOND_SEED}); m = builder.build_sppe(edict({c!r}), preset_cfg=edict({dict(preset)!r}))\n"
                     f"torch.save(m.state_dict(), {path!r})\n")
            subprocess.run([sys.executable, "-c", child], check=True)
            sd = torch.load(path)
        m = builder.build_sppe(EasyDict(c), preset_cfg=preset)
        m.load_state_dict(sd, strict=True)
        m.train()
        outp = m(x.clone().requires_grad_())
        loss = 0.5 * torch.nn.MSELoss()(outp.mul(masks), labels.mul(masks))
        loss.backward()
        res[f"{name}_loss"] = np.float64(loss.item())
        res[f"{name}_out_absmean"] = np.float64(outp.detach().abs().mean().item())
        params = list(m.named_parameters())
        res[f"{name}_wsum"] = np.float64(sum(float(p.detach().double().abs().sum()) for _, p in params))
        res[f"{name}_w0"] = params[0][1].detach().reshape(-1)[:8].numpy().copy()
        picked = [k for i, (k, _) in enumerate(params) if i % 10 == 0] + [params[-1][0], params[-2][0]]
        for k in dict.fromkeys(picked):
            gr = dict(params)[k].grad
            ii = _sample_idx(gr.numel(), "wc" + k, 512)
            res[f"{name}_grad_idx::{k}"] = ii
            res[f"{name}_grad_val::{k}"] = gr.reshape(-1)[ii].numpy()
            res[f"{name}_grad_norm::{k}"] = np.float64(gr.double().norm().item())
        res[f"{name}_bn1_running_mean"] = m.preact.bn1.running_mean.numpy().copy()
        # the same step in float64 (oracle graph, bit-identical to the reference in fp32): how far the REFERENCE's own fp32 step
        # is from exact arithmetic — the tolerance the parity test grants, tensor by tensor
        from oracle import nets
        ref64 = (nets.SimplePoseRef(50) if name == "simplepose" else nets.FastPoseRef(50))
        ref64.load_state_dict(sd, strict=True)
        ref64 = ref64.double().train()
        o64 = ref64(x.double())
        l64 = 0.5 * torch.nn.MSELoss()(o64 * masks.double(), labels.double() * masks.double())
        l64.backward()
        g64 = {k: p.grad for k, p in ref64.named_parameters()}
        for k in dict.fromkeys(picked):
            ii = res[f"{name}_grad_idx::{k}"]
            res[f"{name}_grad_f64::{k}"] = g64[k].reshape(-1)[ii].numpy()
        res[f"{name}_loss_f64"] = np.float64(l64.item())
        print("wellcond", name, float(loss), float(l64), len(dict.fromkeys(picked)), "tensors")
    np.savez_compressed(os.path.join(out, "wellcond_step.npz"), **res)


def gen_hostaug(out: str):
    """Host-side augmentation arithmetic of the reference's SimpleTransform, called unbound on seeded joints:
    half_body_transform (simple_transform.py:253-304, one np.random.randn() draw each) and _integral_target_generator
    (:160-177, incl. the doubled leading weights of the 136 / 133 / 68-joint layouts)."""
    from alphapose.utils.presets.simple_transform import SimpleTransform          # the reference's
    r = np.random.RandomState(77)
    me = types.SimpleNamespace(num_joints=17, upper_body_ids=(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10), lower_body_ids=(11, 12, 13, 14, 15, 16),
                               _aspect_ratio=192.0 / 256, pixel_std=1)
    n = 64
    joints = (r.rand(n, 17, 3) * 300).astype(np.float32)
    vis = (r.rand(n, 17, 1) > r.rand(n, 1, 1)).astype(np.float32)
    centers, scales, ok, after = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros(n, bool), np.zeros(n)
    for i in range(n):
        np.random.seed(1000 + i)
        c, s = SimpleTransform.half_body_transform(me, joints[i], vis[i])
        after[i] = np.random.rand()                                              # the next value of the stream: pins the number of draws
        if c is not None:
            ok[i], centers[i], scales[i] = True, c, s
    res = {"hb_joints": joints, "hb_vis": vis, "hb_ok": ok, "hb_center": centers, "hb_scale": scales, "hb_next_draw": after}
    for nj in (17, 136, 133, 68):
        j3 = (r.rand(nj, 3, 2) * 250).astype(np.float32)
        j3[:, 0, 1] = j3[:, 1, 1] = (r.rand(nj) > 0.3)
        t, w = SimpleTransform._integral_target_generator(me, j3, nj, 256, 192)
        res[f"int{nj}_joints"], res[f"int{nj}_target"], res[f"int{nj}_weight"] = j3, t, w
    print("hostaug", int(ok.sum()), "of", n, "half-body crops;", float(scales.sum()))
    np.savez_compressed(os.path.join(out, "hostaug.npz"), **res)


# ----------------------------------------------------------------------------
# MPE / Margin pinned on the real scikit-image found in this image (row a13)
# ----------------------------------------------------------------------------

SKIMAGE_ROOT = os.environ.get("VATL_SKIMAGE_ROOT", "/opt/conda/lib/python3.9/site-packages/skimage")   # a scikit-image SOURCE tree (0.18.3 in the build image; the reference pins 0.24: regenerate with VATL_SKIMAGE_ROOT=<its tree> when one is at hand)


def load_real_skimage_peak(root: str = SKIMAGE_ROOT):
    """``skimage.feature.peak.peak_local_max`` of the scikit-image source tree that sits in the build image
    (0.18.3; its compiled parts do not load under this interpreter, but ``feature/peak.py`` and ``_shared/coord.py``
    are pure Python over numpy / scipy).  Both files are executed FROM WHERE THEY LIE, by path; the three names they
    import from the rest of the package and never reach on this call path are stubbed."""
    ver = "unknown"
    for line in open(os.path.join(root, "__init__.py"), encoding="utf-8"):
        if line.startswith("__version__"):
            ver = line.split("=")[1].strip().strip("'\"")
    for name in ("skimage", "skimage._shared", "skimage.feature", "skimage.util", "skimage.measure", "skimage._shared.utils"):
        m = types.ModuleType(name); m.__path__ = []
        sys.modules[name] = m
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]

    def remove_arg(*_a, **_k):                                 # decorator factory: warns about `indices=`; no-op here
        return lambda f: f
    sys.modules["skimage._shared.utils"].remove_arg = remove_arg
    load_by_path("skimage._shared.coord", os.path.join(root, "_shared/coord.py"))
    pk = load_by_path("skimage.feature.peak", os.path.join(root, "feature/peak.py"))
    return pk.peak_local_max, ver


def _peak_cases():
    """name -> (H,W) float32 plane.  Besides seeded network-like maps: the excluded border, spacing boundaries,
    truncation to five peaks, degenerate planes, and the input images of scikit-image's own
    feature/tests/test_peak.py (their sizes and constructions) run with the reference's arguments."""
    c = {}
    z = lambda h=64, w=48: np.zeros((h, w), np.float32)
    m = z(); m[4, 10] = 3; m[5, 20] = 2; m[58, 30] = 1.5; m[59, 40] = 4; m[30, 4] = 5; m[31, 5] = 1.2; m[40, 42] = 1.1; m[41, 43] = 6
    c["border_rows_cols"] = m                                  # rows/cols 4, 59 / 4, 43 are border; 5, 58 / 5, 42 are not
    m = z(); m[4, 10] = 2.0; m[5, 10] = 1.0; m[40, 5] = 0.5
    c["border_shadow"] = m                                     # the border maximum is dropped but still shadows (5,10)
    for d in (4, 5, 6, 7):
        m = z(); m[20, 20] = 1.0; m[20, 20 + d] = 0.75; m[40, 10] = 0.5; m[40 + d, 10 + d] = 0.25
        c[f"distinct_{d}_apart"] = m
    m = z()
    for k, (y, x) in enumerate([(8, 8), (8, 24), (8, 40), (24, 8), (24, 24), (24, 40), (44, 8), (44, 24), (56, 40)]):
        m[y, x] = 1.0 + 0.125 * ((k * 5) % 9)
    c["nine_distinct_peaks"] = m                               # num_peaks = 5 keeps the five highest
    c["constant"] = np.full((64, 48), 0.25, np.float32)
    c["zero"] = z()
    m = np.full((64, 48), -1.0, np.float32); m[30, 30] = -0.5; m[12, 12] = -0.75
    c["negative_with_bumps"] = m
    m = z(); m[32, 24] = 1e-30
    c["tiny_positive"] = m
    m = z(); yy, xx = np.mgrid[0:64, 0:48]; m[:] = (yy * 48 + xx) * 1e-3
    c["ramp"] = m                                              # strictly increasing: the only 11x11 maxima are at the border
    m = z(); m[20, 20] = m[20, 24] = 1.0
    c["tie_4_apart"] = m                                       # equal values: the order among them is version dependent
    m = z(); m[20, 20] = m[20, 25] = 1.0
    c["tie_5_apart"] = m
    m = z(); m[30:33, 20:23] = 0.8
    c["tie_plateau_3x3"] = m
    # scikit-image feature/tests/test_peak.py inputs
    r = np.random.RandomState(21)
    m = 0.8 * r.rand(20, 20)
    for y, x in [(7, 7), (7, 13), (13, 7), (13, 13)]:
        m[y, x] = 1
    c["sk_noisy_peaks_20x20"] = m.astype(np.float32)           # four equal peaks 6 apart
    c["sk_constant_20x20"] = np.full((20, 20), 128, np.float32)
    m = np.zeros((7, 7), np.float32); m[1, 1] = 10; m[1, 3] = 11; m[1, 5] = 12; m[3, 5] = 8; m[5, 3] = 7
    c["sk_num_peaks_7x7"] = m                                  # smaller than the border: nothing survives
    c["sk_uniform_20x30"] = np.random.RandomState(21).uniform(size=(20, 30)).astype(np.float32)
    c["sk_uniform_40x60"] = np.random.RandomState(22).uniform(size=(40, 60)).astype(np.float32)
    c["sk_uniform_10x20"] = np.random.RandomState(23).uniform(size=(10, 20)).astype(np.float32)
    m = np.zeros((10, 20), np.float32); m[5, 5] = 1; m[5, 8] = .5
    c["sk_not_adjacent_10x20"] = m
    m = np.zeros((15, 15), np.float32); m[8, 12] = 10; m[2, 2] = 8; m[13, 5] = 10; m[7, 7] = 9; m[9, 5] = 7
    c["sk_isolated_15x15"] = m
    m = np.zeros((30, 30), np.float32); m[15, 15] = 1; m[5, 5] = 1
    c["sk_two_points_30x30"] = m
    return c


def gen_peaks(ref: str, out: str):
    from scipy.special import softmax
    from scipy.stats import entropy
    plm, ver = load_real_skimage_peak()
    ns = _extract_methods(os.path.join(ref, "active_learning/ActiveLearning.py"), ("compute_mpe", "compute_margin"))
    ns.update(peak_local_max=plm, softmax=softmax, entropy=entropy)

    def run(plane):
        loc = np.asarray(plm(plane, min_distance=5, num_peaks=5), np.int64).reshape(-1, 2)
        # candidates of the selection = what _get_peak_mask + _exclude_border leave; a plane is "tied" when two of them are equal
        from scipy import ndimage
        mx = ndimage.maximum_filter(plane, size=11, mode="constant")
        cand = (plane == mx) & (plane > plane.min())
        cand[:5] = False; cand[-5:] = False; cand[:, :5] = False; cand[:, -5:] = False
        v = plane[cand]
        tied = v.size != np.unique(v).size
        pad = np.full((5, 2), -1, np.int64); pad[:len(loc)] = loc
        return pad, len(loc), tied

    N = 24
    hm = synth.peak_items(N, seed=77)
    loc = np.zeros((N, 17, 5, 2), np.int64); cnt = np.zeros((N, 17), np.int64); tied = np.zeros((N, 17), bool)
    for n in range(N):
        for j in range(17):
            loc[n, j], cnt[n, j], tied[n, j] = run(hm[n, j])
    mpe = np.array([ns["compute_mpe"](None, h) for h in hm], np.float64)
    margin = np.array([ns["compute_margin"](None, h) for h in hm], np.float64)
    res = {"skimage_version": np.array(ver), "items_seed": np.int64(77), "loc": loc, "cnt": cnt, "tied": tied, "mpe": mpe, "margin": margin,
           "item_tied": tied.any(1)}
    names = []
    for name, plane in _peak_cases().items():
        l, k, t = run(plane)
        names.append(name)
        res["case_" + name] = plane
        res["loc_" + name] = l[:k]
        res["tied_" + name] = np.bool_(t)
    res["case_names"] = np.array(names)
    np.savez_compressed(os.path.join(out, "peaks.npz"), **res)
    print(f"peaks.npz written (scikit-image {ver}; {int(tied.sum())} of {tied.size} item planes tied, "
          f"cases tied: {[n for n in names if res['tied_' + n]]})")


# ----------------------------------------------------------------------------
# wide arg-max pin of the timed (Winograd F(4x4)) route: >= 64 crops per network
# ----------------------------------------------------------------------------

WIDE_HRNET = {"TYPE": "PoseHighResolutionNet", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50, "FINAL_CONV_KERNEL": 1,
              "PRETRAINED_LAYERS": ["*"],
              "STAGE2": {"NUM_MODULES": 1, "NUM_BRANCHES": 2, "NUM_BLOCKS": [4, 4], "NUM_CHANNELS": [32, 64], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
              "STAGE3": {"NUM_MODULES": 4, "NUM_BRANCHES": 3, "NUM_BLOCKS": [4, 4, 4], "NUM_CHANNELS": [32, 64, 128], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"},
              "STAGE4": {"NUM_MODULES": 3, "NUM_BRANCHES": 4, "NUM_BLOCKS": [4, 4, 4, 4], "NUM_CHANNELS": [32, 64, 128, 256], "BLOCK": "BASIC", "FUSE_METHOD": "SUM"}}

WIDE_CASES = {   # name -> (MODEL cfg, input size, heat-map size, crops, crops whose full heat-maps are kept)
    "simplepose_r50": ({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50}, (256, 192), (64, 48), 64, 4),
    "fastpose_r50": ({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 50}, (256, 192), (64, 48), 64, 4),
    "hrnet_w32": (WIDE_HRNET, (256, 192), (64, 48), 64, 4),
    "fastpose_r152_384": ({"TYPE": "FastPose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_LAYERS": 152}, (384, 288), (96, 72), 8, 1),
}
WIDE_SEED = 2024     # crops = synth.crops(n, seed=WIDE_SEED, hw=...), boxes = synth.bboxes(n, seed=WIDE_SEED): not the seed of any other fixture


def gen_widepin(EasyDict, out: str):
    """The reference's own decode inputs on MANY crops: per network the arg-max index of every joint plane (get_max_pred,
    transforms.py:710-727), its value, the gap to the plane's second largest value (how close each index is to moving), the decoded
    key-points (heatmap_to_coord_simple, transforms.py:550-583), the same indices from the reference modules run in float64 (what
    exact arithmetic would pick), and full heat-maps for a few crops only (the fixture stays small)."""
    from alphapose.models import builder                         # the reference's
    from alphapose.utils import transforms as T
    res = {}
    for name, (c, in_hw, hm_hw, n, keep) in WIDE_CASES.items():
        preset = EasyDict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": list(in_hw), "HEATMAP_SIZE": list(hm_hw)})
        torch.manual_seed(synth.SEED)
        m = builder.build_sppe(EasyDict(c), preset_cfg=preset)
        m.load_state_dict(synth.state_dict_for(m), strict=True)
        m.eval()
        x = torch.from_numpy(synth.crops(n, seed=WIDE_SEED, hw=in_hw))
        bb = synth.bboxes(n, seed=WIDE_SEED)
        hm = np.empty((n, 17) + hm_hw, np.float32)
        with torch.no_grad():
            for i in range(0, n, 8):
                hm[i:i + 8] = m(x[i:i + 8]).numpy()
        flat = hm.reshape(n, 17, -1)
        idx = flat.argmax(2)
        part = np.partition(flat, -2, axis=2)
        top1, top2 = part[..., -1], part[..., -2]
        kp = np.zeros((n, 17, 2), np.float32); mv = np.zeros((n, 17), np.float32)
        for i in range(n):
            cds, mx = T.heatmap_to_coord_simple(torch.from_numpy(hm[i]), bb[i].tolist(), hm_shape=hm_hw, norm_type=None)
            kp[i], mv[i] = cds, mx.reshape(-1)
        assert np.array_equal(mv, top1)
        m64 = m.double()
        idx64 = np.empty((n, 17), np.int64); gap64 = np.empty((n, 17), np.float64); d32 = 0.0; amax = 0.0
        with torch.no_grad():
            for i in range(0, n, 4):
                h64 = m64(x[i:i + 4].double()).numpy().reshape(-1, 17, hm_hw[0] * hm_hw[1])
                idx64[i:i + 4] = h64.argmax(2)
                p64 = np.partition(h64, -2, axis=2)
                gap64[i:i + 4] = p64[..., -1] - p64[..., -2]
                d32 = max(d32, float(np.abs(h64 - flat[i:i + 4]).max())); amax = max(amax, float(np.abs(h64).max()))
        res[f"{name}_n"] = np.int64(n)
        res[f"{name}_idx"] = idx.astype(np.int16)
        res[f"{name}_maxval"] = top1.astype(np.float32)
        res[f"{name}_gap"] = (top1.astype(np.float64) - top2.astype(np.float64)).astype(np.float32)
        res[f"{name}_keypoints"] = kp
        res[f"{name}_idx_f64"] = idx64.astype(np.int16)
        res[f"{name}_gap_f64"] = gap64.astype(np.float32)
        res[f"{name}_absmax"] = np.float64(np.abs(hm).max())
        res[f"{name}_ref_fp32_vs_f64"] = np.float64(d32 / amax)
        res[f"{name}_heatmaps"] = hm[:keep].copy()
        print(name, hm.shape, "absmax", float(np.abs(hm).max()), "min gap", float(res[f"{name}_gap"].min()),
              "fp32 != f64 indices", int((idx != idx64).sum()), "of", idx.size, "fp32 vs f64", d32 / amax, flush=True)
    np.savez_compressed(os.path.join(out, "widepin.npz"), seed=np.int64(WIDE_SEED), **res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    EasyDict = install_shims()
    sys.path.insert(0, a.ref)
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    if a.only in ("", "scorers"):
        gen_scorers(a.ref, a.out)
    if a.only in ("", "simplepose"):
        gen_simplepose(EasyDict, a.out)
    if a.only in ("", "nets2"):
        gen_fastpose_hrnet(EasyDict, a.out)
    if a.only in ("", "targets"):
        gen_targets(a.out)
    if a.only in ("", "l1"):
        gen_l1_loss(a.out)
    if a.only in ("", "crop"):
        gen_crop(a.out)
    if a.only in ("", "wellcond"):
        gen_wellcond(EasyDict, a.out)
    if a.only in ("", "hostaug"):
        gen_hostaug(a.out)
    if a.only in ("", "peaks"):
        gen_peaks(a.ref, a.out)
    if a.only in ("", "r152"):
        gen_fastpose_r152(EasyDict, a.out)
    if a.only in ("", "widepin"):
        gen_widepin(EasyDict, a.out)


if __name__ == "__main__":
    main()
