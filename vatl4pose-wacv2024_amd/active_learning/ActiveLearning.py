"""Active-learning loop around the MI355X hot path (reference: active_learning/ActiveLearning.py).

Same entry points as the reference class — ``ActiveLearning(cfg, opt)``, ``eval_and_query()``,
``outcome()``, ``retrain_model()`` — with the two hot loops rebuilt around batched device kernels:

* ``eval_and_query`` (:253-429): one backbone forward per item (prev/next heat-maps are the neighbours'
  when the dataset is an id-sorted stream, else three forwards like the reference), then ONE
  ``score_batch`` launch sequence per batch instead of a per-item Python loop with D2H copies.
* ``retrain_model`` (:651-686): train-mode forward, fused masked-MSE loss+gradient, HIP backward, AdamW.

In scope: uncertainty None | HP | TPC | THC* (THC_L1, THC_L2, ...: always the L1 norm, ActiveLearning.py:346) | *WPU* (WPU,
WPU_hybrid, WPU_raw: the reference matches these by substring, :371, and always takes the hybrid feature, :372) | THC+WPU |
MPE | Margin | Entropy; representativeness None | Influence | Random; filter None | Random | Diversity | Coreset (device
kernels, active_learning/query.py) and weighted | K-Means (sklearn on the host, exactly like the reference).  COCO mAP /
OSPA evaluation, plots and the dead VL4Pose branch stay out.

Multi-GPU: one process per GPU (active_learning/distributed.py).  Under torchrun every rank runs the driver; as one plain
process on an N-GPU node (``opt.num_gpu`` = N, Run_active_learning.py:92-94) the constructor starts N-1 worker processes
that mirror every public call — the replacement for the reference's ``nn.DataParallel`` (ActiveLearning.py:233).
"""
from __future__ import annotations

import functools
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

import vatl_hip as vh
from alphapose.models import builder
from alphapose.utils.bbox import bbox_xyxy_to_xywh
from alphapose.utils.metrics import DataLogger, calc_accuracy, calc_accuracy_begin
from alphapose.utils.transforms import get_func_heatmap_to_coord

from .al_metric import compute_OKS_batch
from .optim import SGD, Adam, AdamW
from .scoring import multi_peak_scores, score_batch


def uncertainty_kind(name: str) -> str:
    """The branch of the reference's per-item dispatch (ActiveLearning.py:329-401) an ``--uncertainty`` string takes.
    Exact names first, then the substring tests in the reference's order: anything containing "THC" is THC with the L1 norm
    (:345-346) — paired with WPU only under the exact name "THC+WPU" (:403, :494) — and anything else containing "WPU"
    ("WPU", "WPU_hybrid", "WPU_raw": backrun_active_learning.sh:6) is the hybrid-feature WPU (:371-386)."""
    if name in ("None", "HP", "TPC", "MPE", "Margin", "Entropy"):
        return name
    if name == "VL4Pose":
        raise ValueError("Uncertainty type is not supported")          # dead branch in the reference
    if "THC" in name:
        return "THC+WPU" if name == "THC+WPU" else "THC"
    if "WPU" in name:
        return "WPU"
    raise ValueError("Uncertainty type is not supported")


def _collective(fn):
    """Public entry points run on every rank in lock-step.  On the driver process that owns worker processes, the outermost
    call is announced to them first (active_learning/worker.py); under torchrun every rank calls the method itself."""
    @functools.wraps(fn)
    def wrapper(self, *a, **k):
        from . import distributed as D
        if self._depth == 0 and D.have_workers():
            D.command("call", fn.__name__)
        self._depth += 1
        try:
            return fn(self, *a, **k)
        except BaseException:
            if self._depth == 1:
                D.mark_failed()                # the workers are inside some other collective now: never broadcast to them again
            raise
        finally:
            self._depth -= 1
    return wrapper


class ActiveLearning:
    def __init__(self, cfg, opt, eval_dataset=None, train_dataset=None):
        self.cfg, self.opt = cfg, opt
        self.round_cnt = 0
        self.uncertainty = getattr(opt, "uncertainty", "None")
        self.representativeness = getattr(opt, "representativeness", "None")
        self.filter = getattr(opt, "filter", "None")
        self.strategy = getattr(opt, "strategy", self.uncertainty)
        self.video_id = getattr(opt, "video_id", "synthetic")
        self.get_prenext = bool(getattr(opt, "get_prenext", "THC" in self.uncertainty or "TPC" in self.uncertainty))
        self.unc_kind = uncertainty_kind(self.uncertainty)
        if self.representativeness not in ("None", "Influence", "Random"):
            raise ValueError("Representativeness type is not supported")
        if self.filter not in ("None", "Random", "Diversity", "Coreset", "weighted", "K-Means"):
            raise ValueError("Filter type is not supported")
        # ActiveLearning.py:284: embeddings are needed by every representativeness / filter except None and Random
        self.need_embedding = self.representativeness not in ("None", "Random") or self.filter not in ("None", "Random")
        self.w_unc = float(cfg.VAL.get("W_UNC", 1.0))
        self.unc_lambda = float(cfg.VAL.get("UNC_LAMBDA", 1.0))
        self.finish_margin = 0.05
        # one process per GPU: join the torchrun group, or (plain single process, opt.num_gpu > 1) start the worker ranks
        from . import distributed as D
        self._depth = 0
        ngpu = max(1, int(getattr(opt, "num_gpu", 1)))
        world = D.ensure_workers(ngpu)
        if D.have_workers() and D.is_main():
            payload = {k: v for k, v in (("eval_dataset", eval_dataset), ("train_dataset", train_dataset)) if v is not None}
            D.command("new", cfg, opt, D.dump_payload(payload) if payload else "")
        try:                                           # after the ('new', ...) broadcast the workers are building their replicas and will wait in the
            self._build(cfg, opt, eval_dataset, train_dataset, ngpu, world)   # barrier below: a failure here must not be followed by an 'exit' broadcast
        except BaseException:
            if D.mirrored():
                D.mark_failed()
            raise

    def _build(self, cfg, opt, eval_dataset, train_dataset, ngpu, world):
        from . import distributed as D
        self.device = torch.device("cuda", torch.cuda.current_device())
        # the reference's DataParallel scatters every mini-batch over opt.num_gpu replicas (:233); `replicas` keeps that
        # partition (and with it the per-replica BatchNorm statistics) whatever the number of processes actually running
        self.replicas = max(ngpu, world)
        per_rank = max(1, self.replicas // world)

        self.eval_dataset = eval_dataset if eval_dataset is not None else builder.build_dataset(
            cfg.DATASET.EVAL, preset_cfg=cfg.DATA_PRESET, train=False, get_prenext=self.get_prenext)
        self.train_dataset = train_dataset if train_dataset is not None else builder.build_dataset(
            cfg.DATASET.TRAIN, preset_cfg=cfg.DATA_PRESET, train=True, get_prenext=False)
        self.collate_fn = self.eval_dataset.my_collate_fn
        workers = int(getattr(opt, "num_workers", 0))
        if getattr(self.eval_dataset, "DEVICE_ITEMS", False):
            workers = 0                                    # items are made on the device by this process: no worker processes
        self.eval_loader = DataLoader(self.eval_dataset, batch_size=cfg.VAL.BATCH_SIZE * per_rank, shuffle=False, num_workers=workers,
                                      drop_last=False, pin_memory=not getattr(self.eval_dataset, "DEVICE_ITEMS", False), collate_fn=self.collate_fn)
        self.eval_len = len(self.eval_dataset)
        self.dedup = bool(getattr(self.eval_dataset, "ID_SORTED_STREAM", False))
        if self.dedup and hasattr(self.eval_dataset, "emit_neighbour_crops"):
            self.eval_dataset.emit_neighbour_crops = False     # the neighbours' heat-maps come from the stream: no prev / next crops

        self.query_ratio = list(cfg.VAL.QUERY_RATIO)
        self.query_sizes = [int(self.eval_len * x) for x in self.query_ratio]
        self.query_size = self.query_sizes[0]
        self.unlabeled_id = list(range(self.eval_len))
        self.labeled_id = []
        self.retrain_id = []
        self.percentage, self.performance, self.performance_ann = [], [], []
        self.ospa_list, self.ospa_list_ann, self.combine_weight, self.uncertainty_mean, self.moksQ_list = [], [], [], [], []
        self.query_list_list, self.uncertainty_dict, self.influence_dict = {}, {}, {}
        self.spearmanr_list, self.corr_list = [], []
        self.true_labeled_dict, self.false_labeled_dict, self.true_unlabeled_dict, self.false_unlabeled_dict = {}, {}, {}, {}
        self.actual_finish = self.finished_minerror = self.finished_oursc = 100
        self.finish_acc = getattr(opt, "retrain_thresh", 1)
        self.is_early_stop = False
        self.one_by_one = bool(getattr(opt, "onebyone", False))
        if self.one_by_one:
            self.query_size = 3                            # ActiveLearning.py:117-118
        self.continual = bool(getattr(opt, "continual", True))

        self.retrain_epoch = cfg.RETRAIN.BASE
        self.lr = cfg.RETRAIN.LR
        self.moks_queried = 0
        self.model, self.optimizer, self.scheduler = self.initialize_estimator()
        self.criterion = builder.build_loss(cfg.LOSS)
        self.AE = self.initialize_AE() if "WPU" in self.strategy or "WPU" in self.uncertainty else None
        self.eval_joints = self.eval_dataset.EVAL_JOINTS
        self.norm_type = cfg.LOSS.get("NORM_TYPE", None)
        self.hm_size = cfg.DATA_PRESET.HEATMAP_SIZE
        self.heatmap_to_coord = get_func_heatmap_to_coord(cfg)
        if D.mirrored():                           # every worker has built its replica: the pickled datasets can go
            D.barrier()
            D.release_payloads()
        if getattr(opt, "gc_freeze", False):
            # OPT-IN (round 6; it was the default in round 5).  The model, the data sets (one annotation dict per item) and the loaders built above live as long
            # as this object.  A full pass of the cyclic collector over them costs 50 - 100 ms and — triggered by the allocation count, i.e. every few evaluation
            # rounds — lands inside one (tools/al_eval_bench.py --trace: single rounds of 120 - 200 ms among rounds of 79).  gc.freeze() moves what exists NOW to
            # the permanent generation: later passes only look at what the rounds allocate.  It is a change to the HOST PROGRAM's collector — every object alive
            # at this point, the host's own included, is exempt from cycle collection until gc.unfreeze() — so a library constructor must not make it on its
            # own: the host asks for it (opt.gc_freeze = True) and `close()` undoes it.  Reference counting is unaffected.
            import gc
            gc.collect()
            gc.freeze()
            self.__dict__["_gc_frozen"] = True

    def close(self):
        """Finish what runs in the background (the record thread) and undo the process-global change an opt-in made (`opt.gc_freeze`: gc.unfreeze(), so
        that this object, its model and its device memory can be collected like anything else).  Safe to call more than once; the reference has no
        counterpart (its objects hold no threads and change no global state)."""
        self.flush_records()
        if self.__dict__.pop("_gc_frozen", False):
            import gc
            gc.unfreeze()

    # ------------------------------------------------------------------ estimator
    def initialize_estimator(self):
        cfg = self.cfg
        model = builder.build_sppe(cfg.MODEL, preset_cfg=cfg.DATA_PRESET)
        if getattr(self.opt, "from_scratch", False):
            pass
        elif cfg.MODEL.PRETRAINED:
            model.load_state_dict(torch.load(cfg.MODEL.PRETRAINED, map_location="cpu"))
        else:
            raise ValueError("No pretrained model is given!")
        model = model.to(self.device)
        kind = cfg.RETRAIN.OPTIMIZER
        if kind == "AdamW":
            if cfg.MODEL.TYPE == "SimplePose":
                groups = [{"params": model.final_layer.parameters(), "lr": self.lr * 10}, {"params": model.preact.parameters(), "lr": self.lr},
                          {"params": model.deconv_layers.parameters(), "lr": self.lr * 5}]
            elif cfg.MODEL.TYPE == "FastPose":
                groups = [{"params": model.conv_out.parameters(), "lr": self.lr * 10}, {"params": model.preact.parameters(), "lr": self.lr},
                          {"params": model.duc1.parameters(), "lr": self.lr * 5}, {"params": model.duc2.parameters(), "lr": self.lr * 5}]
            else:
                raise ValueError("Optimizer not supported!")          # the reference leaves `optimizer` unbound here
            optimizer = AdamW(params=groups, weight_decay=cfg.RETRAIN.WEIGHT_DECAY)
        elif kind == "Adam":
            optimizer = Adam(model.parameters(), lr=self.lr)
        elif kind == "SGD":
            optimizer = SGD(model.parameters(), lr=self.lr, momentum=0.9, weight_decay=0.0005)
        else:
            raise ValueError("Optimizer not supported!")
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=cfg.RETRAIN.LR_GAMMA)
        from . import distributed as D
        D.broadcast_module_(model)                 # identical replicas on every rank (random init / checkpoint of rank 0)
        return model, optimizer, scheduler

    def initialize_AE(self):
        from .Whole_body_AE.AutoEncoder import WholeBodyAE
        ae = WholeBodyAE(z_dim=self.cfg.AE.Z_DIM, input_dim=int(self.cfg.AE.get("INPUT_DIM", 42)))
        path = self.cfg.AE.get("PRETRAINED", "")
        if path:
            ae.load_state_dict(torch.load(path, map_location="cpu"))
        ae = ae.to(self.device).eval()
        from . import distributed as D
        D.broadcast_module_(ae)
        return ae

    # ------------------------------------------------------------------ hot loop 1
    def _heatmaps(self, inps, emb_out=None, out=None):
        m = self.model
        x = inps[:, 0].to(self.device, non_blocking=True)
        from alphapose.models import hip_engine

        def heatmaps(t):
            # the stream entry point, not `m(t)`: a crop's bits must not depend on the size of the loader batch or of the shard
            # it arrives in (a short last batch would otherwise take the small-batch split-K route of the module call)
            t = t.to(self.device)
            return hip_engine.forward_into(m, t, torch.empty((t.shape[0], self.cfg.DATA_PRESET.NUM_JOINTS, *self.hm_size), device=self.device))
        with torch.no_grad():
            cur = out if out is not None else torch.empty((x.shape[0], self.cfg.DATA_PRESET.NUM_JOINTS, *self.hm_size), device=self.device)
            if emb_out is not None:                   # heat-maps and get_embedding from one trunk pass
                hip_engine.forward_with_embedding(m, x, cur, emb_out)
            else:
                hip_engine.forward_into(m, x, cur)
            if not self.get_prenext or self.dedup:
                return cur, None, None
            return cur, heatmaps(inps[:, 1]), heatmaps(inps[:, 2])

    def _score_range(self, lo, hi):
        """Forward + score the id-sorted items lo..hi-1 (one shard plus its halo); returns (hi-lo, 55 [+ 2048]) float32
        rows: 51 key-point values, 2 uncertainty values, local-peak mean, OKS [, the get_embedding vector]."""
        n = hi - lo
        J, (hh, hw) = self.cfg.DATA_PRESET.NUM_JOINTS, self.hm_size
        # the shard's heat-maps stay on the device (209 KB per item) and are scored in one pass, so THC/TPC
        # neighbours across loader batches need no special casing
        hm_all = torch.empty((n, J, hh, hw), device=self.device)
        thc_ref = torch.zeros(n, device=self.device)
        # per-item host columns are collected on the host and uploaded ONCE after the loop (pinned, non-blocking): an upload from pageable
        # memory inside the loop would block the host until the batch's forward pass has finished, and the next batch's crops are prepared
        # by the host — the loop must only enqueue
        bb_h = np.zeros((n, 4), np.float32)
        bb_dev = None                                                     # crop boxes that arrive as DEVICE tensors (FrameVideo crops on the GPU) stay there: a `.cpu()` here
                                                                          # would wait for the batch's forward pass and serialise the host's next batch with the device
        ip_h, in_h = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        gt_all = np.zeros((n, 3 * J), np.float64)
        ann_all = np.zeros((n, 4), np.float64)
        ids_all = np.zeros((n, 2), np.float64)                            # image id, annotation id (exact in float64)
        emb_all = torch.empty((n, self.emb_dim), device=self.device) if self.need_embedding else None
        thc_norm = "L1" if self.unc_kind in ("THC", "THC+WPU") else None          # `norm_type = 'L1'` for every THC* (:346)
        if getattr(self.eval_dataset, "DEVICE_ITEMS", False) and callable(getattr(self.eval_dataset, "collated", None)) and \
                getattr(self.opt, "device_batches", True):
            # items made on the device by this process: the data set hands over whole batches (same columns as its collate function makes
            # from `__getitems__`, without 11 Python objects per item — the first batch's host time is the one the device cannot hide)
            bs = self.eval_loader.batch_size
            loader = (self.eval_dataset.collated(range(a, min(a + bs, hi))) for a in range(lo, hi, bs))
        else:
            loader = self.eval_loader if (lo == 0 and hi == self.eval_len) else DataLoader(
                Subset(self.eval_dataset, list(range(lo, hi))), batch_size=self.eval_loader.batch_size, shuffle=False, num_workers=0,
                collate_fn=self.collate_fn)
        for (idxs, inps, labels, label_masks, GTkpts, img_ids, ann_ids, bboxes_crop, bboxes_ann, isPrev, isNext) in loader:
            loc = np.asarray(idxs) - lo
            a0, b0 = int(loc[0]), int(loc[0]) + len(loc)
            run = bool(np.array_equal(loc, np.arange(a0, b0)))           # a sequential loader hands over consecutive items: plain slices, no index tensor
            idx = slice(a0, b0) if run else vh.upload(loc, self.device)
            emb_b = (emb_all[idx] if run else torch.empty((len(idxs), self.emb_dim), device=self.device)) if emb_all is not None else None
            cur, prev, nxt = self._heatmaps(inps, emb_b, out=hm_all[idx] if run else None)
            assert cur.dim() == 4, "the dimension of output must be 4"
            if not run:
                hm_all[idx] = cur
                if emb_all is not None:                                   # ActiveLearning.py:284-286, without the second trunk pass
                    emb_all[idx] = emb_b
            if bboxes_crop.is_cuda:
                if bb_dev is None:
                    bb_dev = torch.zeros((n, 4), device=self.device)
                bb_dev[idx] = bboxes_crop.to(self.device, torch.float32)
            else:
                bb_h[loc] = bboxes_crop.numpy()
            ip_h[loc], in_h[loc] = np.asarray(isPrev, np.uint8), np.asarray(isNext, np.uint8)
            if thc_norm is not None and not self.dedup:                   # reference-faithful: explicit prev/next forwards
                ip, inx = vh.upload(ip_h[loc], self.device), vh.upload(in_h[loc], self.device)
                tp, tn = vh.thc_pairs(cur, prev, thc_norm), vh.thc_pairs(cur, nxt, thc_norm)
                one = (ip ^ inx).float()
                thc_ref[idx] = (tp * ip + tn * inx) * (1 + one)
            gt_all[loc] = GTkpts.reshape(len(idxs), -1).numpy()
            ids_all[loc, 0], ids_all[loc, 1] = np.asarray(img_ids, np.float64), np.asarray(ann_ids, np.float64)
            ann_all[loc] = bbox_xyxy_to_xywh(bboxes_ann.numpy().astype(np.float64))
            self._mark(f"batch@{a0}")
        bb_all, ip_all, in_all = (bb_dev if bb_dev is not None else vh.upload(bb_h, self.device)), vh.upload(ip_h, self.device), vh.upload(in_h, self.device)
        ae_flat = self.AE.packed() if self.AE is not None else None
        s = score_batch(hm_all, bb_all, ip_all, in_all, thc_norm=thc_norm if self.dedup else None, ae_flat=ae_flat,
                        ae_dims=(self.AE.input_dim, self.AE.z_dim) if self.AE is not None else (42, 4),
                        wpu_only38=(self.unc_kind == "WPU"))
        if thc_norm is not None and not self.dedup:
            s.thc = thc_ref
        unc = torch.zeros((n, 2), device=self.device)
        if self.unc_kind == "HP":
            unc[:, 0] = s.hp
        elif self.unc_kind == "TPC":
            if not self.dedup:
                raise ValueError("TPC needs an id-sorted stream dataset in this build")
            unc[:, 0] = vh.tpc_stream(hm_all, bb_all, s.keypoints[:, :, :2].contiguous(), ip_all, in_all)
        elif thc_norm is not None:
            unc[:, 0] = s.thc
            if self.unc_kind == "THC+WPU":
                self._wpu_status = s.wpu_status           # checked once the rows are on the host (a read-back here would drain the stream inside the loop)
                unc[:, 1] = s.wpu
        elif self.unc_kind == "WPU":
            self._wpu_status = s.wpu_status
            unc[:, 0] = s.wpu
        elif self.unc_kind in ("MPE", "Margin", "Entropy"):
            unc[:, 0] = multi_peak_scores(hm_all, self.unc_kind).float()
        kp = s.keypoints.reshape(n, -1)
        # compute_OKS on the device (al_metric.py:42-69): no D2H of the key-points inside the evaluation loop
        oks = vh.oks(s.keypoints.contiguous(), vh.upload(gt_all, self.device), vh.upload(ann_all, self.device))
        # float64 side rows for the result records (ActiveLearning.py:310-327): ids, annotation box (xywh), ground truth
        self._side = (lo, hi, vh.upload(np.concatenate([ids_all, ann_all, gt_all], 1), self.device))
        self._mark("scored")
        cols = [kp, unc, s.localpeak[:, None], oks.float()[:, None]]
        if emb_all is not None:
            cols.append(emb_all)
        return torch.cat(cols, 1).contiguous()

    @_collective
    def eval_and_query(self):
        self.model.eval()
        n = self.eval_len
        self._host_critical(True)                      # the record thread works only while this one waits for the device (see _write_records)
        return self._eval_and_query(n)

    def _eval_and_query(self, n):
        from . import distributed as D
        from . import query as Q
        if self.need_embedding and not hasattr(self.model, "get_embedding"):
            raise ValueError("this pose network has no get_embedding: representativeness / filters need it (SURVEY.md §9 item 3)")
        self.emb_dim = 2048
        width = 55 + (self.emb_dim if self.need_embedding else 0)
        # one process per GPU: every rank scores a contiguous shard (+ a one-item halo for the temporal scores)
        # and the result rows are all-gathered (active_learning/distributed.py); world size 1 = whole stream
        self._mark("eval: start")
        rows_dev = D.sharded_rows(n, self._score_range, width, self.device, halo=1 if self.dedup else 0)
        self._mark("eval: every batch enqueued")
        fvecs = rows_dev[:, 55:].contiguous() if self.need_embedding else None          # (n, 2048) stays on the device

        def side_rows(lo, hi):                     # the rows _score_range stashed for exactly this range
            assert self._side[0] == lo and self._side[1] == hi
            return self._side[2]
        side = D.sharded_rows(n, side_rows, 2 + 4 + 3 * self.cfg.DATA_PRESET.NUM_JOINTS, self.device, halo=1 if self.dedup else 0)
        self._host_critical(False)                     # everything is enqueued: while the main thread waits for the device, the record thread runs
        side = side.cpu().numpy()
        self._mark("eval: side rows on the host")
        self._side = None
        rows = rows_dev[:, :55].cpu().numpy()
        self._check_wpu(self.__dict__.pop("_wpu_status", None))      # compute_hybrid's two asserts (hybrid_feature.py:25,31), for this rank's shard
        self._host_critical(True)
        self._mark("eval: rows on the host")
        kp_all = rows[:, :51].copy()
        unc = rows[:, 51:53].astype(np.float64)
        lp = rows[:, 53].astype(np.float64)
        oks = rows[:, 54].astype(np.float64)
        self.keypoints, self.oks = kp_all, oks
        self._write_records(kp_all, oks, side)
        self._mark("eval: records handed over")
        evaluate = getattr(self.opt, "evaluate_fn", None)
        work_dir = getattr(self.opt, "work_dir", None)
        tp = self._third_party_scores(work_dir) if (work_dir and not evaluate and D.is_main()) else {}
        fallback = {"AP": None, "mOKS": float(oks.mean())}         # device OKS of every item: always available
        if evaluate and work_dir:
            self.flush_records()                                   # a user evaluator may read predicted_kpt.json / GT_kpt.json of THIS round
        res = evaluate(kp_all, self) if evaluate else (dict(tp["res"], mOKS=fallback["mOKS"]) if tp.get("res") else fallback)
        res_ann = res if (evaluate or not tp.get("res_ann")) else dict(tp["res_ann"], mOKS=fallback["mOKS"])
        self.percentage.append(len(self.labeled_id) / n * 100)
        self.performance.append(res)
        self.performance_ann.append(res_ann)
        self.ospa_list.append(tp.get("ospa"))
        self.ospa_list_ann.append(tp.get("ospa_ann"))
        self.uncertainty_mean.append(float(unc[:, 0].sum() / n))
        un = np.asarray(self.unlabeled_id, int)                       # ascending, like IndexCollection.index here
        nun = len(un)

        # ---- representativeness (ActiveLearning.py:465-480)
        influence = None
        if self.representativeness != "None":
            if nun in (0, 1):
                influence = np.zeros(nun)
            elif self.representativeness == "Influence":
                influence = Q.influence_scores(fvecs[torch.as_tensor(un, device=self.device)])
            else:
                influence = np.random.rand(nun)
            self.influence_dict[f"Round{self.round_cnt}"] = {int(i): float(v) for i, v in zip(un, influence)}
        combine_weight = 0.0
        if nun > 0:
            combine_weight = float(np.sum(lp[un]) / nun)              # mean local-peak value of the unlabeled items (:411-414, 486-488)
            self.combine_weight.append(combine_weight)
        self.uncertainty_dict[f"Round{self.round_cnt}"] = {int(i): (unc[i].tolist() if self.unc_kind == "THC+WPU" else float(unc[i, 0])) for i in range(n)}

        # ---- total score (:490-527)
        if nun in (0, 1):
            total = np.zeros(nun)
        elif self.uncertainty != "None":
            u = self._total_score(unc[un])
            total = combine_weight * u + (1 - combine_weight) * influence if self.representativeness != "None" else u
        elif self.representativeness == "None":
            total = np.zeros(nun)
        else:
            total = influence
        order = sorted(range(nun), key=lambda k: total[k], reverse=True)       # stable, like sorted(score_dict.items(), reverse=True)
        ranked = [int(un[k]) for k in order]

        # ---- candidates and filter (:529-617)
        if self.filter == "None":
            candidates = sorted(ranked[: self.query_size])
        elif self.filter in ("weighted", "K-Means", "Coreset"):
            candidates = sorted(ranked)
        else:
            candidates = sorted(ranked[: 8 * self.query_size])
        score_of = {int(un[k]): float(total[k]) for k in range(nun)}
        if nun in (0, 1) or self.filter == "None":
            query = candidates
        elif self.filter == "Random":
            query = self.random_query(list(candidates), self.query_size)
        elif self.filter == "Diversity":
            query = Q.diversity_queries(fvecs[torch.as_tensor(candidates, device=self.device)], candidates, self.query_size)
        elif self.filter == "Coreset":
            unc_list = np.zeros(n)
            unc_list[candidates] = total                               # `np.array(list(total_score))`: unlabeled order, as the reference assigns it
            if self.uncertainty == "None" or self.unc_lambda == 0:
                mode = "kcenter"
            else:
                mode = "fixed" if getattr(self.opt, "fixed_lambda", False) else "moks"
            query = Q.coreset_selection(fvecs, self.labeled_id, unc_list, self.query_size, mode, self.moks_queried, self.unc_lambda)
        else:                                                          # "weighted" / "K-Means": sklearn on the host, like the reference
            emb = fvecs[torch.as_tensor(candidates, device=self.device)].double().cpu().numpy()
            if self.filter == "weighted":
                _, first = np.unique(emb, axis=0, return_index=True)   # drop duplicate embeddings
                emb = emb[first]
                cand = [candidates[i] for i in first]
                weight = (1 + self.w_unc * combine_weight * np.array([score_of[c] for c in candidates]))[first]
                if nun <= self.query_size:
                    self.query_size = nun
                self.query_size = min(self.query_size, len(emb))
                query, _ = Q.kmeans_queries(emb, cand, self.query_size, weight)
            else:
                if nun < self.query_size:
                    self.query_size = nun
                query, _ = Q.kmeans_queries(emb, candidates, self.query_size)

        # Random representativeness / Random filter / the first k-center pick draw from the process-local numpy RNG: every
        # rank adopts rank 0's selection, or the ranks' labeled sets (and with them the fine-tune collectives) would diverge
        query, self.query_size = D.broadcast_object(([int(q) for q in query], int(self.query_size)), src=0)

        # ---- book-keeping (:619-650)
        thr = self.finish_acc + self.finish_margin
        lab, unl = set(self.labeled_id), set(self.unlabeled_id)
        key = f"Round{self.round_cnt}"
        self.true_labeled_dict[key] = [i for i in range(n) if i in lab and oks[i] >= thr]
        self.true_unlabeled_dict[key] = [i for i in range(n) if i in unl and oks[i] >= thr]
        self.false_labeled_dict[key] = [i for i in range(n) if i in lab and oks[i] < thr]
        self.false_unlabeled_dict[key] = [i for i in range(n) if i in unl and oks[i] < thr]
        if nun == 0:
            return
        query = [int(q) for q in query]
        self.moks_queried = float(np.mean(oks[query])) if query else 0.0          # get_retrain_id (:852-871)
        self.moksQ_list.append(self.moks_queried)
        self.retrain_id = [i for i in self.labeled_id if oks[i] <= thr] + query
        self.labeled_id = sorted(lab | set(query))
        self.unlabeled_id = [i for i in self.unlabeled_id if i not in set(query)]
        self.query_list_list[key] = query
        self._is_finished(query, oks)
        if self.actual_finish < 100:
            self.is_early_stop = True

    # ------------------------------------------------------------------ result records (ActiveLearning.py:310-327, 438-447, 693-705)
    # Formatting ~110 floats per item and writing three files costs the host ~20 ms per 1024 items — as much as a quarter of the device's
    # evaluation pass — and nothing in eval_and_query needs the result.  The records of a round are therefore built, encoded and written
    # by a host thread (`_RecordsJob`).  The two threads share the interpreter lock, so the thread works only inside WINDOWS: the main thread
    # opens the thread's gate where it waits for the device (the read-backs at the end of eval_and_query, the per-epoch read-backs of
    # retrain_model) and closes it again afterwards; the thread checks the gate (and yields the lock) every ~100 records, so it stops within a
    # fraction of a millisecond.  In an evaluate -> retrain -> evaluate loop the files of round r are produced inside the waits of what
    # follows.  A job that finds no window for 0.25 s stops waiting for one (an idle or foreign main thread must not starve it).
    # `flush_records()` (called by the next `_write_records`, `_third_party_scores`, `outcome`, and by every reader of `kpt_json` /
    # `kpt_json_ann` / `GT_json`) opens the gate and waits for the job.
    def _gate(self):
        g = self.__dict__.get("_records_gate")
        if g is None:
            import threading
            g = self.__dict__["_records_gate"] = threading.Event()
            g.set()
        return g

    def _mark(self, label):
        """Wall-clock check-points of the entry points for tools/al_eval_bench.py --trace (``self._trace`` = a list, or absent: nothing recorded)."""
        t = self.__dict__.get("_trace")
        if t is not None:
            import time
            t.append((label, time.perf_counter()))

    def _host_critical(self, on: bool):
        """on: the main thread is about to prepare / enqueue device work (the record thread pauses at its next check-point); off: it waits."""
        (self._gate().clear if on else self._gate().set)()

    def flush_records(self):
        """Wait until the record lists of the last evaluated round exist and its files are on disk (re-raises what the thread raised)."""
        job = self.__dict__.get("_records_job")
        if job is not None:
            gate = self._gate()
            was_open = gate.is_set()
            gate.set()
            job.join()
            if not was_open:
                gate.clear()
            self.__dict__["_records_job"] = None
            if job.error is not None:
                raise job.error

    kpt_json = property(lambda self: self._record_lists()[0])
    kpt_json_ann = property(lambda self: self._record_lists()[1])
    GT_json = property(lambda self: self._record_lists()[2])

    def _write_records(self, kp_all, oks, side):
        """The reference's result records (ActiveLearning.py:310-327, 438-447): one COCO-style dict per item in
        ``self.kpt_json`` (predictions), ``self.kpt_json_ann`` (labeled items carry their ground truth) and ``self.GT_json``;
        written as predicted_kpt.json / predicted_kpt_ann.json / GT_kpt.json when ``opt.work_dir`` is set, so the
        third-party evaluate_mAP / ospa_for_loc tools can be run on them (they stay outside this package).  Asynchronous: see above."""
        import threading
        from . import distributed as D
        self.flush_records()                                       # one round at a time, files in round order
        work_dir = getattr(self.opt, "work_dir", None)
        al, gate = self, self._gate()
        args = (np.array(kp_all, np.float32), np.array(oks, np.float64), np.array(side, np.float64), set(self.labeled_id),
                work_dir if (work_dir and D.is_main()) else None)

        import time

        class _RecordsJob(threading.Thread):
            error = None
            gated = True

            def pause(self):
                if self.gated and not gate.wait(0.25):
                    self.gated = False                             # no window in sight: run to completion beside whatever the main thread does
                    al._mark("records: ungated")
                time.sleep(0)                                      # hand the interpreter lock to a main thread that is waiting for it

            def run(self):
                try:
                    al._mark("records: job starts")
                    al._build_records(*args, pause=self.pause)
                    al._mark("records: job done")
                except BaseException as e:                         # surfaced by flush_records() on the main thread
                    self.error = e
        job = self.__dict__["_records_job"] = _RecordsJob(name="vatl-records", daemon=False)
        job.start()

    def _build_records(self, kp32, oks, side, labeled, work_dir, pause=lambda: None):
        """The record thread's work: the record COLUMNS of the round (python lists straight from the arrays), and — when there is a work
        directory — the three files' text from those columns.  The per-item dicts of ``kpt_json`` / ``kpt_json_ann`` / ``GT_json`` are only built when
        somebody reads them (``_record_lists``): 3 x 1024 dicts per round are ~10 k collector-tracked objects, and the full collections they
        trigger every few rounds (50 - 100 ms each over a process that holds a model and a data set) landed inside evaluation rounds."""
        pause()
        conf = kp32[:, 2::3]
        # whole columns converted once (a numpy call per item costs more than the record itself).  score = float(np.mean(s) + 1.25 * np.max(s))
        # (ActiveLearning.py:314) with the promotion of the reference's pinned numpy 1.23.5: float32 mean, then float64 product and sum
        cols = {"score": (conf.mean(1).astype(np.float64) + 1.25 * conf.max(1).astype(np.float64)).tolist() if len(kp32) else [],
                "keypoints": kp32.astype(np.float64).tolist(), "GT_keypoints": side[:, 6:].tolist(), "bbox": side[:, 2:6].tolist(),
                "image_id": side[:, 0].astype(np.int64).tolist(), "id": side[:, 1].astype(np.int64).tolist(), "OKS": oks.tolist(),
                "category_id": [1] * len(kp32)}
        self.__dict__["_records_cols"], self.__dict__["_records_labeled"] = cols, labeled
        self.__dict__["_records_lists"] = None
        self._mark("records: columns")
        if work_dir:
            import os
            os.makedirs(work_dir, exist_ok=True)
            texts = self._records_text(cols, labeled, pause)
            self._mark("records: text")
            pause()
            for name, text in zip(("predicted_kpt.json", "predicted_kpt_ann.json", "GT_kpt.json"), texts):
                tmp = os.path.join(work_dir, f".{name}.{os.getpid()}.tmp")         # a reader never sees a half-written file: the previous round's or this one's
                with open(tmp, "w") as f:
                    f.write(text)
                os.replace(tmp, os.path.join(work_dir, name))

    def _record_lists(self):
        """(kpt_json, kpt_json_ann, GT_json) of the last evaluated round: one dict per item (ActiveLearning.py:310-327), built on first use."""
        self.flush_records()
        lists = self.__dict__.get("_records_lists")
        if lists is None:
            cols, labeled = self.__dict__.get("_records_cols"), self.__dict__.get("_records_labeled", ())
            kpt_json, kpt_json_ann, GT_json = [], [], []
            if cols is not None:
                keys = ("bbox", "image_id", "id", "score", "category_id", "keypoints", "GT_keypoints", "OKS")
                for i in range(len(cols["id"])):
                    rec = {k: cols[k][i] for k in keys}
                    kpt_json.append(rec)
                    kpt_json_ann.append(dict(rec, keypoints=rec["GT_keypoints"]) if i in labeled else dict(rec))
                    GT_json.append(dict(rec, keypoints=rec["GT_keypoints"]))
            lists = self.__dict__["_records_lists"] = (kpt_json, kpt_json_ann, GT_json)
        return lists

    def _records_json(self):
        """The three files' text of the last evaluated round (waits for the record thread)."""
        self.flush_records()
        cols = self.__dict__.get("_records_cols")
        if cols is None:
            cols = {k: [] for k in ("bbox", "image_id", "id", "score", "category_id", "keypoints", "GT_keypoints", "OKS")}
        return self._records_text(cols, self.__dict__.get("_records_labeled", ()))

    def _records_text(self, cols, labeled, pause=lambda: None):
        """json.dumps of ``kpt_json``, ``kpt_json_ann`` and ``_gt_dict()`` — the same text, character for character — with every
        record's shared fields (box, ids, score, the two key-point lists, OKS) encoded ONCE instead of once per file: the three files differ only
        in which of the two lists a record's "keypoints" holds, and formatting ~110 floats per record is what writing them costs.  A field is
        encoded for 128 records per encoder call (a column of the records as one nested list, cut at the inner brackets), and the fields that
        come from the data set alone (box, ids, ground truth: the same in every round of a run) are kept from the previous call and reused when
        their values are unchanged.  ``pause`` is called between the encoder calls (the record thread's check-point)."""
        import json
        enc = json.dumps

        def column(key, nested):
            vals = cols[key]
            pieces = []
            for a in range(0, len(vals), 128):                    # (a call holds the interpreter lock: ~0.4 ms for 128 records)
                pause()
                text = enc(vals[a:a + 128])
                pieces.extend(("[" + t + "]" for t in text[2:-2].split("], [")) if nested else text[1:-1].split(", "))
            return pieces
        fixed_vals, fixed_text = [], []
        cached = self.__dict__.get("_records_fixed")
        for k, (key, nested) in enumerate((("bbox", True), ("image_id", False), ("id", False), ("category_id", False), ("GT_keypoints", True))):
            vals = cols[key]
            if cached is not None and cached[0][k] == vals:          # (list comparison: a NaN in the ground truth re-encodes, which is only slower)
                text = cached[1][k]
            else:
                text = column(key, nested)
            fixed_vals.append(vals); fixed_text.append(text)
        self._records_fixed = (fixed_vals, fixed_text)
        box_s, img_s, id_s, cat_s, gt_s = fixed_text
        kp_s, score_s, oks_s = column("keypoints", True), column("score", False), column("OKS", False)
        n = len(cols["id"])
        assert all(len(t) == n for t in (box_s, img_s, id_s, cat_s, gt_s, kp_s, score_s, oks_s)), "a record field does not encode to one piece per record"
        pause()
        pred, ann, gt = [], [], []
        for i in range(n):
            if i % 128 == 127:
                pause()
            head = '{"bbox": %s, "image_id": %s, "id": %s, "score": %s, "category_id": %s, "keypoints": ' % (box_s[i], img_s[i], id_s[i], score_s[i], cat_s[i])
            tail = ', "GT_keypoints": %s, "OKS": %s}' % (gt_s[i], oks_s[i])
            pred.append(head + kp_s[i] + tail)
            ann.append(head + (gt_s[i] if i in labeled else kp_s[i]) + tail)
            gt.append(head + gt_s[i] + tail)
        g = self._gt_dict(image_ids=cols["image_id"])
        assert list(g) == ["images", "categories", "annotations"]
        return ("[" + ", ".join(pred) + "]", "[" + ", ".join(ann) + "]",
                '{"images": %s, "categories": %s, "annotations": [%s]}' % (enc(g["images"]), enc(g["categories"]), ", ".join(gt)))

    def _gt_dict(self, image_ids=None):
        """``save_GT_dict`` (ActiveLearning.py:693-705): the ground truth in COCO layout — ``images`` / ``categories`` copied from the
        evaluation set's annotation file when there is one, the records of this round as ``annotations`` (``image_ids`` given: header only,
        ``annotations`` left empty — the record thread assembles that part from its column text)."""
        import json
        import os
        ev = self.cfg.DATASET.EVAL
        path = os.path.join(str(ev.get("ROOT", "")), str(ev.get("ANN", ""))) if ev.get("ANN") else ""
        if path and os.path.isfile(path):
            with open(path) as f:
                src = json.load(f)
            images, cats = src.get("images", []), src.get("categories", [])
        else:                                                      # datasets without an annotation file (synthetic / in-memory videos)
            ids = image_ids if image_ids is not None else [r["image_id"] for r in self.GT_json]
            images = [{"id": i, "image_id": i} for i in sorted(set(ids))]
            cats = [{"id": 1, "name": "person"}]
        return {"images": images, "categories": cats, "annotations": [] if image_ids is not None else self.GT_json}

    def _third_party_scores(self, work_dir):
        """mAP / OSPA of the written records through the reference's third-party tools (ActiveLearning.py:442-447) when they are
        installed (pycocotools / halpecocotools, JRDB_toolkit); None for whatever is missing."""
        import os
        from alphapose.utils.metrics import evaluate_mAP, have_coco_tools
        gt, out = os.path.join(work_dir, "GT_kpt.json"), {}
        have_ospa = True
        try:
            import JRDB_toolkit.pose_eval  # noqa: F401
        except ImportError:
            have_ospa = False
        if have_coco_tools() or have_ospa:                        # somebody is about to read the files: they must be complete
            self.flush_records()
        for key, name in (("res", "predicted_kpt.json"), ("res_ann", "predicted_kpt_ann.json")):
            try:
                out[key] = evaluate_mAP(os.path.join(work_dir, name), ann_type="keypoints", ann_file=gt, silence=True)
            except NotImplementedError:
                out[key] = None
        try:
            from JRDB_toolkit.pose_eval import ospa_for_loc
            out["ospa"] = ospa_for_loc(ann_json_path=gt, pr_json_path=os.path.join(work_dir, "predicted_kpt.json"))
            out["ospa_ann"] = ospa_for_loc(ann_json_path=gt, pr_json_path=os.path.join(work_dir, "predicted_kpt_ann.json"))
        except ImportError:
            out["ospa"] = out["ospa_ann"] = None
        return out

    def _is_finished(self, query, oks):
        """ActiveLearning.py:707-725: the three stopping-criterion bookmarks (label percentage at which each first held)."""
        time = len(self.labeled_id) / self.eval_len * 100
        if np.all(oks >= self.finish_acc) and time < self.actual_finish:
            self.actual_finish = time
        if query and np.mean(oks[query]) >= self.finish_acc and time < self.finished_minerror:
            self.finished_minerror = time
        if np.all(oks[self.labeled_id] >= self.finish_acc) and time < self.finished_oursc:
            self.finished_oursc = time

    @staticmethod
    def random_query(candidate_list, query_size):
        """ActiveLearning.py:727-734."""
        out = []
        while len(out) < query_size and len(candidate_list) > 0:
            q = int(np.random.choice(candidate_list))
            out.append(q)
            candidate_list.remove(q)
        return out

    @staticmethod
    def _check_wpu(status):
        if status is None:
            return
        st = status.cpu().numpy()
        assert not (st == 1).any(), "height of human body must be positive!"
        assert not (st == 2).any(), "at least one visible keypoint is required!"

    def _total_score(self, u):
        """Min-max normalised uncertainty of the unlabeled items (ActiveLearning.py:492-519)."""
        def norm(v):
            rng = v.max() - v.min()
            return (v - v.min()) / rng if rng > 0 else np.zeros_like(v)
        if len(u) < 2 or self.uncertainty == "None":
            return np.zeros(len(u))
        if self.unc_kind == "THC+WPU":
            t, w = norm(u[:, 0]), norm(u[:, 1])
            mode = getattr(self.opt, "THCvsWPU", "const")
            r = len(self.labeled_id) / self.eval_len
            mix = t + w if mode == "const" else (r * t + (1 - r) * w if mode == "increase" else (1 - r) * t + r * w)
            return norm(mix)
        return norm(u[:, 0])

    # ------------------------------------------------------------------ hot loop 2
    @_collective
    def retrain_model(self):
        """ActiveLearning.py:651-686.  Data parallel like the reference's nn.DataParallel (:233, 667): every rank sees the same
        shuffled mini-batches (shared seed); a mini-batch is cut into ``self.replicas`` DataParallel chunks (``Tensor.chunk``
        sizes) and rank r takes chunks r, r + world, ... — one chunk per rank when there is a process per GPU.  Every chunk is a
        forward/backward of its own (per-replica BatchNorm statistics, exactly DataParallel's), its loss gradient scaled by
        the chunk's share of the mini-batch, so the SUM over chunks and ranks is the gradient of the mean loss over the whole
        mini-batch (:669).  Gradients live in one flat arena that is all-reduced in place while the backward pass is still
        running (active_learning/distributed.py: GradArena)."""
        from alphapose.models import hip_train
        from . import distributed as D
        loss_logger, acc_logger = DataLogger(), DataLogger()
        subset = Subset(self.train_dataset, self.retrain_id)
        world, rank = D.world_rank()
        gen = torch.Generator()
        gen.manual_seed(D.shared_seed())
        loader = DataLoader(subset, batch_size=self.cfg.RETRAIN.BATCH_SIZE * self.replicas, shuffle=True, num_workers=0, drop_last=False,
                            collate_fn=self.collate_fn, generator=gen)
        self.model.train()
        trainer = hip_train.trainer_for(self.model)
        arena = hip_train.arena_for(self.model)
        # the per-step loss / accuracy read-backs wait until the EPOCH's steps are enqueued: read inside the loop they drain the stream every step, and the host's
        # preparation of the next mini-batch (crops, targets) then runs behind the step instead of beside it.  Flushed once per epoch (one synchronisation, before
        # the scheduler step): the host never runs more than an epoch ahead of the device, a kernel error surfaces in the epoch that produced it (a NaN loss reaches the logger of that epoch), and the
        # loggers hold every finished epoch if a later one raises.  Same values, same order into the loggers.
        pending = []
        self._host_critical(True)

        def flush():
            self._host_critical(False)                                                     # a wait for the device: the record thread's turn
            for loss, acc_finish, cnt in pending:
                loss_logger.update(float(loss), cnt)                                   # (a NaN is logged, not raised: the reference's behaviour)
                acc_logger.update(acc_finish(), cnt)
            pending.clear()
            self._host_critical(True)
        for _ in range(self.retrain_epoch):
            for (idxs, inps, labels, label_masks, *_rest) in loader:
                nb = len(idxs)
                mine = D.chunk_bounds(nb, self.replicas)[rank::world]
                acc, kept = None, None
                arena.begin()
                if not mine:                                                           # fewer chunks than ranks: contribute zeros
                    arena.flat.zero_()
                for k, (lo, hi) in enumerate(mine):
                    x = vh.upload(inps[lo:hi, 0], self.device, torch.float32).contiguous()
                    lab, msk = vh.upload(labels[lo:hi], self.device, torch.float32).contiguous(), vh.upload(label_masks[lo:hi], self.device, torch.float32)
                    with torch.no_grad():
                        out = trainer.forward(x)
                        loss, dout = vh.masked_mse_fwd_bwd(out, lab, msk)              # 0.5 * MSE(out*m, label*m) and its gradient
                        if hi - lo != nb:
                            dout.mul_((hi - lo) / nb)
                        if len(mine) > 1 and k == 0:                                   # DataParallel keeps replica 0's BN statistics only
                            kept = [b.clone() for b in self.model.buffers()]
                        trainer.backward(dout, arena=arena, overlap=len(mine) == 1)
                        if len(mine) > 1:                                              # several replicas walked on one GPU: accumulate
                            acc = arena.flat.clone() if acc is None else acc.add_(arena.flat)
                    m = msk.reshape(msk.shape[0], -1, 1, 1)
                    pending.append((loss, calc_accuracy_begin(out * m, lab * m), hi - lo))
                if acc is not None:
                    arena.flat.copy_(acc)
                    with torch.no_grad():
                        for b, kb in zip(self.model.buffers(), kept):
                            b.copy_(kb)
                arena.finish()
                arena.attach()
                self.optimizer.step()
            flush()
            self.scheduler.step()
        D.broadcast_buffers_(self.model)           # BN statistics are per rank; rank 0's survive (DataParallel semantics)
        self.last_train_loss, self.last_train_acc = self._global_avg(loss_logger), self._global_avg(acc_logger)
        if "WPU" in self.uncertainty:              # ActiveLearning.py:680-684: a fresh AE is fine-tuned on the labeled poses
            self.AE = self.initialize_AE()
            self.last_ae_loss = self.retrain_AE()
            D.broadcast_module_(self.AE)           # the fit shuffles with the rank's own RNG: rank 0's AE is the one every shard scores with

    def _global_avg(self, logger):
        """Item-weighted average of a DataLogger over all ranks (the reference logs the loss of the gathered mini-batch)."""
        from . import distributed as D
        t = torch.tensor([logger.sum, logger.cnt], dtype=torch.float64, device=self.device)
        D.allreduce_sum_(t)
        return float(t[0] / t[1]) if float(t[1]) > 0 else 0.0

    def retrain_AE(self):
        """ActiveLearning.py:905-925.  The reference reads the hybrid features of the labeled people from its `Wholebody`
        annotation dataset; here they are computed on the device from the ground-truth key-points and annotation boxes of
        the labeled items (`compute_hybrid`, hybrid_feature.py:14-59), people without a visible key-point skipped like
        Whole_body_hybrid.py:54-55."""
        from .Whole_body_AE.AutoEncoder import fit_autoencoder
        epochs = int(self.cfg.AE.get("EPOCH", 0))
        if epochs <= 0 or not self.labeled_id:
            return 0.0
        gts, boxes = [], []
        annotation_of = getattr(self.eval_dataset, "annotation_of", None)     # (GT key-points, annotation box) without making the crops
        for i in self.labeled_id:
            if annotation_of is not None:
                gt, box = annotation_of(i)
            else:
                item = self.eval_dataset[i]
                gt, box = item[4], item[8]
            gt = np.asarray(gt, np.float64).reshape(-1)
            if gt[2::3].sum() == 0:
                continue
            gts.append(gt)
            boxes.append(bbox_xyxy_to_xywh(np.asarray(box, np.float64).tolist()))
        if not gts:
            return 0.0
        feat, status = vh.hybrid_feature_f64(torch.as_tensor(np.stack(gts), device=self.device), torch.as_tensor(np.asarray(boxes, np.float64), device=self.device))
        feat = feat[status == 0].float()[:, :self.AE.input_dim].contiguous()
        return fit_autoencoder(self.AE, feat, epochs, float(self.cfg.AE.get("LR", 1e-3)))

    # ------------------------------------------------------------------ round logic (ActiveLearning.py:166-205)
    @_collective
    def outcome(self):
        if self.is_early_stop or self.one_by_one:
            # ActiveLearning.py:168-178: an early-stopped video still reports len(query_ratio)+1 points per curve — the last
            # entry repeated, the label percentage advanced along the query schedule
            while len(self.performance) <= len(self.query_ratio):
                self.round_cnt += 1
                self.performance.append(self.performance[-1])
                self.performance_ann.append(self.performance_ann[-1])
                self.ospa_list.append(self.ospa_list[-1])
                self.ospa_list_ann.append(self.ospa_list_ann[-1])
                self.uncertainty_mean.append(self.uncertainty_mean[-1])
                self.percentage.append(self.query_ratio[self.round_cnt - 1] * 100)
                self.combine_weight.append(self.combine_weight[-1])
                self.moksQ_list.append(self.moksQ_list[-1])
            finish = True
        else:
            if not self.continual:
                self.model, self.optimizer, self.scheduler = self.initialize_estimator()
                self.retrain_epoch = int(self.cfg.RETRAIN.BASE * len(self.labeled_id) / self.eval_len + self.cfg.RETRAIN.ALPHA * (1 - self.moks_queried))
            else:
                self.retrain_epoch = int(self.cfg.RETRAIN.ALPHA * (1 - self.moks_queried))
            self.retrain_model()
            self.round_cnt += 1
            finish = False
            if len(self.unlabeled_id) == 0:
                self.eval_and_query()
                finish = True
            elif self.round_cnt >= len(self.query_ratio):
                self.query_size = len(self.unlabeled_id)
            else:
                self.query_size = self.query_sizes[self.round_cnt] - len(self.labeled_id)
        if not finish:
            return None
        self.flush_records()                           # the last round's files are complete when the run hands its results back
        return (self.percentage, self.performance, self.performance_ann, self.query_list_list, self.uncertainty_dict, self.uncertainty_mean,
                self.influence_dict, self.combine_weight, self.spearmanr_list, self.corr_list, self.true_labeled_dict, self.true_unlabeled_dict,
                self.false_labeled_dict, self.false_unlabeled_dict, self.actual_finish, self.finished_minerror, self.finished_oursc,
                self.ospa_list, self.ospa_list_ann, self.moksQ_list)
