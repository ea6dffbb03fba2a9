"""One process per GPU around the hot path: process-group bring-up, frame sharding of the
evaluation stream, the flat gradient arena of the data-parallel fine-tune step, and the worker processes that
stand in for the reference's ``nn.DataParallel`` replicas (ActiveLearning.py:233).

* **Bring-up.**  ``init_from_env()`` joins the torchrun rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*) before
  the GPU is touched; backend "nccl" is RCCL over xGMI on MI355X (``VATL_DIST_BACKEND=gloo`` for CPU tests and for
  several ranks sharing one GPU).  When the *unchanged* reference driver runs as one plain process on an N-GPU node
  (``opt.num_gpu = torch.cuda.device_count()``, Run_active_learning.py:92-94), ``ensure_workers()`` starts N-1 fresh
  child processes (``python -m active_learning.worker``; never an exec of this process) that mirror every
  ``ActiveLearning`` call rank 0 makes — the package's replacement for DataParallel's replica threads.
* **Evaluation.**  The id-sorted item stream is cut into contiguous shards; THC/TPC need the heat-maps of the id-adjacent
  items, so a shard is extended by a one-item halo on each interior side and the halo items are simply re-computed
  locally (two extra forwards per rank instead of any heat-map exchange).  The only collective is one all-gather of the
  per-item result rows (~290 bytes per item).  Parameters live on every GPU: nothing is re-broadcast per call (the
  reference's DataParallel re-broadcasts 136 MB per forward, ActiveLearning.py:233,277).
* **Fine-tune.**  ``GradArena``: one flat fp32 buffer whose slices ARE the ``.grad`` tensors the weight-gradient
  kernels write; it is all-reduced in place, bucket by bucket, while the backward pass is still producing the earlier
  layers' gradients (async collectives on RCCL's own stream).  No concatenation, no copy back, no division pass (the loss
  gradient is pre-scaled by the rank's share of the mini-batch, so SUM is the mean-loss gradient).
"""
from __future__ import annotations

import atexit
import os
import pickle
import socket
import subprocess
import sys
import tempfile

import torch
import torch.distributed as dist


# ---------------------------------------------------------------------------------------------------------------------
# process group
# ---------------------------------------------------------------------------------------------------------------------

def backend_name() -> str:
    return os.environ.get("VATL_DIST_BACKEND", "nccl")


def local_device_index() -> int:
    """The GPU of this rank: LOCAL_RANK modulo the visible devices (several gloo ranks may share one GPU in tests)."""
    ndev = max(torch.cuda.device_count(), 1)
    return int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))) % ndev


def init_from_env() -> bool:
    """Join the process group described by the torchrun environment, if there is one.  Must run before this process
    allocates on a GPU: it selects the rank's device first.  Returns True when a group with more than one rank is up."""
    if dist.is_initialized():
        return dist.get_world_size() > 1
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend_name()
    if torch.cuda.device_count() > 0:
        idx = local_device_index()
        torch.cuda.set_device(idx)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", idx))
            return True
    dist.init_process_group(backend)
    return True


def is_main() -> bool:
    return not dist.is_initialized() or dist.get_rank() == 0


def world_rank():
    return (dist.get_world_size(), dist.get_rank()) if dist.is_initialized() else (1, 0)


def _comm_device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def shared_seed() -> int:
    """One random seed agreed by all ranks (rank 0's): the ranks must shuffle the fine-tune set identically before
    each takes its slice of every mini-batch."""
    seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        seed = seed.to(_comm_device())
        dist.broadcast(seed, src=0)
    return int(seed.item())


def broadcast_object(obj, src: int = 0):
    """Rank ``src``'s (small, picklable) object on every rank — the query indices a round selected, for instance:
    anything drawn from a rank-local RNG must be agreed this way before the ranks' book-keeping may depend on it."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, device=_comm_device())
    return box[0]


# ---------------------------------------------------------------------------------------------------------------------
# evaluation: contiguous shards + halo, one all-gather of result rows
# ---------------------------------------------------------------------------------------------------------------------

def shard_bounds(n: int, rank: int, world: int):
    """Contiguous balanced shard [lo, hi) of n items."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_bounds(n: int, replicas: int):
    """``torch.Tensor.chunk`` cuts of a mini-batch of n items — how nn.DataParallel scatters it over its replicas
    (ceil(n / replicas) items each, possibly fewer chunks than replicas): [(lo, hi), ...], never empty chunks."""
    if n <= 0:
        return []
    size = -(-n // max(1, replicas))
    return [(lo, min(lo + size, n)) for lo in range(0, n, size)]


def halo_bounds(n: int, lo: int, hi: int, halo: int = 1):
    """Shard extended by the halo, clipped to the stream: (lo_ext, hi_ext, skip_front, skip_back)."""
    lo_e, hi_e = max(0, lo - halo), min(n, hi + halo)
    return lo_e, hi_e, lo - lo_e, hi_e - hi


def sharded_rows(n: int, score_fn, row_width: int, device, halo: int = 1) -> torch.Tensor:
    """Every rank scores its shard (+halo) with ``score_fn(lo_ext, hi_ext) -> (hi_ext-lo_ext, row_width)`` rows (any
    one dtype) on ``device``; returns the (n, row_width) result rows of the whole stream on every rank."""
    world, rank = world_rank()
    lo, hi = shard_bounds(n, rank, world)
    lo_e, hi_e, front, back = halo_bounds(n, lo, hi, halo)
    rows = score_fn(lo_e, hi_e)
    rows = rows[front:rows.shape[0] - back].contiguous()
    assert rows.shape == (hi - lo, row_width), (rows.shape, hi - lo, row_width)
    if world == 1:
        return rows
    sizes = [shard_bounds(n, r, world) for r in range(world)]
    pad = max(h - l for l, h in sizes)
    buf = torch.zeros((pad, row_width), device=device, dtype=rows.dtype)
    buf[:hi - lo] = rows
    out = torch.empty((world * pad, row_width), device=device, dtype=rows.dtype)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, buf)
    else:                                                  # gloo: per-rank destination views of the same buffer
        dist.all_gather(list(out.view(world, pad, row_width).unbind(0)), buf)
    if all(h - l == pad for l, h in sizes):
        return out
    keep = [r * pad + k for r, (l, h) in enumerate(sizes) for k in range(h - l)]       # drop the padding rows of the short shards
    return out[torch.as_tensor(keep, device=device)]


# ---------------------------------------------------------------------------------------------------------------------
# fine-tune: flat gradient arena, bucketed in-place all-reduce overlapped with the backward pass
# ---------------------------------------------------------------------------------------------------------------------

class GradArena:
    """One flat fp32 buffer holding every parameter gradient of a model, in ``parameters()`` order.

    ``view(p)`` is the slice the weight-gradient kernel of parameter ``p`` writes (and what becomes ``p.grad``).  The
    backward pass produces gradients from the last layer to the first, i.e. from the high end of the arena downwards.

    The arena is reduced as a FIXED sequence of buckets, cut from the top down every ``bucket_bytes`` — a function of
    (total, bucket_bytes) only: ``cuts = [total, total - bucket, ..., 0]``.  Every rank therefore issues exactly the same
    collectives (count, order, sizes) whatever it did locally — a rank that overlaps, a rank that walked several
    DataParallel chunks and reduces only at the end, and a rank that had no chunk of a short last mini-batch all run the
    same list; WHEN a bucket starts is the only thing that differs.  ``done_offset(o)`` tells the arena that every slice
    at or above element offset ``o`` is final: each not-yet-started bucket that lies wholly at or above ``o`` starts as an
    asynchronous in-place SUM all-reduce (on RCCL it runs on the communicator's stream and overlaps the rest of the
    backward pass).  ``finish()`` starts whatever is left, in the same order, and waits."""

    def __init__(self, params, device=None, bucket_bytes: int = 32 << 20):
        self.params = [p for p in params]
        self.offset, off = {}, 0
        for p in self.params:
            self.offset[p] = off
            off += p.numel()
        self.total = off
        dev = device if device is not None else (self.params[0].device if self.params else torch.device("cpu"))
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.views = {p: self.flat[o:o + p.numel()].view(p.shape) for p, o in self.offset.items()}
        self.bucket = max(1, bucket_bytes // 4)
        self.cuts = list(range(self.total, 0, -self.bucket)) + [0]       # descending bucket edges; the lowest bucket takes the remainder
        self._next = 0                                                   # cuts[_next] = upper edge of the first bucket not started yet
        self._works = []
        self._group = None
        self._live = False
        self.launches = 0
        self.fired = []                                                  # (lo, hi) of every collective of this step, in issue order

    def view(self, p):
        return self.views[p]

    def begin(self, group=None):
        """Start of a backward pass.  ``group`` None = the default process group (nothing to do at world size 1)."""
        self._next, self._works, self._group = 0, [], group
        self._live = dist.is_initialized() and dist.get_world_size(group) > 1
        self.launches = 0
        self.fired = []

    def _fire_next(self):
        hi, lo = self.cuts[self._next], self.cuts[self._next + 1]
        self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self._group, async_op=True))
        self.fired.append((lo, hi))
        self.launches += 1
        self._next += 1

    def would_fire(self, off: int) -> bool:
        """Whether ``done_offset(off)`` would start an all-reduce (callers with work on other streams join them first)."""
        return self._live and self._next + 1 < len(self.cuts) and self.cuts[self._next + 1] >= off

    def done_offset(self, off: int):
        """Every slice at or above element offset ``off`` holds its final value for this step."""
        if not self._live:
            return
        while self._next + 1 < len(self.cuts) and self.cuts[self._next + 1] >= off:
            self._fire_next()

    def finish(self):
        """Start the buckets that have not started yet and wait for all of them (the optimizer step comes next)."""
        if self._live:
            while self._next + 1 < len(self.cuts):
                self._fire_next()
            for w in self._works:
                w.wait()
        self._works = []

    def attach(self):
        """``p.grad`` = the arena slices (no copies)."""
        for p, v in self.views.items():
            p.grad = v


def allreduce_sum_(flat: torch.Tensor, group=None):
    """In-place SUM all-reduce of one flat buffer (no-op at world size 1)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)


def broadcast_buffers_(module, src: int = 0):
    """BatchNorm running statistics are computed per rank during a data-parallel fine-tune (like the replicas of
    nn.DataParallel); the copy that survives is rank ``src``'s."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for b in module.buffers():
        dist.broadcast(b, src=src)


def broadcast_module_(module, src: int = 0):
    """Parameters and buffers of rank ``src`` to every rank (after a random initialisation or a rank-local fit, so
    that the replicas start identical — what nn.DataParallel's per-forward replicate does for the reference)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
        torch.autograd.graph.increment_version(t)


# ---------------------------------------------------------------------------------------------------------------------
# worker processes behind the unchanged single-process driver
# ---------------------------------------------------------------------------------------------------------------------

_workers: list = []


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ensure_workers(num_gpu: int) -> int:
    """Called by ``ActiveLearning.__init__`` on the driver's process.  Under torchrun: join the group.  As one plain
    process with ``opt.num_gpu`` > 1 (the reference's DataParallel case): become rank 0 of a fresh group and start the
    other ranks as child processes running ``active_learning.worker`` (once per process; later ``ActiveLearning``
    objects reuse them).  Returns the world size.  ``VATL_SPAWN=0`` keeps everything on this process's GPU: the
    DataParallel replicas are then walked one after the other (same numerics, ActiveLearning.retrain_model)."""
    if dist.is_initialized():
        return dist.get_world_size()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        init_from_env()
        return dist.get_world_size()
    want = max(1, int(num_gpu))
    if backend_name() == "nccl":
        want = min(want, max(torch.cuda.device_count(), 1))          # RCCL refuses two ranks on one device
    if want <= 1 or os.environ.get("VATL_SPAWN", "1") == "0":
        return 1
    port = _free_port()
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = dict(os.environ, WORLD_SIZE=str(want), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VATL_WORKER_PARENT=str(os.getpid()))
    base["PYTHONPATH"] = os.pathsep.join([pkg_root] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p])
    for r in range(1, want):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        _workers.append(subprocess.Popen([sys.executable, "-m", "active_learning.worker"], env=env, stdin=subprocess.DEVNULL))
    os.environ.update(WORLD_SIZE=str(want), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0")
    init_from_env()
    atexit.register(shutdown_workers)
    return dist.get_world_size()


def have_workers() -> bool:
    return bool(_workers)


def command(*msg):
    """Rank 0 -> workers: the next thing every rank does together.  A worker that died (an exception in its replica) would leave
    rank 0 waiting in the next collective until the backend's timeout: fail here instead."""
    if _workers:
        dead = [(i + 1, p.returncode) for i, p in enumerate(_workers) if p.poll() is not None]
        if dead:
            raise RuntimeError(f"active_learning worker rank(s) exited: {dead} (rank, exit code); see their stderr above")
        broadcast_object(msg, src=0)


_failed = False                # a mirrored call raised on this rank: the workers are somewhere inside another collective
_payload_files: list = []


def mark_failed():
    """The driver's outermost mirrored call ended with an exception: from here on no collective may be assumed to match
    (the workers are blocked in whatever the call was doing), so ``shutdown_workers`` ends them directly."""
    global _failed
    _failed = True


def shutdown_workers():
    """At interpreter exit of the driver.  After a clean run: one ("exit",) broadcast, the workers leave by themselves.  After a
    failed call the exit broadcast would itself be a mismatched collective (it can block until RCCL's watchdog fires), so the
    children are terminated directly and the group is torn down without another collective."""
    global _workers
    if not _workers:
        return
    clean = not _failed and all(p.poll() is None for p in _workers)
    if clean:
        try:
            if dist.is_initialized():
                broadcast_object(("exit",), src=0)
        except Exception:                                  # a worker died meanwhile: fall through to the kill below
            clean = False
    for p in _workers:
        if not clean and p.poll() is None:
            p.terminate()
        try:
            p.wait(timeout=30 if clean else 5)
        except subprocess.TimeoutExpired:
            p.kill()
    _workers = []
    release_payloads()
    if dist.is_initialized():
        try:
            if clean:
                dist.destroy_process_group()
            # else: peers are gone — no collective teardown; the process is exiting anyway
        except Exception:
            pass


def dump_payload(obj) -> str:
    """Objects too large for a broadcast (datasets handed to the constructor) go through a temp file, removed by
    ``release_payloads`` once every worker has built its replica (end of the constructor) or at shutdown."""
    fd, path = tempfile.mkstemp(prefix="vatl_payload_", suffix=".pkl")
    with os.fdopen(fd, "wb") as f:
        pickle.dump(obj, f)
    _payload_files.append(path)
    return path


def release_payloads():
    while _payload_files:
        try:
            os.unlink(_payload_files.pop())
        except OSError:
            pass


def mirrored() -> bool:
    """This process is the driver of worker ranks, or one of those workers (every constructor is mirrored)."""
    return bool(_workers) or bool(os.environ.get("VATL_WORKER_PARENT"))


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
