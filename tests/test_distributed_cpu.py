"""N > 1 host logic on CPU with the gloo backend (world_size 2): contiguous shards with a one-item halo,
results all-gathered, equal to the single-process stream.  The per-item compute here is the numpy oracle
(checker role only) — the sharding code under test is the product's active_learning/distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import scorers, synth


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rows(hm, bb, is_prev, is_next, lo, hi):
    """Result rows of items lo..hi-1 computed from the local window only (neighbours inside the window)."""
    out = np.zeros((hi - lo, 51 + 2), np.float32)
    for k, i in enumerate(range(lo, hi)):
        d = scorers.decode_heatmaps(hm[i], bb[i])
        out[k, :51] = scorers.keypoints_51(d["coords"], d["maxvals"])
        prev = hm[i - 1] if (i - 1 >= lo and is_prev[i]) else None
        nxt = hm[i + 1] if (i + 1 < hi and is_next[i]) else None
        out[k, 51] = scorers.thc_item(hm[i], prev, nxt, prev is not None, nxt is not None)
        out[k, 52] = scorers.localpeak_mean(hm[i])
    return out


def _worker(rank, world, port, n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vatl4pose-wacv2024_amd")]
    import torch.distributed as dist
    from active_learning import distributed as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    hm = synth.blob_heatmaps(n, seed=3)
    bb = synth.bboxes(n, seed=3)
    is_prev, is_next = synth.video_flags(n, 2)
    got = D.sharded_rows(n, lambda lo, hi: torch.from_numpy(_rows(hm, bb, is_prev, is_next, lo, hi)), 53, torch.device("cpu"))
    flat = torch.cat([torch.ones(5) * (rank + 1), torch.ones(3) * 10 * (rank + 1)]) / world      # each rank's share, pre-scaled
    D.allreduce_sum_(flat)
    g = [flat[:5], flat[5:]]
    bn = torch.nn.BatchNorm2d(4)                                   # per-rank running statistics -> rank 0's survive
    bn.running_mean.fill_(float(rank + 1)); bn.running_var.fill_(float(10 * (rank + 1)))
    D.broadcast_buffers_(bn)
    assert float(bn.running_mean[0]) == 1.0 and float(bn.running_var[0]) == 10.0, (rank, bn.running_mean)
    if rank == 0:
        q.put((got.numpy(), [t.numpy() for t in g]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_and_halo():
    import sys
    from active_learning import distributed as D
    for n, w in ((10, 3), (1024, 8), (7, 8), (1, 2)):
        cuts = [D.shard_bounds(n, r, w) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
    assert D.halo_bounds(10, 0, 4) == (0, 5, 0, 1) and D.halo_bounds(10, 4, 7) == (3, 8, 1, 1) and D.halo_bounds(10, 7, 10) == (6, 10, 1, 0)


def test_sharded_stream_equals_single_process_world2():
    n = 11                                       # odd: uneven shards
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, grads = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    hm = synth.blob_heatmaps(n, seed=3); bb = synth.bboxes(n, seed=3)
    is_prev, is_next = synth.video_flags(n, 2)
    want = _rows(hm, bb, is_prev, is_next, 0, n)
    np.testing.assert_array_equal(got, want)     # the halo makes the shard boundary invisible, bit for bit
    np.testing.assert_allclose(grads[0], 1.5) ; np.testing.assert_allclose(grads[1], 15.0)


# ---------------------------------------------------------------------------------------------------------------------
# round 2: DataParallel chunking, the flat gradient arena, process-group bring-up and the worker processes
# ---------------------------------------------------------------------------------------------------------------------

def test_chunk_bounds_are_tensor_chunk_cuts():
    """nn.DataParallel scatters a mini-batch with Tensor.chunk: the fine-tune step must cut it the same way."""
    from active_learning import distributed as D
    for n in (1, 3, 10, 17, 119, 120, 960, 961):
        for rep in (1, 2, 4, 8):
            want = [t.shape[0] for t in torch.arange(n).chunk(rep)]
            got = D.chunk_bounds(n, rep)
            assert [h - l for l, h in got] == want and got[0][0] == 0 and got[-1][1] == n
    assert D.chunk_bounds(0, 8) == []


def _arena_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vatl4pose-wacv2024_amd")]
    import torch.distributed as dist
    from active_learning import distributed as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    params = _arena_params()
    arena = D.GradArena(params, bucket_bytes=4 * 100)          # small buckets: several async all-reduces per step
    for step in range(2):                                      # the arena is reused step after step
        arena.begin()
        filled = 0
        for p in reversed(params):                             # the backward pass fills the arena from the top down
            arena.view(p).copy_(_rank_grad(p, rank, step))
            filled += p.numel()
            arena.done_offset(arena.total - filled)
        arena.finish()
        arena.attach()
        if rank == 0:
            q.put((step, arena.launches, [p.grad.clone().numpy() for p in params], all(p.grad.data_ptr() == arena.view(p).data_ptr() for p in params)))
    dist.barrier()
    dist.destroy_process_group()


def _arena_params():
    g = torch.Generator().manual_seed(5)
    return [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((64, 3, 7, 7), (64,), (64,), (17, 256, 1, 1), (17,), (300,), (1,))]


def _rank_grad(p, rank, step):
    g = torch.Generator().manual_seed(1000 * step + 10 * rank + p.numel() % 7)
    return torch.randn(p.shape, generator=g)


def test_grad_arena_bucketed_allreduce_equals_single_process_sum_world2():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params = _arena_params()
    for step, launches, grads, views in sorted(got, key=lambda t: t[0]):
        assert launches >= 3 and views                         # bucketed, and p.grad IS the arena slice (no copy back)
        for p, g in zip(params, grads):
            want = (_rank_grad(p, 0, step) + _rank_grad(p, 1, step)).numpy()      # what one process holding both shares computes
            np.testing.assert_array_equal(g, want)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vatl4pose-wacv2024_amd", "active_learning", "distributed.py")).read()
    assert "torch.cat" not in src                              # the gradient path has no concatenate / copy-back passes


def test_grad_arena_single_process_is_a_plain_buffer():
    from active_learning import distributed as D
    params = _arena_params()
    arena = D.GradArena(params)
    assert arena.total == sum(p.numel() for p in params) and arena.flat.numel() == arena.total
    arena.begin()
    arena.view(params[3]).fill_(2.0)
    arena.done_offset(0)
    arena.finish()
    arena.attach()
    assert arena.launches == 0 and float(params[3].grad.sum()) == 2.0 * params[3].numel()
    off = arena.offset[params[3]]
    assert float(arena.flat[off:off + params[3].numel()].sum()) == 2.0 * params[3].numel() and float(arena.flat.sum()) == 2.0 * params[3].numel()


# ---------------------------------------------------------------------------------------------------------------------
# round 3: rank-invariant collective sequences (ragged shards at world 4 / 8, ranks that reduce at different times)
# ---------------------------------------------------------------------------------------------------------------------

def _ragged_worker(rank, world, port, ns, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vatl4pose-wacv2024_amd")]
    import torch.distributed as dist
    from active_learning import distributed as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    out = []
    for n in ns:
        # a row = (item id, id of the item before it inside the window, id of the item after it): what a halo must provide
        def score(lo, hi):
            r = torch.full((hi - lo, 3), -1.0)
            for k, i in enumerate(range(lo, hi)):
                r[k, 0] = i
                r[k, 1] = i - 1 if i - 1 >= lo else -1
                r[k, 2] = i + 1 if i + 1 < hi else -1
            return r
        out.append(D.sharded_rows(n, score, 3, torch.device("cpu")).numpy())
        # DataParallel chunk walk of a mini-batch of n items over `world` replicas: every item belongs to exactly one rank
        mine = D.chunk_bounds(n, world)[rank::world]
        owned = torch.zeros(max(n, 1))
        for lo, hi in mine:
            owned[lo:hi] += 1
        D.allreduce_sum_(owned)
        assert n == 0 or bool((owned[:n] == 1).all()), (n, world, owned)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_ragged_streams_shard_the_same_at_world_4_and_8(world):
    """N % world != 0, N < world, N == 1: every rank issues the same all-gather, empty shards included."""
    ns = (5, 13, 1, 3, 8, 33)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, ns, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for n, rows in zip(ns, got):
        assert rows.shape == (n, 3)
        ids = np.arange(n)
        np.testing.assert_array_equal(rows[:, 0], ids)
        np.testing.assert_array_equal(rows[:, 1], ids - 1)                              # -1 only at the stream's start
        np.testing.assert_array_equal(rows[:, 2], np.where(ids + 1 < n, ids + 1, -1))   # and at its end


def _mixed_arena_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vatl4pose-wacv2024_amd")]
    import torch.distributed as dist
    from active_learning import distributed as D
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    params = _arena_params()
    arena = D.GradArena(params, bucket_bytes=4 * 100)
    for step in range(2):
        arena.begin()
        if rank == 2:                                          # no DataParallel chunk of a short last mini-batch: zeros, reduce at the end
            arena.flat.zero_()
        else:
            filled = 0
            for p in reversed(params):
                arena.view(p).copy_(_rank_grad(p, rank, step))
                filled += p.numel()
                if rank == 0:                                  # one chunk: buckets start while the backward pass goes on
                    arena.done_offset(arena.total - filled)
                elif rank == 3 and p is params[3]:             # reports progress once, somewhere in the middle
                    arena.done_offset(arena.total - filled)
            # rank 1 walked several chunks and reduces only at the end
        arena.finish()
        q.put((rank, step, list(arena.fired), arena.flat.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_arena_collective_sequence_is_rank_invariant_world4():
    """a rank that overlaps, a rank that reduces at the end, a rank with no chunk and a rank that reports
    progress once must issue the SAME all-reduce sequence (count, order, sizes); mismatched sequences abort on gloo and hang on
    RCCL.  The bucket cuts depend on (total, bucket_bytes) only."""
    world = 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mixed_arena_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params = _arena_params()
    total = sum(p.numel() for p in params)
    for step in range(2):
        rows = sorted([g for g in got if g[1] == step], key=lambda t: t[0])
        fired = rows[0][2]
        assert len(fired) >= 3 and fired[0][1] == total and fired[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(fired, fired[1:]))                  # contiguous, from the top down
        want = np.concatenate([sum(_rank_grad(p, r, step) for r in (0, 1, 3)).reshape(-1).numpy() for p in params])
        for r in rows:
            assert r[2] == fired, (r[0], r[2], fired)
            np.testing.assert_allclose(r[3], want, rtol=0, atol=1e-6)
        np.testing.assert_array_equal(rows[0][3], rows[2][3])                       # every rank ends with the same bits


def test_grad_arena_cuts_depend_on_sizes_only():
    from active_learning import distributed as D
    params = _arena_params()
    a = D.GradArena(params, bucket_bytes=4 * 100)
    assert a.cuts[0] == a.total and a.cuts[-1] == 0 and all(x - y == 100 for x, y in zip(a.cuts[:-2], a.cuts[1:-1]))
    assert 0 < a.cuts[-2] <= 100
    b = D.GradArena([torch.nn.Parameter(torch.zeros(200))], bucket_bytes=4 * 100)
    assert b.cuts == [200, 100, 0]
    assert D.GradArena([], bucket_bytes=400).cuts == [0]


def _run_driver(tmp_path, script, extra_env=None, timeout=180):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VATL_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join([root, os.path.join(root, "vatl4pose-wacv2024_amd")]))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    path = tmp_path / "driver.py"
    path.write_text(script)
    return subprocess.run([sys.executable, str(path)], env=env, capture_output=True, text=True, timeout=timeout)


def test_single_process_driver_gets_worker_ranks(tmp_path):
    """The unchanged driver is ONE process with opt.num_gpu = N: constructing the AL object must bring up N ranks (fresh child
    processes), every public call must run on all of them in lock-step (announced once, by the outermost call), and the workers
    must leave when the driver does."""
    import json
    script = f'''
import json, sys
from tests.dist_stub import StubAL
from active_learning import distributed as D
al = StubAL({{"token": 42}}, {{"num_gpu": 2, "dir": {str(tmp_path)!r}}})
assert al.world == 2 and D.have_workers()
al.eval_and_query()
al.outcome()
al2 = StubAL({{"token": 7}}, {{"num_gpu": 2, "dir": {str(tmp_path)!r}}})       # a second object reuses the same workers
al2.retrain_model()
print("driver done")
'''
    out = _run_driver(tmp_path, script, {"VATL_WORKER_CLASS": "tests.dist_stub:StubAL"})
    assert out.returncode == 0 and "driver done" in out.stdout, out.stderr[-3000:]
    for r in (0, 1):
        rows = [json.loads(l) for l in open(tmp_path / f"rank{r}.jsonl")]
        assert [x["call"] for x in rows] == ["eval_and_query", "retrain_model", "outcome", "retrain_model"], rows
        assert rows[0]["value"] == 3.0 and rows[0]["world"] == 2            # the all-reduce saw both ranks
        assert rows[1]["value"] == 42 and rows[3]["value"] == 7             # rank 0's objects reached the worker


def test_spawn_can_be_switched_off_and_torchrun_env_is_joined(tmp_path):
    script = f'''
from tests.dist_stub import StubAL
from active_learning import distributed as D
al = StubAL({{"token": 1}}, {{"num_gpu": 4, "dir": {str(tmp_path)!r}}})
assert al.world == 1 and not D.have_workers()
al.eval_and_query()
print("solo ok")
'''
    out = _run_driver(tmp_path, script, {"VATL_SPAWN": "0"})
    assert out.returncode == 0 and "solo ok" in out.stdout, out.stderr[-3000:]
    # torchrun-style: two ranks both run the driver script, the constructor joins the rendezvous from the environment
    import subprocess
    import sys
    script2 = f'''
from tests.dist_stub import StubAL
from active_learning import distributed as D
al = StubAL({{"token": 5}}, {{"num_gpu": 1, "dir": {str(tmp_path / "tr")!r}}})
assert al.world == 2 and not D.have_workers()
al.eval_and_query()
'''
    os.makedirs(tmp_path / "tr")
    (tmp_path / "d2.py").write_text(script2)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, VATL_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join([root, os.path.join(root, "vatl4pose-wacv2024_amd")]),
                   RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(tmp_path / "d2.py")], env=env))
    assert [p.wait(timeout=180) for p in procs] == [0, 0]
    import json
    for r in (0, 1):
        assert json.loads(open(tmp_path / "tr" / f"rank{r}.jsonl").readline())["value"] == 3.0


def test_worker_shutdown_after_a_failed_call_skips_the_collective(monkeypatch):
    """after an exception inside a mirrored call the workers are blocked in some other collective; the exit broadcast
    would be a mismatched collective that can hang until the backend's watchdog.  `mark_failed()` (set by the `_collective`
    wrapper) makes `shutdown_workers()` end the children directly; payload files are released either way."""
    from active_learning import distributed as D

    class FakeProc:
        def __init__(self):
            self.terminated = self.killed = False
            self.returncode = None

        def poll(self):
            return self.returncode

        def terminate(self):
            self.terminated = True
            self.returncode = -15

        def wait(self, timeout=None):
            return self.returncode

        def kill(self):
            self.killed = True
    calls = []
    monkeypatch.setattr(D, "broadcast_object", lambda *a, **k: calls.append(a))
    monkeypatch.setattr(D.dist, "is_initialized", lambda: False)
    path = D.dump_payload({"x": 1})
    assert os.path.exists(path)
    # clean run: one exit broadcast would be attempted (is_initialized False here -> skipped), nothing terminated
    procs = [FakeProc(), FakeProc()]
    monkeypatch.setattr(D, "_workers", procs)
    monkeypatch.setattr(D, "_failed", False)
    D.shutdown_workers()
    assert not any(p.terminated for p in procs) and not os.path.exists(path) and D._workers == []
    # failed run: children terminated, no collective
    procs = [FakeProc(), FakeProc()]
    monkeypatch.setattr(D, "_workers", procs)
    D.mark_failed()
    monkeypatch.setattr(D.dist, "is_initialized", lambda: True)
    monkeypatch.setattr(D.dist, "destroy_process_group", lambda: calls.append("destroy"))
    D.shutdown_workers()
    assert all(p.terminated for p in procs) and calls == [] and D._workers == []
    monkeypatch.setattr(D, "_failed", False)


def test_collective_wrapper_marks_the_failure():
    import importlib
    AL = importlib.import_module("active_learning.ActiveLearning")
    from active_learning import distributed as D

    class Obj:
        _depth = 0

        @AL._collective
        def boom(self):
            raise ValueError("inside a mirrored call")
    D._failed = False
    with pytest.raises(ValueError):
        Obj().boom()
    assert D._failed
    D._failed = False
