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
    g = [torch.ones(5) * (rank + 1), torch.ones(3) * 10 * (rank + 1)]
    D.allreduce_mean_(g)
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
