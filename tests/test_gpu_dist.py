"""The N > 1 path on real kernels: two ranks (gloo, both on cuda:0 — the box has one GPU) run the sharded
evaluation and one data-parallel fine-tune step; results must equal the single-process run."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_al(filter_="None", rep="None"):
    from active_learning import ActiveLearning
    from tests.test_gpu_al import _cfg
    opt = types.SimpleNamespace(uncertainty="THC+WPU", representativeness=rep, filter=filter_, strategy="THC+WPU", video_id="syn", get_prenext=True,
                                from_scratch=True, continual=True, num_gpu=1, onebyone=False, retrain_thresh=1, THCvsWPU="const", fixed_lambda=False)
    torch.manual_seed(0); np.random.seed(0)
    return ActiveLearning(_cfg(), opt)


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "vatl4pose-wacv2024_amd")]
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    al = _make_al("Coreset", "Influence")
    al.eval_and_query()
    out = {"kp": al.keypoints.copy(), "oks": al.oks.copy(), "unc": al.uncertainty_dict["Round0"], "query": al.query_list_list["Round0"],
           "influence": al.influence_dict["Round0"]}
    # one data-parallel fine-tune pass: parameters must stay identical across ranks (same averaged gradients)
    al.outcome()
    flat = torch.cat([p.detach().reshape(-1) for p in al.model.parameters()]).double()
    out["param_sum"] = float(flat.sum()); out["param_abs"] = float(flat.abs().sum())
    out["bn_mean"] = al.model.preact.bn1.running_mean.cpu().numpy().copy()
    out["ae_sum"] = float(torch.cat([p.detach().reshape(-1) for p in al.AE.parameters()]).double().sum())
    out["train_loss"] = al.last_train_loss
    al.eval_and_query()                                               # second round on the fine-tuned replicas
    out["kp2"] = al.keypoints.copy(); out["unc2"] = al.uncertainty_dict["Round1"]
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process():
    single = _make_al("Coreset", "Influence")
    single.eval_and_query()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in (0, 1):
        # sharded evaluation (12 + 12 items, one-item halo) == whole stream: bit-identical key-points, scores, queries
        assert np.array_equal(res[r]["kp"], single.keypoints) and np.array_equal(res[r]["oks"], single.oks)
        assert res[r]["unc"] == single.uncertainty_dict["Round0"]
        assert res[r]["query"] == single.query_list_list["Round0"]
        assert res[r]["influence"] == single.influence_dict["Round0"]
    # data-parallel step: both ranks hold the same parameters afterwards; BN statistics are rank 0's
    assert res[0]["param_sum"] == res[1]["param_sum"] and res[0]["param_abs"] == res[1]["param_abs"]
    assert np.array_equal(res[0]["bn_mean"], res[1]["bn_mean"])
    assert res[0]["ae_sum"] == res[1]["ae_sum"]                        # the re-trained auto-encoder is rank 0's everywhere
    assert np.isfinite(res[0]["train_loss"]) and np.isfinite(res[1]["train_loss"])
    assert np.array_equal(res[0]["kp2"], res[1]["kp2"]) and res[0]["unc2"] == res[1]["unc2"]


# ---------------------------------------------------------------------------------------------------------------------
# round 2: the package brings the ranks up itself (single-process driver, opt.num_gpu = 2), random selections stay
# rank-consistent, and a 2-process run equals the 1-process run that walks the two DataParallel replicas in turn
# ---------------------------------------------------------------------------------------------------------------------

_DRIVER = '''
import sys, types, json
import numpy as np, torch
from tests.test_gpu_al import _cfg
from active_learning import ActiveLearning
from active_learning import distributed as D
out, unc, flt, rep = sys.argv[1:5]
opt = types.SimpleNamespace(uncertainty=unc, representativeness=rep, filter=flt, strategy=unc, video_id="syn", get_prenext=True, from_scratch=True,
                            continual=True, num_gpu=2, onebyone=False, retrain_thresh=1, THCvsWPU="const", fixed_lambda=False)
torch.manual_seed(0); np.random.seed(0)
al = ActiveLearning(_cfg(), opt)
world = D.world_rank()[0]
al.eval_and_query()
q0 = list(al.query_list_list["Round0"])
kp0 = al.keypoints.copy()
assert al.outcome() is None                                  # one data-parallel fine-tune (2 DataParallel replicas of 4 crops each per step)
al.eval_and_query()
flat = torch.cat([p.detach().reshape(-1) for p in al.model.parameters()])
bufs = torch.cat([b.detach().reshape(-1).float() for b in al.model.buffers()])
np.savez(out, world=world, q0=q0, q1=list(al.query_list_list["Round1"]), kp0=kp0, kp1=al.keypoints, params=flat.cpu().numpy(), bufs=bufs.cpu().numpy(),
         loss=al.last_train_loss, labeled=al.labeled_id)
print("driver ok", world, D.have_workers())
'''


def _drive(tmp_path, name, unc, flt, rep, env_extra):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VATL_DIST_BACKEND="gloo", PYTHONPATH=os.pathsep.join([root, os.path.join(root, "vatl4pose-wacv2024_amd")]))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    script = tmp_path / "driver.py"
    script.write_text(_DRIVER)
    out = tmp_path / f"{name}.npz"
    r = subprocess.run([sys.executable, str(script), str(out), unc, flt, rep], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), r.stdout


@pytest.mark.parametrize("unc,flt,rep", [("THC+WPU", "Coreset", "Influence"), ("WPU_hybrid", "Random", "Random")])
def test_single_process_driver_spawns_ranks_and_matches_the_replica_walk(tmp_path, unc, flt, rep):
    """opt.num_gpu = 2 from ONE driver process (the reference's DataParallel case): the constructor starts the second rank; the
    run must equal — bit for bit — the same driver with VATL_SPAWN=0, where one process walks the two DataParallel replicas of
    every mini-batch in turn.  The Random filter / Random representativeness case draws from rank-local numpy RNGs: the
    worker's stream differs from rank 0's (it never sees the driver's seeding), so this only passes when rank 0's
    selection is the one every rank adopts."""
    two, log2 = _drive(tmp_path, "two", unc, flt, rep, {})
    one, log1 = _drive(tmp_path, "one", unc, flt, rep, {"VATL_SPAWN": "0"})
    assert "driver ok 2 True" in log2 and "driver ok 1 False" in log1
    assert int(two["world"]) == 2 and int(one["world"]) == 1
    for k in ("q0", "q1", "kp0", "kp1", "params", "bufs", "labeled"):
        assert np.array_equal(two[k], one[k]), k
    assert abs(float(two["loss"]) - float(one["loss"])) < 1e-6 * max(1.0, abs(float(one["loss"])))


def test_arena_buckets_start_only_after_the_side_stream_joined(monkeypatch):
    """Weight gradients are written on a second HIP stream; a bucket's all-reduce is issued from the main stream.  Every bucket
    must start (a) only after `_side.join()` — no side-stream launch outstanding — and (b) in stream order after everything
    that writes into it: a copy of the bucket taken AT the moment the collective is issued (same stream) must already hold the
    bucket's final values.  The collective itself is replaced by that copy (one process; RCCL is not needed for the ordering)."""
    import torch.distributed as dist
    import vatl_hip as vh
    from active_learning import distributed as D
    from alphapose.models import builder, hip_train
    from alphapose.utils.config import edict
    cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
    preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    m = builder.build_sppe(cfg, preset_cfg=preset).to(dev).train()
    tr = hip_train.trainer_for(m)
    arena = D.GradArena([p for p in m.parameters() if p.requires_grad], device=dev, bucket_bytes=8 << 20)
    assert hip_train._side.enabled
    log, snaps = [], []

    class _Done:
        def wait(self):
            return True

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        log.append(("fire", hip_train._side.used))
        snaps.append((t.data_ptr(), t.numel(), t.clone()))          # on the issuing stream, like the collective's read of the bucket
        return _Done()
    real_join = hip_train._side.join

    def join():
        log.append(("join", hip_train._side.used))
        real_join()
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(hip_train._side, "join", join)
    g = torch.Generator(device=dev); g.manual_seed(5)
    x = torch.rand((24, 3, 256, 192), device=dev, generator=g) - 0.45
    labels = torch.rand((24, 17, 64, 48), device=dev, generator=g) * 0.1
    masks = torch.ones((24, 17, 1, 1), device=dev)
    for _ in range(3):
        log.clear(); snaps.clear()
        with torch.no_grad():
            out = tr.forward(x)
            _, dout = vh.masked_mse_fwd_bwd(out, labels, masks)
            arena.begin()
            arena._live = True                                      # one process standing in for a rank of a larger group
            tr.backward(dout, arena=arena, overlap=True)
            early = len(snaps)
            arena.finish()
        torch.cuda.synchronize()
        assert arena.fired == [(arena.cuts[i + 1], arena.cuts[i]) for i in range(len(arena.cuts) - 1)]
        assert early >= 8, early                                    # most of the 17 buckets start during the backward pass
        fires = [e for e in log if e[0] == "fire"]
        assert len(fires) == len(arena.cuts) - 1 and not any(used for _, used in fires)      # (a) nothing outstanding on the side stream
        first_fire = next(i for i, e in enumerate(log) if e[0] == "fire")
        assert any(e == ("join", True) for e in log[:first_fire])   # the side stream really was in use and was joined first
        base = arena.flat.data_ptr()
        for ptr, n, snap in snaps:                                  # (b) the bucket was final when its collective was issued
            lo = (ptr - base) // 4
            assert torch.equal(snap, arena.flat[lo:lo + n]), (lo, n)


_RCCL_ONE_RANK = r"""
import os, sys
import torch, torch.distributed as dist
root = sys.argv[1]
sys.path[:0] = [root, os.path.join(root, "vatl4pose-wacv2024_amd")]
import vatl_hip as vh
from active_learning import distributed as D
from alphapose.models import builder, hip_train
from alphapose.utils.config import edict
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
cfg = edict({"TYPE": "SimplePose", "PRETRAINED": "", "TRY_LOAD": "", "NUM_DECONV_FILTERS": [256, 256, 256], "NUM_LAYERS": 50})
preset = edict({"TYPE": "simple", "SIGMA": 2, "NUM_JOINTS": 17, "IMAGE_SIZE": [256, 192], "HEATMAP_SIZE": [64, 48]})
torch.manual_seed(3)
m = builder.build_sppe(cfg, preset_cfg=preset).to(dev).train()
tr = hip_train.trainer_for(m)
params = [p for p in m.parameters() if p.requires_grad]
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.rand((24, 3, 256, 192), device=dev, generator=g) - 0.45
labels = torch.rand((24, 17, 64, 48), device=dev, generator=g) * 0.1
masks = torch.ones((24, 17, 1, 1), device=dev)
out = []
for live in (False, True, True):
    arena = D.GradArena(params, device=dev, bucket_bytes=8 << 20)
    with torch.no_grad():
        o = tr.forward(x)
        _, dout = vh.masked_mse_fwd_bwd(o, labels, masks)
        arena.begin()
        arena._live = live                      # a one-rank RCCL group: the collectives are real launches on RCCL's stream, the sum is the identity
        tr.backward(dout, arena=arena, overlap=True)
        early = arena.launches
        arena.finish()
    torch.cuda.synchronize()
    out.append((arena.flat.clone(), early, arena.launches))
assert out[0][2] == 0 and out[1][2] == 17 and out[1][1] >= 8, [o[1:] for o in out]
assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[1][0], out[2][0])
dist.destroy_process_group()
print("rccl one rank ok", out[1][1], out[1][2])
"""


def test_rccl_one_rank_group_runs_the_bucketed_allreduce_beside_the_backward_pass(tmp_path):
    """The only RCCL the builder's one-GPU box can run: a world-size-1 "nccl" group.  The gradient arena's buckets are issued as
    real asynchronous RCCL all-reduces (its own stream, work objects, in-place on arena slices) while the backward pass keeps
    launching on the main stream and the weight gradients on the side stream; with one rank the sum is the identity, so the
    arena must end with exactly the bits of a pass without collectives — which it only does if every bucket was final when its
    collective read it and nothing overwrote a bucket afterwards."""
    import subprocess
    import sys
    script = tmp_path / "rccl_one_rank.py"
    script.write_text(_RCCL_ONE_RANK)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script), root, str(_free_port())], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "rccl one rank ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
