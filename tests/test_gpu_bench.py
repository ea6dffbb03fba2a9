"""bench.py keeps its contract: one JSON line with the metric, the roofline and (at N = 1) the CPU baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # `frac` = MFMA FLOPs the conv launches execute (library meter, padding included) / event time / peak: a roofline fraction, never above 1;
    # the direct-sum throughput figure lives in `algorithmic_frac` (the Winograd launches run 2.25x fewer multiplies, so that one may exceed 1)
    assert 0.4 < r["frac"] < 1.0 and d["value"] > 5000 and r["frac"] <= r["algorithmic_frac"] < 1.6
    assert r["winograd"]["launches"] >= 16 and r["metered_launches"] >= r["timed_calls"] == r["launches"]
    assert 0.3 < r["winograd"]["executed_frac"] < 1.0 and 0.3 < r["implicit_gemm"]["executed_frac"] < 1.0
    assert abs(r["executed_flops_per_step"] - (r["executed_direct_flops"] + r["executed_winograd_flops"])) < 1e6
    assert 0.4 < r["e2e_frac"] <= r["frac"] + 0.02 and abs(r["e2e_algorithmic_frac"] - d["value"] * 10.853e-3 / 157.3) < 2e-3   # whole step vs conv launches only
    assert d["world_size_seen"] == 1 and len(d["rank_devices"]) == 1
    ex = d["extra"]                                                     # configs[2..4] measured in the same run
    assert set(ex) == {"cfg3_simplepose_r50_finetune", "cfg4_hrnet_w32_thc_wpu", "cfg5_fastpose_r152_384_finetune", "headline_variants", "product_entry_points"}
    pe = ex["product_entry_points"]                                     # the reference-shaped entry points, host work included (verdict r04 item 5)
    assert pe["items"] == 1024 and 3000 < pe["eval_and_query_items_per_s"] <= d["value"] * 1.02 and pe["eval_and_query_single_call_items_per_s"] > 3000
    assert 10 < pe["retrain_model_ms_per_step"] < 200 and pe["retrain_model_ms_per_step"] >= ex["cfg3_simplepose_r50_finetune"]["ms_per_step"] * 0.9
    hv = ex["headline_variants"]                                        # batches of 256; faithful 3 forwards vs de-duplicated
    assert hv["thc_bit_identical"] is True and hv["batch256_frames_per_s"] > 5000
    assert 2.0 < hv["dedup_frames_per_s"] / hv["faithful_3fwd_frames_per_s"] < 3.5
    for k in ("cfg3_simplepose_r50_finetune", "cfg5_fastpose_r152_384_finetune"):
        e = ex[k]
        assert e["ms_per_step"] > 0 and 0.2 < e["executed_frac"] < 1.0 and e["executed_frac"] <= e["algorithmic_frac"] and e["allreduce_alone_ms"] is None and e["allreduce_buckets"] == 0
        assert e["overlap_hidden_frac"] is None and e["step_without_allreduce_ms"] is None
        assert abs(e["crops_per_s"] - e["batch_per_gpu"] * 1000.0 / e["ms_per_step"]) / e["crops_per_s"] < 0.01
    assert 130e6 < ex["cfg3_simplepose_r50_finetune"]["grad_bytes"] < 140e6 and 295e6 < ex["cfg5_fastpose_r152_384_finetune"]["grad_bytes"] < 305e6
    assert ex["cfg4_hrnet_w32_thc_wpu"]["frames_per_s"] > 3000 and ex["cfg4_hrnet_w32_thc_wpu"]["halo_frames"] == 0
    assert 0.2 < ex["cfg4_hrnet_w32_thc_wpu"]["executed_frac"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "frames/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # round 6: the thread count is calibrated at the measured batch within the CPUs the container is GRANTED (cgroup quota), the all-granted-cores figure beside it
    assert 1 <= c["cores"] <= c["cores_available"] <= c["cores_visible"] and str(c["cores"]) in c["calibration_forward_s"]
    assert c["all_cores"]["cores"] == c["cores_available"] and 0 < c["all_cores"]["value"] <= c["value"] * 1.25
    # round 6: the shader clock sampled over the timed region and the fractions against the peak at THAT clock (the hwmon files are readable on the GPU boxes)
    assert r["sclk_samples"] >= 3 and 500 < r["sclk_mhz_min"] <= r["sclk_mhz_mean"] <= r["sclk_mhz_max"] <= 2600 and r["nominal_sclk_mhz"] == 2400.0
    assert abs(r["peak_at_sclk"] - 157.3 * r["sclk_mhz_mean"] / 2400.0) < 0.02 and abs(r["frac_at_sclk"] - r["achieved"] / r["peak_at_sclk"]) < 1e-3
    assert r["frac"] <= r["frac_at_sclk"] < 1.0
    assert len(pe["eval_and_query_round_ms"]) == pe["eval_and_query_rounds"] and pe["gc_freeze"] is False and pe["eval_and_query_p90_round_ms"] >= pe["eval_and_query_median_round_ms"]
    # whole-job rate is consistent with the step time
    assert abs(d["value"] - 1024 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.01


def test_bench_two_ranks_launch_contract():
    """The driver's N > 1 launch line (torch.distributed.run, one rank per GPU) on a 1-GPU box: two ranks share the device and
    talk over gloo (VATL_DIST_BACKEND; RCCL refuses two ranks on one device).  Checks the rendezvous, the per-step result
    gather, the barrier / max-over-ranks timing and that only rank 0 prints the line, with whole-job frames = 2 x 1024."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, VATL_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d and d["roofline"]["bound"] == "mfma"
    assert abs(d["value"] - 2 * 1024 * 1000.0 / d["ms_per_step"]) / d["value"] < 0.01
    assert d["config"]["parallelism"] == "frame-sharded x2"


def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` run bare (how the driver calls it): the script itself must start two ranks as fresh child
    processes and relay rank 0's line; the data-parallel fine-tune configurations run with the gradient arena all-reduced
    across the ranks (gloo here: two ranks share the box's one GPU, RCCL needs a device per rank)."""
    env = dict(os.environ, VATL_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], capture_output=True, text=True,
                         timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "frame-sharded x2" and "cpu_baseline" not in d
    assert d["world_size_seen"] == 2 and [r["rank"] for r in d["rank_devices"]] == [0, 1] and all(r["backend"] == "gloo" for r in d["rank_devices"])
    ex = d["extra"]
    for k in ("cfg3_simplepose_r50_finetune", "cfg5_fastpose_r152_384_finetune"):
        assert ex[k]["allreduce_buckets"] >= 2 and ex[k]["allreduce_alone_ms"] > 0      # bucketed, overlapped with the backward pass
        assert ex[k]["allreduce_buckets"] == -(-ex[k]["grad_bytes"] // ex[k]["allreduce_bucket_bytes"])   # the fixed, rank-invariant cut list
        assert 0.0 <= ex[k]["overlap_hidden_frac"] <= 1.0 and ex[k]["step_without_allreduce_ms"] > 0
        assert abs(ex[k]["crops_per_s"] - 2 * ex[k]["batch_per_gpu"] * 1000.0 / ex[k]["ms_per_step"]) / ex[k]["crops_per_s"] < 0.01
    assert ex["cfg4_hrnet_w32_thc_wpu"]["halo_frames"] == 1                              # rank 0 of 2: one interior edge
