"""bench.py launch logic that needs no GPU: a contradictory rank environment and an impossible --gpus fail loudly."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
    # RCCL needs one device per rank: asking for more ranks than GPUs fails loudly instead of printing n_gpus: 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "VATL_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
