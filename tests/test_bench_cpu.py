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


def test_clock_sampler_and_granted_cpus_read_what_the_box_offers(tmp_path, monkeypatch):
    """bench.ClockSampler on a fake amdgpu hwmon tree (the build container has no /sys/class/drm): the card is found by PCI address, the mean is taken over the
    samples inside the timed region with its first tenth skipped, and a box without the files reports None with the reason instead of failing;
    bench.granted_cpus cuts the visible CPUs down to the cgroup quota."""
    import builtins
    import time
    sys.path.insert(0, ROOT)
    import bench
    # granted_cpus: quota "1600000 100000" -> 16 of whatever is visible
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            import io
            return io.StringIO("1600000 100000\n")
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    assert bench.granted_cpus() == (16, 256, 16.0)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(8)), raising=False)
    assert bench.granted_cpus() == (8, 8, 16.0)
    monkeypatch.setattr(builtins, "open", real_open)
    # ClockSampler without sysfs: nothing to read, a reason instead of an exception
    c = bench.ClockSampler(0)
    if c.hwmon is None:
        with c:
            time.sleep(0.02)
        s = c.summary()
        assert s["sclk_mhz_mean"] is None and s["sclk_samples"] == 0 and s["sclk_source"]
    # ... and with a readable hwmon directory: samples, statistics, ramp-up skipped
    hw = tmp_path / "hwmon0"
    hw.mkdir()
    (hw / "freq1_input").write_text("2100000000\n")
    (hw / "power1_input").write_text("1250000000\n")
    c = bench.ClockSampler.__new__(bench.ClockSampler)
    c.period, c.samples, c._stop, c._thread, c.hwmon, c.why, c.pci = 0.002, [], __import__("threading").Event(), None, str(hw), None, "0000:f1:00.0"
    with c:
        time.sleep(0.05)
        (hw / "freq1_input").write_text("2300000000\n")
        time.sleep(0.05)
    s = c.summary()
    assert s["sclk_samples"] >= 10 and 2100.0 <= s["sclk_mhz_min"] <= s["sclk_mhz_mean"] <= s["sclk_mhz_max"] == 2300.0 and s["power_w_mean"] == 1250.0
    late = c.summary(t0=c.t0 + 0.06, t1=c.t1)
    assert late["sclk_mhz_min"] == 2300.0
