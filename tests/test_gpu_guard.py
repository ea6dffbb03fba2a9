"""The parameter guard of the inference plans (round 6): `vatl_checksum_multi` against its numpy restatement, and the hole it closes —
a parameter / buffer written in place through `.data` bumps no torch version counter, so the plan's key cannot see it
(reference idiom: alphapose/models/layers/dcn/deform_conv.py:232,255; loaders beside ActiveLearning.py:217)."""
import numpy as np
import pytest
import torch

from oracle import checksum as oc
from oracle import synth
from tests.gpu_util import dev, to_dev
from tests.test_gpu_conv import _build_simplepose

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vh():
    import vatl_hip
    vatl_hip.lib()
    return vatl_hip


def test_checksum_kernel_matches_the_published_formula(vh):
    """Sizes around the block (16 384 words) and vector (4 words) boundaries, unaligned starts (views into a larger buffer), int64 and
    float tensors, one launch over all of them; bit-identical to oracle/checksum.py, and independent of the order of the rows."""
    r = np.random.RandomState(3)
    bw = int(vh.lib().vatl_checksum_block_words())
    assert bw == 16384
    sizes = [1, 2, 3, 4, 5, 63, 64, 257, bw - 1, bw, bw + 1, 2 * bw + 7, 5 * bw + 3, 1_000_003]
    big = to_dev(r.standard_normal(sum(sizes) + 3 * len(sizes) + 8).astype(np.float32))
    ts, off = [], 1                                                   # start one word in: the first tensors are NOT 16-byte aligned
    for k, n in enumerate(sizes):
        ts.append(big[off:off + n]); off += n + (k % 3)
    ts.append(torch.arange(-50, 50, device=dev(), dtype=torch.int64))
    ts.append(torch.zeros(7, device=dev(), dtype=torch.float32))
    want = np.array([oc.checksum_tensor(t) for t in ts], np.uint64)
    got = vh.ChecksumTable(ts).launch().cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want)
    got_rev = vh.ChecksumTable(ts[::-1]).launch().cpu().numpy().view(np.uint64)
    assert np.array_equal(got_rev, want[::-1])
    ts[5].view(torch.int32)[11] ^= 1                                  # one bit
    again = vh.ChecksumTable(ts).launch().cpu().numpy().view(np.uint64)
    assert (again != want).sum() == 1 and again[5] != want[5]
    with pytest.raises(vh.VatlError):
        vh.ChecksumTable([torch.zeros(3, dtype=torch.uint8, device=dev())])


def test_data_writes_are_noticed_and_the_next_forward_equals_a_fresh_model(vh, monkeypatch):
    from alphapose.models import hip_engine
    assert hip_engine.PARAM_GUARD and 0 < hip_engine.GUARD_MIN_INTERVAL_S <= 0.05
    monkeypatch.setattr(hip_engine, "GUARD_MIN_INTERVAL_S", 0.0)          # every call guarded: the sequence below is call-exact (the interval has its own test)
    m = _build_simplepose()
    x = to_dev(synth.crops(20, seed=4))
    out = torch.empty((20, 17, 64, 48), device=dev())
    with torch.no_grad():
        hip_engine.forward_into(m, x, out)
        hip_engine.forward_into(m, x, out)                            # guard launches, nothing to report
        hip_engine.verify(m)
        before = out.clone()
        # the torch-1.12-era idiom: overwrite through .data — no version counter moves
        w_new = m.final_layer.weight.data * 0.5
        key = hip_engine._version_key(m, dev())
        m.final_layer.weight.data.copy_(w_new)
        m.preact.bn1.running_mean.data.zero_()
        assert hip_engine._version_key(m, dev()) == key
        hip_engine.forward_into(m, x, out)                            # runs the OLD packed values (asserted: that is the failure being guarded)
        torch.cuda.synchronize()
        assert torch.equal(out, before)
        with pytest.raises(hip_engine.StalePlanError) as ei:
            hip_engine.forward_into(m, x, out)                        # ... and the next call into the engine says so, by name
        assert "final_layer.weight" in str(ei.value) and "preact.bn1.running_mean" in str(ei.value)
        hip_engine.forward_into(m, x, out)                            # the plan was dropped: this call packs the current values
        torch.cuda.synchronize()
    fresh = _build_simplepose()
    with torch.no_grad():
        fresh.final_layer.weight.copy_(w_new)
        fresh.preact.bn1.running_mean.zero_()
        ref = torch.empty_like(out)
        hip_engine.forward_into(fresh, x, ref)
    assert torch.equal(out, ref) and not torch.equal(out, before)
    # invalidate(): the documented call after a .data write — no error at all, correct values at once
    with torch.no_grad():
        m.final_layer.bias.data.add_(1.0)
        hip_engine.invalidate(m)
        hip_engine.forward_into(m, x, out)
        fresh.final_layer.bias.add_(1.0)
        hip_engine.forward_into(fresh, x, ref)
        hip_engine.verify(m)
    assert torch.equal(out, ref)
    # verify(): the synchronous form
    with torch.no_grad():
        m.deconv_layers[1].running_var.data.mul_(1.5)
        with pytest.raises(hip_engine.StalePlanError, match="deconv_layers.1.running_var"):
            hip_engine.verify(m)
    # module calls (`model(x)`, <= 16 crops) go through the same guard
    with torch.no_grad():
        y0 = m(x[:4]).clone()
        m.final_layer.weight.data.mul_(2.0)
        m(x[:4])
        with pytest.raises(hip_engine.StalePlanError):
            torch.cuda.synchronize(); m(x[:4])
        y1 = m(x[:4])
    assert not torch.equal(y0, y1)
    # tracked writes (version bump) never raise: the key changes and the plan is rebuilt
    with torch.no_grad():
        m.final_layer.weight.mul_(0.5)
        y2 = m(x[:4]); m(x[:4]); hip_engine.verify(m)
    assert torch.allclose(y2, y0, rtol=1e-5, atol=1e-5)


def test_guard_interval_bounds_how_long_a_write_can_go_unnoticed(vh, monkeypatch):
    """With GUARD_MIN_INTERVAL_S = t a plan's checksums are launched at most once per t (latency loops of ~1 ms calls pay the guard's ~60 us of stream time on one call in
    twenty): a write is noticed by the first call that is more than t after the last guarded one (+ one call for the read-back), and by `verify` at once."""
    import time
    from alphapose.models import hip_engine
    monkeypatch.setattr(hip_engine, "GUARD_MIN_INTERVAL_S", 0.25)
    m = _build_simplepose()
    x = to_dev(synth.crops(4, seed=4))
    with torch.no_grad():
        m(x); m(x)                                                        # plan built (its own reference checksums), first guarded call
        torch.cuda.synchronize()
        g = m.__dict__["_vatl_plan"][2]
        n0 = len(g.pending)
        for _ in range(5):
            m(x)
        assert len(g.pending) <= n0 + 1                                   # calls inside the interval launch no checksums (arrived read-backs are consumed)
        m.final_layer.weight.data.mul_(3.0)
        m(x); m(x)                                                        # still inside the interval: nobody looked
        with pytest.raises(hip_engine.StalePlanError, match="final_layer.weight"):
            hip_engine.verify(m)                                          # ... but verify() looks at once
        m(x); m(x)                                                        # new plan, new reference
        m.final_layer.weight.data.mul_(0.5)
        time.sleep(0.3)
        m(x)                                                              # past the interval: guarded (stale values, as designed) ...
        torch.cuda.synchronize()
        with pytest.raises(hip_engine.StalePlanError):
            m(x)                                                          # ... and reported by the next call


def test_guard_cost_is_small_next_to_a_stream_call(vh):
    """One checksum launch over SimplePose-R50's 136 MB next to a 256-crop forward: < 2 % (measured ~30 us vs ~16 ms)."""
    from alphapose.models import hip_engine
    m = _build_simplepose()
    x = to_dev(synth.crops(64, seed=4)).repeat(4, 1, 1, 1)
    out = torch.empty((256, 17, 64, 48), device=dev())

    def timed(k=5):
        with torch.no_grad():
            hip_engine.forward_into(m, x, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(k):
                hip_engine.forward_into(m, x, out)
            e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k
    on = min(timed() for _ in range(3))
    hip_engine.PARAM_GUARD = False
    try:
        hip_engine.invalidate(m)
        off = min(timed() for _ in range(3))
    finally:
        hip_engine.PARAM_GUARD = True
        hip_engine.invalidate(m)
    from tests.gpu_util import record
    record("param_guard_cost_256_crops", ms_on=on, ms_off=off)
    assert on < off * 1.02 + 0.05, (on, off)
