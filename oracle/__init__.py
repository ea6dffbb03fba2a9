"""CPU oracle for the VATL4Pose pose-inference + uncertainty hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.
The product path (``vatl4pose-wacv2024_amd/``) never imports this package and
fails loudly when the HIP library is missing.

Pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md §4, §8c).  The restatement here is pinned against outputs of the
reference itself, generated in the build container by ``tools/make_golden.py``
(which imports ``/root/reference`` with three import shims) and committed as
``tests/golden/*.npz``.  ``tests/test_oracle_golden.py`` re-checks every
function below against those fixtures on CPU.

Modules
-------
``scorers``  numpy restatement of decode / TPC / THC / local-peak / hybrid
             feature / WPU / masked MSE / AdamW / OKS / heat-map accuracy.
``nets``     plain ``torch.nn`` (CPU, fp32) restatement of the SimplePose,
             FastPose and HRNet graphs and of the whole-body auto-encoder.
``synth``    seeded, platform-stable synthetic weights / inputs shared by the
             fixtures, the parity tests and ``bench.py``.
"""
