#!/usr/bin/env python3
"""Register / LDS budget of every kernel of a csrc file, from hipcc's own remarks (no GPU needed):

    python tools/kernel_resources.py conv_igemm [conv_wgrad ...]
prints arch VGPRs, accumulator VGPRs, spills, waves per SIMD and LDS bytes per block.  A kernel whose waves leave part of the
SIMD's 512 registers free lets a small element-wise wave of another stream share the CU (profiles/r05_notes.md)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vatl4pose-wacv2024_amd", "csrc")
OCC, LDS = r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for name in sys.argv[1:]:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c",
                        os.path.join(CSRC, name + ".hip"), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    print("==", name)
    for blk in r.stderr.split("Function Name: ")[1:]:
        mangled = blk.split("\n")[0].strip()
        def g(k):
            m = re.search(k + r": (\d+)", blk)
            return m.group(1) if m else "?"
        dn = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()[:120]
        print("V%4s A%4s spill%4s occ%2s LDS%7s  %s" % (g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(OCC), g(LDS), dn))
