#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_round.sh pmc).

    python tools/pmc_summary.py gpurun_out profiles/r01_pmc_summary.json

Units and corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced read stream, so it is
doubled (calibrated here on nchw_to_nhwc_kernel, whose read volume is known exactly:
N*3*H*W*4 bytes); WRITE_SIZE matched known store volumes (stem output) without correction.
Each pass ran `bench.py --steps 1 --warmup 0` = 2 forwards of the 1024-frame video
(one timed step + the roofline pass), so per-step values are totals / 2.
"""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch = load(f"{root}/pmc_FETCH_SIZE/r1_counter_collection.csv")
    write = load(f"{root}/pmc_WRITE_SIZE/r1_counter_collection.csv")
    steps = 2
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("vatl::"):
            continue
        n = fetch[k][0] or write[k][0]
        kernels[k] = {"launches_per_step": n // steps,
                      "read_bytes_per_step": int(fetch[k][1] * 1024 * 2 / steps),      # x2: gfx950 FETCH_SIZE correction
                      "write_bytes_per_step": int(write[k][1] * 1024 / steps)}
    conv = [v for k, v in kernels.items() if "conv_igemm" in k or "gemm1x1_persistent" in k or "winograd_kernel" in k or "winograd_persist_kernel" in k or "winograd_f4_kernel" in k or "stem_pool_kernel" in k or "bottleneck_chain_kernel" in k or "conv1x1_rows_kernel" in k or "conv1x1_rows256_kernel" in k or "winograd_c32_kernel" in k]   # all conv launches
    cal, cal_name = kernels.get("vatl::nchw_to_nhwc_kernel"), "nchw_to_nhwc_kernel"
    if cal is None:                     # round 4: the fused stem reads the NCHW crops itself (same known read volume; it writes the pooled 64-channel rows: N x 64 x 48 x 64 x 4 bytes)
        for k, v in kernels.items():
            if "stem_pool_kernel<3, 7>" in k:
                cal, cal_name = v, "stem_pool_kernel<3, 7>"
    summary = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 0, 1024 frames/step",
        "fetch_correction": 2.0,
        "calibration": {"kernel": cal_name, "expected_read_bytes": 1024 * 3 * 256 * 192 * 4,
                        "measured_read_bytes_corrected": cal["read_bytes_per_step"] if cal else None,
                        "expected_write_bytes": 1024 * 4 * 256 * 192 * 4, "measured_write_bytes": cal["write_bytes_per_step"] if cal else None},
        "conv_igemm": {"launches_per_step": sum(v["launches_per_step"] for v in conv),
                       "read_bytes_per_step": sum(v["read_bytes_per_step"] for v in conv),
                       "write_bytes_per_step": sum(v["write_bytes_per_step"] for v in conv)},
        "kernels": kernels,
    }
    summary["conv_igemm"]["hbm_bytes_per_step"] = summary["conv_igemm"]["read_bytes_per_step"] + summary["conv_igemm"]["write_bytes_per_step"]
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps(summary["conv_igemm"]), json.dumps(summary["calibration"]))


if __name__ == "__main__":
    main()
