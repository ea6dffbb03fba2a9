#!/usr/bin/env python3
"""GPU busy / idle accounting of a rocprofv3 kernel trace (fine-tune step analysis).

    python tools/gap_report.py <..._kernel_trace.csv> [steps]

Takes the dispatches between the first and the last `adamw_multi_kernel` launch of the trace (whole optimizer steps),
and prints per step: wall time, union of kernel intervals (busy), idle time, the number of launches, the idle time
split by gap size, and kernel time by category (conv fwd+dgrad / wgrad / BatchNorm + element-wise / micro-launches
shorter than 15 us / optimizer / other).
"""
import csv
import json
import sys


def category(name: str) -> str:
    if "conv_wgrad" in name:
        return "wgrad"
    if "conv_igemm" in name or "gemm1x1" in name or "conv3x3_halo" in name or "streamk" in name or "winograd_kernel" in name or "winograd_persist_kernel" in name or "stem_pool_kernel" in name or "bottleneck_chain_kernel" in name or "conv1x1_rows_kernel" in name or "conv1x1_rows256_kernel" in name or "winograd_c32_kernel" in name or "winograd_f4_kernel" in name:
        return "conv_fwd_dgrad"
    if "adamw" in name or "adam_" in name or "sgd" in name:
        return "optimizer"
    for k in ("bn_", "scale_bias", "col_reduce", "maxpool", "relu", "nchw", "se_", "pixel", "fuse_up", "col_sum", "hw_reduce", "masked_mse", "mse_finish"):
        if k in name:
            return "bn_elementwise"
    if "wgrad_reduce" in name or "pack_" in name or "splitk_reduce" in name or "wino_pack" in name:
        return "reduce_pack"
    return "other"


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "adamw_multi_kernel" in r["Kernel_Name"] or "adam_step" in r["Kernel_Name"]]
    if len(opt) < 2:
        sys.exit("need at least two optimizer steps in the trace")
    # optimizer launches come in groups (one per parameter group, a pointer-table copy between them): a step ends with the last
    # launch of its group — the next optimizer launch is hundreds of dispatches away
    ends = [i for k, i in enumerate(opt) if k + 1 == len(opt) or opt[k + 1] - i > 50]
    # the first step of a process also initialises the optimizer state between its optimizer launches (it can look like two
    # steps): start counting after the SECOND step boundary
    if len(ends) < 3:
        sys.exit("need at least three optimizer steps in the trace")
    lo, hi = ends[1] + 1, ends[-1] + 1
    steps = len(ends) - 2
    win = rows[lo:hi]
    t0, t1 = int(win[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in win)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win)
    busy, gaps, cur_s, cur_e = 0, [], iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    cat, micro_n, micro_t = {}, 0, 0
    for r in win:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = category(r["Kernel_Name"])
        cat[c] = cat.get(c, 0) + d
        if d < 15000:
            micro_n += 1
            micro_t += d
    ms = lambda ns: round(ns / steps / 1e6, 3)
    out = {"steps": steps, "launches_per_step": round(len(win) / steps, 1), "wall_ms": ms(t1 - t0), "busy_ms": ms(busy), "idle_ms": ms(t1 - t0 - busy),
           "gaps_per_step": round(len(gaps) / steps, 1),
           "idle_ms_in_gaps_under_5us": ms(sum(g for g in gaps if g < 5000)), "idle_ms_in_gaps_5_20us": ms(sum(g for g in gaps if 5000 <= g < 20000)),
           "idle_ms_in_gaps_over_20us": ms(sum(g for g in gaps if g >= 20000)),
           "kernel_ms_by_category": {k: ms(v) for k, v in sorted(cat.items(), key=lambda kv: -kv[1])},
           "kernel_ms_sum": ms(sum(cat.values())), "launches_under_15us_per_step": round(micro_n / steps, 1), "their_ms": ms(micro_t)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
