#!/usr/bin/env python3
"""Long HIP API calls of a `rocprofv3 --hip-trace --output-format csv` run, with the calls around them.

    python tools/hip_stalls.py <..._hip_api_trace.csv> [--over-ms 5] [--context 3]
A host thread that "stalls" while enqueueing is inside one of these (a synchronous copy, an allocation, a launch that waits for queue space)."""
import csv
import sys

path = sys.argv[1]
over = float(sys.argv[sys.argv.index("--over-ms") + 1]) * 1e6 if "--over-ms" in sys.argv else 5e6
ctx = int(sys.argv[sys.argv.index("--context") + 1]) if "--context" in sys.argv else 3
rows = list(csv.DictReader(open(path)))
name = "Function" if "Function" in rows[0] else "Name"
rows = [r for r in rows if not r[name].startswith("__hip")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
counts = {}
for i, r in enumerate(rows):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d >= over and r[name] not in ("hipGetDeviceCount",):
        counts[r[name]] = counts.get(r[name], 0) + 1
        print("-- %.1f ms in %s at t = %.1f ms (thread %s)" % (d / 1e6, r[name], (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("Thread_Id", "?")))
        for q in rows[max(0, i - ctx):i + ctx + 1]:
            print("     %10.3f ms  %8.3f ms  %s" % ((int(q["Start_Timestamp"]) - t0) / 1e6, (int(q["End_Timestamp"]) - int(q["Start_Timestamp"])) / 1e6, q[name]))
print(counts)
