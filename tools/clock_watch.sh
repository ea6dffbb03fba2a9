#!/bin/bash
# Shader clock / power of the GPUs while a command runs: samples every amdgpu hwmon directory every 50 ms and prints, per
# card that did any work, the clock distribution seen while it drew more than 60 % of its peak sampled power.
#   tools/clock_watch.sh python bench.py --steps 5 --warmup 2 --no-extra
out=${CLOCK_WATCH_OUT:-gpurun_out/clock_watch.txt}
mkdir -p "$(dirname "$out")"
hws=""
for d in /sys/class/drm/card*/device/hwmon/hwmon*; do
  [ -r "$d/freq1_input" ] && hws="$hws $d"
done
if [ -z "$hws" ]; then echo "no readable amdgpu hwmon directory" | tee "$out"; "$@"; exit $?; fi
( while :; do
    for d in $hws; do echo "$d $(cat $d/freq1_input 2>/dev/null) $(cat $d/power1_input 2>/dev/null || cat $d/power1_average 2>/dev/null)"; done
    sleep 0.05
  done ) > "$out.raw" &
watcher=$!
"$@"; rc=$?
kill $watcher 2>/dev/null; wait $watcher 2>/dev/null
python3 - "$out.raw" > "$out" <<'PY'
import collections, sys
cards = collections.defaultdict(list)
for l in open(sys.argv[1]):
    t = l.split()
    if len(t) == 3:
        cards[t[0]].append((int(t[1]) / 1e6, int(t[2]) / 1e6))
for d, f in sorted(cards.items()):
    pmax, pmin = max(p for _, p in f), min(p for _, p in f)
    if pmax < 1.5 * pmin:
        continue                                           # this card did not run the command
    busy = sorted(c for c, p in f if p > 0.6 * pmax)
    q = lambda x: busy[min(len(busy) - 1, int(x * len(busy)))]
    print(f"{d.split('/')[4]}: samples {len(f)}, busy {len(busy)} (power > {0.6 * pmax:.0f} W; max {pmax:.0f} W, min {pmin:.0f} W)")
    print(f"  shader clock while busy, MHz: min {busy[0]:.0f}  p10 {q(0.1):.0f}  median {q(0.5):.0f}  p90 {q(0.9):.0f}  max {busy[-1]:.0f};  overall {min(c for c, _ in f):.0f} - {max(c for c, _ in f):.0f}")
PY
rm -f "$out.raw"; cat "$out"; exit $rc
