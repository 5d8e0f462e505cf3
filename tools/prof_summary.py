#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats --output-format csv` directory of bench.py:
per-kernel totals per training step, and -- for the roofline kernel -- the average duration split into the launches
made INSIDE the training steps (where the two-stream step lets side-stream GEMMs run concurrently and stretch it) and
the back-to-back ISOLATED launches of bench.py's roofline section (first half rotating through > Infinity-Cache
buffers = HBM-cold, second half one cache-resident pair), which is what `roofline.us_per_launch` times with HIP events.

    python3 tools/prof_summary.py <dir> <steps profiled> [rows] [roofline kernel substring]
"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 18
roof = sys.argv[4] if len(sys.argv) > 4 else "attn_fwd_fast_kernel"
f = max(glob.glob(d + '/*/*kernel_stats.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.3f ms  (%.3f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:nrows]:
    print("%-78s calls/step %6.1f  ms/step %7.3f  avg %8.2f us  %5.1f%%" % (
        r['Name'][:78], float(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, float(r['Percentage'])))

traces = glob.glob(os.path.dirname(f) + '/*kernel_trace.csv')
if traces:
    ev = []
    for r in csv.DictReader(open(traces[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), roof in r['Kernel_Name']))
    ev.sort()
    in_step, isolated = [], []
    for i, (s, e, is_roof) in enumerate(ev):
        if not is_roof:
            continue
        prev_roof = i > 0 and ev[i - 1][2]
        next_roof = i + 1 < len(ev) and ev[i + 1][2]
        (isolated if (prev_roof or next_roof) else in_step).append((e - s) / 1e3)
    def avg(v):
        return sum(v) / len(v) if v else float('nan')
    half = len(isolated) // 2
    print("\n%s: %d launches" % (roof, len(in_step) + len(isolated)))
    print("  inside training steps (co-running with the side stream) : n=%4d  avg %6.2f us" % (len(in_step), avg(in_step)))
    print("  isolated, rotating >256 MiB of inputs (HBM-cold)        : n=%4d  avg %6.2f us" % (half, avg(isolated[:half])))
    print("  isolated, one cache-resident pair (roofline.us_per_launch): n=%4d  avg %6.2f us" % (len(isolated) - half, avg(isolated[half:])))
