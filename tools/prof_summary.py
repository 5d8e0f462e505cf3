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
    # CU-time: a launch of W workgroups holds at most min(W, 256) of the 256 CUs, so its claim on the chip is duration x min(W, 256) /
    # 256 "chip-microseconds".  The step's BPTT window is bound by CU time: this, not the kernel's own duration, is
    # what a kernel that deliberately runs on few CUs (gemm_tn_pp.hip: 56 workgroups per LSTM chunk) costs the step.
    cu = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for r in csv.DictReader(open(traces[0])):
        try:
            wgs = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * \
                  (int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_Z'])))
        except (KeyError, ValueError):
            continue
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        a = cu[r['Kernel_Name']]
        a[0] += dur * min(wgs, 256) / 256.0
        a[1] += dur
        a[2] += 1
    print("\nchip-time per step (duration x min(workgroups, 256) / 256), the kernels that claim most of the chip:")
    for k, a in sorted(cu.items(), key=lambda kv: -kv[1][0])[:14]:
        print("%-78s chip-ms/step %7.3f   (kernel ms/step %7.3f, mean CUs %5.0f)" % (k[:78], a[0] / 1e3 / steps, a[1] / 1e3 / steps, 256.0 * a[0] / a[1] if a[1] else 0))
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
