"""Summarise a rocprofv3 --kernel-trace --stats output directory: per-kernel totals per training step."""
import csv, glob, sys, collections
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
import os
f = max(glob.glob(d + '/*/*kernel_stats.csv'), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time %.3f ms  (%.3f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("%-78s calls/step %6.1f  ms/step %7.3f  avg %8.2f us  %5.1f%%" % (
        r['Name'][:78], float(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
