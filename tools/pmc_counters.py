#!/usr/bin/env python3
"""Per-launch averages of arbitrary rocprofv3 --pmc counters for one kernel (every *counter_collection.csv below a directory).
    python3 tools/pmc_counters.py <dir> <kernel-name substring>"""
import collections
import csv
import glob
import sys

d, kernel = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            vals[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
for name in sorted(vals):
    v = sorted(vals[name].values())
    v = v[len(v) // 8: len(v) - len(v) // 8] if len(v) >= 16 else v
    print("%-32s per launch %.4g   (%d launches)" % (name, sum(v) / len(v), len(vals[name])))
