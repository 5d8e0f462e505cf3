#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: kernels in start order with their queue, duration and grid,
then per-queue busy time.    python tools/step_timeline.py <kernel_trace.csv> [step index] [--full]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
step = rows[idx[k] + 1: idx[k + 1] + 1]
t0 = int(step[0]["Start_Timestamp"])
end = int(step[-1]["End_Timestamp"])
print("step wall %.1f us, %d kernels" % ((end - t0) / 1e3, len(step)))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", "")
    return n[:48]


busy = {}
agg = {}
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[r["Queue_Id"]] = busy.get(r["Queue_Id"], 0) + e - s
    key = short(r["Kernel_Name"]) + " g" + str(int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])) + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"]
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += e - s
    if "--full" in sys.argv:
        print("%9.1f %8.1f q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], key))
for q, b in busy.items():
    print("queue %s busy %.1f us" % (q, b / 1e3))
for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us  x%-3d %s" % (a[1] / 1e3, a[0], key))
