#!/usr/bin/env python3
"""Compact per-step timeline from a rocprofv3 kernel trace of bench.py: picks one training step in the middle of the
trace (delimited by adam_kernel launches) and prints, per stream, runs of kernels with start offset, span and busy time.

    python3 tools/timeline.py <dir> [step index from the end, default 3]
"""
import csv
import glob
import re
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
win = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else None   # [ms, ms): print every kernel inside
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0")), r.get("Queue_Id", "0")))
ev.sort()
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
step = ev[lo:hi]
t0 = step[0][0]
print("step: %d kernels, wall %.3f ms" % (len(step), (step[-1][1] - t0) / 1e6))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9A-Z]+?)I", n)
    if m:
        return m.group(1)
    if "uic_gemm_kernel" in n:
        m = re.search(r"uic_gemm_kernel<[^,]*, [^,]*, (.*?)>", n)
        return "gemm<" + (re.sub(r"\s+", "", n.split("uic_gemm_kernel<")[1].split(">")[0])[-22:]) + ">"
    return n[:40]


streams = sorted(set(e[4] for e in step))
busy_all = 0
for q in streams:
    ks = [e for e in step if e[4] == q]
    busy = sum(e[1] - e[0] for e in ks)
    print("queue %s: %d kernels, busy %.3f ms, first @%.3f ms, last end @%.3f ms" % (q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, (ks[-1][1] - t0) / 1e6))
    # runs of the same kernel name
    runs = []
    for e in ks:
        n = short(e[2])
        if runs and runs[-1][0] == n:
            runs[-1][2] = e[1]; runs[-1][3] += e[1] - e[0]; runs[-1][4] += 1
        else:
            runs.append([n, e[0], e[1], e[1] - e[0], 1])
    # merge into phases by printing only runs > 20 us or counts
    for n, s, e, b, c in runs:
        if b > 15000 or c > 1:
            print("   @%7.3f ms  span %7.3f ms  busy %7.3f ms  x%-3d %s" % ((s - t0) / 1e6, (e - s) / 1e6, b / 1e6, c, n))

if win:
    print("\nall kernels starting in [%.3f, %.3f) ms:" % win)
    for e in step:
        st = (e[0] - t0) / 1e6
        if win[0] <= st < win[1]:
            print("   q%s @%7.3f ms  %7.1f us  %s" % (e[4], st, (e[1] - e[0]) / 1e3, short(e[2])))
