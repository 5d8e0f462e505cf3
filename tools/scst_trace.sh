#!/bin/bash
# kernel trace of the self-critical step (resident inputs): gpurun -- bash tools/scst_trace.sh ; the trace of ONE step lands in gpurun_out/scst_trace.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ps -- python3 $GRAFT_REPO_ROOT/tools/scst_bench.py --only resident --steps 6 --rounds 1 > /tmp/ps.log 2>&1
tail -3 /tmp/ps.log
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/ps/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the Adam kernel; take the second-to-last complete step
ends = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
a, b = ends[-3] + 1, ends[-2] + 1
t0 = int(rows[a]["Start_Timestamp"])
out = open("$GRAFT_REPO_ROOT/gpurun_out/scst_trace.txt", "w")
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    out.write("%9.1f us  +%7.1f  dur %7.1f  q%-3s %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
out.write("step: %d kernels, %.3f ms from first start to last end\n" % (b - a, (prev_end - t0) / 1e6))
out.close()
print(open("$GRAFT_REPO_ROOT/gpurun_out/scst_trace.txt").read()[-400:])
PY
