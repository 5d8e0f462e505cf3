#!/bin/bash
# kernel mix of the joint pivot decode (captioner beam search + translateBatch): gpurun -- bash tools/pivot_profile.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $GRAFT_REPO_ROOT/tools/pivot_decode_bench.py --iters 3 > /tmp/pp.log 2>&1
tail -1 /tmp/pp.log
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.3f (captioner x8, translator x8 passes)" % (tot/1e6))
for r in rows[:18]: print(r["Name"][:72].ljust(72), "%7d calls"%int(r["Calls"]), "%8.1f us avg"%(float(r["AverageNs"])/1e3), "%8.3f ms total"%(float(r["TotalDurationNs"])/1e6))
PY
