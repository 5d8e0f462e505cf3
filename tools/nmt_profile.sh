#!/bin/bash
# kernel mix of the pivot NMT training step: gpurun -- bash tools/nmt_profile.sh
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pn; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -- python3 $GRAFT_REPO_ROOT/tools/nmt_bench.py --steps 10 > /tmp/pn.log 2>&1
tail -1 /tmp/pn.log
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pn/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows); n=sum(int(r["Calls"]) for r in rows)
print("total kernel ms per step %.3f, launches per step %.0f" % (tot/1e6/25, n/25))
for r in rows[:16]: print(r["Name"][:64].ljust(64), "%6.1f/step"%(int(r["Calls"])/25), "%8.1f us avg"%(float(r["AverageNs"])/1e3), "%7.3f ms/step"%(float(r["TotalDurationNs"])/25e6))
PY
