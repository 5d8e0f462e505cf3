#!/bin/bash
# kernel mix of the decode passes: gpurun -- bash tools/decode_profile.sh
cd /tmp && export TMPDIR=/tmp
for w in sample greedy; do
rm -rf /tmp/pd; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -- python3 $GRAFT_REPO_ROOT/tools/decode_profile.py $w > /tmp/pd.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/pd/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("== $w: total kernel ms per pass %.3f" % (tot/1e6/6))
for r in rows[:9]: print(r["Name"][:64].ljust(64), "%6.1f/pass"%(int(r["Calls"])/6), "%8.1f us avg"%(float(r["AverageNs"])/1e3), "%7.3f ms/pass"%(float(r["TotalDurationNs"])/6e6))
PY
done
