#!/bin/bash
# Kernel trace of the training step with the one-GPU stand-in for RCCL (tools/comm_proxy.py): which hardware queue every stream's
# kernels run on, and when the stand-in's pieces start relative to the step.  gpurun -- bash tools/comm_trace.sh [out]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-commtrace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct     # (a box can be handed out again with its /tmp: never read an earlier run's trace)
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -- python3 $R/tools/comm_proxy.py --steps 6 --only sharded > /tmp/ct.log 2>&1
tail -3 /tmp/ct.log
python3 - <<PY > $O/comm_timeline.txt
import csv, glob
f = glob.glob('/tmp/ct/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
adam = [i for i, r in enumerate(rows) if 'adam_ranges_kernel' in r['Kernel_Name'] or 'adam_kernel' in r['Kernel_Name']]
a, b = adam[-6], adam[-5]      # (a timed step: the last three are the stamped ones, each behind a host sync)
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a:b + 1]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    print('%9.1f %8.1f q%s %s g%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], n, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))))
PY
grep -c . $O/comm_timeline.txt
grep -n "comm_proxy\|adam" $O/comm_timeline.txt | head -20
