cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/tools/_probe.py > /tmp/pq.log 2>&1
tail -2 /tmp/pq.log
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $(find /tmp/pq -name "*kernel_trace.csv" | head -1) 8 --full > $GRAFT_REPO_ROOT/gpurun_out/r4_proxy_timeline.txt 2>&1
grep -n "comm_proxy\|persist\|adam\|step wall\|queue" $GRAFT_REPO_ROOT/gpurun_out/r4_proxy_timeline.txt | head -40
