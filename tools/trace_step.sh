#!/bin/bash
# One step's kernels per queue in start order (rocprofv3 kernel trace of the bench; tracing makes the host launch-bound: kernel mix,
# order and durations only).  Run as: gpurun -- bash tools/trace_step.sh [out name]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p7     # (a box can be handed out again with its /tmp: never read an earlier run's trace)
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p7 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-f32 > /tmp/p7.log 2>&1
python3 $R/tools/step_timeline.py $(find /tmp/p7 -name "*kernel_trace.csv" | head -1) 4 --full > $O/step_timeline.txt 2>&1
python3 $R/tools/prof_summary.py /tmp/p7 24 30 > $O/bench_summary.txt 2>&1
tail -3 /tmp/p7.log
