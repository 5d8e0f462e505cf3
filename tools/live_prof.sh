R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/live
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa /tmp/pl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --long-run 0 --all-positions > /tmp/pa.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --long-run 0 > /tmp/pl.log 2>&1
python3 $R/tools/step_timeline.py $(find /tmp/pa -name "*kernel_trace.csv" | head -1) 4 --full > $O/timeline_all.txt 2>&1
python3 $R/tools/step_timeline.py $(find /tmp/pl -name "*kernel_trace.csv" | head -1) 4 --full > $O/timeline_live.txt 2>&1
python3 $R/tools/prof_summary.py /tmp/pa 32 45 > $O/summary_all.txt 2>&1
python3 $R/tools/prof_summary.py /tmp/pl 32 45 > $O/summary_live.txt 2>&1
grep '"metric"' /tmp/pa.log | tail -1 | cut -c1-300
grep '"metric"' /tmp/pl.log | tail -1 | cut -c1-300
