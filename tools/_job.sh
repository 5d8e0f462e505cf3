cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_suite3.log 2>&1
grep -E "passed|failed" gpurun_out/r4_suite3.log | tail -3
grep -E "^E |Error" gpurun_out/r4_suite3.log | head -10
for i in 1 2; do python bench.py --no-f32 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done
