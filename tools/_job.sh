cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "256_tile" > gpurun_out/r4_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r4_tests.log | tail -3
for i in 1 2 3; do python bench.py --no-f32 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done
timeout 200 python tools/host_time.py 2>&1 | tail -18 | head -11
