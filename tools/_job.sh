cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_topdown.py tests/test_gpu_boundary.py tests/test_gpu_dp2.py tests/test_gpu_fullsize_decode.py -m gpu -x -q > gpurun_out/r4_tests.log 2>&1
grep -E "passed|failed" gpurun_out/r4_tests.log | tail -3
for i in 1 2 3; do python bench.py --no-f32 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])"; done
