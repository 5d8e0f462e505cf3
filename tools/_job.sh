cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nmt.py -m gpu -x -q -k "persistent_launches or configs2 or real_width" 2>&1 | tail -25 > gpurun_out/r4_nmt_tests.log
tail -8 gpurun_out/r4_nmt_tests.log
timeout 600 python -m pytest tests/test_gpu_nmt.py tests/test_gpu_pivot.py -m gpu -x -q 2>&1 | tail -3
