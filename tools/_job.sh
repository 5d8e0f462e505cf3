cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nmt.py -m gpu -x -q -k "persistent_decoder or configs2" 2>&1 | tail -25 > gpurun_out/r4_nmt_tests.log
tail -5 gpurun_out/r4_nmt_tests.log
timeout 300 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_topdown.py -m gpu -x -q -k "reproducible or embed or golden" 2>&1 | tail -4
timeout 300 bash tools/nmt_profile.sh > gpurun_out/r4_nmt_profile.txt 2>&1
cat gpurun_out/r4_nmt_profile.txt | cut -c1-160
