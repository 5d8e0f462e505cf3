#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_topdown.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_decode.py tests/test_gpu_boundary.py tests/test_gpu_dp2.py -x -q > gpurun_out/td.log 2>&1
grep -E "passed|failed" gpurun_out/td.log | tail -2
for i in 1 2 3; do
python tools/host_time.py 2>&1 | grep -E "prologue|recurrence done|joined" | tr -s ' ' | tr '\n' ' '
python bench.py --no-cpu-baseline --no-f32 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done
