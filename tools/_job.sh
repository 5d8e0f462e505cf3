#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_topdown.py tests/test_gpu_fullsize.py -x -q > gpurun_out/td.log 2>&1
tail -2 gpurun_out/td.log
for i in 1 2; do
python tools/host_time.py 2>&1 | grep -E "wall|prologue|recurrence done|joined"
python bench.py --no-cpu-baseline --no-f32 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done
timeout 300 python tools/phase_times.py 2>/dev/null | tail -20
