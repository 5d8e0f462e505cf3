#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_topdown.py tests/test_gpu_fullsize.py -x -q > gpurun_out/td.log 2>&1
tail -5 gpurun_out/td.log
for m in 0 3 0 3; do
echo "== mode $m"
UIC_EXP_MODE=$m python tools/host_time.py 2>&1 | grep -E "wall|prologue|recurrence done|BPTT|tail|joined|wgrads done"
UIC_EXP_MODE=$m python bench.py --no-cpu-baseline --no-f32 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done
