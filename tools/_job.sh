#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fullsize_decode.py -x -q > gpurun_out/td.log 2>&1
tail -3 gpurun_out/td.log
timeout 200 python tools/rnn_persist_probe.py --dbg > gpurun_out/rnn_persist_probe.txt 2>&1
grep -E "phase|step  |forward" gpurun_out/rnn_persist_probe.txt
for i in 1 2; do
python tools/host_time.py 2>&1 | grep -E "wall|prologue|recurrence done|joined"
python bench.py --no-cpu-baseline --no-f32 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
done
