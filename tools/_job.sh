#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.txt 2>&1
tail -3 gpurun_out/gpu_suite.txt
for i in 1 2; do python bench.py --no-cpu-baseline --no-f32 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"; done
python tools/nmt_bench.py --steps 20 2>/dev/null | tail -1
python tools/gemm_headroom.py 2>&1 | grep -A3 "f32 input"
