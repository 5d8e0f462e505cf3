#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmt.py tests/test_gpu_pivot.py -x -q > gpurun_out/nmt.log 2>&1
grep -E "passed|failed" gpurun_out/nmt.log | tail -2
grep -E "^E " gpurun_out/nmt.log | head -8
for i in 1 2 3; do python tools/nmt_bench.py --steps 20 2>/dev/null | tail -1; done
python tools/pivot_decode_bench.py --iters 20 2>/dev/null | tail -1
