#!/bin/bash
timeout 900 python tools/_dbg.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "f32_input or 256_tile" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_topdown.py -x -q 2>&1 | tail -2
python tools/gemm_headroom.py 2>&1 | grep -A3 "f32 input"
python bench.py --no-cpu-baseline --no-f32 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"
