#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_nmt.py tests/test_gpu_ops.py tests/test_gpu_topdown.py tests/test_gpu_fc.py tests/test_gpu_discriminator.py -x -q > gpurun_out/t.log 2>&1
grep -E "passed|failed" gpurun_out/t.log | tail -2
python tools/nmt_bench.py --steps 20 2>/dev/null | tail -1
python tools/nmt_bench.py --steps 20 2>/dev/null | tail -1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
bash $R/tools/nmt_profile.sh 2>&1 | tail -19 | cut -c1-150 | grep -E "total|embed_scan"
