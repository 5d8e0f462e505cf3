#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_nmt.py tests/test_gpu_pivot.py -x -q > gpurun_out/nmt.log 2>&1
tail -3 gpurun_out/nmt.log
python tools/nmt_bench.py --steps 20 2>/dev/null | tail -1
python tools/nmt_bench.py --steps 20 2>/dev/null | tail -1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
bash $R/tools/nmt_profile.sh 2>&1 | tail -19 | cut -c1-150 | head -4
