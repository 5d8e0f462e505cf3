#!/bin/bash
# HBM traffic of the batch-assembly kernel: two separate counter passes (never combined with other trace domains)
R="$GRAFT_REPO_ROOT"
O="$R/gpurun_out/loader_pmc"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lpm
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/lpm/fetch -- python3 $R/tools/loader_kernel_only.py > /tmp/lpmf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/lpm/write -- python3 $R/tools/loader_kernel_only.py > /tmp/lpmw.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/lpm att_batch_assemble_wave_kernel $O/loader_pmc.json
tail -2 /tmp/lpmf.log
