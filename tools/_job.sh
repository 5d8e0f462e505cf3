#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in base xe512 base xe512; do
  if [ "$v" = base ]; then unset UIC_LIB; else export UIC_LIB=$R/variants/libuic_$v.so; fi
  echo "=== $v"
  python $R/tools/nmt_bench.py --steps 20 2>/dev/null | tail -1
done
export UIC_LIB=$R/variants/libuic_xe512.so
cd /tmp && export TMPDIR=/tmp
bash $R/tools/nmt_profile.sh 2>&1 | tail -19 | cut -c1-150 | grep -E "total|xe_"
timeout 300 python -m pytest $R/tests/test_gpu_nmt.py -x -q -k "large_vocab or configs2" 2>&1 | tail -1
