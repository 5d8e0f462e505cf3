#!/bin/bash
# A/B of library variants on one box: tools/ab_step.sh <rounds> <variant>...   ("base" = the in-tree library, else variants/libuic_<name>.so
# from tools/build_variant.sh); alternates the variants <rounds> times and prints the fused step's wall time and timing marks.
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset UIC_LIB; else export UIC_LIB=$PWD/variants/libuic_$v.so; fi
    echo "=== $v (round $r)"
    timeout 200 python tools/host_time.py --steps 30 2>&1 | grep -v amdgpu.ids
  done
done
