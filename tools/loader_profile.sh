#!/bin/bash
# rocprofv3 kernel stats of the input-pipeline bench (the program itself right after `--`)
mkdir -p "$GRAFT_REPO_ROOT/gpurun_out/prof_loader"
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_loader"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$GRAFT_REPO_ROOT/tools/loader_bench.py" --batches 3 > "$OUT/bench.json" 2> "$OUT/err.txt"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
head -8 "$f" > "$OUT/kernel_stats_head.csv"
cat "$OUT/kernel_stats_head.csv"
