#!/bin/bash
# The round's last GPU job (VERDICT process rule): the whole GPU suite and the smoke entry on a fresh box, log under gpurun_out/.
#   gpurun --timeout 3000 -- 'bash tools/final_suite.sh'      then copy gpurun_out/final_gpu_suite.txt to profiles/
mkdir -p gpurun_out
{
  echo "HEAD ${UIC_HEAD:-$(cat .git_head 2>/dev/null)}"      # (no .git on the box: pass UIC_HEAD=$(git rev-parse --short HEAD) in the gpurun command)
  python -m pytest tests -m gpu -x -q 2>&1 | tail -15
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -i "smoke"
} > gpurun_out/final_gpu_suite.txt 2>&1
tail -8 gpurun_out/final_gpu_suite.txt
