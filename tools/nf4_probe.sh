#!/bin/bash
# nf4 dequantise: tiles per workgroup  -> gpurun_out/<tag>_nf4_steps.txt     (bash tools/nf4_probe.sh r04)
F=gpurun_out/${1:-r04}_nf4_steps.txt
{
echo "# tools/kernel_bench.py --only nf4: DALM_NF4_STEPS = tiles of 2048 weights per workgroup (1 = the round-3 kernel)   $(date -u +%F)"
for nt in 0 1; do for st in 1 2 4 8; do
  echo "## DALM_NF4_STEPS=$st DALM_NF4_NT=$nt"
  DALM_NF4_NT=$nt DALM_NF4_STEPS=$st python tools/kernel_bench.py --only nf4 2>&1 | grep -A3 "^nf4" | grep "^nf4\|dequantize"
done; done
echo "## default"
python tools/kernel_bench.py --only nf4 2>&1 | grep -A3 "^nf4"
} > $F 2>&1
cat $F
