#!/bin/bash
# pool kernels at the sizes where they are an HBM problem: workgroup size of the forward, store kind of the backward
# -> gpurun_out/<tag>_pool_probe.txt      (bash tools/pool_probe.sh r04)
F=gpurun_out/${1:-r04}_pool_probe.txt
{
echo "# tools/kernel_bench.py --only pool on one MI355X, $(date -u +%F)"
for nt in 1024 256; do for st in 0 1; do
  echo "## DALM_POOL_NT=$nt DALM_POOL_BWD_NT=$st"
  DALM_POOL_NT=$nt DALM_POOL_BWD_NT=$st python tools/kernel_bench.py --only pool 2>&1 | grep -A2 "^pool cfg2 p B150 T128 D1024 bf16\|^pool B1200\|^pool cfg2 p B150 T128 D1024 f32"
done; done
echo "## defaults"
python tools/kernel_bench.py --only pool 2>&1 | grep -A2 "^pool"
} > $F 2>&1
cat $F
