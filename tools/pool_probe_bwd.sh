#!/bin/bash
F=gpurun_out/${1:-r04}_pool_bwd_kernels.txt
{
echo "# pool backward, all bench shapes: d-chunk kernel (DALM_POOL_BWD_ROWS=0, the round-3 kernel) vs the row-major default   $(date -u +%F)"
echo "## DALM_POOL_BWD_ROWS=0"
DALM_POOL_BWD_ROWS=0 python tools/kernel_bench.py --only pool 2>&1 | grep -A2 "^pool" | grep "^pool\|bwd"
echo "## default (row-major, <= 8 token slices)"
python tools/kernel_bench.py --only pool 2>&1 | grep -A2 "^pool"
} > $F 2>&1
cat $F
