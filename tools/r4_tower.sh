#!/bin/bash
# round 4: rotary + SwiGLU kernels - parity tests, per-kernel lines, and the step with / without them
mkdir -p gpurun_out
python -m pytest tests/test_tower_ops_gpu.py -q -m gpu 2>&1 | tail -8
python tools/kernel_bench.py --only tower 2>&1 | grep -A8 "^tower" | tee gpurun_out/r04_tower_kernels.txt
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc"
{
echo "# bench.py cfg3, 20 steps: generator-tower kernels on / off   $(date -u +%F)"
for cfg in "1 1" "0 1" "1 0" "0 0" "1 1"; do set -- $cfg
  echo "## DALM_ROPE_KERNEL=$1 DALM_SWIGLU_KERNEL=$2"
  DALM_ROPE_KERNEL=$1 DALM_SWIGLU_KERNEL=$2 $B 2>&1 | grep '^{"metric"' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(f\"   {d['value']:.2f} pairs/s  {d['ms_per_step']:.2f} ms/step\")"
done
} 2>&1 | tee gpurun_out/r04_tower_step.txt
