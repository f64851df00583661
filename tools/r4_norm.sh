#!/bin/bash
# round 4: residual add + RMSNorm kernels - parity tests, the step with / without them
mkdir -p gpurun_out
python -m pytest tests/test_tower_ops_gpu.py tests/test_lora_ops_gpu.py -q -m gpu -x 2>&1 | tail -12
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc"
{
echo "# bench.py cfg3, 20 steps: residual add + RMSNorm as dalm_rms_norm_* (default) vs torch's kernels   $(date -u +%F)"
for v in 1 0 1; do
  echo "## DALM_NORM_KERNEL=$v"
  DALM_NORM_KERNEL=$v $B 2>&1 | grep '^{"metric"\|Error\|error' | python -c "
import json,sys
for l in sys.stdin:
    try:
        d=json.loads(l); print(f\"   {d['value']:.2f} pairs/s  {d['ms_per_step']:.2f} ms/step\")
    except Exception: print('   ', l.strip()[:300])"
done
} 2>&1 | tee gpurun_out/r04_norm_step.txt
