#!/bin/bash
# round 4, fifth GPU call: LoRA output as one addmm (A/B on the bench lines) + the tests that changed since the last full suite
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
: > gpurun_out/r04/lora_addmm.txt
for w in cfg2 cfg3 cfg5; do
for v in 0 1 0 1; do
  DALM_LORA_ADDMM=$v timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$w DALM_LORA_ADDMM=$v', round(d['value'], 2), 'pairs/s', round(d['ms_per_step'], 2), 'ms/step', 'final_loss', d['config'].get('final_loss'))
" >> gpurun_out/r04/lora_addmm.txt
done
done
cat gpurun_out/r04/lora_addmm.txt
( time timeout 900 python -m pytest tests -m gpu -q -k "realwidth or step_parity or one_launch or lora" ) > gpurun_out/r04/gpu_subset5.log 2>&1
tail -6 gpurun_out/r04/gpu_subset5.log
