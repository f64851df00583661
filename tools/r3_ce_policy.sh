#!/bin/bash
# round 3: cache policy of the fused CE kernel's two streams (DALM_CE_VARIANT) x rows per workgroup (DALM_CE_BS), all-ones
# mask micro-benchmark (out of place and in place) and the in-step launch (bench.py's live roofline probe)
mkdir -p gpurun_out
out=gpurun_out/ce_policy.txt
: > $out
for v in n c l w; do
  for V in 32000 65024; do
    echo "== DALM_CE_VARIANT=$v V=$V" >> $out
    DALM_CE_VARIANT=$v timeout 200 python tools/kernel_bench.py --only ce --vocab $V 2>&1 | grep -A6 "bf16" | grep -v "^ce.*f32" >> $out
  done
done
for v in n c; do
  echo "== DALM_CE_BS=1024 DALM_CE_VARIANT=$v V=32000" >> $out
  DALM_CE_BS=1024 DALM_CE_VARIANT=$v timeout 200 python tools/kernel_bench.py --only ce --vocab 32000 2>&1 | grep -A6 "bf16" >> $out
done
for v in n c l w; do
  echo "== bench.py DALM_CE_VARIANT=$v" >> $out
  DALM_CE_VARIANT=$v timeout 400 python bench.py --no-pmc --steps 8 --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line); r=d['roofline']
        print('value', d['value'], 'ms', d['ms_per_step'], 'avg_launch_us', r['avg_launch_us'], 'frac', r['frac'])
" >> $out
done
cat $out
