#!/bin/bash
# round 3: streaming row-statistics kernel at mid sizes - K-sliced workgroups (DALM_STREAM_KS = 1 / 2 / 4, unset = the plan's
# choice) and the LDS-tiled form for reference.  The run committed as profiles/r03_sim_midsize_experiments.txt also swept
# three variants that were removed afterwards (XCD rectangles + L2 warm-up, LDS-padded occupancy cap, 4-k granule copies).
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_r2_gpu.py tests/test_hip_parity.py -q -m gpu -k "rowstats or contrastive or fused or sim" > gpurun_out/sim_ks_tests.log 2>&1
tail -3 gpurun_out/sim_ks_tests.log
out=gpurun_out/sim_ks.txt
: > $out
for ks in 0 1 2 4; do
  echo "== KS=$ks" >> $out
  DALM_STREAM_KS=$ks timeout 300 python tools/kernel_bench.py --only sim --sizes 768,1200,1536,2048,3072 2>&1 | grep -A1 "^sim" | grep -v "^--" | paste - - | awk '{print $1,$2,$4,$5,$6}' >> $out
done
echo "== LDS-tiled form (DALM_SIM_ROWSTATS=g)" >> $out
DALM_SIM_ROWSTATS=g timeout 300 python tools/kernel_bench.py --only sim --sizes 768,1200,1536,2048,3072 2>&1 | grep -A1 "^sim" | grep -v "^--" | paste - - | awk '{print $1,$2,$4,$5,$6}' >> $out
cat $out
