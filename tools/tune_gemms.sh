#!/bin/bash
# Regenerate dalm_amd/tuning/tunableop_gfx950.csv on an MI355X (about 3-4 minutes of GPU time).
set -e
OUT=${1:-gpurun_out/tunableop_results.csv}
export DALM_TUNED_GEMMS=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$OUT \
       PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
python bench.py --workload cfg2 --steps 2 --warmup 1 --no-graph
ls -la gpurun_out/tunableop_results*.csv
