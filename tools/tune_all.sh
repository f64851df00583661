#!/bin/bash
# Regenerate / extend dalm_amd/tuning/tunableop_gfx950.csv on an MI355X: padded cfg3 / cfg2 / cfg5 shapes and every packed row count
# the trainers can produce (tools/tune_packed.py).  Results land in gpurun_out/tune/*.csv; merge with
#   python tools/tune_packed.py --merge gpurun_out/tune/<file>.csv
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/tune; mkdir -p $O
python tools/tune_packed.py --out $O/packed_llama.csv > $O/packed_llama.log 2>&1; tail -2 $O/packed_llama.log
python tools/tune_packed.py --generator falcon-7b --out $O/packed_falcon.csv > $O/packed_falcon.log 2>&1; tail -2 $O/packed_falcon.log
export DALM_TUNED_GEMMS=0 PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=20 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
PYTORCH_TUNABLEOP_FILENAME=$O/cfg2.csv python bench.py --workload cfg2 --steps 2 --warmup 1 --no-graph > $O/cfg2.log 2>&1
PYTORCH_TUNABLEOP_FILENAME=$O/cfg2_packed.csv python bench.py --workload cfg2 --data-path packed --steps 2 --warmup 4 --no-graph > $O/cfg2_packed.log 2>&1
ls -la $O
