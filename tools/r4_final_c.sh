#!/bin/bash
mkdir -p gpurun_out/final_r04
P=gpurun_out/final_r04
python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $P/r04_pytest_gpu_tail.txt
python bench.py --workload cfg5 --no-cpu-baseline --no-pmc > $P/r04_bench_cfg5.json 2>> $P/bench.err
python bench.py --workload cfg2 --no-cpu-baseline --no-pmc > $P/r04_bench_cfg2.json 2>> $P/bench.err
for f in cfg5 cfg2; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$P/r04_bench_$f.json") if l.startswith("{")][-1]); print("$f", d["value"], d["unit"], d["ms_per_step"], "ms/step")
except Exception as e: print("$f", "FAILED", e)
PY
done
