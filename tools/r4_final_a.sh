#!/bin/bash
# final session A of round 4: the whole GPU suite, then the bench lines
mkdir -p gpurun_out/final_r04
P=gpurun_out/final_r04
python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $P/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $P/smoke.txt
python bench.py > $P/r04_bench_default.json 2> $P/bench_default.err
tail -c 2500 $P/r04_bench_default.json
python bench.py --workload cfg5 --no-cpu-baseline --no-pmc > $P/r04_bench_cfg5.json 2>> $P/bench.err
python bench.py --workload cfg2 --no-cpu-baseline --no-pmc > $P/r04_bench_cfg2.json 2>> $P/bench.err
python bench.py --dtype fp32 --no-cpu-baseline --no-pmc > $P/r04_bench_cfg3_fp32.json 2>> $P/bench.err
python bench.py --use-bnb both --no-cpu-baseline --no-pmc > $P/r04_bench_cfg3_nf4.json 2>> $P/bench.err
for f in cfg5 cfg2 cfg3_fp32 cfg3_nf4; do python - <<PY
import json
try:
    d=json.loads([l for l in open("$P/r04_bench_$f.json") if l.startswith("{")][-1]); print("$f", d["value"], d["unit"], d["ms_per_step"], "ms/step")
except Exception as e: print("$f", "FAILED", e)
PY
done
tail -5 $P/bench.err
