#!/bin/bash
# round 3, after the small-path backward change: whole GPU suite again, the headline line, the per-shape traces that name the
# changed kernels.  Outputs: gpurun_out/final2_r03/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/final2_r03; rm -rf $P; mkdir -p $P
timeout 1500 python -m pytest tests -q -m gpu > $P/gpu_suite.log 2>&1; tail -3 $P/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $P/smoke.log 2>&1; tail -1 $P/smoke.log
timeout 900 python bench.py > $P/r03_bench_default.json 2> $P/bench.err; python -c "
import json; d=json.loads(open('$P/r03_bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['loss_path_us'], d['parity']['rel'])"
timeout 600 bash tools/prof_small.sh > $P/prof_small.log 2>&1
cp gpurun_out/prof_small/small_per_shape.txt $P/r03_small_path_per_shape.txt
cp gpurun_out/prof_small/pool_per_shape.txt $P/r03_pool_per_shape.txt
timeout 900 bash tools/pmc_bench.sh r03 > $P/pmc_bench.log 2>&1
cp gpurun_out/pmc_bench_r03/pmc_loss_kernels.txt $P/r03_bench_pmc_loss_kernels.txt
cp gpurun_out/pmc_bench_r03/bench_kernel_stats.txt $P/r03_bench_step_kernel_stats.txt
cp gpurun_out/pmc_bench_r03/dalm_kernels_per_shape.txt $P/r03_bench_dalm_kernels_per_shape.txt
python tools/pmc_summary.py gpurun_out/pmc_bench_r03 --json $P/roofline_traffic.json --workload cfg3 --dtype bf16 \
  --source "profiles/r03_bench_pmc_loss_kernels.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py, real masks; FETCH doubled per the gfx950 guide)" > /dev/null
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out -name "*counter_collection.csv" -size +4M -delete
head -12 $P/r03_small_path_per_shape.txt
