#!/bin/bash
# round 3, closing run: whole GPU suite, smoke, the headline bench line (+ the nf4 line), kernel micro-benchmarks, the
# in-situ kernel trace / PMC of the loss kernels and the similarity MfmaUtil passes.  Outputs: gpurun_out/final_r03/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/final_r03; rm -rf $P; mkdir -p $P
timeout 1500 python -m pytest tests -q -m gpu > $P/gpu_suite.log 2>&1; tail -4 $P/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $P/smoke.log 2>&1; tail -2 $P/smoke.log
timeout 900 python bench.py > $P/r03_bench_default.json 2> $P/bench.err; tail -c 600 $P/r03_bench_default.json
timeout 600 python bench.py --use-bnb both --no-cpu-baseline --no-pmc > $P/r03_bench_cfg3_nf4.json 2>> $P/bench.err; tail -c 300 $P/r03_bench_cfg3_nf4.json
timeout 600 python tools/kernel_bench.py > $P/r03_kernel_bench.txt 2>/dev/null; grep -A2 "1200x1200\|nf4 11008" $P/r03_kernel_bench.txt
timeout 900 bash tools/pmc_bench.sh r03 > $P/pmc_bench.log 2>&1
cp gpurun_out/pmc_bench_r03/pmc_loss_kernels.txt $P/r03_bench_pmc_loss_kernels.txt
cp gpurun_out/pmc_bench_r03/bench_kernel_stats.txt $P/r03_bench_step_kernel_stats.txt
cp gpurun_out/pmc_bench_r03/dalm_kernels_per_shape.txt $P/r03_bench_dalm_kernels_per_shape.txt
python tools/pmc_summary.py gpurun_out/pmc_bench_r03 --json $P/roofline_traffic.json --workload cfg3 --dtype bf16 \
  --source "profiles/r03_bench_pmc_loss_kernels.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py, real masks; FETCH doubled per the gfx950 guide)" > /dev/null
timeout 600 bash tools/pmc_sim.sh > $P/pmc_sim.log 2>&1
sed -n '/== util/,$p' $P/pmc_sim.log > $P/r03_sim_mfma_util_and_durations.txt
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out -name "*counter_collection.csv" -size +4M -delete
ls -la $P
