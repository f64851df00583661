#!/bin/bash
# One GPU session that regenerates everything under profiles/ for a round (run on the GPU box, repo root):
#   bash tools/refresh_profiles.sh r04
# Output lands in gpurun_out/profiles_<tag>/ ; copy the summaries into profiles/ afterwards.
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=gpurun_out/profiles_$TAG
rm -rf $P; mkdir -p $P
# 1. bench lines (HIP-event roofline inside)
python bench.py > $P/${TAG}_bench_default.json 2> $P/bench_default.err
python bench.py --dtype fp32 --no-cpu-baseline > $P/${TAG}_bench_cfg3_fp32.json 2>> $P/bench_default.err
python bench.py --workload cfg5 --no-cpu-baseline > $P/${TAG}_bench_cfg5.json 2>> $P/bench_default.err
python bench.py --workload cfg2 > $P/${TAG}_bench_cfg2.json 2>> $P/bench_default.err
python bench.py --workload cfg1 > $P/${TAG}_bench_cfg1.json 2>> $P/bench_default.err
python bench.py --steps 20 --warmup 5 --data-path loader --no-cpu-baseline --no-pmc > $P/${TAG}_bench_cfg3_loader.json 2>> $P/bench_default.err
python bench.py --steps 30 --warmup 6 --fuse-lm-head --no-cpu-baseline > $P/${TAG}_bench_cfg3_fuse_lm_head.json 2>> $P/bench_default.err
python bench.py --data-path bucketed --steps 24 --warmup 12 --no-cpu-baseline > $P/${TAG}_bench_cfg3_bucketed_trimmed.json 2>> $P/bench_default.err
# 2. in-situ kernel trace + PMC traffic of the loss kernels in bench.py (cfg3 bf16 and cfg5)
bash tools/pmc_bench.sh $TAG > $P/pmc_bench.log 2>&1
cp gpurun_out/pmc_bench_$TAG/pmc_loss_kernels.txt $P/${TAG}_bench_pmc_loss_kernels.txt
cp gpurun_out/pmc_bench_$TAG/bench_kernel_stats.txt $P/${TAG}_bench_step_kernel_stats.txt
cp gpurun_out/pmc_bench_$TAG/dalm_kernels_per_shape.txt $P/${TAG}_bench_dalm_kernels_per_shape.txt
python tools/pmc_summary.py gpurun_out/pmc_bench_$TAG --json $P/roofline_traffic.json --workload cfg3 --dtype bf16 \
  --source "profiles/${TAG}_bench_pmc_loss_kernels.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py, real masks; FETCH doubled per the gfx950 guide)" > /dev/null
# 3. kernel micro-benchmarks (HIP events / hipGraph timing) + per-shape rocprof durations + MFMA utilisation
python tools/kernel_bench.py > $P/${TAG}_kernel_bench.txt 2>/dev/null
bash tools/pmc_sim.sh > $P/pmc_sim.log 2>&1
sed -n '/== util/,$p' $P/pmc_sim.log > $P/${TAG}_sim_mfma_util_and_durations.txt
bash tools/prof_small.sh > $P/prof_small.log 2>&1
cp gpurun_out/prof_small/small_per_shape.txt $P/${TAG}_small_path_per_shape.txt
cp gpurun_out/prof_small/pool_per_shape.txt $P/${TAG}_pool_per_shape.txt
# 4. CE kernels: per-shape durations (one trace) and HBM traffic at all-ones masks (one vocabulary size per PMC process)
OUT=gpurun_out/pmc_ce_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/kernel_bench.py --only ce > $OUT/trace.log 2>&1
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py "$t" "marg_ce|ce_" 40 > $P/${TAG}_ce_kernels_per_shape.txt
bash tools/pmc_ce.sh > $P/pmc_ce.log 2>&1
cp gpurun_out/pmc_ce/summary.txt $P/${TAG}_ce_pmc_traffic_per_shape.txt
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out -name "*counter_collection.csv" -size +4M -delete
ls -la $P
