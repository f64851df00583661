#!/bin/bash
# In-situ profile of bench.py: (1) kernel-trace of the full-depth eager step -> totals and PER-SHAPE durations,
# (2) PMC FETCH_SIZE and WRITE_SIZE (separate passes, kernel-trace only alongside) of the loss-path kernels at reduced
# tower depth - the loss kernels see the same shapes/masks at any depth, and the CSV stays small.
#   bash tools/pmc_bench.sh [round-tag] [extra bench args, e.g. "--workload cfg5"]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02}; EXTRA=${2:-}
OUT=gpurun_out/pmc_bench_$TAG$(echo "$EXTRA" | tr -c 'a-zA-Z0-9\n' '_')
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline $EXTRA > $OUT/trace.log 2>&1 || true
D="--retriever-layers 2 --generator-layers 2 --steps 4 --warmup 1 --no-graph --no-cpu-baseline $EXTRA"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py $D > $OUT/fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py $D > $OUT/write.log 2>&1 || true
python tools/pmc_summary.py $OUT > $OUT/pmc_loss_kernels.txt 2>&1 || true
cat $OUT/pmc_loss_kernels.txt
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
# the TIMED steps only (round 4): initialisation kernels, warm-up steps and post-run probes are cut out of the trace
python tools/summarize_step_window.py "$t" --warmup 2 --steps 5 --top 50 > $OUT/bench_kernel_stats.txt; tail -25 $OUT/bench_kernel_stats.txt
python tools/summarize_trace.py "$t" "dalm|marg_ce|small_|pool_|flash|gemm_f32|rag_loss|ce_|lora_|rope_|swiglu_|rms_norm" 80 > $OUT/dalm_kernels_per_shape.txt; cat $OUT/dalm_kernels_per_shape.txt
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json || true
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
