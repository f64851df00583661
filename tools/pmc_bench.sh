#!/bin/bash
# In-situ profile of bench.py: (1) kernel-trace stats of the full-depth eager step, (2) PMC FETCH_SIZE and
# WRITE_SIZE (separate passes) of the loss-path kernels at reduced tower depth - the loss kernels see the
# same shapes/masks at any depth, and the CSV stays small.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $OUT/trace.log 2>&1 || true
D="--retriever-layers 2 --generator-layers 2 --steps 4 --warmup 1 --no-graph --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py $D > $OUT/fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py $D > $OUT/write.log 2>&1 || true
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); python tools/summarize_rocprof.py "$f" 50 > $OUT/bench_kernel_stats.txt; tail -22 $OUT/bench_kernel_stats.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +6M -delete
