#!/bin/bash
# Kernel-level durations (rocprofv3 --kernel-trace --stats) of the latency-bound kernels: small-batch contrastive
# path and the fused pool.  Usage (GPU box, repo root): bash tools/prof_small.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_small
rm -rf $OUT; mkdir -p $OUT
for what in small pool; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$what -- python tools/kernel_bench.py --only $what > $OUT/$what.log 2>&1 || true
  f=$(find $OUT/$what -name "*kernel_stats.csv" | head -1)
  python tools/summarize_rocprof.py "$f" 40 > $OUT/${what}_kernel_stats.txt || true
  t=$(find $OUT/$what -name "*kernel_trace.csv" | head -1)
  python tools/summarize_trace.py "$t" "small_|pool_|l2norm|flash|gemm_f32|rag_loss|splitk|rowstats" 60 > $OUT/${what}_per_shape.txt || true
  find $OUT/$what -name "*kernel_trace.csv" -size +8M -delete
done
cat $OUT/small_per_shape.txt $OUT/pool_per_shape.txt
