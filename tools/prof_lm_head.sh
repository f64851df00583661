#!/bin/bash
# rocprofv3 kernel trace of the lm_head + CE sub-problem at the cfg3 shapes: materialised logits vs live-row chunks
# (tuned GEMM table).  Writes gpurun_out/prof_lm_head/lm_head_kernels_per_shape.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_lm_head; rm -rf $OUT; mkdir -p $OUT
for arm in "materialised" "2048"; do
  tag=$(echo $arm | tr -d ' ')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -- python tools/lm_head_bench.py --tuned --only=cfg3 "--arms=$arm" > $OUT/$tag.log 2>&1
  t=$(find $OUT/$tag -name "*kernel_trace.csv" | head -1)
  echo "## arm: $arm" >> $OUT/lm_head_kernels_per_shape.txt
  grep "ms " $OUT/$tag.log >> $OUT/lm_head_kernels_per_shape.txt
  python tools/summarize_trace.py "$t" "" 16 >> $OUT/lm_head_kernels_per_shape.txt
  echo >> $OUT/lm_head_kernels_per_shape.txt
done
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
cat $OUT/lm_head_kernels_per_shape.txt
