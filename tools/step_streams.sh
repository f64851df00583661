#!/bin/bash
# kernel trace of 5 timed bench steps, summarised per HIP stream -> gpurun_out/<tag>_step_by_stream.txt
TAG=${1:-r04}
EXTRA=${2:-}          # e.g. "--workload cfg2"
MARK=${3:-marg_ce_row|marg_ce_stream}     # a kernel launched once per step (cfg2: small_grad_kernel)
OUT=gpurun_out/step_trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-pmc $EXTRA > $OUT/trace.log 2>&1 || true
T=$(ls -S $OUT/trace/*/*_kernel_trace.csv | head -1)
python tools/summarize_step_window.py $T --warmup 2 --steps 5 --top 60 --by-stream --sequence 90 --marker "$MARK" > gpurun_out/${TAG}_step_by_stream.txt 2>&1
head -c 200 $T > $OUT/columns.txt
rm -rf $OUT/trace
grep -A95 'consecutive launches' gpurun_out/${TAG}_step_by_stream.txt
