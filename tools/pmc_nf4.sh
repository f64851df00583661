#!/bin/bash
# HBM traffic of the nf4 kernels (rocprofv3 --pmc, FETCH_SIZE and WRITE_SIZE in separate passes) next to their algorithmic bytes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_nf4; rm -rf $OUT; mkdir -p $OUT
{
echo "# nf4 kernels: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (KB per launch, mean; FETCH x2 per the gfx950 guide) - python tools/kernel_bench.py --only nf4"
echo "# algorithmic bytes, n = 11008 x 4096 = 45.09 M weights: quantise reads n*el (bf16 90.2 MB, f32 180.4 MB), writes 0.5625 n = 25.4 MB;"
echo "#                    dequantise reads 25.4 MB, writes n*el;  n = 4096^2 = 16.78 M: 33.6 MB bf16, 9.4 MB packed"
PMC_MATCH="nf4_" python tools/pmc_run.py $OUT "FETCH_SIZE" "WRITE_SIZE" -- python tools/kernel_bench.py --only nf4
echo "# rocprofv3 --kernel-trace durations"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python tools/kernel_bench.py --only nf4 > $OUT/trace.log 2>&1
python tools/summarize_trace.py "$(find $OUT/trace -name '*kernel_trace.csv' | head -1)" "nf4_" 12
} > gpurun_out/r03_nf4_kernels.txt 2>&1
find $OUT -name "*.csv" -delete
cat gpurun_out/r03_nf4_kernels.txt
