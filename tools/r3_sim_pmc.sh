#!/bin/bash
# round 3: what bounds the streaming row-statistics kernel at 1200^2 - kernel trace + PMC passes (clock, MFMA, waits)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/sim_pmc; rm -rf $OUT; mkdir -p $OUT
F=gpurun_out/sim_pmc_1200.txt
{
for ks in 1 2; do
echo "## DALM_STREAM_KS=$ks   rocprofv3 --kernel-trace"
DALM_STREAM_KS=$ks rocprofv3 --kernel-trace --output-format csv -d $OUT/trace$ks -- python tools/kernel_bench.py --only sim --sizes 1200 > $OUT/trace$ks.log 2>&1
python tools/summarize_trace.py "$(find $OUT/trace$ks -name '*kernel_trace.csv' | head -1)" "sim_|transpose|rowstats|flash" 12
echo "## DALM_STREAM_KS=$ks   PMC"
DALM_STREAM_KS=$ks PMC_MATCH="sim_rowstats_stream|transpose_pad" python tools/pmc_run.py $OUT/pmc$ks "MfmaUtil" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" -- python tools/kernel_bench.py --only sim --sizes 1200
done
} > $F 2>&1
find $OUT -name "*.csv" -delete
cat $F
