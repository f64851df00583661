#!/bin/bash
# Round-3 evidence for the lm_head + log-sum-exp kernel: ablations, rocprofv3 kernel durations next to the library path,
# PMC MfmaUtil / wait counters / L2 hit rate (separate passes) -> gpurun_out/r03_lm_head_mfma_kernel.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_lm_head; rm -rf $OUT; mkdir -p $OUT
F=gpurun_out/r03_lm_head_mfma_kernel.txt
{
echo "# lm_head + log-sum-exp kernel (dalm_lm_head_lse_fwd), round 3 - $(date -u +%F) - MI355X"
echo "## tools/lm_head_kernel_bench.py (GPU time inside a hipGraph; library path = hipBLASLt GEMM + forward CE kernel)"
python tools/lm_head_kernel_bench.py 2>&1 | grep -v amdgpu.ids
echo
echo "## the round-2 kernel on this box (DALM_LM_HEAD_GEN=2)"
DALM_LM_HEAD_GEN=2 python tools/lm_head_kernel_bench.py 2>&1 | grep "dalm_lm_head"
echo
echo "## tools/lm_head_ablate.py (what each part of the loop costs; ABL != 0 computes garbage)"
python tools/lm_head_ablate.py 2>&1 | grep -v amdgpu.ids
echo
echo "## rocprofv3 --kernel-trace, per (kernel, grid)"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python tools/lm_head_kernel_bench.py > $OUT/trace.log 2>&1
python tools/summarize_trace.py "$(find $OUT/trace -name '*kernel_trace.csv' | head -1)" "lm_head|Cijk|marg_ce" 12
echo
echo "## rocprofv3 --pmc, one pass per group (MfmaUtil | SQ wait counters | L2 hits), means over launches per (kernel, grid)"
PMC_MATCH="lm_head_lse4w|Cijk" python tools/pmc_run.py $OUT/pmc "MfmaUtil" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" -- python tools/lm_head_kernel_bench.py
} > $F 2>&1
find $OUT -name "*.csv" -delete
cat $F
