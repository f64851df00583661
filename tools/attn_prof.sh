#!/bin/bash
# per-kernel durations of tools/attn_bench.py -> gpurun_out/attn_kernels.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
rm -rf gpurun_out/at; mkdir -p gpurun_out/at
(cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/at -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 5 "$@" > $GRAFT_REPO_ROOT/gpurun_out/at/log.txt 2>&1)
python tools/summarize_trace.py "$(find gpurun_out/at -name '*kernel_trace.csv' | head -1)" "attn|bwd_" 20 > gpurun_out/attn_kernels.txt 2>&1
cat gpurun_out/at/log.txt | tail -4 >> gpurun_out/attn_kernels.txt
rm -rf gpurun_out/at
cat gpurun_out/attn_kernels.txt
