#!/bin/bash
# round 3: small-batch backward with every operand fragment of a wave in flight before its first MFMA (small_grad_kernel<true>):
# parity, then GPU time per call
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_r2_gpu.py tests/test_hip_parity.py -q -m gpu -k "small or contrastive or fused or sharded" > gpurun_out/small_tests.log 2>&1
tail -3 gpurun_out/small_tests.log
timeout 200 python tools/kernel_bench.py --only small 2>&1 | grep -A2 "^small"
