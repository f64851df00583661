#!/bin/bash
# round 3: nf4 tests + the accumulation test whose tolerance changed + nf4 kernel bandwidth
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_nf4.py tests/test_step_parity_gpu.py -x -q -m gpu -k "nf4 or accumulation or kernels or bnb or linear" > gpurun_out/nf4_tests.log 2>&1
tail -15 gpurun_out/nf4_tests.log
timeout 300 python tools/kernel_bench.py --only nf4 2>&1 | tail -12
