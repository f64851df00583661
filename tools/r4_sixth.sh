#!/bin/bash
# round 4, sixth GPU call: bf16x3 first pass of the exact top-k (tests + timing), depth-2 tolerance, full suite
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r04/gpu_suite6.log 2>&1
tail -8 gpurun_out/r04/gpu_suite6.log
timeout 600 python tools/topk_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/topk_bench.txt
cat gpurun_out/r04/topk_bench.txt
