#!/bin/bash
# round 4, third GPU call: the whole GPU suite again (one ill-posed test fixed; the k-context tests have not run yet), and the
# pipelined small-path partial kernel: rounds in flight 0 (round-3 kernel) / 2 / 3 / 4
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q ) > gpurun_out/r04/gpu_suite3.log 2>&1
tail -12 gpurun_out/r04/gpu_suite3.log
: > gpurun_out/r04/small_pipe.txt
for rep in 1 2; do
for d in 0 2 3 4; do
  echo "== DALM_SMALL_PIPE=$d (pass $rep)" >> gpurun_out/r04/small_pipe.txt
  DALM_SMALL_PIPE=$d timeout 300 python tools/kernel_bench.py --only small 2>&1 | grep -v "amdgpu.ids" >> gpurun_out/r04/small_pipe.txt
done
done
cat gpurun_out/r04/small_pipe.txt
