#!/bin/bash
# round 4, eighth GPU call: the remaining helper launches moved in-launch (streaming statistics merge, sliced small backward)
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_kernels_r2_gpu.py tests/test_hip_parity.py tests/test_sharded_two_ranks_gpu.py -m gpu -q ) > gpurun_out/r04/merge_tests.log 2>&1
tail -6 gpurun_out/r04/merge_tests.log
timeout 300 python tools/kernel_bench.py --only small 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/small_bwd_one_launch.txt
cat gpurun_out/r04/small_bwd_one_launch.txt
timeout 300 python tools/kernel_bench.py --only sim --sizes 1200,1536,2048,3072 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/sim_midsize_merge_inlaunch.txt
cat gpurun_out/r04/sim_midsize_merge_inlaunch.txt
