#!/bin/bash
# round 4, fourth GPU call: one-launch small forward (tests + timing), bf16x3 kernel under rocprofv3 (durations, MfmaUtil,
# traffic), the timed-window kernel statistics of the bench step
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_kernels_r2_gpu.py -m gpu -q -k "one_launch or small_path" ) > gpurun_out/r04/one_launch_tests.log 2>&1
tail -8 gpurun_out/r04/one_launch_tests.log
timeout 300 python tools/kernel_bench.py --only small 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/small_one_launch.txt
cat gpurun_out/r04/small_one_launch.txt
cd /tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_x3; mkdir -p gpurun_out/pmc_x3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_x3/trace -- python tools/x3_probe.py 4096 16384 > gpurun_out/pmc_x3/trace.log 2>&1
t=$(find gpurun_out/pmc_x3/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py "$t" "lm_head|split3|lse_merge" 20 > gpurun_out/r04/x3_per_shape.txt; cat gpurun_out/r04/x3_per_shape.txt
PMC_MATCH="lm_head_lse4w|split3|lse_merge" timeout 600 python tools/pmc_run.py gpurun_out/pmc_x3/pmc "MfmaUtil" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" -- python tools/x3_probe.py 4096 16384 > gpurun_out/r04/x3_pmc.txt 2>&1
cat gpurun_out/r04/x3_pmc.txt
find gpurun_out/pmc_x3 -name "*.csv" -size +2M -delete
timeout 900 bash tools/pmc_bench.sh r04 > gpurun_out/r04/pmc_bench.log 2>&1
tail -40 gpurun_out/pmc_bench_r04/bench_kernel_stats.txt
