#!/bin/bash
# round 5, GPU call 2: GPU-side times of the LoRA kernels (graph replay + rocprofv3 trace) and the in-situ step trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_lora2_gpu.py -q -m gpu -p no:cacheprovider -k "protocol or training" > $O/lora2_tests_b.log 2>&1; tail -3 $O/lora2_tests_b.log
timeout 200 python tools/lora_bench.py --json $O/lora_bench_4608x4096.json > $O/lora_bench_4608x4096.txt 2>&1
timeout 200 python tools/lora_bench.py --rows 19200 --cols 1024 --json $O/lora_bench_19200x1024.json > $O/lora_bench_19200x1024.txt 2>&1
cat $O/lora_bench_4608x4096.txt $O/lora_bench_19200x1024.txt
rm -rf $O/lora_trace; (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/lora_trace -- python $GRAFT_REPO_ROOT/tools/lora_bench.py --iters 10 > /dev/null 2>&1)
t=$(find $O/lora_trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py "$t" "lora" 40 > $O/lora_kernels_per_shape.txt; cat $O/lora_kernels_per_shape.txt | cut -c1-60,100-200
rm -rf $O/lora_trace
bash tools/step_streams.sh r05a > /dev/null 2>&1
grep -E "lora|rms_norm|swiglu|rope|wall span|^## stream" gpurun_out/r05a_step_by_stream.txt | head -60
