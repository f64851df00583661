#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_lora2_gpu.py -q -m gpu -p no:cacheprovider > $O/lora2_tests.log 2>&1; tail -6 $O/lora2_tests.log
timeout 200 python tools/lora_bench.py --json $O/lora_bench_4608x4096.json > $O/lora_bench_4608x4096.txt 2>&1
cat $O/lora_bench_4608x4096.txt | grep "r5"
DALM_LORA2_ABL=4 timeout 100 python tools/lora_bench.py --only rowdot2 2>&1 | grep "r5 rowdot2" | sed 's/^/ABL4 /'
