#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_lm_head_backward_gpu.py -q -m gpu -p no:cacheprovider > $O/lm_head_bwd_tests.log 2>&1; tail -40 $O/lm_head_bwd_tests.log
timeout 300 python tools/lm_head_train_bench.py --json $O/lm_head_train_bench.json > $O/lm_head_train_bench.txt 2>&1; cat $O/lm_head_train_bench.txt | tail -5
timeout 300 python tools/lm_head_train_bench.py --tuned --json $O/lm_head_train_bench_tuned.json > $O/lm_head_train_bench_tuned.txt 2>&1; cat $O/lm_head_train_bench_tuned.txt | tail -5
