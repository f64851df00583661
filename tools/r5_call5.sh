#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_lora2_gpu.py tests/test_lora_ops_gpu.py -q -m gpu -p no:cacheprovider > $O/lora2_tests.log 2>&1; tail -4 $O/lora2_tests.log
timeout 200 python tools/lora_bench.py --json $O/lora_bench_4608x4096.json > $O/lora_bench_4608x4096.txt 2>&1
timeout 200 python tools/lora_bench.py --rows 19200 --cols 1024 --json $O/lora_bench_19200x1024.json > $O/lora_bench_19200x1024.txt 2>&1
grep "r5" $O/lora_bench_4608x4096.txt
rm -f $O/lora_step_ab.txt
for v in "DALM_LORA_V2=1" "DALM_LORA_V2=0" "DALM_LORA_V2=1"; do
  echo "## $v" >> $O/lora_step_ab.txt
  env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc 2> $O/bench_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')" >> $O/lora_step_ab.txt 2>&1
done
cat $O/lora_step_ab.txt
