#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_frozen_linear_gpu.py -q -m gpu -p no:cacheprovider > $O/frozen_linear_tests.log 2>&1; tail -5 $O/frozen_linear_tests.log
rm -f $O/dgrad_step_ab.txt
for v in "DALM_DGRAD_T=1" "DALM_DGRAD_T=0" "DALM_DGRAD_T=1"; do
  echo "## $v" >> $O/dgrad_step_ab.txt
  env $v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pmc 2> $O/bench_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  ', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')" >> $O/dgrad_step_ab.txt 2>&1
  tail -2 $O/bench_ab.err | grep -v amdgpu >> $O/dgrad_step_ab.txt
done
cat $O/dgrad_step_ab.txt
(time timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider -x) > $O/gpu_suite.log 2>&1; tail -12 $O/gpu_suite.log
