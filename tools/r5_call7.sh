#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/gemm_layout_probe.py > $O/gemm_layout_probe.txt 2>&1; cat $O/gemm_layout_probe.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_headline_config_gpu.py -q -m gpu -p no:cacheprovider -x > $O/headline_tests.log 2>&1; tail -15 $O/headline_tests.log
cat gpurun_out/headline_parity.json | python -c "import sys,json; d=json.load(sys.stdin); [print(k, json.dumps({kk: v[kk] for kk in v if kk.startswith('rel')})) for k,v in d.items()]"
