#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
(time timeout 1800 python -m pytest tests -q -m gpu -p no:cacheprovider) > $O/gpu_suite.log 2>&1; tail -15 $O/gpu_suite.log
