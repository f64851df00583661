#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for a in 0 1 2 4 8 5 6 7 15; do
  echo "## DALM_LORA2_ABL=$a"
  DALM_LORA2_ABL=$a timeout 100 python tools/lora_bench.py --only rowdot2 2>&1 | grep "r5 rowdot2"
done > $O/lora_rowdot_abl.txt 2>&1
cat $O/lora_rowdot_abl.txt
