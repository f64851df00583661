#!/bin/bash
# HBM traffic of the CE kernels from PMC counters (separate passes, no tracing domains mixed in):
#   FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2 TCC slots) cannot share a pass.
# One vocabulary size per process: launches of one instantiation with the same grid (B*Tg rows) are indistinguishable in
# the counter CSV, so V = 32000 and V = 65024 must never share a run.
# Usage (on the GPU box, from the repo root): bash tools/pmc_ce.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ce
rm -rf $OUT; mkdir -p $OUT
: > $OUT/summary.txt
for V in 32000 65024; do
  D=$OUT/V$V
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/fetch -- python tools/kernel_bench.py --only ce --vocab $V > $D.fetch.log 2>&1 || true
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/write -- python tools/kernel_bench.py --only ce --vocab $V > $D.write.log 2>&1 || true
  python tools/pmc_summary.py $D --label "B18 Tg256 V$V" >> $OUT/summary.txt 2>&1 || true
done
cat $OUT/summary.txt
find $OUT -name "*.csv" -delete
