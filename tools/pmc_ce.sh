#!/bin/bash
# HBM traffic of the CE kernel from PMC counters (separate passes, no tracing domains mixed in):
#   FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2 TCC slots) cannot share a pass.
# Usage (on the GPU box, from the repo root): bash tools/pmc_ce.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_ce
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/kernel_bench.py --only ce > $OUT/trace.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python tools/kernel_bench.py --only ce > $OUT/fetch.log 2>&1 || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python tools/kernel_bench.py --only ce > $OUT/write.log 2>&1 || true
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
# keep the merge small
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -exec sh -c 'head -2000 "$1" > "$1.head"; rm "$1"' _ {} \;
