#!/bin/bash
# rocprofv3 kernel durations + MfmaUtil (own PMC pass) of the bf16 MFMA lm_head kernel and of the library path beside it.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lm_head
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/lm_head_kernel_bench.py > $OUT/trace.log 2>&1 || true
rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/util -- python tools/lm_head_kernel_bench.py > $OUT/util.log 2>&1 || true
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py "$t" "lm_head|Cijk|marg_ce" 12 > $OUT/lm_head_kernel_per_shape.txt
python - >> $OUT/lm_head_kernel_per_shape.txt <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(list)
for f in glob.glob("gpurun_out/pmc_lm_head/util/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        if not any(t in k for t in ("lm_head_lse_kernel", "Cijk")): continue
        k = re.sub(r"\(.*", "", k).replace("void ", "").replace("dalm::", "")[:70]
        agg[(k, int(r["Grid_Size"]) // int(r["Workgroup_Size"]))].append(float(r["Counter_Value"]))
print("\n# MfmaUtil (rocprofv3 --pmc MfmaUtil, own pass), mean over launches")
for (k, g), v in sorted(agg.items(), key=lambda kv: kv[0][1]):
    print(f"{k:72s} blocks={g:6d} n={len(v):3d} MfmaUtil={sum(v)/len(v):6.2f} %")
PY
grep "us " $OUT/trace.log >> $OUT/lm_head_kernel_per_shape.txt || true
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/lm_head_kernel_per_shape.txt
