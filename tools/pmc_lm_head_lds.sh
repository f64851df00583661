#!/bin/bash
# LDS-side PMC counters of the bf16 MFMA lm_head kernel (own passes): bank conflicts, LDS utilisation, wait cycles.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_lm_head_lds; rm -rf $OUT; mkdir -p $OUT
for set in "LdsBankConflict LdsUtil" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/lm_head_kernel_bench.py > $OUT/$tag.log 2>&1 || true
done
python - <<'PY'
import csv, glob, re
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmc_lm_head_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lm_head_lse_kernel" not in k: continue
        agg[int(r["Grid_Size"]) // int(r["Workgroup_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g in sorted(agg):
    print("lm_head_lse_kernel blocks=%d" % g, "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(agg[g].items())))
PY
