#!/bin/bash
# MFMA utilisation of the f32-MFMA similarity kernel from PMC counters (own pass, kernel-trace only alongside).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_sim
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/kernel_bench.py --only sim --sizes ${SIM_SIZES:-4096,16384} > $OUT/trace.log 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $OUT/mfma -- python tools/kernel_bench.py --only sim --sizes ${SIM_SIZES:-4096,16384} > $OUT/mfma.log 2>&1 || true
rocprofv3 --pmc MfmaUtil --output-format csv -d $OUT/util -- python tools/kernel_bench.py --only sim --sizes ${SIM_SIZES:-4096,16384} > $OUT/util.log 2>&1 || true
python - <<'PY'
import csv, glob, re
from collections import defaultdict
def load(pat):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if not any(t in k for t in ("gemm_f32_mfma", "sim_flash", "sim_rowstats_stream", "lm_head_lse4w", "split3_bf16")): continue
            k = re.sub(r"\(.*", "", k).replace("void ", "").replace("dalm::", "")
            grid = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            agg[(k, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
for name, pat in (("mfma", "gpurun_out/pmc_sim/mfma/**/*counter_collection.csv"), ("util", "gpurun_out/pmc_sim/util/**/*counter_collection.csv")):
    agg = load(pat)
    print("==", name)
    for k in sorted(agg, key=lambda x: x[1]):
        c = agg[k]
        line = f"{k[0]} blocks={k[1]} n={len(next(iter(c.values())))} " + " ".join(f"{cn}={sum(v)/len(v):.4g}" for cn, v in c.items())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"]); act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
            line += f"  MFMA busy/(GUI_ACTIVE*1024 SIMDs)={busy/(act*1024):.3f}"
        print(line)
PY
t=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py "$t" "gemm_f32|flash|rowstats|splitk|lm_head|split3" 40 > $OUT/per_shape.txt; cat $OUT/per_shape.txt
find $OUT -name "*kernel_trace.csv" -delete
