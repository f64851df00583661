#!/bin/bash
# SQ / TCC counters of the attention backward kernels (tools/attn_bench.py) -> gpurun_out/attn_pmc.txt
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/atp; rm -rf $O; mkdir -p $O
(cd /tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 3 "$@" > $O/sq.log 2>&1)
(cd /tmp; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 3 "$@" > $O/fetch.log 2>&1)
(cd /tmp; rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 3 "$@" > $O/write.log 2>&1)
python - <<'PY' > gpurun_out/attn_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/atp/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" not in k and "bwd_kernel" not in k:
            continue
        name = next((n for n in ("attn_bwd_dq", "attn_bwd_dkdv", "attn_fwd_kernel", "attn_fwd2_kernel", "attn_mask_bits") if n in k), k.split("(")[0][-40:])
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}   (n={len(v)})")
PY
cat gpurun_out/attn_pmc.txt; tail -3 $O/sq.log
rm -rf $O
