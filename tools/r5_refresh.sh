#!/bin/bash
# Round 5: every measured line under profiles/r05_* from ONE tree.  Run on the GPU box from the repo root:
#   bash tools/r5_refresh.sh [part]        part = bench | trainer | prof | kernels | all (default)
# Output: gpurun_out/profiles_r05/ ; copy into profiles/ afterwards.
PART=${1:-all}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
P=gpurun_out/profiles_r05; mkdir -p $P
date -u +"# %Y-%m-%dT%H:%M:%SZ tree $(cat .tree_id 2>/dev/null)" >> $P/r05_refresh_stamp.txt
if [ "$PART" = bench ] || [ "$PART" = all ]; then
  python bench.py > $P/r05_bench_default.json 2> $P/bench.err
  python bench.py --workload cfg5 --no-cpu-baseline > $P/r05_bench_cfg5.json 2>> $P/bench.err
  python bench.py --workload cfg2 --no-cpu-baseline > $P/r05_bench_cfg2.json 2>> $P/bench.err
  python bench.py --workload cfg1 --no-cpu-baseline > $P/r05_bench_cfg1.json 2>> $P/bench.err
  python bench.py --dtype fp32 --no-cpu-baseline > $P/r05_bench_cfg3_fp32.json 2>> $P/bench.err
  python bench.py --steps 20 --warmup 5 --data-path loader --no-cpu-baseline --no-pmc > $P/r05_bench_cfg3_loader.json 2>> $P/bench.err
  python bench.py --steps 30 --warmup 6 --fuse-lm-head --no-cpu-baseline > $P/r05_bench_cfg3_fuse_lm_head.json 2>> $P/bench.err
  DALM_LM_HEAD_TRAIN_KERNEL=0 python bench.py --steps 30 --warmup 6 --fuse-lm-head --no-cpu-baseline > $P/r05_bench_cfg3_fuse_lm_head_library_path.json 2>> $P/bench.err
  python bench.py --data-path bucketed --steps 24 --warmup 12 --no-cpu-baseline > $P/r05_bench_cfg3_bucketed_trimmed.json 2>> $P/bench.err
  for f in $P/r05_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[1].split('/')[-1]:52s} {d['value']:9.2f} {d['unit']:16s} {d['ms_per_step']:8.2f} ms/step  roofline.frac {d.get('roofline', {}).get('frac')}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
  done | tee $P/r05_bench_lines.txt
fi
if [ "$PART" = trainer ] || [ "$PART" = all ]; then
  for w in cfg3 cfg2 cfg5; do
    python bench.py --workload $w --through-trainer --bench-line $P/r05_bench_$( [ $w = cfg3 ] && echo default || echo $w ).json > $P/r05_trainer_$w.json 2> $P/trainer_$w.err
    tail -1 $P/r05_trainer_$w.json | cut -c1-400
  done
fi
if [ "$PART" = prof ] || [ "$PART" = all ]; then
  bash tools/pmc_bench.sh r05 > $P/pmc_bench.log 2>&1
  cp gpurun_out/pmc_bench_r05/pmc_loss_kernels.txt $P/r05_bench_pmc_loss_kernels.txt
  cp gpurun_out/pmc_bench_r05/bench_kernel_stats.txt $P/r05_bench_step_kernel_stats.txt
  cp gpurun_out/pmc_bench_r05/dalm_kernels_per_shape.txt $P/r05_bench_dalm_kernels_per_shape.txt
  python tools/pmc_summary.py gpurun_out/pmc_bench_r05 --json $P/roofline_traffic.json --workload cfg3 --dtype bf16 \
    --source "profiles/r05_bench_pmc_loss_kernels.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py, real masks; FETCH doubled per the gfx950 guide)" > /dev/null
  bash tools/step_streams.sh r05 > /dev/null 2>&1; cp gpurun_out/r05_step_by_stream.txt $P/r05_step_by_stream.txt
  bash tools/step_streams.sh r05cfg2 "--workload cfg2" "small_grad_kernel|sim_small|small_" > /dev/null 2>&1; cp gpurun_out/r05cfg2_step_by_stream.txt $P/r05cfg2_step_by_stream.txt 2>/dev/null
  bash tools/step_streams.sh r05cfg5 "--workload cfg5" > /dev/null 2>&1; cp gpurun_out/r05cfg5_step_by_stream.txt $P/r05cfg5_step_by_stream.txt 2>/dev/null
fi
if [ "$PART" = kernels ] || [ "$PART" = all ]; then
  python tools/kernel_bench.py > $P/r05_kernel_bench.txt 2>/dev/null
  python tools/lora_bench.py --json $P/r05_lora_bench_4608x4096.json > $P/r05_lora_bench_4608x4096.txt 2>/dev/null
  python tools/lora_bench.py --rows 19200 --cols 1024 > $P/r05_lora_bench_19200x1024.txt 2>/dev/null
  python tools/lm_head_train_bench.py --json $P/r05_lm_head_train_bench.json > $P/r05_lm_head_train_kernels_vs_library.txt 2>/dev/null
  python tools/lm_head_train_bench.py --tuned > $P/r05_lm_head_train_kernels_vs_library_tuned_table.txt 2>/dev/null
  python tools/attn_bench.py > $P/r05_attn_bench.txt 2>/dev/null
  python tools/attn_bench.py --lens full >> $P/r05_attn_bench.txt 2>/dev/null
  bash tools/attn_prof.sh > /dev/null 2>&1; cp gpurun_out/attn_kernels.txt $P/r05_attn_kernels_per_launch.txt
  bash tools/attn_pmc.sh > /dev/null 2>&1; cp gpurun_out/attn_pmc.txt $P/r05_attn_pmc.txt
  python tools/sim_grad_x3_bench.py > $P/r05_sim_grad_x3.txt 2>/dev/null
  rm -rf $P/lt; (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$P/lt -- python $GRAFT_REPO_ROOT/tools/lora_bench.py --iters 10 > /dev/null 2>&1)
  python tools/summarize_trace.py "$(find $P/lt -name '*kernel_trace.csv' | head -1)" "lora" 40 > $P/r05_lora_kernels_per_shape.txt; rm -rf $P/lt
fi
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out -name "*counter_collection.csv" -size +4M -delete
ls $P
