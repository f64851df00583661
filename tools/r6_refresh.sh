#!/bin/bash
# Round 6: measured lines under profiles/r06_* from ONE tree.  Run on the GPU box from the repo root:
#   bash tools/r6_refresh.sh [part ...]     parts: bench packed trainer prof kernels simpmc dist all (default: all)
# Output: gpurun_out/profiles_r06/ ; copy into profiles/ afterwards.
# Every step goes through `run`: stdout -> the named file, stderr -> <file>.err (kept, never /dev/null), a non-zero exit status is
# recorded in r06_refresh_failures.txt and the script itself exits non-zero at the end (round 5's script discarded stderr and
# a crashing tools/kernel_bench.py went unnoticed - VERDICT r5 weak 2).
PARTS="${*:-all}"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
P=gpurun_out/profiles_r06; mkdir -p $P
FAIL=$P/r06_refresh_failures.txt
date -u +"# %Y-%m-%dT%H:%M:%SZ parts: $PARTS" >> $P/r06_refresh_stamp.txt
has() { [[ " $PARTS " == *" $1 "* ]] || [[ " $PARTS " == *" all "* ]]; }
run() {   # run <output file> <command ...>
  local out=$1; shift
  "$@" > "$out" 2> "$out.err"
  local rc=$?
  if [ $rc -ne 0 ]; then
    echo "rc=$rc  $*  (stderr: $out.err)" | tee -a $FAIL
    tail -5 "$out.err"
  else
    [ -s "$out.err" ] || rm -f "$out.err"
  fi
  return $rc
}
lines() {
  for f in "$@"; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(f"{sys.argv[1].split('/')[-1]:52s} {d['value']:9.2f} {d['unit']:10s} {d['ms_per_step']:8.2f} ms/step  roofline.frac {r.get('frac')}  model TF {d['config'].get('step_model_tflops')}")
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
  done
}
if has bench; then
  run $P/r06_bench_default.json python bench.py
  run $P/r06_bench_cfg3_whole_step_graph.json python bench.py --whole-step-graph --steps 20 --warmup 5 --no-cpu-baseline --no-pmc
  run $P/r06_bench_cfg3_packed_whole_step_graph.json python bench.py --data-path packed --whole-step-graph --steps 20 --warmup 5 --no-cpu-baseline --no-pmc
  run $P/r06_bench_cfg5.json python bench.py --workload cfg5 --no-cpu-baseline
  run $P/r06_bench_cfg2.json python bench.py --workload cfg2 --no-cpu-baseline
  run $P/r06_bench_cfg1.json python bench.py --workload cfg1 --no-cpu-baseline
  run $P/r06_bench_cfg3_fp32.json python bench.py --dtype fp32 --no-cpu-baseline
  run $P/r06_bench_cfg3_loader.json python bench.py --steps 20 --warmup 5 --data-path loader --no-cpu-baseline --no-pmc
  run $P/r06_bench_cfg3_bucketed_trimmed.json python bench.py --data-path bucketed --steps 24 --warmup 12 --no-cpu-baseline
fi
if has packed; then
  run $P/r06_bench_cfg3_packed.json python bench.py --data-path packed --steps 20 --warmup 4 --no-cpu-baseline
  run $P/r06_bench_cfg5_packed.json python bench.py --workload cfg5 --data-path packed --steps 20 --warmup 4 --no-cpu-baseline
  run $P/r06_bench_cfg2_packed.json python bench.py --workload cfg2 --data-path packed --no-cpu-baseline
fi
if has bench || has packed; then lines $P/r06_bench_*.json | tee $P/r06_bench_lines.txt; fi
if has trainer; then
  for w in cfg3 cfg2 cfg5; do
    run $P/r06_trainer_$w.json python bench.py --workload $w --through-trainer --bench-line $P/r06_bench_$( [ $w = cfg3 ] && echo default || echo $w ).json
    tail -1 $P/r06_trainer_$w.json | cut -c1-400
    run $P/r06_trainer_${w}_packed.json python bench.py --workload $w --through-trainer --data-path packed --bench-line $P/r06_bench_${w}_packed.json
    tail -1 $P/r06_trainer_${w}_packed.json | cut -c1-400
  done
fi
if has prof; then
  run $P/pmc_bench.log bash tools/pmc_bench.sh r06
  cp gpurun_out/pmc_bench_r06/pmc_loss_kernels.txt $P/r06_bench_pmc_loss_kernels.txt
  cp gpurun_out/pmc_bench_r06/bench_kernel_stats.txt $P/r06_bench_step_kernel_stats.txt
  cp gpurun_out/pmc_bench_r06/dalm_kernels_per_shape.txt $P/r06_bench_dalm_kernels_per_shape.txt
  run $P/pmc_summary.log python tools/pmc_summary.py gpurun_out/pmc_bench_r06 --json $P/roofline_traffic.json --workload cfg3 --dtype bf16 \
    --source "profiles/r06_bench_pmc_loss_kernels.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py, real masks; FETCH doubled per the gfx950 guide)"
  run $P/step_streams.log bash tools/step_streams.sh r06; cp gpurun_out/r06_step_by_stream.txt $P/r06_step_by_stream.txt
  run $P/step_streams_packed.log bash tools/step_streams.sh r06packed "--data-path packed"; cp gpurun_out/r06packed_step_by_stream.txt $P/r06packed_step_by_stream.txt
  run $P/step_streams_cfg2.log bash tools/step_streams.sh r06cfg2 "--workload cfg2" "small_grad_kernel|sim_small|small_"; cp gpurun_out/r06cfg2_step_by_stream.txt $P/
  run $P/step_streams_cfg5.log bash tools/step_streams.sh r06cfg5 "--workload cfg5"; cp gpurun_out/r06cfg5_step_by_stream.txt $P/
  run $P/step_streams_cfg5packed.log bash tools/step_streams.sh r06cfg5packed "--workload cfg5 --data-path packed"; cp gpurun_out/r06cfg5packed_step_by_stream.txt $P/
fi
if has kernels; then
  run $P/r06_kernel_bench.txt python tools/kernel_bench.py --quick
  cp gpurun_out/kernel_bench.json $P/r06_kernel_bench.json
  run $P/r06_lora_bench_4608x4096.txt python tools/lora_bench.py --json $P/r06_lora_bench_4608x4096.json
  run $P/r06_attn_bench.txt python tools/attn_bench.py
  DALM_ATTN_FWD=1 DALM_ATTN_DKDV=1 run $P/r06_attn_bench_first_forms.txt python tools/attn_bench.py
  DALM_ATTN_FWD=1 DALM_ATTN_DKDV=1 run $P/attn_ab_first.log python tools/attn_ab.py --out /tmp/attn_first.pt
  run $P/attn_ab_second.log python tools/attn_ab.py --out /tmp/attn_second.pt
  run $P/r06_attn_forms_ab.txt python tools/attn_ab.py --compare /tmp/attn_first.pt /tmp/attn_second.pt
  run $P/attn_prof.log bash tools/attn_prof.sh; cp gpurun_out/attn_kernels.txt $P/r06_attn_kernels.txt
  run $P/attn_pmc.log bash tools/attn_pmc.sh; cp gpurun_out/attn_pmc.txt $P/r06_attn_pmc.txt
  run $P/r06_sim_grad_x3.txt python tools/sim_grad_x3_bench.py
fi
if has simpmc; then
  run $P/r06_sim_pmc_mfma.txt bash tools/pmc_sim.sh
  cp gpurun_out/pmc_sim/per_shape.txt $P/r06_sim_pmc_per_shape.txt 2>/dev/null
fi
if has dist; then
  # the W > 1 launch mode with ONE rank (DALM_FORCE_DIST=1): torch.distributed communicator, then the library's own
  DALM_FORCE_DIST=1 run $P/r06_bench_one_rank_dist_torch.json python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pmc
  DALM_FORCE_DIST=1 DALM_NATIVE_COMM=1 run $P/r06_bench_one_rank_dist_native.json python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-pmc
  DALM_FORCE_DIST=1 run $P/r06_bench_one_rank_dist_torch_packed.json python bench.py --data-path packed --steps 20 --warmup 6 --no-cpu-baseline --no-pmc
  lines $P/r06_bench_one_rank_dist_*.json | tee $P/r06_bench_one_rank_dist_lines.txt
fi
find gpurun_out -name "*kernel_trace.csv" -size +4M -delete
find gpurun_out -name "*counter_collection.csv" -size +4M -delete
ls $P
if [ -s $FAIL ]; then echo "FAILED STEPS:"; cat $FAIL; exit 1; fi
