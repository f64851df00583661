#!/bin/bash
# round 4, first GPU call: the GPU suite (incl. the new trainer test), the CE row-order / fill experiment, the trainer
# bench at reduced depth (debug) and, when that works, at full cfg3 size next to a default bench line of the same box
mkdir -p gpurun_out/r04
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04/gpu_suite.log 2>&1
tail -5 gpurun_out/r04/gpu_suite.log
timeout 500 python tools/ce_row_order_probe.py --sweep --iters 40 > gpurun_out/r04/ce_row_order.txt 2>&1
cat gpurun_out/r04/ce_row_order.txt
timeout 300 python bench.py --through-trainer --trainer-rows 1500 --retriever-layers 2 --generator-layers 2 \
    > gpurun_out/r04/trainer_debug.json 2> gpurun_out/r04/trainer_debug.err
rc=$?
tail -c 1500 gpurun_out/r04/trainer_debug.json; echo "trainer debug rc=$rc"
if [ $rc -ne 0 ]; then tail -30 gpurun_out/r04/trainer_debug.err; exit 0; fi
timeout 300 python bench.py --steps 20 --warmup 3 --no-pmc > gpurun_out/r04/bench_default.json 2> gpurun_out/r04/bench_default.err
tail -c 600 gpurun_out/r04/bench_default.json
timeout 600 python bench.py --through-trainer --bench-line gpurun_out/r04/bench_default.json \
    > gpurun_out/r04/trainer_cfg3.json 2> gpurun_out/r04/trainer_cfg3.err
echo "trainer cfg3 rc=$?"; tail -c 2500 gpurun_out/r04/trainer_cfg3.json; tail -5 gpurun_out/r04/trainer_cfg3.err
