#!/bin/bash
# round 4, second GPU call: full GPU suite (bf16x3 similarity, k-context loss end to end), similarity bench f32 vs bf16x3,
# the trainer entry points at cfg5 (Falcon architecture) and cfg2 (retriever-only) size
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04/gpu_suite2.log 2>&1
tail -15 gpurun_out/r04/gpu_suite2.log
timeout 600 python tools/kernel_bench.py --only sim --sizes 1536,4096,8192,16384 > gpurun_out/r04/sim_bf16x3.txt 2>&1
cat gpurun_out/r04/sim_bf16x3.txt | head -60
timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 3 > gpurun_out/r04/bench_cfg2.json 2> gpurun_out/r04/bench_cfg2.err
timeout 400 python bench.py --workload cfg2 --through-trainer --bench-line gpurun_out/r04/bench_cfg2.json \
    > gpurun_out/r04/trainer_cfg2.json 2> gpurun_out/r04/trainer_cfg2.err
echo "trainer cfg2 rc=$?"; tail -c 1800 gpurun_out/r04/trainer_cfg2.json; tail -3 gpurun_out/r04/trainer_cfg2.err
timeout 400 python bench.py --workload cfg5 --steps 20 --warmup 3 --no-pmc --no-cpu-baseline > gpurun_out/r04/bench_cfg5.json 2> gpurun_out/r04/bench_cfg5.err
timeout 900 python bench.py --workload cfg5 --through-trainer --bench-line gpurun_out/r04/bench_cfg5.json \
    > gpurun_out/r04/trainer_cfg5.json 2> gpurun_out/r04/trainer_cfg5.err
echo "trainer cfg5 rc=$?"; tail -c 1200 gpurun_out/r04/trainer_cfg5.json; tail -3 gpurun_out/r04/trainer_cfg5.err
