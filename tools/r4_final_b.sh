#!/bin/bash
# final session B of round 4: rocprofv3 summaries of the bench command, per-kernel lines, the trainer entry point
mkdir -p gpurun_out/final_r04
P=gpurun_out/final_r04
bash tools/pmc_bench.sh r04 > $P/pmc_bench.log 2>&1
cp gpurun_out/pmc_bench_r04/pmc_loss_kernels.txt $P/r04_bench_pmc_loss_kernels.txt
cp gpurun_out/pmc_bench_r04/bench_kernel_stats.txt $P/r04_bench_step_kernel_stats.txt
cp gpurun_out/pmc_bench_r04/dalm_kernels_per_shape.txt $P/r04_bench_dalm_kernels_per_shape.txt
bash tools/step_streams.sh r04 > /dev/null 2>&1
cp gpurun_out/r04_step_by_stream.txt $P/
python tools/kernel_bench.py --quick > $P/r04_kernel_bench.txt 2>/dev/null
grep -A12 "^tower\|^pool B1200\|^nf4 11008x4096 -> bf16" $P/r04_kernel_bench.txt | head -50
python bench.py --through-trainer > $P/r04_trainer_cfg3.json 2> $P/trainer_cfg3.err
tail -c 1500 $P/r04_trainer_cfg3.json
head -30 $P/r04_bench_dalm_kernels_per_shape.txt
