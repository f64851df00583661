timeout 300 python -m pytest tests/test_kernels_r2_gpu.py -x -q -k "lm_head" 2>&1 | tail -3
for v in 0 1 2 3; do echo "PIECES=$v"; DALM_LM_HEAD_PIECES=$v timeout 200 python tools/lm_head_kernel_bench.py 2>&1 | grep -E "dalm_lm_head|hipBLASLt" ; done
