for gh in 14 7 4 2; do echo "GH=$gh"; DALM_LM_HEAD_GH=$gh timeout 300 python tools/_r3_ab.py 2 2>&1 | head -2; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for gh in 14 7; do DALM_LM_HEAD_GH=$gh DALM_LM_HEAD_PIECES=2 PMC_MATCH="lm_head_lse4w" timeout 300 python tools/pmc_run.py gpurun_out/pmc_gh "TCC_HIT_sum TCC_MISS_sum" "MfmaUtil" -- python tools/lm_head_ablate.py 2 2>&1 | tail -1; done
DALM_LM_HEAD_PIECES=2 timeout 300 python -m pytest tests/test_kernels_r2_gpu.py -x -q -k "lm_head" 2>&1 | tail -1
