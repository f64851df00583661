# The W > 1 code path on a 1-GPU box (DALM_FORCE_DIST=1: live RCCL communicator, bucketed gradient all-reduce, tower graphs):
# torch.distributed, the native binding, packed rows, the retriever-only step (eager at W > 1), and the plain one-rank default.
run() { echo -n "$1 $2 : "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['config'].get('gpu_max_hw_queues'))"; }
run "DALM_FORCE_DIST=1" ""
run "DALM_FORCE_DIST=1 DALM_NATIVE_COMM=1" ""
run "DALM_FORCE_DIST=1" "--data-path packed"
run "DALM_FORCE_DIST=1" "--workload cfg2"
run "X=1" ""
