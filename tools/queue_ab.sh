# A/B on ONE box: launch structures and hardware-queue settings.
#   tower graphs + eager loss / optimizer (bench.py's default since round 6) vs the whole step as ONE hipGraph (--whole-step-graph)
#   a one-rank RCCL communicator (DALM_FORCE_DIST=1: the W > 1 code path) with 3 / 4 hardware queues, with and without claiming the
#   compute streams' queues before the communicator exists (DALM_CLAIM_QUEUES)
run() { echo -n "$1 $2 : "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['config'].get('launch'), d['config'].get('hw_queues'))"; }
run "X=1" ""
run "X=1" "--whole-step-graph"
run "DALM_FORCE_DIST=1" ""
run "DALM_FORCE_DIST=1 DALM_HW_QUEUES=0" ""
run "DALM_FORCE_DIST=1 DALM_HW_QUEUES=0 DALM_CLAIM_QUEUES=0" ""
run "DALM_FORCE_DIST=1 DALM_CLAIM_QUEUES=0" ""
run "DALM_FORCE_DIST=1 DALM_HW_QUEUES=0 DALM_NATIVE_COMM=1" ""
