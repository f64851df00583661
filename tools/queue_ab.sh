# A/B of launch structures on ONE box: tower graphs + eager loss / optimizer (bench.py's default since round 6) vs the whole step as
# ONE hipGraph (--whole-step-graph, the default of rounds 2-5) vs the tower graphs with a one-rank RCCL communicator (DALM_FORCE_DIST=1)
run() { echo -n "$1 $2 : "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['config'].get('launch'), d.get('roofline',{}).get('frac'))"; }
run "X=1" ""
run "X=1" "--whole-step-graph"
run "DALM_FORCE_DIST=1" ""
run "X=1" "--data-path packed"
run "X=1" "--data-path packed --whole-step-graph"
run "X=1" "--workload cfg5"
run "X=1" "--workload cfg5 --whole-step-graph"
run "X=1" "--workload cfg5 --data-path packed"
