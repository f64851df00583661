run() { echo -n "$1 $2 : "; env $1 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-pmc $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['config'].get('launch'))"; }
run "X=1" ""
run "X=1" "--no-overlap"
run "X=1" "--data-path packed"
run "X=1" "--data-path packed --no-overlap"
run "DALM_PACK_PAIR=0" "--data-path packed"
run "DALM_PACK_PAIR=0" "--data-path packed --no-overlap"
