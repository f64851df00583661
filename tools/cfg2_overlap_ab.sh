# cfg2 / cfg1 (retriever-only step) launch structures on ONE box: encoder-call graphs + eager loss / optimizer (--graph-towers, padded
# batches) vs the whole step as one two-stream hipGraph vs one stream
run() { echo -n "$1 $2 : "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],2), round(d['ms_per_step'],2), d['config'].get('launch'))"; }
run "X=1" "--workload cfg2 --graph-towers"
run "X=1" "--workload cfg2"
run "X=1" "--workload cfg2 --no-overlap"
run "X=1" "--workload cfg2 --data-path packed"
run "X=1" "--workload cfg1 --graph-towers"
run "X=1" "--workload cfg1"
