# kernel-trace view of launch structures (rocprofv3 distorts absolute times; read the gap counts): bash tools/graph_vs_towers_gaps.sh [bench args]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo; export TMPDIR=/tmp
ARGS="${*:-}"
for mode in whole towers; do
  extra="--whole-step-graph"; [ $mode = towers ] && extra="--graph-towers"
  rm -rf /tmp/gp_$mode; (cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$mode -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-pmc $ARGS $extra > /tmp/gp_$mode.log 2>&1)
  f=$(find /tmp/gp_$mode -name "*kernel_trace.csv" | head -1)
  python tools/graph_step_gaps.py $f --steps 4 --skip-last 1 > gpurun_out/gaps_$mode.txt 2>&1
  head -14 gpurun_out/gaps_$mode.txt
done
