#!/bin/bash
# round 4, seventh GPU call: the bench line with the back-to-back CE probe, smoke(), the full GPU suite on the final tree
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 600 python bench.py > gpurun_out/r04/bench_default_final.json 2> gpurun_out/r04/bench_default_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_default_final.json") if l.startswith("{")][-1])
r = d["roofline"]
print("value", d["value"], "ms", d["ms_per_step"], "frac", r["frac"], "avg_launch_us", r["avg_launch_us"], "b2b_us", r["back_to_back_us"],
      "b2b_frac", r["back_to_back_frac"], "traffic", r["traffic"], "loss_path_us", r["loss_path_us"])
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r04/gpu_suite7.log 2>&1
tail -5 gpurun_out/r04/gpu_suite7.log
