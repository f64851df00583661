"""Run a command under rocprofv3 --pmc, one pass per counter group, and print per-kernel means keyed on (kernel, grid).
    python tools/pmc_run.py OUTDIR "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "MfmaUtil" -- python tools/lm_head_ablate.py 0
Kernel names are matched against $PMC_MATCH (regex, default: everything)."""
import csv
import glob
import os
import re
import subprocess
import sys
from collections import defaultdict

args = sys.argv[1:]
cut = args.index("--")
out, groups, cmd = args[0], args[1:cut], args[cut + 1:]
match = re.compile(os.environ.get("PMC_MATCH", "."))
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
agg = defaultdict(lambda: defaultdict(list))
for i, g in enumerate(groups):
    d = os.path.join(out, f"pass{i}")
    subprocess.run(["rocprofv3", "--pmc", *g.split(), "--output-format", "csv", "-d", d, "--"] + cmd, env=env,
                   stdout=open(os.path.join(out, f"pass{i}.log"), "w"), stderr=subprocess.STDOUT)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if not match.search(k):
                continue
            k = re.sub(r"\(.*", "", k).replace("void ", "").replace("dalm::", "")[:60]
            agg[(k, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        os.remove(f)
for (k, g), cs in sorted(agg.items()):
    print(f"{k:62s} blocks={g:6d}  " + "  ".join(f"{c}={sum(v)/len(v):.4g} (n={len(v)})" for c, v in sorted(cs.items())))
