"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the dalm kernels, PER SHAPE.

Launches are grouped by (kernel instantiation, grid).  Two problem sizes CAN share both (marg_ce_bwd_kernel<bf16_t,256> runs
grid 4608 at V = 32000 and at V = 65024 - VERDICT r2), and the counter CSV carries no kernel arguments: tools/pmc_ce.sh therefore
profiles ONE vocabulary size per process and passes it here as --label.  FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE under-reports wide
(16 B/lane) coalesced streaming reads by exactly 2x (MI355X_MICROARCH.md, HBM section): the corrected read bytes
are 2 * FETCH_SIZE * 1024.  WRITE_SIZE is uncalibrated; it is reported raw.

    python tools/pmc_summary.py <dir with fetch/ and write/> [--json profiles/roofline_traffic.json
                                                              --workload cfg3 --dtype bf16 --source "..."]
With --json the dominant loss-path kernel (largest bytes per launch) is written out as the `traffic` record
bench.py quotes for that workload/dtype.
"""
import argparse
import csv
import glob
import json
import re
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name).replace("dalm::", "")
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*(\[.*\])?$", "", name)


def load(pattern, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            if "dalm" not in r["Kernel_Name"]:
                continue
            blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            a = agg[(short(r["Kernel_Name"]), blocks)]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--json", default=None)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--source", default=None)
    ap.add_argument("--label", default=None, help="printed with every line (e.g. V=32000: the counter CSV has no kernel arguments, "
                                                  "so shapes that share an instantiation AND a grid are profiled in separate processes)")
    a = ap.parse_args()
    fetch = load(f"{a.root}/fetch/**/*counter_collection.csv", "FETCH_SIZE")
    write = load(f"{a.root}/write/**/*counter_collection.csv", "WRITE_SIZE")
    lab = f"{a.label} | " if a.label else ""
    print(("shape | " if a.label else "") + "kernel | blocks | launches | FETCH_SIZE KB/launch (raw) | read bytes/launch (x2 gfx950) | WRITE_SIZE KB/launch (raw) | total bytes/launch")
    best = None
    for k in sorted(set(fetch) | set(write), key=lambda x: (x[0], x[1])):
        fc, fv = fetch.get(k, [0, 0.0])
        wc, wv = write.get(k, [0, 0.0])
        f1 = fv / fc if fc else 0.0
        w1 = wv / wc if wc else 0.0
        total = 2 * f1 * 1024 + w1 * 1024
        print(f"{lab}{k[0]} | {k[1]} | {fc or wc} | {f1:.1f} | {2 * f1 * 1024:.4g} | {w1:.1f} | {total:.4g}")
        if "marg_ce" in k[0] and (best is None or total > best[1]):
            best = (k, total, 2 * f1 * 1024, w1 * 1024)
    if a.json and best:
        rec = {"workload": a.workload, "dtype": a.dtype, "kernel": best[0][0], "blocks": best[0][1],
               "bench_marg_ce_bytes_per_launch": round(best[1]), "read_bytes": round(best[2]), "written_bytes": round(best[3]),
               "source": a.source or f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of bench.py ({a.root}); FETCH doubled (gfx950)"}
        with open(a.json, "w") as f:
            json.dump(rec, f, indent=1)
        print("wrote", a.json, rec)


if __name__ == "__main__":
    main()
