"""Per-SHAPE kernel durations from a rocprofv3 `*_kernel_trace.csv`: launches are grouped by
(kernel name, grid, workgroup) so that one template instantiation run at several problem sizes is not
averaged into one meaningless number.

    python tools/summarize_trace.py <kernel_trace.csv> [name-filter-regex] [top]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name).replace("dalm::", "")
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*(\[.*\])?$", "", name)      # drop the parameter list
    return name[:110]


def main(path, flt=None, top=60):
    agg = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = short(r["Kernel_Name"])
            if flt and not re.search(flt, name):
                continue
            grid = tuple(int(r.get(f"Grid_Size_{a}", r.get(f"Grid_Size{a}", 1)) or 1) for a in "XYZ")
            wg = tuple(int(r.get(f"Workgroup_Size_{a}", r.get(f"Workgroup_Size{a}", 1)) or 1) for a in "XYZ")
            blocks = tuple(g // max(w, 1) for g, w in zip(grid, wg))
            agg[(name, blocks, wg[0] * wg[1] * wg[2], r.get("VGPR_Count", "?"))].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"# source: {path}")
    print(f"{'kernel':112s} {'blocks':>16s} {'thr':>5s} {'vgpr':>5s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'med_us':>9s}")
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]
    for (name, blocks, thr, vg), d in rows:
        d.sort()
        b = "x".join(str(x) for x in blocks)
        print(f"{name:112s} {b:>16s} {thr:5d} {vg:>5s} {len(d):6d} {sum(d)/len(d)/1e3:9.2f} {d[0]/1e3:9.2f} {d[len(d)//2]/1e3:9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None, int(sys.argv[3]) if len(sys.argv) > 3 else 60)
