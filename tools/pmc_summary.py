"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes for the dalm kernels.

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE under-reports wide (16 B/lane) coalesced
streaming reads by exactly 2x (MI355X_MICROARCH.md, HBM section): the corrected read bytes are
2 * FETCH_SIZE * 1024.  WRITE_SIZE is uncalibrated; it is reported raw.
"""
import csv
import glob
import sys
from collections import defaultdict


def load(pattern, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            if "dalm" not in k:
                continue
            k = k.split("(")[0].replace("void ", "").replace("dalm::", "")
            a = agg[k]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main(root):
    fetch = load(f"{root}/fetch/**/*counter_collection.csv", "FETCH_SIZE")
    write = load(f"{root}/write/**/*counter_collection.csv", "WRITE_SIZE")
    print("kernel | launches | FETCH_SIZE KB/launch (raw) | read bytes/launch (x2 gfx950 correction) | WRITE_SIZE KB/launch (raw)")
    for k in sorted(set(fetch) | set(write)):
        fc, fv = fetch.get(k, [0, 0.0])
        wc, wv = write.get(k, [0, 0.0])
        f1 = fv / fc if fc else float("nan")
        w1 = wv / wc if wc else float("nan")
        print(f"{k} | {fc or wc} | {f1:.1f} | {2 * f1 * 1024:.4g} | {w1:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
