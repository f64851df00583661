"""Condense a rocprofv3 `*_kernel_stats.csv` into a short, committed summary (profiles/)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"void (dalm::)?(\w+)<", name)
    if name.startswith("Cijk_"):
        mt = re.search(r"MT(\d+x\d+x\d+)", name)
        return "hipBLASLt " + name[:14] + (" MT" + mt.group(1) if mt else "")
    if "at::native::" in name:
        m2 = re.search(r"at::native::(\w+)", name)
        f = re.findall(r"(\w+Functor|\w+_kernel_cuda|\w+_kernel_impl|\w+_kernel)\b", name)
        tag = f[1] if len(f) > 1 else (f[0] if f else "")
        dt = "bf16" if "BFloat16" in name else ("f32" if "float" in name else "")
        return f"aten::{m2.group(1) if m2 else '?'}[{tag}:{dt}]"[:90]
    return name[:90]


def main(path, top=40):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    print(f"# source: {path}")
    print(f"# total kernel time {total/1e6:.2f} ms over {sum(v[0] for v in agg.values())} launches")
    print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:92s} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f} {100*t/total:6.2f}")
    print("# --- dalm kernels ---")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if any(t in k for t in ("dalm", "marg_ce", "gemm_f32_mfma", "pool_", "rowstats", "l2norm", "ce_", "contrastive", "small_", "flash", "rag_loss", "transpose_pad")):
            print(f"{k:92s} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f} {100*t/total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
