"""Kernel-time breakdown of the TIMED steps only, from a rocprofv3 `*_kernel_trace.csv` of bench.py.

rocprofv3's own `*_kernel_stats.csv` aggregates the whole process: model initialisation (normal_ / uniform_ kernels, hundreds
of launches), warm-up steps and the post-run probes dilute every percentage.  Here the window is cut out of the
trace itself: step boundaries are the fused-Adam launches (`multi_tensor_apply`) that end every step - the window runs from
the end of the last warm-up step's optimizer kernels to the end of the last timed step's, located through the launches of the
marginalised-CE kernel (one per step).

    python tools/summarize_step_window.py <kernel_trace.csv> --warmup 2 --steps 5 [--top 45]
"""
import argparse
import csv
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_rocprof import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--warmup", type=int, required=True)
    ap.add_argument("--steps", type=int, required=True)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--by-stream", action="store_true",
                    help="also: kernel time per HIP stream / queue inside the window, its busiest kernels, and the small launches "
                         "(< 8 us) by name and grid - which stream is the long pole of the step, and what fills it")
    ap.add_argument("--marker", default="marg_ce_row|marg_ce_stream",
                    help="regex of a kernel launched exactly once per step (retriever-only steps: small_grad_kernel)")
    ap.add_argument("--sequence", type=int, default=0, help="with --by-stream: print this many consecutive launches of the busiest stream, forward and backward")
    a = ap.parse_args()
    rows, lanes = [], {}
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
            lane = r.get("Stream_Id") or r.get("Queue_Id") or "?"
            grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
            lanes[(rows[-1][0], rows[-1][1])] = (lane, grid)
    rows.sort()
    import re

    mark = re.compile(a.marker)
    ce = [s for s, e, n in rows if mark.search(n)]
    adam = [(s, e) for s, e, n in rows if "multi_tensor_apply" in n]
    need = a.warmup + a.steps
    if len(ce) < need:
        sys.exit(f"only {len(ce)} CE launches in the trace, expected >= {need}")
    ce = ce[:need]          # later launches belong to post-run probes

    def step_end(i):        # end of the last optimizer kernel between CE launch i and CE launch i + 1 (or the next 2 s)
        lo, hi = ce[i], (ce[i + 1] if i + 1 < len(ce) else ce[i] + 2_000_000_000)
        ends = [e for s, e in adam if lo < s < hi]
        return max(ends) if ends else hi

    t0 = step_end(a.warmup - 1) if a.warmup > 0 else rows[0][0]
    t1 = step_end(need - 1)
    agg, total, launches = {}, 0.0, 0
    for s, e, n in rows:
        if s < t0 or e > t1:
            continue
        k = short(n)
        v = agg.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += e - s
        total += e - s
        launches += 1
    span = t1 - t0
    print(f"# source: {a.trace}")
    print(f"# window: the {a.steps} timed steps only (after {a.warmup} warm-up steps; initialisation and post-run probes excluded)")
    print(f"# wall span {span/1e6:.2f} ms = {span/1e6/a.steps:.2f} ms per step; kernel time {total/1e6:.2f} ms "
          f"({100*total/span:.1f} % of the span: >100 % means kernels overlap on several streams) over {launches} launches "
          f"= {launches // a.steps} per step")
    print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f"{k:92s} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f} {100*t/total:6.2f}")
    groups = {"hipBLASLt / rocBLAS GEMM": 0.0, "aten elementwise / reduce / copy": 0.0, "attention (torch SDPA)": 0.0,
              "dalm_* (this library)": 0.0, "optimizer": 0.0, "other": 0.0}
    for k, (c, t) in agg.items():
        if k.startswith("hipBLASLt") or "Cijk" in k or "gemm" in k.lower() and "dalm" not in k and "mfma" not in k:
            groups["hipBLASLt / rocBLAS GEMM"] += t
        elif "multi_tensor_apply" in k:
            groups["optimizer"] += t
        elif "dalm::" not in k and any(x in k for x in ("attention", "attn", "fmha", "flash_fwd", "flash_bwd", "sdpa", "bwd_kernel_dk_dv", "bwd_kernel_dq",
                                  "bwd_kernel_fuse", "bwd_preprocess")):
            groups["attention (torch SDPA)"] += t
        elif "dalm::" in k or any(x in k for x in ("marg_ce", "pool_", "small_", "rag_loss", "ce_prep", "ce_finalize", "rms_norm", "nf4", "l2norm",
                                  "sim_", "scale_inplace", "contrastive")):
            groups["dalm_* (this library)"] += t
        elif k.startswith("aten::") or "at::native" in k or "elementwise" in k:
            groups["aten elementwise / reduce / copy"] += t
        else:
            groups["other"] += t
    print("# --- by family ---")
    for g, t in sorted(groups.items(), key=lambda kv: -kv[1]):
        print(f"# {g:40s} {t/1e6:10.3f} ms {100*t/total:6.2f} %")
    if not a.by_stream:
        return
    per = {}
    for s, e, n in rows:
        if s < t0 or e > t1:
            continue
        lane, grid = lanes[(s, e)]
        d = per.setdefault(lane, {"t": 0.0, "n": 0, "k": {}, "small": {}})
        d["t"] += e - s
        d["n"] += 1
        v = d["k"].setdefault(short(n), [0, 0.0])
        v[0] += 1
        v[1] += e - s
        if e - s < 8000:
            w = d["small"].setdefault((short(n), grid), [0, 0.0])
            w[0] += 1
            w[1] += e - s
    if a.sequence:
        # the launches of the busiest stream around the middle of the first timed step's forward and backward: what runs
        # between two GEMMs of a layer, in order
        busiest = max(per.items(), key=lambda kv: kv[1]["t"])[0]
        seq = [(s, e, n) for s, e, n in rows if t0 <= s and e <= t1 and lanes[(s, e)][0] == busiest]
        per_step = len(seq) // a.steps
        for label, frac in (("forward", 0.15), ("backward", 0.65)):
            i0 = int(per_step * frac)
            print(f"\n# --- {a.sequence} consecutive launches of stream {busiest}, {label} (launch {i0} of {per_step} in step 1) ---")
            for s, e, n in seq[i0:i0 + a.sequence]:
                print(f"   {(e - s) / 1e3:9.2f} us  grid {lanes[(s, e)][1]:16s} {short(n)[:110]}")
    print("\n# --- by stream (per step) ---")
    for lane, d in sorted(per.items(), key=lambda kv: -kv[1]["t"]):
        print(f"## stream {lane}: {d['t']/1e6/a.steps:.2f} ms of kernels per step, {d['n'] // a.steps} launches per step")
        for k, (c, t) in sorted(d["k"].items(), key=lambda kv: -kv[1][1])[:28]:
            print(f"   {k:88s} {c / a.steps:8.1f} {t/1e6/a.steps:9.3f} ms {t/c/1e3:9.2f} us")
        sm = sorted(d["small"].items(), key=lambda kv: -kv[1][1])
        tot = sum(v[1] for _, v in sm)
        print(f"   -- launches under 8 us: {sum(v[0] for _, v in sm) // a.steps} per step, {tot/1e6/a.steps:.2f} ms per step")
        for (k, grid), (c, t) in sm[:14]:
            print(f"      {k:70s} grid {grid:14s} {c / a.steps:8.1f} {t/1e6/a.steps:9.3f} ms")


if __name__ == "__main__":
    main()
