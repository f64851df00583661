"""Where a hipGraph-replayed step spends its wall time: from a rocprofv3 `*_kernel_trace.csv` of a GRAPHED bench.py run, cut the
last `--steps` steps at the fused-Adam launches and report, per step: wall time, time with at least one kernel running on ANY
stream (union), idle time, per-stream busy time, and the largest idle gaps with the kernels on either side.

    rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py --data-path packed --steps 6 --warmup 4 --no-cpu-baseline --no-pmc
    python tools/graph_step_gaps.py out/*/*_kernel_trace.csv --steps 5
"""
import argparse
import csv
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from summarize_rocprof import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--skip-last", type=int, default=0, help="steps at the end of the trace to leave out (post-run probes)")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id") or "?"))
    rows.sort()
    adam_end = []
    last = None
    for s, e, n, _ in rows:                       # a step ends with a burst of multi_tensor_apply launches
        if "multi_tensor_apply" in n:
            if last is not None and s - last < 2_000_000:
                adam_end[-1] = e
            else:
                adam_end.append(e)
            last = e
    if a.skip_last:
        adam_end = adam_end[:-a.skip_last]
    if len(adam_end) < a.steps + 1:
        sys.exit(f"only {len(adam_end)} optimizer bursts in the trace")
    cuts = adam_end[-(a.steps + 1):]
    print(f"# {a.trace}: last {a.steps} steps")
    for i in range(a.steps):
        t0, t1 = cuts[i], cuts[i + 1]
        ks = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
        busy = defaultdict(float)
        for s, e, n, q in ks:
            busy[q] += (e - s) / 1e6
        # union of intervals
        union, gaps, cur_s, cur_e, prev_name = 0.0, [], None, None, None
        for s, e, n, q in ks:
            if cur_e is None:
                cur_s, cur_e = s, e
            elif s <= cur_e:
                cur_e = max(cur_e, e)
            else:
                union += cur_e - cur_s
                gaps.append((s - cur_e, prev_name, n))
                cur_s, cur_e = s, e
            prev_name = n
        if cur_e is not None:
            union += cur_e - cur_s
        wall = (t1 - t0) / 1e6
        print(f"step {i}: wall {wall:8.2f} ms  kernels {len(ks):5d}  some kernel running {union / 1e6:8.2f} ms  idle {wall - union / 1e6:7.2f} ms  "
              + "  ".join(f"stream {q}: {v:7.2f} ms" for q, v in sorted(busy.items(), key=lambda x: -x[1])[:4]))
        if i == a.steps - 1:
            small = sum(1 for g in gaps if g[0] < 3000)
            print(f"  gaps: {len(gaps)} (of which {small} under 3 us: sum {sum(g[0] for g in gaps if g[0] < 3000) / 1e6:.2f} ms); sum of all {sum(g[0] for g in gaps) / 1e6:.2f} ms")
            for g, pn, nn in sorted(gaps, reverse=True)[:12]:
                print(f"    {g / 1e3:8.1f} us   after {short(pn)[:70]:70s} before {short(nn)[:70]}")


if __name__ == "__main__":
    main()
