"""Exact top-k (eval retrieval): dalm_sim_topk with the f32 and the bf16x3 first pass against materialise +
torch.topk, HIP events, on a synthetic unit-norm corpus.
    python tools/topk_bench.py [--queries 1024,4096] [--corpus 262144] [--k 10]
"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from dalm_amd.ops import default_ops  # noqa: E402


def timed(fn, iters=5, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return sorted(ts)[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", default="1024,4096")
    ap.add_argument("--corpus", type=int, default=262144)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--dim", type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = default_ops()
    g = torch.Generator(device="cpu").manual_seed(0)
    C = torch.nn.functional.normalize(torch.randn(a.corpus, a.dim, generator=g), dim=1).to(dev)
    for nq in [int(x) for x in a.queries.split(",")]:
        Q = torch.nn.functional.normalize(C[:nq] + 0.05 * torch.randn(nq, a.dim, device=dev), dim=1)
        flop = 2.0 * nq * a.corpus * a.dim
        res = {}
        for tag, env in (("f32 first pass", "0"), ("bf16x3 first pass", "1")):
            os.environ["DALM_TOPK_BF16X3"] = env
            t = timed(lambda: ops.sim_topk(Q, C, a.k))
            v, i, o = ops.sim_topk(Q, C, a.k)
            res[tag] = (t, i, int(o))
            print(f"top-{a.k}  {nq} x {a.corpus} x {a.dim}  fused, {tag:18s} {t*1e3:8.2f} ms  {flop/t/1e12:7.1f} TF f32-equivalent  overflow={int(o)}")
        os.environ.pop("DALM_TOPK_BF16X3", None)
        same = torch.equal(res["f32 first pass"][1], res["bf16x3 first pass"][1])
        if nq * a.corpus * 4 < 20e9:
            t = timed(lambda: torch.topk(Q @ C.t(), a.k, dim=1), iters=3, warmup=1)
            print(f"top-{a.k}  {nq} x {a.corpus} x {a.dim}  materialise + torch.topk        {t*1e3:8.2f} ms")
        print(f"   identical picks with either first pass: {same}")


if __name__ == "__main__":
    main()
