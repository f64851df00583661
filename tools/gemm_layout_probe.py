"""Backward (dx = g @ W) GEMMs of the frozen projections: the library's NN kernel on W [N, K] as stored, against the same
product through a TRANSPOSED copy W^T [K, N] (then it is the forward's TN layout), at the cfg3 generator shapes, with the
pre-tuned solution table replayed as bench.py does.  hipGraph replay timing.

    python tools/gemm_layout_probe.py"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * iters)


def main():
    from dalm_amd.tuning import enable_tuned_gemms

    print("tuned table loaded:", enable_tuned_gemms())
    dev = torch.device("cuda:0")
    R = 4608
    for name, N, K in (("q/k/v/o_proj", 4096, 4096), ("gate/up_proj", 11008, 4096), ("down_proj", 4096, 11008)):
        x = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
        g = torch.randn(R, N, device=dev, dtype=torch.bfloat16)
        W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
        Wt = W.t().contiguous()
        dx = torch.empty(R, K, device=dev, dtype=torch.bfloat16)
        fl = 2 * R * N * K
        rows = [("forward  F.linear(x, W)            [TN]", lambda: F.linear(x, W)),
                ("backward torch.mm(g, W)            [NN]", lambda: torch.mm(g, W)),
                ("backward F.linear(g, W^T copy)     [TN]", lambda: F.linear(g, Wt)),
                ("backward dx.addmm_(g, W)           [NN, beta 1]", lambda: dx.addmm_(g, W)),
                ("backward dx.addmm_(g, W^T.t())     [TN, beta 1]", lambda: dx.addmm_(g, Wt.t()))]
        for label, fn in rows:
            t = timed(fn)
            print(f"{name:14s} N={N:5d} K={K:5d}  {label:48s} {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
