"""One attempt each at the two tower kernels furthest below the HBM rate: rms_norm_bwd (all three streams
requested before the reduction: DALM_RMS_BWD_V2=1) and SwiGLU (8 tiles per workgroup: DALM_SWIGLU_STEPS=8), at the cfg3 shapes,
hipGraph replay timing.  Run once per variant (the switches are read when the library loads):
    for v in "" DALM_RMS_BWD_V2=1 DALM_SWIGLU_STEPS=8; do env $v python tools/tower_attempts.py; done"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dalm_amd import hip  # noqa: E402


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
    dev = torch.device("cuda:0")
    R, D, I = 4608, 4096, 11008
    bf = torch.bfloat16
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("DALM_RMS_BWD_V2", "DALM_SWIGLU_STEPS") if k in os.environ) or "defaults"
    dy, h, dres = (torch.randn(R, D, device=dev, dtype=bf) for _ in range(3))
    w = torch.randn(D, device=dev, dtype=bf)
    rstd = torch.rand(R, device=dev) + 0.5
    dx = torch.empty_like(dy)
    t = timed(lambda: hip.call("dalm_rms_norm_bwd", hip.ptr(dy), hip.ptr(h), hip.ptr(w), hip.ptr(rstd), hip.ptr(dres), hip.BF16, R, D,
                               hip.ptr(dx), hip.stream()))
    mb = 4 * R * D * 2 / 1e6
    print(f"[{tag}] rms_norm_bwd + residual gradient [{R}, {D}] bf16: {t:7.2f} us  {mb / t:5.2f} TB/s  {mb / t / 8:5.3f} of 8 TB/s")
    gate, up, da = (torch.randn(R, I, device=dev, dtype=bf) for _ in range(3))
    act, dg, du = torch.empty_like(gate), torch.empty_like(gate), torch.empty_like(gate)
    n = R * I
    t = timed(lambda: hip.call("dalm_swiglu_fwd", hip.ptr(gate), hip.ptr(up), hip.ptr(act), hip.BF16, n, hip.stream()))
    print(f"[{tag}] swiglu_fwd [{R}, {I}] bf16: {t:7.2f} us  {3 * n * 2 / 1e6 / t:5.2f} TB/s  {3 * n * 2 / 1e6 / t / 8:5.3f} of 8 TB/s")
    t = timed(lambda: hip.call("dalm_swiglu_bwd", hip.ptr(da), hip.ptr(gate), hip.ptr(up), hip.ptr(dg), hip.ptr(du), hip.BF16, n, hip.stream()))
    print(f"[{tag}] swiglu_bwd [{R}, {I}] bf16: {t:7.2f} us  {5 * n * 2 / 1e6 / t:5.2f} TB/s  {5 * n * 2 / 1e6 / t / 8:5.3f} of 8 TB/s")


if __name__ == "__main__":
    main()
