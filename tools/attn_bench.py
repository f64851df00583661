"""Attention backward at the cfg3 layer shape (B 18, H 32, T 256, hd 128, HF causal + left-padding mask): `dalm_attn_bwd`
against torch's memory-efficient backward, hipGraph replay timing.  -> stdout (profiles/r05_attn_bwd.txt)"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dalm_amd.models import attention  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=18)
ap.add_argument("--H", type=int, default=32)
ap.add_argument("--T", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--lens", default="random", help="random (T/2..T) | full | <n> (every sequence n tokens, left-padded)")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, H, T, hd = a.B, a.H, a.T, 128
g = torch.Generator().manual_seed(0)
q, k, v, go = [torch.randn(B, T, H, hd, generator=g).bfloat16().to(dev).transpose(1, 2) for _ in range(4)]
lens = torch.randint(T // 2, T + 1, (B,), generator=g)
lens[0] = T
if a.lens == "full":
    lens[:] = T
elif a.lens != "random":
    lens[:] = int(a.lens)
col = torch.arange(T, device=dev)
st = (T - lens).to(dev)
mask = ((col[None, None, :] <= col[None, :, None]) & (col[None, None, :] >= st[:, None, None]))[:, None]
scale = hd ** -0.5
live_frac = float(mask.float().mean())


def timed(fn, label, flops):
    qq, kk, vv = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):                       # forward on the capture stream: autograd replays the backward on it
        o = fn(qq, kk, vv)
        for _ in range(3):
            torch.autograd.grad(o, (qq, kk, vv), go, retain_graph=True)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(a.iters):
                torch.autograd.grad(o, (qq, kk, vv), go, retain_graph=True)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * a.iters)
    print(f"{label:48s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s on the live tiles' 5 products")
    return us


def timed_fwd(fn, label, flops):
    qq, kk, vv = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn(qq, kk, vv)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(a.iters):
                fn(qq, kk, vv)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * a.iters)
    print(f"{label:48s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s on the live tiles' 2 products")
    return us


flops = 5 * 2.0 * B * H * T * T * hd * live_frac
print(f"# B {B} H {H} T {T} hd {hd}; live fraction of the mask {live_frac:.3f}; algorithmic bytes {8 * B * H * T * hd * 2 / 1e6:.0f} MB")
t_ours = timed(lambda x, y, z: attention._SdpaHipBackward.apply(x, y, z, mask, scale, False), "dalm_attn_bwd (2 launches)", flops)
t_torch = timed(lambda x, y, z: torch.nn.functional.scaled_dot_product_attention(x, y, z, attn_mask=mask, scale=scale),
                "torch memory-efficient backward (3 launches)", flops)
print(f"# ratio {t_torch / t_ours:.2f}x")
f_ours = timed_fwd(lambda x, y, z: attention._SdpaHipBackward.apply(x, y, z, mask, scale, False), "dalm_attn_fwd (+ mask bits, cached)", flops * 0.4)
f_torch = timed_fwd(lambda x, y, z: torch.nn.functional.scaled_dot_product_attention(x, y, z, attn_mask=mask, scale=scale),
                    "torch memory-efficient forward", flops * 0.4)
print(f"# forward ratio {f_torch / f_ours:.2f}x")
