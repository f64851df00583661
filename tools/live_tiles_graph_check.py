"""Does the live-tile byte array of dalm_attn_mask_bits_packed follow the batch under hipGraph replay?  (It is cleared by a
hipMemsetAsync in front of the kernel: captured as a memset node.)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dalm_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
nseq, T = 12, 128
W = (T + 31) // 32


def make(lens):
    cu = torch.zeros(nseq + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(lens).cumsum(0)
    return cu


cus = [make([128, 0, 40, 0, 90, 7, 0, 0, 33, 64, 0, 1]), make([0, 100, 0, 65, 0, 0, 128, 12, 0, 0, 31, 0])]
n = 512
key_live = torch.ones(n, dtype=torch.uint8, device=dev)
cu_static = cus[0].to(dev)
rows = torch.empty(nseq * 32 * W * W, dtype=torch.int32, device=dev)
cols = torch.empty_like(rows)
live = torch.empty(nseq * W * W, dtype=torch.uint8, device=dev)


def call():
    hip.call("dalm_attn_mask_bits_packed", hip.ptr(key_live), hip.ptr(cu_static), nseq, T, 0, hip.ptr(rows), hip.ptr(cols), hip.ptr(live), hip.stream())


def eager(cu):
    cu_static.copy_(cu.to(dev))
    call()
    torch.cuda.synchronize()
    return live.clone()


want = [eager(c) for c in cus]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    cu_static.copy_(cus[0].to(dev))
    call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        call()
for k in (0, 1, 0, 1):
    cu_static.copy_(cus[k].to(dev))
    live.fill_(1) if k == 0 else None
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    same = torch.equal(live, want[k])
    extra = int(((live != 0) & (want[k] == 0)).sum())
    print(f"replay with batch {k}: live tiles equal to the eager result: {same} (stale bytes: {extra})")
