"""Both forms of the attention kernels on the same inputs -> one .pt file of outputs per process; `--compare a b` says whether
every tensor is BIT-identical.  The second forms (LDS-DMA staging, transpose reads: attn_fwd2 / attn_bwd_dq2 / attn_bwd_dkdv2 in
dalm_amd/csrc/attn.hip) restate the first forms' arithmetic instruction for instruction, so they must be.
    DALM_ATTN_FWD=1 DALM_ATTN_DKDV=1 python tools/attn_ab.py --out /tmp/a.pt     # first forms
    python tools/attn_ab.py --out /tmp/b.pt && python tools/attn_ab.py --compare /tmp/a.pt /tmp/b.pt"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

ap = argparse.ArgumentParser()
ap.add_argument("--out")
ap.add_argument("--compare", nargs=2)
a = ap.parse_args()

if a.compare:
    x, y = torch.load(a.compare[0]), torch.load(a.compare[1])
    bad = 0
    for k in x:
        same = torch.equal(x[k], y[k])
        if not same:
            bad += 1
            d = (x[k].double() - y[k].double()).abs()
            print(f"DIFFERENT {k}: {int((d > 0).sum())} of {d.numel()} elements, max |diff| {float(d.max()):.3e}")
    print(f"{len(x)} tensors, {bad} differ" + ("" if bad else ": bit-identical"))
    sys.exit(1 if bad else 0)

from dalm_amd import packed  # noqa: E402
from dalm_amd.models import attention  # noqa: E402

torch.manual_seed(0)
dev = torch.device("cuda:0")
res = {}


def mask2d(B, T, lens, left):
    m = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lens):
        if n:
            if left:
                m[b, T - n:] = 1
            else:
                m[b, :n] = 1
    return m


def run(tag, fn, q, k, v, go):
    qq, kk, vv = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    o = fn(qq, kk, vv)
    o.backward(go if o.shape == go.shape else go.transpose(1, 2))
    for n, t in (("o", o.detach()), ("dq", qq.grad), ("dk", kk.grad), ("dv", vv.grad)):
        res[f"{tag}.{n}"] = t.float().cpu()


# padded layout: (B, H, T, lens, left, causal, hd, dropout)
CASES = [(4, 8, 256, [256, 130, 77, 200], True, True, 128, 0.0), (3, 4, 200, [200, 64, 1], True, True, 128, 0.0),
         (2, 2, 448, [448, 300], True, True, 128, 0.0), (6, 16, 128, [128, 30, 77, 5, 100, 128], False, False, 64, 0.1),
         (5, 16, 50, [5, 15, 9, 50, 1], False, False, 64, 0.1), (2, 4, 320, [320, 191], True, True, 64, 0.0),
         (3, 4, 256, [256, 100, 31], True, True, 128, 0.1), (1, 4, 2048, [2048], True, True, 128, 0.0),
         (2, 2, 1000, [1000, 517], True, True, 64, 0.1), (1, 2, 1536, [1300], False, False, 128, 0.0), (2, 3, 96, [96, 33], False, True, 64, 0.0)]
for ci, (B, H, T, lens, left, causal, hd, p) in enumerate(CASES):
    g = torch.Generator().manual_seed(100 + ci)
    q, k, v, go = [(0.7 * torch.randn(B, T, H, hd, generator=g)).to(dev, torch.bfloat16).transpose(1, 2) for _ in range(4)]
    m2 = mask2d(B, T, lens, left).to(dev).bool()
    mask = m2[:, None, None, :].expand(B, 1, T, T)
    if causal:
        col = torch.arange(T, device=dev)
        mask = mask & (col[None, None, None, :] <= col[None, None, :, None])
    mask = mask.contiguous()
    run(f"padded{ci}", lambda x, y, z: attention.sdpa(x, y, z, mask, hd ** -0.5, False, p, 11 + ci), q, k, v, go)

# rotary embedding fused into the attention node (Llama path: dq / dk leave the backward kernels as gradients of the UN-rotated tensors)
for ci, (B, H, T, lens, hd) in enumerate([(3, 4, 256, [256, 140, 61], 128), (2, 2, 320, [320, 191], 64)]):
    g = torch.Generator().manual_seed(300 + ci)
    q, k, v, go = [(0.7 * torch.randn(B, T, H, hd, generator=g)).to(dev, torch.bfloat16).transpose(1, 2) for _ in range(4)]
    m2 = mask2d(B, T, lens, True).to(dev).bool()
    col = torch.arange(T, device=dev)
    mask = (m2[:, None, None, :].expand(B, 1, T, T) & (col[None, None, None, :] <= col[None, None, :, None])).contiguous()
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    ang = col.float()[:, None] * inv[None, :]
    cos = torch.cat((ang.cos(), ang.cos()), -1).to(torch.bfloat16)[None]
    sin = torch.cat((ang.sin(), ang.sin()), -1).to(torch.bfloat16)[None]
    assert attention.rope_fusable(q, k, cos, sin)
    run(f"rope{ci}", lambda x, y, z: attention.rope_sdpa(x, y, z, cos, sin, mask, hd ** -0.5, False), q, k, v, go)

# packed layout
PACKED = [(4, 8, 256, [256, 130, 77, 200], True, True, 128), (3, 2, 256, [200, 0, 129], False, True, 128),
          (4, 4, 128, [128, 30, 77, 5], False, False, 64), (5, 2, 50, [5, 15, 9, 50, 1], False, False, 64),
          (3, 2, 1024, [1024, 0, 700], True, True, 128), (40, 16, 50, [((7 * i) % 50) + (i % 3 == 0) for i in range(40)], False, False, 64)]
for ci, (B, H, T, lens, left, causal, hd) in enumerate(PACKED):
    g = torch.Generator().manual_seed(200 + ci)
    m2 = mask2d(B, T, lens, left)
    rows, cu = packed.pack_plan(m2, shifted=causal, multiple=64)
    n = rows.numel()
    ids = torch.zeros(B, T, dtype=torch.long, device=dev)
    _i, _p, desc, _v = packed.packed_inputs(ids, m2.to(dev), rows.to(dev), cu.to(dev), causal)
    q, k, v, go = [(0.7 * torch.randn(1, n, H, hd, generator=g)).to(dev, torch.bfloat16).transpose(1, 2) for _ in range(4)]
    run(f"packed{ci}", lambda x, y, z: attention.sdpa(x, y, z, desc, hd ** -0.5, False), q, k, v, go)
torch.cuda.synchronize()
torch.save(res, a.out)
print(f"{len(res)} tensors -> {a.out}")
