"""Shared helpers for the test-suite (golden loading, seeded synthetic batches)."""
from pathlib import Path

import numpy as np
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"
LOSS_CASES = sorted(p.stem for p in GOLDEN.glob("loss_*.npz"))
POOL_CASES = sorted(p.stem for p in GOLDEN.glob("pool_*.npz"))


def load_npz(name: str):
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def norm_rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def synth_batch(seed, B, D, Tg, V, *, pad_side="right", dtype=torch.float32, logit_gain=1.0, full_mask=False):
    """Seeded synthetic (q, p, logits, ids, mask, qlen) as SURVEY section 8d describes."""
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
    p = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
    logits = (logit_gain * torch.randn(B, Tg, V, generator=g)).to(dtype)
    ids = torch.randint(0, V, (B, Tg), generator=g)
    if full_mask:
        lens = torch.full((B,), Tg)
    else:
        lens = torch.randint(max(2, Tg // 4), Tg + 1, (B,), generator=g)
    ar = torch.arange(Tg).unsqueeze(0)
    mask = (ar < lens.unsqueeze(1)).long() if pad_side == "right" else (ar >= (Tg - lens).unsqueeze(1)).long()
    qlen = torch.clamp((lens.float() * 0.8).floor().long(), min=1)
    return q, p, logits, ids, mask, qlen
