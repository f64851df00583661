"""A LoRA-wrapped Linear as ONE autograd node on the `dalm_lora_*` kernels (dalm_amd/csrc/lora.hip).

peft (what the reference configures: dalm/models/rag_e2e_base_model.py:145-160) evaluates
    out = W x + s * B(A(dropout(x)))
as eager ops; per wrapped projection of the cfg3 generator that was 6 launches forward and 11 backward around the base GEMM.
Here: forward = base GEMM (the library's) + `rowdot` (z = dropout(x) A^T / (1-p)) + `rankupd` (out += s z B^T, in place);
backward = base GEMM (dx = g W) + `rowdot` (dz = s g B) + 2 x `colacc` (dB = s g^T z, dA = dz^T dropout(x) / (1-p)) +
`rankupd` (dx += mask (dz A) / (1-p), in place).  A, B and every [rows, r] tensor stay in float32: no autocast casts of the
adapter weights, no gradient casts back.

Dropout: the mask is never stored; the kernels regenerate it from (a 64-bit seed word in DEVICE memory, a per-call salt, the
element index).  `advance_dropout_seed()` bumps the seed word with a device op, so it can be captured into the step's
hipGraph: every replay draws new masks.  The per-call salt (module id and a host call counter) separates the modules and -
outside graphs - successive calls.  The random stream is this library's own, not torch's (dropout masks never agreed
between devices or libraries anyway); `p` and the 1 / (1-p) rescaling are peft's.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .. import hip

_seed_words: Dict[int, torch.Tensor] = {}


def dropout_seed(device: torch.device) -> torch.Tensor:
    """The device-resident 64-bit seed word of `device` (created from torch's CPU generator on first use)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    t = _seed_words.get(idx)
    if t is None:
        # derived from torch's seed WITHOUT drawing from the global generator (a draw here would shift every later random
        # number of the program - shuffles, initialisations - depending on whether a LoRA layer happened to run first)
        g = torch.Generator().manual_seed((torch.initial_seed() ^ 0x5DA1A0D5EED) & (2 ** 63 - 1))
        first = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, generator=g).item())
        t = torch.full((1,), first, dtype=torch.int64, device=torch.device("cuda", idx))
        _seed_words[idx] = t
    return t


def advance_dropout_seed(device: Optional[torch.device] = None) -> None:
    """One device op (capturable): the next forward draws new dropout masks.  The training steps call this once per step."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    dropout_seed(dev).add_(0x9E3779B97F4A7C15 - (1 << 64))     # += golden-ratio increment (as a signed 64-bit value)


def branch_supported(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> bool:
    """The low-rank branch alone (any base layer): shapes and dtypes the kernels take."""
    r = a.shape[0]
    return (x.is_cuda and r in (8, 16) and a.shape[1] % 8 == 0 and b.shape[0] % 8 == 0
            and a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
            and x.dtype in (torch.float32, torch.bfloat16))


def supported(x: torch.Tensor, base, a: torch.Tensor, b: torch.Tensor) -> bool:
    """Base GEMM and branch as one autograd node: plain nn.Linear bases only (a subclass may override forward)."""
    return (branch_supported(x, a, b) and type(base) is torch.nn.Linear
            and base.weight.dtype in (torch.float32, torch.bfloat16))


def _rowdot(x2, w, kmajor, rank, scale, p, seed, salt):
    out = torch.empty(x2.shape[0], rank, device=x2.device, dtype=torch.float32)
    hip.call("dalm_lora_rowdot", hip.ptr(x2), hip.dtype_code(x2), hip.ptr(w), int(kmajor), x2.shape[0], x2.shape[1], rank,
             float(scale), float(p), hip.ptr(seed) if p > 0 else None, salt, hip.ptr(out), hip.stream())
    return out


def _rankupd_(y2, z, w, cmajor, rank, scale, p, seed, salt):
    hip.call("dalm_lora_rankupd", hip.ptr(y2), hip.dtype_code(y2), hip.ptr(z), hip.ptr(w), int(cmajor), y2.shape[0], y2.shape[1],
             rank, float(scale), float(p), hip.ptr(seed) if p > 0 else None, salt, hip.stream())
    return y2


def _colacc(x2, z, rank, scale, p, seed, salt, jmajor):
    R, C = x2.shape
    out = torch.empty((rank, C) if jmajor else (C, rank), device=x2.device, dtype=torch.float32)
    nbytes = hip.load().dalm_lora_colacc_workspace_bytes(R, C, rank)
    ws = torch.empty(nbytes, device=x2.device, dtype=torch.uint8)
    hip.call("dalm_lora_colacc", hip.ptr(x2), hip.dtype_code(x2), hip.ptr(z), R, C, rank, float(scale), float(p),
             hip.ptr(seed) if p > 0 else None, salt, hip.ptr(out), int(jmajor), hip.ptr(ws), nbytes, hip.stream())
    return out


class _LoRALinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, a, b, scaling, p, salt):
        K, N, rank = weight.shape[1], weight.shape[0], a.shape[0]
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        x2 = x.reshape(-1, K)
        if cdt is not None and x2.dtype != cdt:
            x2 = x2.to(cdt)                    # what autocast's linear would do to its input
        x2 = x2.contiguous()
        w = weight if weight.dtype == x2.dtype else weight.to(x2.dtype)
        bia = bias if bias is None or bias.dtype == x2.dtype else bias.to(x2.dtype)
        with torch.autocast("cuda", enabled=False):
            out = F.linear(x2, w, bia)                                                    # [R, N], the library's GEMM
        seed = dropout_seed(x2.device) if p > 0 else None
        z = _rowdot(x2, a, True, rank, 1.0 / (1.0 - p), p, seed, salt)                    # dropout(x) A^T / (1-p)
        _rankupd_(out, z, b, True, rank, scaling, 0.0, None, 0)                           # out += s z B^T
        ctx.save_for_backward(x2, w, a, b, z)      # w: the weight in the compute dtype (the parameter itself when they agree)
        ctx.meta = (scaling, p, salt, x.shape, x.dtype, rank)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        x2, weight, a, b, z = ctx.saved_tensors
        scaling, p, salt, xshape, xdtype, rank = ctx.meta
        N = weight.shape[0]
        g2 = g.reshape(-1, N)
        if g2.dtype != x2.dtype:
            g2 = g2.to(x2.dtype)
        g2 = g2.contiguous()
        seed = dropout_seed(x2.device) if p > 0 else None
        keep = 1.0 / (1.0 - p)
        dz = _rowdot(g2, b, False, rank, scaling, 0.0, None, 0)                           # s g B          [R, r]
        db = _colacc(g2, z, rank, scaling, 0.0, None, 0, False) if ctx.needs_input_grad[4] else None      # s g^T z  [N, r]
        da = _colacc(x2, dz, rank, keep, p, seed, salt, True) if ctx.needs_input_grad[3] else None        # [r, K]
        dx = None
        if ctx.needs_input_grad[0]:
            w = weight if weight.dtype == g2.dtype else weight.to(g2.dtype)
            dx = torch.mm(g2, w)                                                          # the library's GEMM
            _rankupd_(dx, dz, a, False, rank, keep, p, seed, salt)                        # += mask (dz A) / (1-p)
            dx = dx.view(xshape)
            if dx.dtype != xdtype:
                dx = dx.to(xdtype)
        dbias = g2.sum(0) if ctx.needs_input_grad[2] else None
        return dx, None, dbias, da, db, None, None, None


class _LoRABranchFn(torch.autograd.Function):
    """out = base_out + s * B(A(dropout(x))) for a base layer that runs as its own module (nf4 storage, a Linear subclass):
    the branch is added to the base layer's output in place; the backward passes g through to the base layer untouched and
    returns the branch's own dx, dA, dB."""

    @staticmethod
    def forward(ctx, base_out, x, a, b, scaling, p, salt):
        K, N, rank = a.shape[1], b.shape[0], a.shape[0]
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        x2 = x.reshape(-1, K)
        if cdt is not None and x2.dtype != cdt:
            x2 = x2.to(cdt)
        x2 = x2.contiguous()
        seed = dropout_seed(x2.device) if p > 0 else None
        z = _rowdot(x2, a, True, rank, 1.0 / (1.0 - p), p, seed, salt)
        _rankupd_(base_out.view(-1, N), z, b, True, rank, scaling, 0.0, None, 0)
        ctx.mark_dirty(base_out)
        ctx.save_for_backward(x2, a, b, z)
        ctx.meta = (scaling, p, salt, x.shape, x.dtype, rank)
        return base_out

    @staticmethod
    def backward(ctx, g):
        x2, a, b, z = ctx.saved_tensors
        scaling, p, salt, xshape, xdtype, rank = ctx.meta
        g2 = g.reshape(-1, b.shape[0]).contiguous()
        seed = dropout_seed(x2.device) if p > 0 else None
        keep = 1.0 / (1.0 - p)
        dz = _rowdot(g2, b, False, rank, scaling, 0.0, None, 0)
        db = _colacc(g2, z, rank, scaling, 0.0, None, 0, False) if ctx.needs_input_grad[3] else None
        da = _colacc(x2, dz, rank, keep, p, seed, salt, True) if ctx.needs_input_grad[2] else None
        dx = None
        if ctx.needs_input_grad[1]:
            dx = _rankupd_(torch.zeros_like(x2), dz, a, False, rank, keep, p, seed, salt).view(xshape)
            if dx.dtype != xdtype:
                dx = dx.to(xdtype)
        return g, dx, da, db, None, None, None


def lora_branch_(base_out, x, a, b, scaling: float, p: float, salt: int):
    """base_out += scaling * B(A(dropout_p(x))), in place, differentiable in x, a, b (and base_out)."""
    if base_out.dtype not in (torch.float32, torch.bfloat16) or not base_out.is_contiguous():
        raise TypeError("lora_branch_: the base layer's output must be a contiguous float32 / bfloat16 tensor")
    return _LoRABranchFn.apply(base_out, x, a, b, float(scaling), float(p), int(salt) & 0xFFFFFFFF)


def lora_linear(x, base: torch.nn.Linear, a: torch.Tensor, b: torch.Tensor, scaling: float, p: float, salt: int):
    """W x + bias + scaling * B(A(dropout_p(x))) with W = base.weight frozen or not (its gradient is not produced here:
    LoRA freezes the base layer)."""
    if base.weight.requires_grad:
        raise RuntimeError("lora_linear: the base weight must be frozen (LoRA trains A and B only)")
    return _LoRALinearFn.apply(x, base.weight, base.bias, a, b, float(scaling), float(p), int(salt) & 0xFFFFFFFF)

