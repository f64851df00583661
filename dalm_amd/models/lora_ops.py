"""A LoRA-wrapped Linear as ONE autograd node on the `dalm_lora_*` kernels (dalm_amd/csrc/lora.hip).

peft (what the reference configures: dalm/models/rag_e2e_base_model.py:145-160) evaluates
    out = W x + s * B(A(dropout(x)))
as eager ops; per wrapped projection of the cfg3 generator that was 6 launches forward and 11 backward around the base GEMM.
Here: forward = base GEMM (the library's) + `rowdot` (z = dropout(x) A^T / (1-p)) + `rankupd` (out += s z B^T, in place);
backward = base GEMM (dx = g W) + `rowdot` (dz = s g B) + 2 x `colacc` (dB = s g^T z, dA = dz^T dropout(x) / (1-p)) +
`rankupd` (dx += mask (dz A) / (1-p), in place).  A, B and every [rows, r] tensor stay in float32: no autocast casts of the
adapter weights, no gradient casts back.

Round 5 (`dalm_lora2_*`, dalm_amd/csrc/lora2.hip): bf16 activations take the stacked kernels - projections that read the
same tensor (q_proj / v_proj of a Llama block, query / key / value of a BERT block) run as ONE autograd node
(`lora_group_forward`): x is streamed once for all their z, once for all their dA, dx gets every rank update in one pass, the
base GEMMs of the backward accumulate into one dx (no autograd add kernels), and the dropout mask is computed once in the
forward and kept as bits (the backward no longer reads the seed word).  lora_B is held in [r, N]-major memory (models/lora.py) so
that every weight operand is K-major.  float32 activations keep the round-4 kernels below.

Dropout (float32 path): the mask is never stored; the kernels regenerate it from (a 64-bit seed word in DEVICE memory, a per-call salt, the
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
    _epoch[0] += 1


_epoch = [0]                 # host count of advances: a snapshot is good until the next one
_snapshots: dict = {}


def dropout_seed_snapshot(device: torch.device) -> torch.Tensor:
    """A copy of the seed word as it is NOW, shared by every caller on this stream until the next `advance_dropout_seed`
    (one 8-byte copy per step and stream instead of one per attention call).  Callers keep it for their backward: the live
    word may have advanced by then.  Never shared between an eager region and a hipGraph capture (the copy must be part of
    the graph that reads it)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(device).cuda_stream, bool(torch.cuda.is_current_stream_capturing()))
    hit = _snapshots.get(key)
    if hit is None or hit[0] != _epoch[0]:
        hit = (_epoch[0], dropout_seed(device).clone())
        _snapshots[key] = hit
    return hit[1]


def branch_supported(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> bool:
    """The low-rank branch alone (any base layer): shapes and dtypes the kernels take."""
    r = a.shape[0]
    return (x.is_cuda and r in (8, 16) and a.shape[1] % 8 == 0 and b.shape[0] % 8 == 0
            and a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous()
            and (b.is_contiguous() or b.t().is_contiguous())
            and x.dtype in (torch.float32, torch.bfloat16))


def _b_cmajor(b: torch.Tensor) -> bool:
    """lora_B.weight [N, r]: True when its memory is [N][r] (a plain contiguous Linear weight), False for the [r][N]-major form
    models/lora.py creates (then `b.t()` is the contiguous [r, N] matrix every kernel wants)."""
    return b.is_contiguous() and not (b.shape[1] > 1 and b.t().is_contiguous())


def _colacc_b(g2, z, rank, scale, b):
    """dB with the memory layout of b."""
    if _b_cmajor(b):
        return _colacc(g2, z, rank, scale, 0.0, None, 0, False)
    return _colacc(g2, z, rank, scale, 0.0, None, 0, True).t()


def supported(x: torch.Tensor, base, a: torch.Tensor, b: torch.Tensor) -> bool:
    """Base GEMM and branch as one autograd node: plain nn.Linear bases only (a subclass may override forward)."""
    return (branch_supported(x, a, b) and _is_plain_linear(base)
            and base.weight.dtype in (torch.float32, torch.bfloat16))


def _is_plain_linear(m) -> bool:
    # nn.Linear itself, or Falcon's FalconLinear (y = x W^T + b spelled as a matmul: the same function; BASELINE config 5)
    return type(m) is torch.nn.Linear or (isinstance(m, torch.nn.Linear) and type(m).__name__ == "FalconLinear")


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
        # the seed word as the forward saw it (8-byte device copy, capturable): the backward regenerates the mask from THIS word,
        # whatever advanced the live one in between
        seed = dropout_seed(x2.device).clone() if p > 0 else None
        z = _rowdot(x2, a, True, rank, 1.0 / (1.0 - p), p, seed, salt)                    # dropout(x) A^T / (1-p)
        _rankupd_(out, z, b, _b_cmajor(b), rank, scaling, 0.0, None, 0)                   # out += s z B^T
        ctx.seed = seed
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
        seed = ctx.seed
        keep = 1.0 / (1.0 - p)
        dz = _rowdot(g2, b, not _b_cmajor(b), rank, scaling, 0.0, None, 0)                # s g B          [R, r]
        db = _colacc_b(g2, z, rank, scaling, b) if ctx.needs_input_grad[4] else None                      # s g^T z  [N, r]
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
        seed = dropout_seed(x2.device).clone() if p > 0 else None
        z = _rowdot(x2, a, True, rank, 1.0 / (1.0 - p), p, seed, salt)
        _rankupd_(base_out.view(-1, N), z, b, _b_cmajor(b), rank, scaling, 0.0, None, 0)
        ctx.mark_dirty(base_out)
        ctx.seed = seed
        ctx.save_for_backward(x2, a, b, z)
        ctx.meta = (scaling, p, salt, x.shape, x.dtype, rank)
        return base_out

    @staticmethod
    def backward(ctx, g):
        x2, a, b, z = ctx.saved_tensors
        scaling, p, salt, xshape, xdtype, rank = ctx.meta
        g2 = g.reshape(-1, b.shape[0]).contiguous()
        seed = ctx.seed
        keep = 1.0 / (1.0 - p)
        dz = _rowdot(g2, b, not _b_cmajor(b), rank, scaling, 0.0, None, 0)
        db = _colacc_b(g2, z, rank, scaling, b) if ctx.needs_input_grad[3] else None
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



# =====================================================================================================================
# round 5: stacked kernels (dalm_lora2_*), bf16 activations
# =====================================================================================================================
_tickets2: Dict[tuple, torch.Tensor] = {}
_FWD_STACKED = __import__("os").environ.get("DALM_LORA_FWD_STACKED", "0") == "1"      # A/B: mode 2 in the forward


def _colacc_tickets(dev: torch.device, words: int) -> torch.Tensor:
    """Arrival tickets of `dalm_lora2_colacc`: zeroed once, left zero by every call.  One buffer per (device, stream): calls that
    share it are ordered on that stream (the retriever towers run on their own streams beside the generator's)."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _tickets2.get(key)
    if buf is None or buf.numel() < words:
        buf = torch.zeros((max(words, 4096),), device=dev, dtype=torch.int32)
        _tickets2[key] = buf
    return buf


def v2_supported(x2: torch.Tensor, rank: int) -> bool:
    import os

    return (x2.is_cuda and x2.dtype == torch.bfloat16 and rank in (8, 16) and x2.shape[1] % 32 == 0
            and os.environ.get("DALM_LORA_V2", "1") != "0")


def _bt(b: torch.Tensor) -> torch.Tensor:
    """lora_B.weight [N, r] -> the contiguous [r, N] matrix (a view when models/lora.py created the parameter)."""
    t = b.t()
    return t if t.is_contiguous() else t.contiguous()


def rowdot2(xs, Ws, rank: int, scale: float, p: float, salts, mode: int):
    """mode 1: ([x], [W]) -> [z];  mode 2: ([x], [W0, W1]) -> [z0, z1] from ONE pass over x;  mode 3: ([x0, x1], [W0, W1]).
    Returns (zs, bits): bits[t] is the [R, K/8] uint8 keep mask of slot t (None without dropout)."""
    x0 = xs[0]
    R, K = x0.shape
    n = 1 if mode == 1 else 2
    dev = x0.device
    zs = [torch.empty(R, rank, device=dev, dtype=torch.float32) for _ in range(n)]
    bits = [torch.empty(R, K // 8, device=dev, dtype=torch.uint8) if p > 0 else None for _ in range(n)]
    seed = dropout_seed(dev) if p > 0 else None
    x1 = xs[1] if mode == 3 else None
    hip.call("dalm_lora2_rowdot", hip.ptr(x0), hip.ptr(x1), hip.ptr(Ws[0]), hip.ptr(Ws[1]) if n == 2 else None,
             hip.ptr(zs[0]), hip.ptr(zs[1]) if n == 2 else None, hip.ptr(bits[0]), hip.ptr(bits[1]) if n == 2 else None,
             R, K, rank, float(scale), float(p), hip.ptr(seed), int(salts[0]) & 0xFFFFFFFF,
             (int(salts[1]) & 0xFFFFFFFF) if n == 2 else 0, mode, hip.stream())
    return zs, bits


def rankupd2_(ys, zs, Ws, bits, rank: int, scale: float, mode: int):
    """mode 1: y += scale m (z W);  mode 2: ONE y += both terms;  mode 3: two independent (y, z, W)."""
    y0 = ys[0]
    R, C = y0.shape
    n = 1 if mode == 1 else 2
    hip.call("dalm_lora2_rankupd", hip.ptr(y0), hip.ptr(ys[1]) if mode == 3 else None, hip.ptr(zs[0]),
             hip.ptr(zs[1]) if n == 2 else None, hip.ptr(Ws[0]), hip.ptr(Ws[1]) if n == 2 else None,
             hip.ptr(bits[0]) if bits is not None else None, hip.ptr(bits[1]) if (bits is not None and n == 2) else None,
             R, C, rank, float(scale), mode, hip.stream())
    return ys


def colacc2(xs, zs, bits, rank: int, scale: float, mode: int):
    """out_t [rank, C] = scale * sum_row m_t x_t[row, :] (x) z_t[row, :]; mode as in rowdot2 (mode 2: one pass over x)."""
    x0 = xs[0]
    R, C = x0.shape
    n = 1 if mode == 1 else 2
    dev = x0.device
    lib = hip.load()
    outs = [torch.empty(rank, C, device=dev, dtype=torch.float32) for _ in range(n)]
    nbytes = lib.dalm_lora2_colacc_workspace_bytes(R, C, rank, mode)
    ws = torch.empty(max(nbytes, 8), device=dev, dtype=torch.uint8)
    tickets = _colacc_tickets(dev, lib.dalm_lora2_colacc_ticket_words(C, mode))
    hip.call("dalm_lora2_colacc", hip.ptr(x0), hip.ptr(xs[1]) if mode == 3 else None, hip.ptr(zs[0]),
             hip.ptr(zs[1]) if n == 2 else None, hip.ptr(bits[0]) if bits is not None else None,
             hip.ptr(bits[1]) if (bits is not None and n == 2) else None, hip.ptr(outs[0]), hip.ptr(outs[1]) if n == 2 else None,
             R, C, rank, float(scale), mode, hip.ptr(ws), nbytes, hip.ptr(tickets), hip.stream())
    return outs


def _pairs(idx):
    """[0, 1, 2] -> [(0, 1), (2,)]: the launches of a group of projections."""
    return [tuple(idx[i:i + 2]) for i in range(0, len(idx), 2)]


class _LoRAGroupFn(torch.autograd.Function):
    """(out_0, ..., out_{n-1}) = (W_i x + b_i [+ s_i B_i A_i dropout_i(x)])_i for projections that read the same x.
    Tensor arguments after x, per member: weight, bias or None, lora_A or None, lora_B^T ([r, N] contiguous) or None.
    meta: per member (scaling, p, salt) or None for a plain Linear."""

    @staticmethod
    def forward(ctx, x, meta, *flat):
        n = len(meta)
        Ws, bs, As, Bts = flat[0::4], flat[1::4], flat[2::4], flat[3::4]
        K = Ws[0].shape[1]
        cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None
        x2 = x.reshape(-1, K)
        if cdt is not None and x2.dtype != cdt:
            x2 = x2.to(cdt)
        x2 = x2.contiguous()
        wc = [w if w.dtype == x2.dtype else w.to(x2.dtype) for w in Ws]
        outs = []
        with torch.autocast("cuda", enabled=False):
            for i in range(n):
                bi = bs[i] if bs[i] is None or bs[i].dtype == x2.dtype else bs[i].to(x2.dtype)
                outs.append(torch.nn.functional.linear(x2, wc[i], bi))
        lora = [i for i in range(n) if meta[i] is not None]
        rank = As[lora[0]].shape[0]
        zs, bits = [None] * n, [None] * n
        # one pass over x per pair of adapters (one pass for q + v); members with different dropout rates cannot share a launch
        for grp in _pairs(lora):
            p = meta[grp[0]][1]
            if len(grp) == 2 and meta[grp[1]][1] == p:
                # both adapters in one launch.  Mode 3 with the SAME x in both slots (16-row tiles, the second read of x comes
                # from the caches) measured faster than mode 2's 8-row stacked tiles, which pull all of A_q and A_v through every
                # workgroup: 37 vs 45 us at [4608, 4096] with dropout (profiles/r05_lora_bench_4608x4096.txt)
                fwd_mode = 2 if (rank == 8 and _FWD_STACKED) else 3
                z, bt = rowdot2([x2, x2], [As[grp[0]], As[grp[1]]], rank, 1.0 / (1.0 - p), p, [meta[grp[0]][2], meta[grp[1]][2]],
                                fwd_mode)
                zs[grp[0]], zs[grp[1]], bits[grp[0]], bits[grp[1]] = z[0], z[1], bt[0], bt[1]
            else:
                for i in grp:
                    pi = meta[i][1]
                    z, bt = rowdot2([x2], [As[i]], rank, 1.0 / (1.0 - pi), pi, [meta[i][2]], 1)
                    zs[i], bits[i] = z[0], bt[0]
        for grp in _pairs(lora):
            same = len(grp) == 2 and outs[grp[0]].shape == outs[grp[1]].shape and meta[grp[0]][0] == meta[grp[1]][0]
            if same:
                rankupd2_([outs[grp[0]], outs[grp[1]]], [zs[grp[0]], zs[grp[1]]], [Bts[grp[0]], Bts[grp[1]]], None, rank,
                          meta[grp[0]][0], 3)
            else:
                for i in grp:
                    rankupd2_([outs[i]], [zs[i]], [Bts[i]], None, rank, meta[i][0], 1)
        ctx.meta, ctx.lora, ctx.n, ctx.rank = meta, lora, n, rank
        ctx.w_params = Ws
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        ctx.set_materialize_grads(False)          # an output nobody differentiated arrives as None, not as a zero tensor
        ctx.save_for_backward(x2, *wc, *[As[i] for i in lora], *[Bts[i] for i in lora], *[zs[i] for i in lora],
                              *[b for b in (bits[i] for i in lora) if b is not None])
        ctx.has_bits = [bits[i] is not None for i in lora]
        return tuple(o.view(*x.shape[:-1], o.shape[-1]) for o in outs)

    @staticmethod
    def backward(ctx, *gs):
        n, lora, meta, rank = ctx.n, ctx.lora, ctx.meta, ctx.rank
        sv = ctx.saved_tensors
        x2, wc = sv[0], sv[1:1 + n]
        m = len(lora)
        As = dict(zip(lora, sv[1 + n:1 + n + m]))
        Bts = dict(zip(lora, sv[1 + n + m:1 + n + 2 * m]))
        zs = dict(zip(lora, sv[1 + n + 2 * m:1 + n + 3 * m]))
        rest = list(sv[1 + n + 3 * m:])
        bits = {i: (rest.pop(0) if hb else None) for i, hb in zip(lora, ctx.has_bits)}
        g2 = []
        for i in range(n):
            g = gs[i]
            if g is None:
                g2.append(None)
                continue
            g = g.reshape(-1, wc[i].shape[0])
            if g.dtype != x2.dtype:
                g = g.to(x2.dtype)
            g2.append(g.contiguous())
        live = [i for i in lora if g2[i] is not None]
        dz, dA, dBt = {}, {}, {}
        for grp in _pairs(live):
            same = len(grp) == 2 and g2[grp[0]].shape == g2[grp[1]].shape and meta[grp[0]][0] == meta[grp[1]][0]
            if same:                                       # dz = s g B and dB^T = s z^T g for both adapters, one launch each
                z2, _ = rowdot2([g2[grp[0]], g2[grp[1]]], [Bts[grp[0]], Bts[grp[1]]], rank, meta[grp[0]][0], 0.0, [0, 0], 3)
                d2 = colacc2([g2[grp[0]], g2[grp[1]]], [zs[grp[0]], zs[grp[1]]], None, rank, meta[grp[0]][0], 3)
                for k, i in enumerate(grp):
                    dz[i], dBt[i] = z2[k], d2[k]
            else:
                for i in grp:
                    dz[i] = rowdot2([g2[i]], [Bts[i]], rank, meta[i][0], 0.0, [0], 1)[0][0]
                    dBt[i] = colacc2([g2[i]], [zs[i]], None, rank, meta[i][0], 1)[0]
        for grp in _pairs(live):                           # dA = dz^T (mask x) / (1-p): x streamed once per pair
            p = meta[grp[0]][1]
            both = len(grp) == 2 and rank == 8 and meta[grp[1]][1] == p and (bits[grp[0]] is None) == (bits[grp[1]] is None)
            if both:
                bt = [bits[grp[0]], bits[grp[1]]] if bits[grp[0]] is not None else None
                d2 = colacc2([x2], [dz[grp[0]], dz[grp[1]]], bt, rank, 1.0 / (1.0 - p), 2)
                dA[grp[0]], dA[grp[1]] = d2[0], d2[1]
            else:
                for i in grp:
                    pi = meta[i][1]
                    dA[i] = colacc2([x2], [dz[i]], [bits[i]] if bits[i] is not None else None, rank, 1.0 / (1.0 - pi), 1)[0]
        dx = None
        if ctx.needs_input_grad[0]:
            from . import frozen_linear

            for i in range(n):                             # the base GEMMs accumulate into one dx
                if g2[i] is None:
                    continue
                # frozen weights: through the transposed copy (the forward's GEMM layout: frozen_linear.py); ctx.w_params holds
                # the PARAMETERS (the cache lives on them), wc the compute-dtype tensors the forward used
                wt = frozen_linear.dgrad_weight(ctx.w_params[i], x2.dtype)
                if dx is None:
                    dx = torch.nn.functional.linear(g2[i], wt) if wt is not None else torch.mm(g2[i], wc[i])
                elif wt is not None:
                    dx.addmm_(g2[i], wt.t())
                else:
                    dx.addmm_(g2[i], wc[i])
            if dx is None:
                dx = torch.zeros_like(x2)
            for grp in _pairs(live):                       # += mask (dz A) / (1-p), every adapter of a pair in one pass
                p = meta[grp[0]][1]
                both = len(grp) == 2 and rank == 8 and meta[grp[1]][1] == p and (bits[grp[0]] is None) == (bits[grp[1]] is None)
                if both:
                    bt = [bits[grp[0]], bits[grp[1]]] if bits[grp[0]] is not None else None
                    rankupd2_([dx], [dz[grp[0]], dz[grp[1]]], [As[grp[0]], As[grp[1]]], bt, rank, 1.0 / (1.0 - p), 2)
                else:
                    for i in grp:
                        pi = meta[i][1]
                        rankupd2_([dx], [dz[i]], [As[i]], [bits[i]] if bits[i] is not None else None, rank, 1.0 / (1.0 - pi), 1)
            dx = dx.view(ctx.xshape)
            if dx.dtype != ctx.xdtype:
                dx = dx.to(ctx.xdtype)
        grads = [dx, None]
        for i in range(n):
            grads += [None, None, dA.get(i), dBt.get(i)]
        return tuple(grads)


def lora_group_forward(x, members):
    """members: list of (weight, bias, lora_A or None, lora_B or None, scaling, p, salt).  Returns the tuple of outputs."""
    from .bert_ops import twin

    x = twin(x)               # an f32 LayerNorm output carrying its bf16 copy (bert_ops): no second cast, same autograd node
    meta, flat = [], []
    for (w, bias, a, b, scaling, p, salt) in members:
        if a is None:
            meta.append(None)
            flat += [w, bias, None, None]
        else:
            meta.append((float(scaling), float(p), int(salt) & 0xFFFFFFFF))
            flat += [w, bias, a, _bt(b)]
    return _LoRAGroupFn.apply(x, tuple(meta), *flat)


def group_supported(x, members) -> bool:
    """All members plain frozen Linears (bf16 / f32 weights), adapters of one rank with f32 contiguous A and [r, N]-major B, and an
    activation the stacked kernels take (bf16, or autocast to bf16)."""
    if not x.is_cuda:
        return False
    cdt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
    if cdt != torch.bfloat16:
        return False
    ranks = set()
    for (w, bias, a, b, scaling, p, salt) in members:
        if w.requires_grad or (bias is not None and bias.requires_grad) or w.dtype not in (torch.float32, torch.bfloat16):
            return False
        if a is not None:
            if not (a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.t().is_contiguous()):
                return False
            if a.shape[1] % 32 != 0 or b.shape[0] % 8 != 0:
                return False
            ranks.add(a.shape[0])
    import os

    return len(ranks) == 1 and next(iter(ranks)) in (8, 16) and os.environ.get("DALM_LORA_V2", "1") != "0"
