"""Autograd wrappers of `dalm_rope_qk` / `dalm_swiglu_{fwd,bwd}` (dalm_amd/csrc/tower.hip): the rotary embedding of q and k
and the SwiGLU activation of a Llama-family decoder layer as ONE HIP launch per direction each.

transformers evaluates both as chains of eager elementwise ops (modeling_llama.py `apply_rotary_pos_emb`,
`LlamaMLP.forward`; the reference reaches them through `self.generator_model(...)`,
dalm/models/rag_e2e_base_model.py:104-106).  The kernels round where those chains round, so forward values and gradients
are the eager chain's (tests/test_tower_ops_gpu.py asserts equality, not closeness).

GPU tensors only: the callers in `fastpath.py` keep transformers' own code for CPU tensors.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import hip


def _strides3(t: torch.Tensor):
    return (C.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def _rope_launch(q, k, cos, sin, backward: bool):
    """q, k: [B, H, T, hd] (any b/h/t strides, contiguous last dim); cos, sin: [B or 1, T, hd]."""
    hip.require_gpu(q, k, cos, sin)
    if q.stride(-1) != 1:
        q = q.contiguous()
    if k.stride(-1) != 1:
        k = k.contiguous()
    qo, ko = torch.empty_like(q), torch.empty_like(k)       # preserve_format: same strides as the (dense) views
    if qo.stride(-1) != 1 or ko.stride(-1) != 1:            # overlapping / exotic input layout: plain contiguous outputs
        qo = torch.empty(q.shape, dtype=q.dtype, device=q.device)
        ko = torch.empty(k.shape, dtype=k.dtype, device=k.device)
    B, Hq, T, hd = q.shape
    Hk = k.shape[1]
    cs = (C.c_int64 * 2)(cos.stride(0) if cos.shape[0] > 1 else 0, cos.stride(1))   # [1, T, hd]: one table for the batch
    hip.call("dalm_rope_qk", hip.ptr(q), hip.ptr(k), hip.ptr(qo), hip.ptr(ko), hip.ptr(cos), hip.ptr(sin), hip.dtype_code(q),
             B, T, Hq, Hk, hd, _strides3(q), _strides3(k), _strides3(qo), _strides3(ko), cs, int(backward), hip.stream())
    return qo, ko


class _RopeQK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, cos, sin):
        ctx.save_for_backward(cos, sin)
        return _rope_launch(q, k, cos, sin, False)

    @staticmethod
    def backward(ctx, gq, gk):
        cos, sin = ctx.saved_tensors
        dq, dk = _rope_launch(gq, gk, cos, sin, True)
        return dq, dk, None, None


def rope_supported(q, k, cos, sin) -> bool:
    return (q.is_cuda and q.dim() == 4 and k.dim() == 4 and cos.dim() == 3 and sin.shape == cos.shape
            and q.dtype in (torch.float32, torch.bfloat16) and k.dtype == q.dtype and cos.dtype == q.dtype
            and sin.dtype == q.dtype and q.shape[-1] % 2 == 0 and k.shape[-1] == q.shape[-1]
            and q.shape[0] == k.shape[0] and q.shape[2] == k.shape[2]
            and cos.shape[0] in (1, q.shape[0]) and cos.shape[1:] == (q.shape[2], q.shape[3]) and cos.stride(-1) == 1
            and sin.stride() == cos.stride()
            and not cos.requires_grad and not sin.requires_grad)


def rope_qk(q, k, cos, sin):
    """(q*cos + rotate_half(q)*sin, k*cos + rotate_half(k)*sin) with cos / sin broadcast over the head dimension."""
    return _RopeQK.apply(q, k, cos, sin)


def _halves_of_one(gate: torch.Tensor, up: torch.Tensor) -> bool:
    """gate and up are the two column halves of ONE contiguous [R, 2 C] buffer (the output of the concatenated gate | up GEMM,
    frozen_linear.pair_forward)."""
    if gate.dim() < 2 or gate.shape != up.shape or gate.stride() != up.stride() or gate.stride(-1) != 1:
        return False
    C = gate.shape[-1]
    el = gate.element_size()
    lead = 1
    for d in range(gate.dim() - 2, -1, -1):            # rows must be laid out as one [R, 2 C] matrix
        if gate.shape[d] != 1 and gate.stride(d) != 2 * C * lead:
            return False
        lead *= gate.shape[d]
    return up.data_ptr() == gate.data_ptr() + C * el and C % (16 // el) == 0 and gate.data_ptr() % 16 == 0


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, up):
        hip.require_gpu(gate, up)
        ctx.halves = _halves_of_one(gate, up)
        if ctx.halves:                     # strided views of one [R, 2 C] GEMM output: read in place, no .contiguous() copies
            C = gate.shape[-1]
            R = gate.numel() // C
            act = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
            hip.call("dalm_swiglu_fwd_2d", hip.ptr(gate), hip.ptr(up), hip.ptr(act), hip.dtype_code(gate), R, C, 2 * C, 2 * C, C,
                     hip.stream())
            ctx.save_for_backward(gate, up)
            return act
        gate, up = gate.contiguous(), up.contiguous()
        act = torch.empty_like(gate)
        hip.call("dalm_swiglu_fwd", hip.ptr(gate), hip.ptr(up), hip.ptr(act), hip.dtype_code(gate), gate.numel(), hip.stream())
        ctx.save_for_backward(gate, up)
        return act

    @staticmethod
    def backward(ctx, d_act):
        gate, up = ctx.saved_tensors
        d_act = d_act.contiguous()
        if ctx.halves:
            # gate / up read in place; d_gate and d_up leave as two CONTIGUOUS tensors: the two dgrad GEMMs then are the shapes
            # (and leading dimensions) the tuned solution table holds - gradient halves of one [R, 2 C] buffer (lda = 2 C) cost
            # the padded cfg3 step 20 ms (148 vs 128 ms/step, library-default solutions for the strided operand)
            C = gate.shape[-1]
            R = gate.numel() // C
            dg = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
            du = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
            hip.call("dalm_swiglu_bwd_2d", hip.ptr(d_act), hip.ptr(gate), hip.ptr(up), hip.ptr(dg), hip.ptr(du),
                     hip.dtype_code(gate), R, C, C, 2 * C, 2 * C, C, C, hip.stream())
            return dg, du
        dg, du = torch.empty_like(gate), torch.empty_like(up)
        hip.call("dalm_swiglu_bwd", hip.ptr(d_act), hip.ptr(gate), hip.ptr(up), hip.ptr(dg), hip.ptr(du),
                 hip.dtype_code(gate), gate.numel(), hip.stream())
        return dg, du


def swiglu_supported(gate, up) -> bool:
    return (gate.is_cuda and gate.dtype in (torch.float32, torch.bfloat16) and up.dtype == gate.dtype
            and up.shape == gate.shape)


def swiglu(gate, up):
    """silu(gate) * up."""
    return _SwiGLU.apply(gate, up)


# ---------------------------------------------------------------------------
# RMSNorm, optionally fused with the residual add in front of it (dalm_rms_norm_{fwd,bwd})
# ---------------------------------------------------------------------------
def rms_norm_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    D = x.shape[-1]
    vec = 4 if x.dtype == torch.float32 else 8
    # w.dtype == x.dtype: eager LlamaRMSNorm with an f32 weight and bf16 activations promotes the product to f32 - another
    # output dtype and rounding than the kernel's: those modules keep transformers' code
    return (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and w.dim() == 1 and w.shape[0] == D
            and w.dtype == x.dtype and D % vec == 0 and D <= 64 * vec * 16)


def _norm_fwd(x2, delta2, w, eps):
    R, D = x2.shape
    y = torch.empty_like(x2)
    h = torch.empty_like(x2) if delta2 is not None else None
    rstd = torch.empty(R, device=x2.device, dtype=torch.float32)
    hip.call("dalm_rms_norm_fwd", hip.ptr(x2), hip.ptr(delta2), hip.ptr(w), hip.dtype_code(x2), R, D, float(eps), hip.ptr(h),
             hip.ptr(y), hip.ptr(rstd), hip.stream())
    return h, y, rstd


def _norm_bwd(dy2, h2, w, rstd, dres2):
    R, D = h2.shape
    dx = torch.empty_like(h2)
    hip.call("dalm_rms_norm_bwd", hip.ptr(dy2), hip.ptr(h2), hip.ptr(w), hip.ptr(rstd), hip.ptr(dres2), hip.dtype_code(h2), R, D,
             hip.ptr(dx), hip.stream())
    return dx


def _weight_grad(dy2, h2, rstd):
    # the normalised value rounded to the activation dtype first, as the eager chain does before its weight multiply
    xh = (h2.float() * rstd.unsqueeze(1)).to(h2.dtype)
    return (dy2.float() * xh.float()).sum(0)


def _as_rows(t, D, dtype):
    t = t.reshape(-1, D)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class _AddRmsNorm(torch.autograd.Function):
    """(h, y) = (x + delta, rmsnorm(x + delta) * w);  with delta None: h IS x (returned as is) and only the backward is fused:
    the gradient that reaches h through the residual path is added inside the norm's backward kernel."""

    @staticmethod
    def forward(ctx, x, delta, w, eps):
        D = x.shape[-1]
        w_c = w if w.dtype == x.dtype else w.to(x.dtype)
        x2 = _as_rows(x, D, x.dtype)
        d2 = _as_rows(delta, D, x.dtype) if delta is not None else None
        h2, y2, rstd = _norm_fwd(x2, d2, w_c, eps)
        ctx.save_for_backward(h2 if h2 is not None else x2, w_c, rstd)
        ctx.shape, ctx.has_delta, ctx.w_dtype = x.shape, delta is not None, w.dtype
        h = h2.view(x.shape) if h2 is not None else x
        return h, y2.view(x.shape)

    @staticmethod
    def backward(ctx, dh, dy):
        h2, w_c, rstd = ctx.saved_tensors
        D = h2.shape[1]
        dy2 = _as_rows(dy, D, h2.dtype)
        dres2 = _as_rows(dh, D, h2.dtype) if dh is not None else None
        d = _norm_bwd(dy2, h2, w_c, rstd, dres2).view(ctx.shape)
        dw = _weight_grad(dy2, h2, rstd).to(ctx.w_dtype) if ctx.needs_input_grad[2] else None
        return d, (d if ctx.has_delta else None), dw, None


def add_rms_norm(x, delta, w, eps):
    """x + delta and its RMSNorm in one launch (one more in the backward, which also folds in the residual-path gradient).
    delta = None: (x itself, rmsnorm(x) * w)."""
    return _AddRmsNorm.apply(x, delta, w, eps)


# ---------------------------------------------------------------------------
# Falcon decoder layer: LayerNorm, erf-GELU and the three-way residual add (dalm_amd/csrc/falcon.hip), bf16 activations
# ---------------------------------------------------------------------------
def layer_norm_supported(x: torch.Tensor, w: torch.Tensor, b) -> bool:
    D = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.bfloat16 and w.dim() == 1 and w.shape[0] == D and w.dtype == torch.bfloat16
            and (b is None or (b.dtype == torch.bfloat16 and b.shape == w.shape)) and D % 8 == 0 and D <= 8192)


class _LayerNormRes(torch.autograd.Function):
    """(x, y) = (x, LayerNorm(x) in the activation dtype); x is handed back so that the gradient reaching it through the
    residual path is added INSIDE the norm's backward kernel (as `_AddRmsNorm` with delta None)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        D = x.shape[-1]
        x2 = _as_rows(x, D, x.dtype)
        R = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(R, device=x.device, dtype=torch.float32)
        rstd = torch.empty(R, device=x.device, dtype=torch.float32)
        hip.call("dalm_layer_norm_fwd", hip.ptr(x2), hip.ptr(w), hip.ptr(b), R, D, float(eps), hip.ptr(y), hip.ptr(mean),
                 hip.ptr(rstd), hip.stream())
        ctx.save_for_backward(x2, w, mean, rstd)
        ctx.shape, ctx.has_bias = x.shape, b is not None
        return x, y.view(x.shape)

    @staticmethod
    def backward(ctx, dres, dy):
        x2, w, mean, rstd = ctx.saved_tensors
        R, D = x2.shape
        if dy is None:
            return dres, None, None, None
        dy2 = _as_rows(dy, D, x2.dtype)
        dres2 = _as_rows(dres, D, x2.dtype) if dres is not None else None
        dx = torch.empty_like(x2)
        hip.call("dalm_layer_norm_bwd", hip.ptr(dy2), hip.ptr(x2), hip.ptr(w), hip.ptr(mean), hip.ptr(rstd), hip.ptr(dres2), R, D,
                 hip.ptr(dx), hip.stream())
        dw = db = None
        if ctx.needs_input_grad[1]:
            xh = (x2.float() - mean.unsqueeze(1)) * rstd.unsqueeze(1)
            dw = (dy2.float() * xh).sum(0).to(w.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.float().sum(0).to(w.dtype)
        return dx.view(ctx.shape), dw, db, None


def layer_norm_res(x, w, b, eps):
    """(x, LayerNorm(x)): one launch forward, one backward (which also folds in the residual-path gradient)."""
    return _LayerNormRes.apply(x, w, b, eps)


def flat_bf16_supported(*ts) -> bool:
    t0 = ts[0]
    return all(t.is_cuda and t.dtype == torch.bfloat16 and t.shape == t0.shape for t in ts) and t0.numel() % 8 == 0


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        hip.require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        hip.call("dalm_gelu_fwd", hip.ptr(x), hip.ptr(y), x.numel(), hip.stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        hip.call("dalm_gelu_bwd", hip.ptr(dy), hip.ptr(x), hip.ptr(dx), x.numel(), hip.stream())
        return dx


def gelu(x):
    """0.5 x (1 + erf(x / sqrt 2)), bf16 in and out, f32 arithmetic."""
    return _Gelu.apply(x)


class _Add3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        hip.require_gpu(a, b, c)
        a, b, c = a.contiguous(), b.contiguous(), c.contiguous()
        out = torch.empty_like(a)
        hip.call("dalm_add3", hip.ptr(a), hip.ptr(b), hip.ptr(c), hip.ptr(out), a.numel(), hip.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g, g


def add3(a, b, c):
    """c + (a + b), the inner sum rounded to bf16 first (what `a += b; c + a` leaves)."""
    return _Add3.apply(a, b, c)
