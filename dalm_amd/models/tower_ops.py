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


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, up):
        hip.require_gpu(gate, up)
        gate, up = gate.contiguous(), up.contiguous()
        act = torch.empty_like(gate)
        hip.call("dalm_swiglu_fwd", hip.ptr(gate), hip.ptr(up), hip.ptr(act), hip.dtype_code(gate), gate.numel(), hip.stream())
        ctx.save_for_backward(gate, up)
        return act

    @staticmethod
    def backward(ctx, d_act):
        gate, up = ctx.saved_tensors
        d_act = d_act.contiguous()
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
