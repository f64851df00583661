"""`dropout + residual add + LayerNorm` of a BERT encoder layer as one HIP launch per direction (`dalm_bert_add_norm_{fwd,bwd}`,
dalm_amd/csrc/bert.hip), for the configuration the trainers and bench.py run the retriever in: bf16 autocast over a frozen
(LoRA) base.

transformers' BertSelfOutput / BertOutput compute `LayerNorm(dropout(dense(h)) + input_tensor)`; the reference reaches them
through `self.retriever_model(...)` (dalm/models/rag_e2e_base_model.py:84-93).  Under autocast the dense output is bf16, the
residual stream and the LayerNorm are f32 (autocast runs layer_norm in f32), and every consumer GEMM casts the f32 result back to
bf16.  The kernel rounds at those points and hands back BOTH forms: `y32` (the module's output, exactly the eager dtype) and
`y16 = bf16(y32)`, attached to `y32` as `_dalm_bf16` - the consumers in this package (`frozen_linear.FrozenLinearT`, the LoRA
group node) take the twin instead of casting again (`twin`), which keeps autograd exact: the twin is a second output of the same
node, its gradient is added to y32's in the backward kernel.

Dropout (BERT's hidden_dropout_prob, training mode) comes from this library's generator (seed word + salt + element index, mask
kept as bits; oracle/lora_mask.py::keep_mask_v2) - torch's philox stream has no reference to match.  Needs its own RNG state to be
re-run: not supported under activation recompute (DALM_BERT_KERNELS=0 restores transformers' modules).
"""
from __future__ import annotations

import os

import torch

from .. import hip


def twin(x: torch.Tensor) -> torch.Tensor:
    """The bf16 copy a `_BertAddNorm` output carries, when the caller is about to cast x to bf16 anyway (bf16 autocast)."""
    t = getattr(x, "_dalm_bf16", None)
    if t is not None and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == t.dtype \
            and t.shape == x.shape and x._version == getattr(x, "_dalm_bf16_version", -1) and t._version == getattr(x, "_dalm_bf16_tversion", -1):
        return t                  # (an in-place write to either tensor since they were made: the twin is stale, cast instead)
    return x


class _BertAddNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, res, w, b, eps, p, salt):
        D = a.shape[-1]
        a2, r2 = a.reshape(-1, D), res.reshape(-1, D)
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        r2 = r2 if r2.is_contiguous() else r2.contiguous()
        R = a2.shape[0]
        dev = a.device
        y32 = torch.empty((R, D), dtype=torch.float32, device=dev)
        y16 = torch.empty((R, D), dtype=torch.bfloat16, device=dev)
        mean = torch.empty(R, dtype=torch.float32, device=dev)
        rstd = torch.empty(R, dtype=torch.float32, device=dev)
        bits, seed = None, None
        if p > 0.0:
            from . import lora_ops

            bits = torch.empty((R, D // 8), dtype=torch.uint8, device=dev)
            seed = lora_ops.dropout_seed(dev)         # read in this launch only: the backward reads the stored bits
        hip.call("dalm_bert_add_norm_fwd", hip.ptr(a2), hip.ptr(r2), hip.ptr(w), hip.ptr(b), int(w.dtype == torch.bfloat16), R, D,
                 float(eps), float(p), hip.ptr(seed), int(salt) & 0xFFFFFFFF, hip.ptr(y32), hip.ptr(y16), hip.ptr(bits), hip.ptr(mean),
                 hip.ptr(rstd), hip.stream())
        ctx.save_for_backward(a2, r2, w, mean, rstd, *([bits] if bits is not None else []))
        ctx.p, ctx.shape = float(p), a.shape
        ctx.set_materialize_grads(False)
        return y32.view(a.shape), y16.view(a.shape)

    @staticmethod
    def backward(ctx, g32, g16):
        sv = ctx.saved_tensors
        a2, r2, w, mean, rstd = sv[:5]
        bits = sv[5] if len(sv) > 5 else None
        R, D = a2.shape
        if g32 is None and g16 is None:
            return None, None, None, None, None, None, None
        if g32 is not None:
            g32 = g32.reshape(R, D)
            g32 = g32 if (g32.dtype == torch.float32 and g32.is_contiguous()) else g32.float().contiguous()
        if g16 is not None:
            g16 = g16.reshape(R, D)
            g16 = g16 if (g16.dtype == torch.bfloat16 and g16.is_contiguous()) else g16.to(torch.bfloat16).contiguous()
        d_res = torch.empty((R, D), dtype=torch.float32, device=a2.device)
        d_a = torch.empty((R, D), dtype=torch.bfloat16, device=a2.device)
        hip.call("dalm_bert_add_norm_bwd", hip.ptr(g32), hip.ptr(g16), hip.ptr(a2), hip.ptr(r2), hip.ptr(w),
                 int(w.dtype == torch.bfloat16), hip.ptr(bits), hip.ptr(mean), hip.ptr(rstd), R, D, ctx.p, hip.ptr(d_res), hip.ptr(d_a),
                 hip.stream())
        return d_a.view(ctx.shape), d_res.view(ctx.shape), None, None, None, None, None


def supported(a: torch.Tensor, res: torch.Tensor, ln: torch.nn.Module) -> bool:
    """bf16 dense output + f32 residual (what bf16 autocast produces), a frozen affine LayerNorm over the last dimension."""
    if os.environ.get("DALM_BERT_KERNELS", "1") == "0":
        return False
    w, b = getattr(ln, "weight", None), getattr(ln, "bias", None)
    if not isinstance(ln, torch.nn.LayerNorm) or w is None or b is None or w.requires_grad or b.requires_grad:
        return False
    D = a.shape[-1]
    return (a.is_cuda and a.dtype == torch.bfloat16 and res.dtype == torch.float32 and res.shape == a.shape and res.device == a.device
            and tuple(ln.normalized_shape) == (D,) and D % 8 == 0 and D <= 2048 and w.dtype == b.dtype
            and w.dtype in (torch.float32, torch.bfloat16) and w.is_contiguous() and b.is_contiguous()
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def add_norm(a, res, ln: torch.nn.LayerNorm, p: float, salt: int) -> torch.Tensor:
    """LayerNorm(dropout_p(a) + res) -> the f32 result carrying its bf16 twin."""
    y32, y16 = _BertAddNorm.apply(a, res, ln.weight, ln.bias, float(ln.eps), float(p), int(salt))
    y32._dalm_bf16 = y16
    y32._dalm_bf16_version, y32._dalm_bf16_tversion = y32._version, y16._version
    return y32
