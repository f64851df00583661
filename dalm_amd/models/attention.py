"""Scaled-dot-product attention of the generator tower with a hand-written HIP BACKWARD (`dalm_attn_bwd`,
dalm_amd/csrc/attn.hip), registered with transformers as the attention implementation "dalm_sdpa".

transformers evaluates a decoder layer's attention through `sdpa_attention_forward`
(transformers/integrations/sdpa_attention.py) -> torch.nn.functional.scaled_dot_product_attention; the reference reaches it through
`self.generator_model(...)` (dalm/models/rag_e2e_base_model.py:104-106) and differentiates it in `loss.backward()`
(dalm/training/rag_e2e/train_rage2e.py:466).  Head widths 128 (Llama-2-7b) and 64 (Falcon-7b, through fastpath's FalconAttention patch).  With HF's boolean mask (causal + left padding) torch runs its memory-efficient
kernels: 60 us forward, 440 us backward per layer at cfg3 (2.7 % of the MFMA peak).  Here

  forward   `dalm_attn_fwd` (one launch, online softmax, writes the rows' log-sum-exp);  DALM_ATTN_FWD_KERNEL=0: torch's own
            memory-efficient kernel called as the aten op so that its log-sum-exp comes back
            (`aten::_scaled_dot_product_efficient_attention`, the values of F.scaled_dot_product_attention bit for bit);
  backward  `dalm_attn_bwd`: two launches.
Both read the mask as packed bits (packed once per mask tensor - every layer passes the same one) and skip dead 32 x 32 tiles.

Attention dropout (BERT's attention_probs_dropout_prob in training mode) is applied inside the kernels, its keep mask regenerated
from (a device seed word, a per-call salt, the element index) and never stored - this library's own generator
(oracle/attn_dropout.py), not torch's philox stream.  DALM_ATTN_DROPOUT=0 sends dropout calls to torch (needed under activation
recompute, torch.utils.checkpoint: the salt is a host call counter, a re-run forward would draw another mask).
Everything the kernels do not take (CPU tensors, other head widths, odd T with dropout, float masks, a KV cache, no gradient wanted) goes
to transformers' own `sdpa_attention_forward`, unchanged.  DALM_ATTN_KERNEL=0 keeps the model on "sdpa".
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from .. import hip
from ..packed import packed_of

NAME = "dalm_sdpa"
_HEAD_DIMS = (64, 128)


class _MaskPack:
    """What the kernels read of one mask tensor: bias (for torch's forward), row / column bit words, live tiles."""

    __slots__ = ("mask", "key", "bias", "rows", "cols", "live", "packed")


_last: list = [None]          # the pack of the mask seen last: one forward pass hands the same tensor object to every layer


def _pack(mask: Optional[torch.Tensor], B: int, H: int, T: int, causal: bool, dtype, device) -> _MaskPack:
    seqs = packed_of(mask)
    if seqs is not None:                       # packed (un-padded) call: the words come from the sequence list (dalm_amd/packed.py)
        pk = _MaskPack()
        pk.mask, pk.key, pk.bias, pk.packed = mask, None, None, seqs
        pk.rows, pk.cols, pk.live = seqs.bits()
        return pk
    key = (None if mask is None else (mask.data_ptr(), mask._version, tuple(mask.shape), tuple(mask.stride())), B, T, causal,
           dtype, device, torch.cuda.current_stream(device).cuda_stream)
    cur = _last[0]
    if cur is not None and cur.mask is mask and cur.key == key:
        return cur
    pk = _MaskPack()
    pk.mask, pk.key = mask, key
    W = (T + 31) // 32
    pk.rows = torch.empty(B * 32 * W * W, dtype=torch.int32, device=device)
    pk.cols = torch.empty_like(pk.rows)
    pk.live = torch.empty(B * W * W, dtype=torch.uint8, device=device)
    pk.bias = None
    pk.packed = None
    if mask is None:
        hip.call("dalm_attn_mask_bits", None, B, T, 0, 0, int(causal), hip.ptr(pk.rows), hip.ptr(pk.cols), hip.ptr(pk.live), hip.stream())
    else:
        m = mask if mask.stride(-1) == 1 else mask.contiguous()
        hip.call("dalm_attn_mask_bits", hip.ptr(m), B, T, m.stride(0), m.stride(2), int(causal), hip.ptr(pk.rows), hip.ptr(pk.cols),
                 hip.ptr(pk.live), hip.stream())
    _last[0] = pk
    return pk


def _torch_bias(pk: _MaskPack, B: int, H: int, T: int, dtype, device):
    """torch's own conversion of a boolean mask (aten convert_boolean_attn_mask): 0 where attended, -inf elsewhere; the last
    dimension's allocation padded to a multiple of 8 elements as its memory-efficient kernel wants the bias aligned."""
    if pk.mask is None:
        return None
    if pk.bias is None:
        m = pk.mask
        Ta = (T + 7) // 8 * 8
        bias = torch.zeros(B, 1, T, Ta, dtype=dtype, device=device)[..., :T]
        bias.masked_fill_(m.logical_not(), float("-inf"))
        pk.bias = bias.expand(B, H, T, T)
    return pk.bias


def _strides3(t: torch.Tensor):
    return [t.stride(0), t.stride(1), t.stride(2)]


def _dense_like(t: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(t)                       # preserve_format: the [B, T, H, hd] memory of the projections' views
    if out.stride(-1) != 1:
        out = torch.empty(t.shape, dtype=t.dtype, device=t.device)
    return out


def _attn_forward(q, k, v, pk, scale, causal, drop=(0.0, None, 0)):
    B, H, T, hd = q.shape
    if pk.packed is not None:                   # q, k, v: [1, H, n, hd] views of [n, H hd] projections; sequences from cu_seqlens
        sq = pk.packed
        out = torch.empty(1, T, H, hd, dtype=q.dtype, device=q.device).transpose(1, 2)
        lse = torch.empty(sq.nseq, H, sq.T, dtype=torch.float32, device=q.device)
        flat = []
        for t in (q, k, v, out):
            flat += _strides3(t)
        hip.call("dalm_attn_fwd_packed", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(pk.rows), hip.ptr(pk.live), hip.ptr(sq.cu),
                 sq.nseq, H, sq.T, hd, float(scale), (C.c_int64 * 12)(*flat), float(drop[0]), hip.ptr(drop[1]),
                 int(drop[2]) & 0xFFFFFFFF, hip.ptr(out), hip.ptr(lse), hip.stream())
        return out, lse
    if os.environ.get("DALM_ATTN_FWD_KERNEL", "1") == "0" and drop[0] == 0.0:    # torch's memory-efficient forward + its log-sum-exp
        out, lse, _, _ = torch.ops.aten._scaled_dot_product_efficient_attention(
            q, k, v, _torch_bias(pk, B, H, T, q.dtype, q.device), True, 0.0, causal, scale=scale)
        return out, lse
    out = torch.empty(B, T, H, hd, dtype=q.dtype, device=q.device).transpose(1, 2)   # torch's layout: the caller's
    lse = torch.empty(B, H, T, dtype=torch.float32, device=q.device)                  # transpose(1, 2).contiguous() is free
    flat = []
    for t in (q, k, v, out):
        flat += _strides3(t)
    hip.call("dalm_attn_fwd", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(pk.rows), hip.ptr(pk.live), B, H, T, hd, float(scale),
             (C.c_int64 * 12)(*flat), float(drop[0]), hip.ptr(drop[1]), int(drop[2]) & 0xFFFFFFFF, hip.ptr(out), hip.ptr(lse),
             hip.stream())
    return out, lse


def _attn_backward(q, k, v, out, lse, d_out, pk, scale, cos=None, sin=None, drop=(0.0, None, 0)):
    B, H, T, hd = q.shape
    if d_out.stride(-1) != 1 or any(s % 8 for s in d_out.stride()[:3]):
        d_out = d_out.contiguous()
    if pk.packed is not None:
        sq = pk.packed
        dq, dk, dv = _dense_like(q), _dense_like(k), _dense_like(v)
        delta = torch.empty(sq.nseq, H, sq.T, dtype=torch.float32, device=q.device)
        flat = []
        for t in (q, k, v, out, d_out, dq, dk, dv):
            flat += _strides3(t)
        hip.call("dalm_attn_bwd_packed", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(out), hip.ptr(d_out), hip.ptr(lse),
                 hip.ptr(pk.rows), hip.ptr(pk.cols), hip.ptr(pk.live), hip.ptr(sq.cu), sq.nseq, H, sq.T, hd, float(scale),
                 (C.c_int64 * 24)(*flat), hip.ptr(cos), hip.ptr(sin), 0 if cos is None else cos.stride(1), float(drop[0]),
                 hip.ptr(drop[1]), int(drop[2]) & 0xFFFFFFFF, hip.ptr(dq), hip.ptr(dk), hip.ptr(dv), hip.ptr(delta), hip.stream())
        return dq, dk, dv
    lse = lse if (lse.is_contiguous() and lse.shape[-1] == T) else lse[..., :T].contiguous()
    dq, dk, dv = _dense_like(q), _dense_like(k), _dense_like(v)
    delta = torch.empty(B, H, T, dtype=torch.float32, device=q.device)
    flat = []
    for t in (q, k, v, out, d_out, dq, dk, dv):
        flat += _strides3(t)
    cs_b = 0 if (cos is None or cos.shape[0] == 1) else cos.stride(0)
    hip.call("dalm_attn_bwd", hip.ptr(q), hip.ptr(k), hip.ptr(v), hip.ptr(out), hip.ptr(d_out), hip.ptr(lse), hip.ptr(pk.rows),
             hip.ptr(pk.cols), hip.ptr(pk.live), B, H, T, hd, float(scale), (C.c_int64 * 24)(*flat), hip.ptr(cos), hip.ptr(sin),
             cs_b, 0 if cos is None else cos.stride(1), float(drop[0]), hip.ptr(drop[1]), int(drop[2]) & 0xFFFFFFFF, hip.ptr(dq),
             hip.ptr(dk), hip.ptr(dv), hip.ptr(delta), hip.stream())
    return dq, dk, dv


class _SdpaHipBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale, causal, dropout_p=0.0, salt=0):
        B, H, T, hd = q.shape
        pk = _pack(mask, B, H, T, causal, q.dtype, q.device)
        drop = (0.0, None, 0)
        if dropout_p > 0.0:
            # this STEP's copy of the device seed word (8 bytes, capturable): the step advances the word itself, a backward that
            # runs after the next advance must still see the forward's value.  Taken per advance, not per mask pack: a pack that
            # is re-used across steps (attention_mask=None: same key every step) must not freeze the seed
            from . import lora_ops

            drop = (float(dropout_p), lora_ops.dropout_seed_snapshot(q.device), int(salt))
        out, lse = _attn_forward(q, k, v, pk, scale, causal, drop)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.pack, ctx.scale, ctx.drop = pk, scale, drop
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = _attn_backward(q, k, v, out, lse, d_out, ctx.pack, ctx.scale, drop=ctx.drop)
        return dq, dk, dv, None, None, None, None, None


class _RopeSdpaHip(torch.autograd.Function):
    """rotary embedding of q and k (`dalm_rope_qk`, one launch) + the attention; the backward of the rotation happens in the
    epilogues of `dalm_attn_bwd` (no launch of its own, the rotated gradients never reach memory)."""

    @staticmethod
    def forward(ctx, q, k, v, cos, sin, mask, scale, causal):
        from . import tower_ops

        B, H, T, hd = q.shape
        q2, k2 = tower_ops._rope_launch(q, k, cos, sin, False)
        pk = _pack(mask, B, H, T, causal, q.dtype, q.device)
        # multi-query (ONE key / value head, Falcon-7B): the kernels read it through stride-0 head views - nothing is broadcast
        # in memory; the backward kernels write per-head dk / dv, summed over the heads below
        ctx.mqa = k.shape[1] == 1 and H > 1
        kx = k2.expand(B, H, T, hd) if ctx.mqa else k2
        vx = v.expand(B, H, T, hd) if ctx.mqa else v
        out, lse = _attn_forward(q2, kx, vx, pk, scale, causal)
        ctx.save_for_backward(q2, k2, v, out, lse, cos, sin)
        ctx.pack, ctx.scale = pk, scale
        return out

    @staticmethod
    def backward(ctx, d_out):
        q2, k2, v, out, lse, cos, sin = ctx.saved_tensors
        if ctx.mqa:
            B, H, T, hd = q2.shape
            dq, dk, dv = _attn_backward(q2, k2.expand(B, H, T, hd), v.expand(B, H, T, hd), out, lse, d_out, ctx.pack, ctx.scale, cos, sin)
            return dq, dk.sum(dim=1, keepdim=True), dv.sum(dim=1, keepdim=True), None, None, None, None, None
        dq, dk, dv = _attn_backward(q2, k2, v, out, lse, d_out, ctx.pack, ctx.scale, cos, sin)
        return dq, dk, dv, None, None, None, None, None


def rope_fusable(q, k, cos, sin) -> bool:
    """cos / sin tables `dalm_attn_bwd` reads in its epilogues: what `tower_ops.rope_supported` takes, bf16, 16-byte rows."""
    from . import tower_ops

    return (tower_ops.rope_supported(q, k, cos, sin) and cos.dtype == torch.bfloat16 and cos.shape[-1] == q.shape[-1]
            and cos.stride(1) % 8 == 0 and (cos.shape[0] == 1 or cos.stride(0) % 8 == 0)
            and cos.data_ptr() % 16 == 0 and sin.data_ptr() % 16 == 0
            and (k.shape == q.shape or (k.shape[1] == 1 and k.shape[0] == q.shape[0] and k.shape[2:] == q.shape[2:])))


def rope_sdpa(query, key, value, cos, sin, mask, scale: float, causal: bool):
    """sdpa(apply_rotary_pos_emb(query, key, cos, sin), value, ...) for what `supported` and `rope_fusable` accept."""
    return _RopeSdpaHip.apply(query, key, value, cos, sin, mask, scale, causal)


def _views_ok(*ts) -> bool:
    return all(t.stride(-1) == 1 and all(s % 8 == 0 for s in t.stride()[:3]) and t.data_ptr() % 16 == 0 for t in ts)


def supported(query, key, value, mask, dropout, causal, kwargs) -> bool:
    if not (query.is_cuda and query.dtype == torch.bfloat16 and key.dtype == query.dtype and value.dtype == query.dtype):
        return False
    if query.dim() != 4 or query.shape[-1] not in _HEAD_DIMS or key.shape != query.shape or value.shape != query.shape:
        return False                                 # a KV cache (kv length != q length) or grouped heads left unexpanded
    if query.shape[2] < 2 or query.shape[2] > 2048 or kwargs.get("position_bias") is not None:
        return False
    if dropout != 0.0 and not (0.0 < dropout < 1.0 and query.shape[2] % 2 == 0 and os.environ.get("DALM_ATTN_DROPOUT", "1") != "0"):
        return False                                 # the in-kernel mask pairs elements (i, j), (i, j + 1): even T
    if not (torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)):
        return False                                 # nothing to differentiate: torch's fused forward alone is the best path
    B, _, T, _ = query.shape
    if mask is None:
        pass                                         # causal, or fully bidirectional (an encoder batch without padding)
    elif not (mask.dtype == torch.bool and mask.dim() == 4 and tuple(mask.shape) == (B, 1, T, T) and mask.is_cuda and not causal):
        return False
    return _views_ok(query, key, value)


def packed_supported(query, key, value, dropout: float = 0.0) -> bool:
    """What `dalm_attn_*_packed` take: [1, H, n, hd] bf16 views with 16-byte aligned rows, head width 64 / 128."""
    if not (query.is_cuda and query.dtype == torch.bfloat16 and key.dtype == query.dtype and value.dtype == query.dtype):
        return False
    if query.dim() != 4 or query.shape[0] != 1 or query.shape[-1] not in _HEAD_DIMS or key.shape != query.shape \
            or value.shape != query.shape:
        return False
    if dropout != 0.0 and not (0.0 < dropout < 1.0 and os.environ.get("DALM_ATTN_DROPOUT", "1") != "0"):
        return False
    return _views_ok(query, key, value) and os.environ.get("DALM_ATTN_KERNEL", "1") != "0"


def _packed_sdpa_torch(query, key, value, seqs, scale: float, dropout: float):
    """The packed attention without the kernels (CPU tensors, fp32, head widths they do not take): re-pad q / k / v to
    [nseq, H, T, hd] by index, torch's SDPA under the mask the sequence list describes, rows gathered back.  A row without a
    live key attends itself and its output is zeroed afterwards - every intermediate stays finite, and the result is what the
    kernels (and torch's memory-efficient kernels on the padded layout) return for such rows: 0."""
    _, H, n, hd = query.shape
    slot, gather = seqs.padded_index()
    nseq, T = seqs.nseq, seqs.T

    def pad(t):                                                   # [1, H, n, hd] -> [nseq, H, T, hd]
        rows = torch.cat((t[0].transpose(0, 1), t.new_zeros((1, H, hd))), dim=0)           # [n + 1, H, hd]
        return rows.index_select(0, gather).view(nseq, T, H, hd).transpose(1, 2)

    kl = torch.cat((seqs.key_live != 0, seqs.key_live.new_zeros((1,), dtype=torch.bool))).index_select(0, gather).view(nseq, 1, 1, T)
    mask = kl.expand(nseq, 1, T, T)
    if seqs.causal:
        mask = mask & torch.ones((T, T), dtype=torch.bool, device=query.device).tril()
    has_key = mask.any(dim=-1, keepdim=True)                                                  # [nseq, 1, T, 1]
    mask = mask | (~has_key & torch.eye(T, dtype=torch.bool, device=query.device))
    out = torch.nn.functional.scaled_dot_product_attention(pad(query), pad(key), pad(value), attn_mask=mask, dropout_p=dropout,
                                                           scale=scale)
    out = torch.where(has_key, out, torch.zeros_like(out))
    return out.transpose(1, 2).reshape(nseq * T, H, hd).index_select(0, slot).unsqueeze(0)  # [1, n, H, hd]


def _packed_attention(module, query, key, value, attention_mask, seqs, dropout, scaling):
    from transformers.integrations.sdpa_attention import repeat_kv

    groups = getattr(module, "num_key_value_groups", 1)
    if groups > 1 and key.shape[1] != query.shape[1]:
        key, value = repeat_kv(key, groups), repeat_kv(value, groups)
    scale = float(scaling) if scaling is not None else float(query.shape[-1]) ** -0.5
    if packed_supported(query, key, value, dropout):
        salt = 0
        if dropout > 0.0:
            module._dalm_attn_calls = getattr(module, "_dalm_attn_calls", 0) + 1
            salt = ((id(module) >> 4) << 12) ^ (module._dalm_attn_calls & 0xFFF)
        out = _SdpaHipBackward.apply(query, key, value, attention_mask, scale, False, float(dropout), salt)
        return out.transpose(1, 2).contiguous(), None
    return _packed_sdpa_torch(query, key, value, seqs, scale, float(dropout)), None


def dalm_sdpa_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0, scaling: Optional[float] = None,
                                is_causal: Optional[bool] = None, **kwargs):
    """Same contract as transformers' `sdpa_attention_forward`: ([B, T, H, hd] output, None)."""
    from transformers.integrations.sdpa_attention import repeat_kv, sdpa_attention_forward

    seqs = packed_of(attention_mask)
    if seqs is not None:                     # packed (un-padded) tower call: `attention_mask` is a descriptor, not a mask
        return _packed_attention(module, query, key, value, attention_mask, seqs, dropout, scaling)
    groups = getattr(module, "num_key_value_groups", 1)
    causal = bool(query.shape[2] > 1 and attention_mask is None
                  and (is_causal if is_causal is not None else getattr(module, "is_causal", True)))
    k2, v2 = (repeat_kv(key, groups), repeat_kv(value, groups)) if (groups > 1 and key.shape[1] != query.shape[1]) else (key, value)
    if os.environ.get("DALM_ATTN_KERNEL", "1") == "0" or not supported(query, k2, v2, attention_mask, dropout, causal, kwargs):
        return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling,
                                      is_causal=is_causal, **kwargs)
    scale = float(scaling) if scaling is not None else float(query.shape[-1]) ** -0.5
    salt = 0
    if dropout > 0.0:      # this module's id in the upper bits, a host call counter below (graph replays re-use the captured salt;
        module._dalm_attn_calls = getattr(module, "_dalm_attn_calls", 0) + 1        # there the device seed word changes the masks)
        salt = ((id(module) >> 4) << 12) ^ (module._dalm_attn_calls & 0xFFF)
    out = _SdpaHipBackward.apply(query, k2, v2, attention_mask, scale, causal, float(dropout), salt)
    return out.transpose(1, 2).contiguous(), None


_registered = [False]


def register() -> bool:
    """Register "dalm_sdpa" with transformers (attention function + the SDPA mask builder).  False when this transformers has
    no such registry or its `sdpa_attention_forward` is not the code the replacement restates."""
    if _registered[0]:
        return True
    try:
        import inspect

        from transformers import AttentionInterface
        from transformers.integrations import sdpa_attention
        from transformers.masking_utils import AttentionMaskInterface, sdpa_mask

        src = inspect.getsource(sdpa_attention.sdpa_attention_forward)
        if not ("torch.nn.functional.scaled_dot_product_attention(" in src and "attn_output.transpose(1, 2).contiguous()" in src
                and "is_causal = q_length > 1 and attention_mask is None and is_causal" in src
                and "return attn_output, None" in src):
            return False
        AttentionInterface.register(NAME, dalm_sdpa_attention_forward)
        AttentionMaskInterface.register(NAME, sdpa_mask)
    except Exception:
        return False
    _registered[0] = True
    return True


def sdpa(query, key, value, mask, scale: float, causal: bool, dropout_p: float = 0.0, salt: int = 0):
    """F.scaled_dot_product_attention(query, key, value, attn_mask=mask, is_causal=causal, scale=scale, dropout_p=dropout_p) for
    what `supported` accepts (the dropout mask comes from this library's own generator, oracle/attn_dropout.py)."""
    return _SdpaHipBackward.apply(query, key, value, mask, scale, causal, dropout_p, salt)


def use_hip_attention_backward(model: torch.nn.Module) -> bool:
    """Switch a Llama-family model (head width 64 or 128) from "sdpa" to "dalm_sdpa".  DALM_ATTN_KERNEL=0 disables."""
    if os.environ.get("DALM_ATTN_KERNEL", "1") == "0":
        return False
    cfg = getattr(model, "config", None)
    if cfg is None or getattr(cfg, "_attn_implementation", None) != "sdpa":
        return False
    if getattr(cfg, "model_type", "") not in ("llama", "mistral", "qwen2", "bert"):
        return False
    hd = getattr(cfg, "head_dim", None) or (cfg.hidden_size // cfg.num_attention_heads)
    if hd not in _HEAD_DIMS or not register():
        return False
    cfg._attn_implementation = NAME
    return True
