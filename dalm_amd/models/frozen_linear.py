"""Frozen projections with a transposed copy of the weight for the backward GEMM.

Under LoRA (the reference's configuration for 7B generators: dalm/models/rag_e2e_base_model.py:61-80) every base projection is
frozen: its backward is one GEMM, dx = g W with W [N, K] as stored - the library's "NN" layout.  hipBLASLt runs that layout
slower than the forward's ("TN"): measured at the cfg3 shapes with the tuned solution table (profiles/r05_gemm_layout_probe.txt)
    q/k/v/o   162.5 us (NN)  ->  148.6 us through a transposed copy (TN), 135.6 us when it accumulates into an existing dx
    down_proj 376.1 us       ->  327.3 us        gate/up_proj 349.6 us -> 334.7 us
A frozen weight never changes, so the copy W^T [K, N] is made once (lazily, on the weight's device; +1x the frozen weights'
bytes - 13.5 GB for Llama-2-7b in bf16, on a 288 GB GPU) and the backward becomes F.linear(g, W^T).  Same values as autograd's
own backward up to the summation order inside the library's kernel.  DALM_DGRAD_T=0 disables (no copies are made).

Nothing in transformers is patched: frozen `nn.Linear` modules get the subclass below by class swap (same parameters, same
state_dict keys); trainable weights, CPU tensors and inference keep `nn.Linear.forward`.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

_ENABLED = os.environ.get("DALM_DGRAD_T", "1") != "0"


def enabled() -> bool:
    return _ENABLED


import weakref

# the copies live in a side table keyed by the parameter OBJECT's id, not on the parameter: `torch.save(model)` / pickling a
# module does not serialise them, and a parameter that goes away takes its copy with it (a finalizer removes the entry; a
# WeakKeyDictionary cannot hold tensors - its key comparison calls Tensor.__eq__)
_WT_CACHE: dict = {}


def dgrad_weight(w: torch.Tensor, dtype: Optional[torch.dtype] = None) -> Optional[torch.Tensor]:
    """W^T [K, N], contiguous, in `dtype` (default: w's own); None when disabled, when w is trainable or not on a GPU, when the
    copy would not fit comfortably (free HBM < 4 x its size: the backward then runs on W as stored), or when the first request
    for it comes from inside a hipGraph capture (an allocation that must outlive the graph's pool).  The cache is keyed by the
    weight's storage, version, device and dtype: a reloaded or moved weight gets a fresh copy."""
    if not _ENABLED or w.requires_grad or not w.is_cuda or w.dim() != 2:
        return None
    dtype = dtype or w.dtype
    key = (w.data_ptr(), w._version, w.device, dtype, tuple(w.shape))
    cached = _WT_CACHE.get(id(w))
    if cached is not None and cached[0] == key and cached[2]() is w:
        return cached[1]
    if torch.cuda.is_current_stream_capturing():
        return None
    need = w.numel() * torch.empty((), dtype=dtype).element_size()
    try:
        free, _total = torch.cuda.mem_get_info(w.device)
        if free < 4 * need:
            return None
    except Exception:
        pass
    with torch.no_grad():
        wt = w.detach().to(dtype).t().contiguous()
    # the copy is made by kernels on THIS stream and then shared through the cache: a backward of the same module running on
    # another stream (the retriever-only step runs its two towers on two streams) would pick it up before it is written - one
    # host sync per weight, once (seen as a wrong first-step gradient at cfg2: 0.62 instead of 0.34)
    torch.cuda.current_stream(w.device).synchronize()
    try:
        fresh = id(w) not in _WT_CACHE
        _WT_CACHE[id(w)] = (key, wt, weakref.ref(w))
        if fresh:
            weakref.finalize(w, _WT_CACHE.pop, id(w), None)
    except TypeError:       # an object that cannot be weakly referenced: recompute next time
        _WT_CACHE.pop(id(w), None)
    return wt


def drop_dgrad_copy(w: torch.Tensor) -> None:
    _WT_CACHE.pop(id(w), None)


def _compute_dtype(x: torch.Tensor) -> torch.dtype:
    return torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype


class _FrozenLinearFn(torch.autograd.Function):
    """y = x W^T + b with W, b frozen: the backward is dx = F.linear(g, W^T copy)."""

    @staticmethod
    def forward(ctx, x, w, b):
        y = F.linear(x, w, b)                              # under autocast this casts exactly as nn.Linear.forward does
        ctx.w = w
        ctx.cdt = y.dtype
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        w = ctx.w
        wt = dgrad_weight(w, ctx.cdt)
        g2 = g.reshape(-1, w.shape[0])
        if g2.dtype != ctx.cdt:
            g2 = g2.to(ctx.cdt)
        dx = F.linear(g2, wt) if wt is not None else torch.mm(g2, w.to(ctx.cdt))
        dx = dx.view(ctx.xshape)
        return (dx if dx.dtype == ctx.xdtype else dx.to(ctx.xdtype)), None, None


class _FrozenPairFn(torch.autograd.Function):
    """(x W0^T, x W1^T) for two frozen bias-free projections of ONE input (gate_proj / up_proj): the two backward GEMMs
    accumulate into one dx - no separate add of two [rows, K] tensors."""

    @staticmethod
    def forward(ctx, x, w0, w1):
        y0, y1 = F.linear(x, w0), F.linear(x, w1)
        ctx.w = (w0, w1)
        ctx.cdt = y0.dtype
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        ctx.set_materialize_grads(False)
        return y0, y1

    @staticmethod
    def backward(ctx, g0, g1):
        dx = None
        for g, w in zip((g0, g1), ctx.w):
            if g is None:
                continue
            g2 = g.reshape(-1, w.shape[0])              # a view for the column halves of one [rows, 2 N] buffer (row stride 2 N)
            if g2.dtype != ctx.cdt:
                g2 = g2.to(ctx.cdt)
            wt = dgrad_weight(w, ctx.cdt)
            if dx is None:
                dx = F.linear(g2, wt) if wt is not None else torch.mm(g2, w.to(ctx.cdt))
            elif wt is not None:
                dx.addmm_(g2, wt.t())
            else:
                dx.addmm_(g2, w.to(ctx.cdt))
        if dx is None:
            return None, None, None
        dx = dx.view(ctx.xshape)
        return (dx if dx.dtype == ctx.xdtype else dx.to(ctx.xdtype)), None, None


# ---------------------------------------------------------------------------------------------------------------------
# two frozen projections of one input as ONE forward GEMM: x [W0 | W1]^T
# ---------------------------------------------------------------------------------------------------------------------
_CAT = os.environ.get("DALM_CAT_GEMM", "1") != "0"


def _cat_weights(m0: torch.nn.Linear, m1: torch.nn.Linear) -> Optional[torch.Tensor]:
    """[W0; W1] as ONE contiguous [N0 + N1, K] tensor that the two modules' weights are VIEWS of (no second copy of the weights:
    the parameters are re-pointed at the halves, state_dict keys and values unchanged).  Made on first use; re-made when a
    parameter was replaced or moved (its storage is then no longer the cat's).  None when the pair cannot share (devices,
    dtypes, trainable, not on a GPU) or DALM_CAT_GEMM=0.

    Why: hipBLASLt runs the [rows, K] x [K, N] projections on 256 x 256 tiles; one GEMM over 2 N columns fills the last wave of
    tiles better than two over N (measured, tools/gemm_concat_probe.py: gate | up forward 672 -> 621 us at 4608 rows,
    493 -> 425 us at 2944 packed rows; q | k | v 408 -> 355 us at 4608 rows)."""
    w0, w1 = m0.weight, m1.weight
    if not _CAT or w0.requires_grad or w1.requires_grad or not w0.is_cuda or w0.device != w1.device or w0.dtype != w1.dtype \
            or w0.dim() != 2 or w1.dim() != 2 or w0.shape[1] != w1.shape[1]:
        return None
    cat = getattr(m0, "_dalm_cat", None)
    n0, el = w0.shape[0], w0.element_size()
    if cat is not None and cat.dtype == w0.dtype and cat.shape[0] == n0 + w1.shape[0] and w0.data_ptr() == cat.data_ptr() \
            and w1.data_ptr() == cat.data_ptr() + n0 * w0.shape[1] * el and w0.is_contiguous() and w1.is_contiguous():
        return cat
    if torch.cuda.is_current_stream_capturing():
        return None                            # never re-point parameters inside a hipGraph capture
    with torch.no_grad():
        cat = torch.cat((w0.detach(), w1.detach()), dim=0)
        torch.cuda.current_stream(w0.device).synchronize()      # other streams may run these modules next (see dgrad_weight)
        w0.data = cat[:n0]
        w1.data = cat[n0:]
    for w in (w0, w1):                         # transposed dgrad copies made from the old storage are still VALUES-correct, but
        drop_dgrad_copy(w)                     # keyed by pointer: drop them, they are rebuilt on the next backward
    try:
        m0._dalm_cat = cat
    except Exception:
        return None
    return cat


def uncat_weights(model: torch.nn.Module) -> int:
    """Give every projection pair that shares ONE concatenated weight tensor (`_cat_weights`) its own storage again - for callers
    that serialise whole modules tensor by tensor (`save_pretrained` of a frozen full model: safetensors refuses tensors that share
    memory).  The next eligible forward concatenates again.  Returns how many pairs were separated."""
    n = 0
    for m in model.modules():
        if getattr(m, "_dalm_cat", None) is None:
            continue
        cat = m._dalm_cat
        for other in model.modules():
            w = getattr(other, "weight", None)
            if isinstance(w, torch.Tensor) and w.dim() == 2 and w.untyped_storage().data_ptr() == cat.untyped_storage().data_ptr():
                with torch.no_grad():
                    w.data = w.detach().clone()
                drop_dgrad_copy(w)
        try:
            del m._dalm_cat
        except Exception:
            m._dalm_cat = None
        n += 1
    return n


class _FrozenCatPairFn(torch.autograd.Function):
    """(x W0^T, x W1^T) as the two column halves of ONE GEMM output x [W0; W1]^T; backward: the two accumulating dgrad GEMMs of
    `_FrozenPairFn` (one GEMM over the 2 N-deep contraction measured slower at 4608 rows, tools/gemm_concat_probe.py), reading the
    gradient halves in place when they arrive as the halves of one buffer (tower_ops._SwiGLU does that)."""

    @staticmethod
    def forward(ctx, x, cat, w0, w1):
        y = F.linear(x, cat)
        n0 = w0.shape[0]
        ctx.w = (w0, w1)
        ctx.cdt = y.dtype
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        ctx.set_materialize_grads(False)
        return y[..., :n0], y[..., n0:]

    @staticmethod
    def backward(ctx, g0, g1):
        return _FrozenPairFn.backward(ctx, g0, g1) + (None,)


def _eligible(x: torch.Tensor, *mods) -> bool:
    if not (_ENABLED and x.is_cuda and torch.is_grad_enabled() and x.requires_grad):
        return False
    cdt = _compute_dtype(x)
    if cdt not in (torch.bfloat16, torch.float16, torch.float32):
        return False
    for m in mods:
        if m.weight.requires_grad or (m.bias is not None and m.bias.requires_grad) or not m.weight.is_cuda:
            return False
    return True


class FrozenLinearT(torch.nn.Linear):
    """nn.Linear whose backward uses the transposed copy while the layer is frozen."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _eligible(x, self):
            from .bert_ops import twin

            return _FrozenLinearFn.apply(twin(x), self.weight, self.bias)       # an f32 LayerNorm output carrying its bf16 copy
        return super().forward(x)


def pair_forward(x: torch.Tensor, m0: torch.nn.Linear, m1: torch.nn.Linear):
    """(m0(x), m1(x)); one autograd node with an accumulating backward when both are frozen and bias-free."""
    plain = (torch.nn.Linear, FrozenLinearT)
    if m0.bias is None and m1.bias is None and type(m0) in plain and type(m1) in plain and _eligible(x, m0, m1):
        cat = _cat_weights(m0, m1)
        if cat is not None:
            return _FrozenCatPairFn.apply(x, cat, m0.weight, m1.weight)
        return _FrozenPairFn.apply(x, m0.weight, m1.weight)
    return m0(x), m1(x)


def _is_plain_linear(m: torch.nn.Module) -> bool:
    # nn.Linear itself, or Falcon's FalconLinear (y = x W^T + b spelled as a matmul: the same function)
    return type(m) is torch.nn.Linear or (isinstance(m, torch.nn.Linear) and type(m).__name__ == "FalconLinear")


def use_transposed_dgrad(model: torch.nn.Module) -> int:
    """Class-swap the FROZEN plain `nn.Linear` modules of `model` to FrozenLinearT (a model that is fine-tuned in full is left
    alone; the forward still checks per call that the layer is frozen); returns how many were swapped.  LoRA-wrapped projections
    and their siblings keep their own modules: the LoRA group node (models/lora_ops.py) asks `dgrad_weight` itself."""
    if not _ENABLED:
        return 0
    n = 0
    for name, m in model.named_modules():
        if not _is_plain_linear(m) or name.endswith("base_layer"):              # LoRALinear.base_layer: the group node's business
            continue
        if m.weight.requires_grad or (m.bias is not None and m.bias.requires_grad):
            continue
        m.__class__ = FrozenLinearT
        n += 1
    return n
