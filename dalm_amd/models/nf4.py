"""`use_bnb`: the frozen base weights in 4-bit NormalFloat storage, on the library's own kernels.

The reference passes `BitsAndBytesConfig(load_in_4bit=True, bnb_4bit_quant_type="nf4",
bnb_4bit_compute_dtype=torch.bfloat16)` to `from_pretrained` (dalm/models/rag_e2e_base_model.py:50-58,137-142;
retriever_only_base_model.py:23-27,84-90): transformers swaps every `nn.Linear` outside the output head for a
bitsandbytes `Linear4bit`, which stores blocks of 64 weights as 4-bit NF4 indices + one f32 absmax and, per call, casts
the activations to bfloat16, dequantises the weight to bfloat16, multiplies, and casts the result back.  bitsandbytes is a
CUDA library that is not in this image; `NF4Linear` is the same storage format and the same compute recipe on
`dalm_nf4_quantize` / `dalm_nf4_dequantize` (dalm_amd/csrc/nf4.hip).  The weight is dequantised again in the backward
pass (only the activation gradient exists - the quantised weight is frozen), so 0.5625 bytes per weight stay resident
and one [out,in] bf16 scratch is live at a time.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
from torch import nn

from .. import hip

BLOCK = 64
COMPUTE_DTYPE = torch.bfloat16


def quantize(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[*] f32/bf16/f16 on the GPU -> (packed uint8 [ceil(n/2)], absmax f32 [ceil(n/64)]) over the flattened tensor."""
    hip.require_gpu(w)
    if w.dtype not in (torch.float32, torch.bfloat16):
        w = w.float()
    w = w.contiguous()
    n = w.numel()
    packed = torch.empty((n + 1) // 2, device=w.device, dtype=torch.uint8)
    absmax = torch.empty((n + BLOCK - 1) // BLOCK, device=w.device, dtype=torch.float32)
    hip.call("dalm_nf4_quantize", hip.ptr(w), hip.dtype_code(w), n, hip.ptr(packed), hip.ptr(absmax), hip.stream())
    return packed, absmax


def dequantize(packed: torch.Tensor, absmax: torch.Tensor, shape, dtype: torch.dtype = COMPUTE_DTYPE) -> torch.Tensor:
    hip.require_gpu(packed, absmax)
    if dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("nf4 dequantises to float32 or bfloat16")
    # the kernel reads raw pointers: a quant state that went through model.half() / .to(bfloat16) would be read as garbage
    if absmax.dtype != torch.float32 or packed.dtype != torch.uint8:
        raise TypeError(f"nf4 quant state must stay uint8 / float32 (got {packed.dtype} / {absmax.dtype}): was the module "
                        "cast with .half() / .to(dtype) through something other than NF4Linear._apply?")
    if not (absmax.is_contiguous() and packed.is_contiguous()):
        raise ValueError("nf4 quant state must be contiguous")
    n = 1
    for d in shape:
        n *= int(d)
    if packed.numel() != (n + 1) // 2 or absmax.numel() != (n + BLOCK - 1) // BLOCK:
        raise ValueError(f"nf4 quant state does not match shape {tuple(shape)}")
    out = torch.empty(tuple(shape), device=packed.device, dtype=dtype)
    hip.call("dalm_nf4_dequantize", hip.ptr(packed), hip.ptr(absmax), out.numel(), hip.dtype_code(out), hip.ptr(out),
             hip.stream())
    return out


class _NF4MatMul(torch.autograd.Function):
    """y = x . dequant(W)^T with W frozen: nothing but the 4-bit storage is kept for the backward pass."""

    @staticmethod
    def forward(ctx, x, packed, absmax, shape, bias):
        w = dequantize(packed, absmax, shape, x.dtype)
        ctx.save_for_backward(packed, absmax)
        ctx.shape, ctx.has_bias = shape, bias is not None
        return torch.nn.functional.linear(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        packed, absmax = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dy.matmul(dequantize(packed, absmax, ctx.shape, dy.dtype))
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_bias and ctx.needs_input_grad[4] else None
        return dx, None, None, None, db


class NF4Linear(nn.Module):
    """`nn.Linear` with the weight held as NF4 (buffers `qweight`, `absmax`); the bias stays in its own dtype."""

    def __init__(self, base: nn.Linear):
        super().__init__()
        self.in_features, self.out_features = base.in_features, base.out_features
        packed, absmax = quantize(base.weight.detach())
        self.register_buffer("qweight", packed)
        self.register_buffer("absmax", absmax)
        self.bias = None if base.bias is None else nn.Parameter(base.bias.detach().clone(), requires_grad=False)

    def _apply(self, fn, recurse=True):
        """Dtype casts (model.half(), .to(torch.bfloat16), .float()) leave the quant state alone - bitsandbytes keeps its
        absmax out of them in the same way; device moves still apply."""
        absmax = self.absmax
        super()._apply(fn, recurse)
        if self.absmax.dtype != torch.float32:
            self.absmax = absmax.to(self.absmax.device)
        return self

    @property
    def weight(self) -> torch.Tensor:
        """The dequantised weight (a fresh bf16 tensor per access) - for code that inspects `.weight` (LoRA merge,
        device / shape probes); the module never keeps it."""
        return dequantize(self.qweight, self.absmax, (self.out_features, self.in_features), COMPUTE_DTYPE)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # bitsandbytes' Linear4bit.forward: activations -> compute dtype, result -> the caller's dtype
        xin = x if x.dtype == COMPUTE_DTYPE else x.to(COMPUTE_DTYPE)
        bias = None if self.bias is None else self.bias.to(COMPUTE_DTYPE)
        y = _NF4MatMul.apply(xin, self.qweight, self.absmax, (self.out_features, self.in_features), bias)
        return y if y.dtype == x.dtype else y.to(x.dtype)

    def to_linear(self, dtype: torch.dtype = COMPUTE_DTYPE) -> nn.Linear:
        lin = nn.Linear(self.in_features, self.out_features, bias=self.bias is not None, device=self.qweight.device,
                        dtype=dtype)
        with torch.no_grad():
            lin.weight.copy_(self.weight)
            if self.bias is not None:
                lin.bias.copy_(self.bias)
        lin.requires_grad_(False)
        return lin

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, nf4"


def _own_keep_rule(model: nn.Module) -> List[str]:
    """The output head (and nothing else that is a Linear): what transformers' rule comes to for the architectures of the
    path - a bare encoder (`AutoModel`) has no head and converts every Linear, a causal LM keeps `lm_head`."""
    head = model.get_output_embeddings() if hasattr(model, "get_output_embeddings") else None
    return [] if head is None else [n for n, m in model.named_modules() if m is head]


def modules_kept_in_full_precision(model: nn.Module) -> List[str]:
    """Name patterns of the modules a 4-bit load leaves alone.  The reference passes no `llm_int8_skip_modules`, so
    transformers decides (`get_keys_to_not_convert`: tied weights, the module of the last parameter, the output embedding);
    its own function is used when this transformers build has it, `_own_keep_rule` otherwise (tests/test_nf4.py checks
    that the two convert the same Linears on the golden BERT / Llama / Falcon models)."""
    try:
        from transformers.quantizers.base import get_keys_to_not_convert
    except ImportError:
        try:
            from transformers.integrations.bitsandbytes import get_keys_to_not_convert   # transformers 4.x
        except ImportError:
            get_keys_to_not_convert = None
    if get_keys_to_not_convert is not None and hasattr(model, "get_output_embeddings"):
        try:
            return list(get_keys_to_not_convert(model))
        except Exception:       # a model class without the bookkeeping that function reads
            pass
    return _own_keep_rule(model)


def _skipped(full_name: str, patterns: Iterable[str]) -> bool:
    """transformers' `should_convert_module`, negated: a pattern is a prefix followed by a dot, a (regex) match from the
    start of the name, or a suffix of it."""
    import re

    for key in patterns:
        try:
            if re.match(f"{key}\\.", full_name) or re.match(f"{key}", full_name):
                return True
        except re.error:
            pass
        if full_name.endswith(key):
            return True
    return False


def quantize_linears(model: nn.Module, skip: Optional[Iterable[str]] = None) -> int:
    """Swap every `nn.Linear` of `model` (already on the GPU) outside `skip` for an `NF4Linear`, freeing the original
    weight as it goes.  Returns the number of converted modules."""
    skip = list(modules_kept_in_full_precision(model) if skip is None else skip)
    done = 0
    for full, parent, child_name, child in linears_to_convert(model, skip):
        setattr(parent, child_name, NF4Linear(child))
        done += 1
    if done == 0:
        raise ValueError("use_bnb: no nn.Linear found to quantise")
    model._dalm_nf4 = True
    return done


def _is_plain_linear(m: nn.Module) -> bool:
    """`nn.Linear` itself, or Falcon's `FalconLinear` (y = x W^T + b spelled as a matmul).  transformers 4.x - the
    reference pins `transformers>4.35` - converted every `isinstance(module, nn.Linear)`; transformers 5.x narrowed that to
    the exact type, which would leave a Falcon generator (BASELINE config 5) unquantised.  Other subclasses may override
    forward and are left alone."""
    return type(m) is nn.Linear or (isinstance(m, nn.Linear) and type(m).__name__ == "FalconLinear")


def linears_to_convert(model: nn.Module, skip: Iterable[str]):
    """(full name, parent, attribute, module) of every plain Linear outside the skip patterns."""
    skip = list(skip)
    out = []
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{parent_name}.{child_name}" if parent_name else child_name
            if _is_plain_linear(child) and not _skipped(full, skip):
                out.append((full, parent, child_name, child))
    return out


def weight_bytes(model: nn.Module) -> int:
    """Resident bytes of parameters + buffers (what `use_bnb` is for)."""
    seen, total = set(), 0
    for t in list(model.parameters()) + list(model.buffers()):
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            total += t.numel() * t.element_size()
    return total
