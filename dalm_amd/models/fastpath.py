"""Plumbing-level speed-ups of the HF towers that keep the math: no new kernels, only PyTorch-ROCm natives.

`use_native_rms_norm(model)`: transformers' *RMSNorm modules (Llama, Mistral, ...) spell the norm as
~7 eager elementwise kernels forward and ~10 backward (upcast, pow, mean, rsqrt, mul, downcast, mul).
`torch.nn.functional.rms_norm` computes the same `w * x * rsqrt(mean(x^2) + eps)` (fp32 internally) in
one fused kernel each way: 458 -> 124 us per norm (fwd+bwd, [4608, 4096] bf16 on MI355X), about 22 ms of
a 230 ms cfg3 step.  Differences are bf16 rounding-order only (max 1 bf16 ulp).  DALM_NATIVE_NORM=0 disables.
"""
from __future__ import annotations

import os
import types

import torch
import torch.nn.functional as F


def _native_forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
    return F.rms_norm(hidden_states, (hidden_states.shape[-1],), self.weight, self._dalm_eps)


def use_native_rms_norm(model: torch.nn.Module) -> int:
    """Patch every `*RMSNorm` module with a plain weight vector; returns how many were patched."""
    if os.environ.get("DALM_NATIVE_NORM", "1") == "0":
        return 0
    n = 0
    for mod in model.modules():
        if not type(mod).__name__.endswith("RMSNorm"):
            continue
        w = getattr(mod, "weight", None)
        eps = getattr(mod, "variance_epsilon", getattr(mod, "eps", None))
        if w is None or w.dim() != 1 or eps is None:
            continue
        if "Gemma" in type(mod).__name__:  # (1 + w) parameterisation: not the same formula
            continue
        mod._dalm_eps = float(eps)
        mod.forward = types.MethodType(_native_forward, mod)
        n += 1
    return n
