"""Minimal in-tree LoRA injector (the image has no `peft`).

Mirrors the configuration the reference asks peft for
(dalm/models/rag_e2e_base_model.py:145-160): r=8, lora_alpha=16, lora_dropout=0.05,
bias="none", on modules whose name ends in one of `target_modules`
(key/query/value for BERT-style retrievers, q_proj/v_proj for Llama/Falcon-style models).
Parameter names follow peft's layout (`base_layer`, `lora_A.default`, `lora_B.default`) so
adapter checkpoints look familiar.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Iterable, List

import torch
from torch import nn

ADAPTER_WEIGHTS = "adapter_model.bin"
ADAPTER_CONFIG = "adapter_config.json"


class LoRALinear(nn.Module):
    def __init__(self, base: nn.Linear, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.05):
        super().__init__()
        self.base_layer = base
        self.r, self.lora_alpha = r, lora_alpha
        self.scaling = lora_alpha / r
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(lora_dropout) if lora_dropout > 0 else nn.Identity()})
        dev = base.weight.device
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False, device=dev, dtype=torch.float32)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False, device=dev, dtype=torch.float32)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B["default"].weight)
        self.merged = False

    @property
    def weight(self):  # some HF code peeks at .weight
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    @property
    def in_features(self):
        return self.base_layer.in_features

    @property
    def out_features(self):
        return self.base_layer.out_features

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.base_layer(x)
        if self.merged:
            return out
        a, b = self.lora_A["default"], self.lora_B["default"]
        z = self.lora_dropout["default"](x)
        if not torch.is_autocast_enabled() and z.dtype != a.weight.dtype:
            z = z.to(a.weight.dtype)  # outside autocast the fp32 adapters need fp32 activations
        return torch.add(out, b(a(z)).to(out.dtype), alpha=self.scaling)  # one kernel for scale + add

    @torch.no_grad()
    def merge(self) -> nn.Linear:
        delta = (self.lora_B["default"].weight @ self.lora_A["default"].weight) * self.scaling
        self.base_layer.weight.add_(delta.to(self.base_layer.weight.dtype))
        return self.base_layer


def _matches(name: str, targets: Iterable[str]) -> bool:
    leaf = name.rsplit(".", 1)[-1]
    return any(leaf == t or name.endswith("." + t) for t in targets)


def inject_lora(model: nn.Module, target_modules: List[str], r: int = 8, lora_alpha: int = 16,
                lora_dropout: float = 0.05) -> nn.Module:
    """Freeze `model`, wrap every matching nn.Linear in a LoRALinear (trainable A/B only)."""
    for prm in model.parameters():
        prm.requires_grad_(False)
    replaced = 0
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{parent_name}.{child_name}" if parent_name else child_name
            if isinstance(child, nn.Linear) and _matches(full, target_modules):
                setattr(parent, child_name, LoRALinear(child, r, lora_alpha, lora_dropout))
                replaced += 1
    if replaced == 0:
        raise ValueError(f"Target modules {target_modules} not found in the base model.")
    model._dalm_lora_config = {"r": r, "lora_alpha": lora_alpha, "lora_dropout": lora_dropout, "bias": "none",
                               "target_modules": list(target_modules), "peft_type": "LORA"}
    return model


def lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in model.state_dict().items() if ".lora_A." in k or ".lora_B." in k}


def has_lora(model: nn.Module) -> bool:
    return any(isinstance(m, LoRALinear) for m in model.modules())


def save_adapter(model: nn.Module, path: str) -> None:
    os.makedirs(path, exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in lora_state_dict(model).items()}, os.path.join(path, ADAPTER_WEIGHTS))
    with open(os.path.join(path, ADAPTER_CONFIG), "w") as f:
        json.dump(getattr(model, "_dalm_lora_config", {}), f, indent=2)


def load_adapter(model: nn.Module, path: str) -> nn.Module:
    with open(os.path.join(path, ADAPTER_CONFIG)) as f:
        cfg = json.load(f)
    if not has_lora(model):
        inject_lora(model, cfg["target_modules"], cfg.get("r", 8), cfg.get("lora_alpha", 16), cfg.get("lora_dropout", 0.05))
    sd = torch.load(os.path.join(path, ADAPTER_WEIGHTS), map_location="cpu")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if unexpected:
        raise RuntimeError(f"unexpected adapter keys: {unexpected[:4]}")
    return model


def merge_and_unload(model: nn.Module) -> nn.Module:
    for parent in list(model.modules()):
        for child_name, child in list(parent.named_children()):
            if isinstance(child, LoRALinear):
                setattr(parent, child_name, child.merge())
    return model


def trainable_parameter_summary(model: nn.Module) -> str:
    t = sum(p.numel() for p in model.parameters() if p.requires_grad)
    a = sum(p.numel() for p in model.parameters())
    return f"trainable params: {t:,} || all params: {a:,} || trainable%: {100.0 * t / max(a, 1):.4f}"
