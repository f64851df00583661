"""Minimal in-tree LoRA injector (the image has no `peft`).

Mirrors the configuration the reference asks peft for
(dalm/models/rag_e2e_base_model.py:145-160): r=8, lora_alpha=16, lora_dropout=0.05,
bias="none", on modules whose name ends in one of `target_modules`
(key/query/value for BERT-style retrievers, q_proj/v_proj for Llama/Falcon-style models).
Parameter names follow peft's in-memory layout (`base_layer`, `lora_A.default`, `lora_B.default`); adapters are
written in peft's ON-DISK format (adapter_model.safetensors with `base_model.model.` keys + a full
adapter_config.json), so `PeftModel.from_pretrained` - what the reference's eval and
`attach_pre_trained_peft_layers` call - can read them, and adapters trained with the reference load here.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Iterable, List

import torch
from torch import nn

# DALM_LORA_KERNEL=0: keep the eager branch (dropout, two skinny GEMMs, an add) on the GPU too
_FUSED = os.environ.get("DALM_LORA_KERNEL", "1") != "0"
ADAPTER_WEIGHTS = "adapter_model.bin"
ADAPTER_CONFIG = "adapter_config.json"


class LoRALinear(nn.Module):
    def __init__(self, base: nn.Linear, r: int = 8, lora_alpha: int = 16, lora_dropout: float = 0.05):
        super().__init__()
        self.base_layer = base
        self.r, self.lora_alpha = r, lora_alpha
        self.scaling = lora_alpha / r
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(lora_dropout) if lora_dropout > 0 else nn.Identity()})
        dev = base.qweight.device if hasattr(base, "qweight") else base.weight.device   # NF4Linear: no resident weight
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False, device=dev, dtype=torch.float32)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False, device=dev, dtype=torch.float32)})
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        # lora_B.weight is [out_features, r] (peft's shape, what the adapter files hold) but lives in [r, out_features]-major
        # memory: every LoRA kernel then reads A and B the same way (rank rows, 16-byte loads along the long dimension - the
        # [N][r] layout cost dalm_lora_rowdot 8 dword loads per MFMA step).  Shape, state_dict keys and values are unchanged;
        # gradients and optimizer state follow the parameter's strides (torch's gradient layout contract).
        self.lora_B["default"].weight = nn.Parameter(torch.zeros(r, base.out_features, device=dev, dtype=torch.float32).t())
        self.merged = False
        self._group = None            # LoRAGroup of the projections that read the same input (build_groups)
        LoRALinear._count += 1
        self._uid = LoRALinear._count

    _count = 0

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
        if self.merged:
            return self.base_layer(x)
        a, b = self.lora_A["default"], self.lora_B["default"]
        if x.is_cuda and _FUSED:
            from . import lora_ops

            if self._group is not None:
                out = self._group.fetch(self, x)          # q / k / v of one block: one autograd node for all of them
                if out is not None:
                    return out
            if lora_ops.branch_supported(x, a.weight, b.weight):
                p, salt = self.next_mask_key()
                base = self.base_layer
                if lora_ops.supported(x, base, a.weight, b.weight) and not base.weight.requires_grad:
                    member = (base.weight, base.bias, a.weight, b.weight, self.scaling, p, salt)
                    if lora_ops.group_supported(x, [member]):                 # bf16 activations: the round-5 kernels
                        return lora_ops.lora_group_forward(x, [member])[0]
                    return lora_ops.lora_linear(x, base, a.weight, b.weight, self.scaling, p, salt)
                if hasattr(base, "qweight") or isinstance(base, nn.Linear):      # nf4 storage / a Linear subclass
                    out = base(x)                                                # its own module, its own backward
                    if out.dtype in (torch.float32, torch.bfloat16) and out.is_contiguous():
                        return lora_ops.lora_branch_(out, x, a.weight, b.weight, self.scaling, p, salt)   # out += ..., in place
                    return self._eager_branch(out, x)
        return self._eager_branch(self.base_layer(x), x)

    def next_mask_key(self):
        """(p, salt) of this call.  salt: this module's id in the upper bits, a host call counter below it (graph replays re-use
        the captured salt; there the device-side seed word, advanced once per step, changes the masks)."""
        drop = self.lora_dropout["default"]
        p = float(getattr(drop, "p", 0.0)) if self.training else 0.0
        self._calls = getattr(self, "_calls", 0) + 1
        return p, (self._uid << 12) ^ (self._calls & 0xFFF)

    def _eager_branch(self, out: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        a, b = self.lora_A["default"], self.lora_B["default"]
        z = self.lora_dropout["default"](x)
        if not torch.is_autocast_enabled() and z.dtype != a.weight.dtype:
            z = z.to(a.weight.dtype)  # outside autocast the fp32 adapters need fp32 activations
        return torch.add(out, b(a(z)).to(out.dtype), alpha=self.scaling)  # one kernel for scale + add

    @torch.no_grad()
    def merge(self) -> nn.Linear:
        delta = (self.lora_B["default"].weight @ self.lora_A["default"].weight) * self.scaling
        base = self.base_layer
        if hasattr(base, "to_linear"):      # nf4 base: merge into the dequantised weight (nothing is re-quantised)
            base = base.to_linear()
        base.weight.add_(delta.to(base.weight.dtype))
        return base


# ---- projections that read the same input ------------------------------------------------------------------------------
# transformers calls q_proj / k_proj / v_proj (query / key / value) of one attention module on the SAME tensor object, one after
# the other.  A LoRAGroup lets the first of those calls evaluate all of them as one autograd node (lora_ops.lora_group_forward:
# x streamed once for every adapter, one dx) and hands the siblings their outputs when they are called with that very tensor.
# Nothing in transformers is patched: a sibling called with another tensor (cross attention) simply computes on its own, and a
# group whose stash is left unclaimed switches itself off.
SHARED_INPUT_NAMES = (("q_proj", "k_proj", "v_proj"), ("query", "key", "value"))
_GROUPS = os.environ.get("DALM_LORA_GROUP", "1") != "0"


class GroupedLinear(nn.Linear):
    """A plain (frozen) nn.Linear that sits between LoRA-wrapped siblings (k_proj between q_proj and v_proj): same parameters,
    same state_dict keys; forward asks the group first."""

    _group = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._group is not None and x.is_cuda:
            out = self._group.fetch(self, x)
            if out is not None:
                return out
        return super().forward(x)


class LoRAGroup:
    def __init__(self, members):
        self.members = list(members)
        self.enabled = True
        self._x = None
        self._outs = {}

    def _member_args(self, m):
        if isinstance(m, LoRALinear):
            p, salt = m.next_mask_key()
            return (m.base_layer.weight, m.base_layer.bias, m.lora_A["default"].weight, m.lora_B["default"].weight,
                    m.scaling, p, salt)
        return (m.weight, m.bias, None, None, 0.0, 0.0, 0)

    def fetch(self, module, x):
        """The output of `module` for input x, or None (the caller computes it itself)."""
        if not self.enabled or not _GROUPS or not _FUSED:
            return None
        if self._x is x:
            out = self._outs.pop(id(module), None)
            if not self._outs:
                self._x = None
                self._misses = 0             # a stash claimed in full: the members do share their input
            return out
        if self._outs:
            # an earlier stash was never claimed (an exception / OOM retry in the middle of a forward, or a model that does not
            # feed these projections the same tensor): drop it and try again; three in a row -> this model does not share
            # inputs here, the group switches itself off WITH a warning (it used to go silent on the first one)
            self._x, self._outs = None, {}
            self._misses = getattr(self, "_misses", 0) + 1
            if self._misses >= 3:
                self.enabled = False
                import warnings

                warnings.warn("dalm_amd.lora: a q / k / v projection group saw three unclaimed outputs in a row - its members do "
                              "not read one shared input in this model; the group node is off (per-projection kernels stay on)")
                return None
        from . import lora_ops

        if any(isinstance(m, LoRALinear) and m.merged for m in self.members):
            return None
        for m in self.members:               # the plain-Linear path of a member must be exactly F.linear
            base = m.base_layer if isinstance(m, LoRALinear) else m
            if type(base) not in (nn.Linear, GroupedLinear) and type(base).__name__ != "FalconLinear":
                return None
        args = [self._member_args(m) for m in self.members]
        if not lora_ops.group_supported(x, args):
            return None
        outs = lora_ops.lora_group_forward(x, args)
        self._x = x
        self._outs = {id(m): o for m, o in zip(self.members, outs)}
        out = self._outs.pop(id(module))
        if not self._outs:
            self._x = None
        return out


def build_groups(model: nn.Module) -> int:
    """Link LoRA-wrapped projections (and the plain Linears between them) that read the same input; returns the group count."""
    n = 0
    for parent in model.modules():
        kids = dict(parent.named_children())
        for names in SHARED_INPUT_NAMES:
            mods = [kids[k] for k in names if k in kids]
            if len(mods) < 2 or not any(isinstance(m, LoRALinear) for m in mods):
                continue
            if any(not (isinstance(m, LoRALinear) or type(m) in (nn.Linear, GroupedLinear)) for m in mods):
                continue
            if len({m.in_features for m in mods}) != 1:
                continue
            grp = LoRAGroup(mods)
            for m in mods:
                if type(m) is nn.Linear:
                    m.__class__ = GroupedLinear
                m._group = grp
            n += 1
    return n


def _is_linear(m: nn.Module) -> bool:
    return isinstance(m, nn.Linear) or hasattr(m, "qweight")     # nn.Linear or its nf4 form (models/nf4.py)


def _matches(name: str, targets: Iterable[str]) -> bool:
    leaf = name.rsplit(".", 1)[-1]
    return any(leaf == t or name.endswith("." + t) for t in targets)


# peft's own per-architecture defaults for models without q_proj / v_proj (TRANSFORMERS_MODELS_TO_LORA_TARGET_MODULES_MAPPING)
FUSED_QKV_FALLBACK = {"falcon": ["query_key_value"], "gpt_neox": ["query_key_value"], "bloom": ["query_key_value"],
                      "gpt2": ["c_attn"], "mpt": ["Wqkv"]}


def resolve_targets(model: nn.Module, target_modules: List[str]) -> List[str]:
    """The reference hard-codes q_proj / v_proj for every generator (rag_e2e_base_model.py:61-80); architectures
    with a fused QKV projection (Falcon - BASELINE config 5) have no such modules and peft would raise.  Fall
    back to peft's default targets for that architecture instead of failing."""
    names = [n for n, m in model.named_modules() if _is_linear(m)]
    if any(_matches(n, target_modules) for n in names):
        return list(target_modules)
    mt = getattr(getattr(model, "config", None), "model_type", None)
    fb = FUSED_QKV_FALLBACK.get(mt)
    if fb and any(_matches(n, fb) for n in names):
        return list(fb)
    return list(target_modules)


def inject_lora(model: nn.Module, target_modules: List[str], r: int = 8, lora_alpha: int = 16,
                lora_dropout: float = 0.05) -> nn.Module:
    """Freeze `model`, wrap every matching nn.Linear in a LoRALinear (trainable A/B only)."""
    target_modules = resolve_targets(model, target_modules)
    for prm in model.parameters():
        prm.requires_grad_(False)
    replaced = 0
    for parent_name, parent in list(model.named_modules()):
        for child_name, child in list(parent.named_children()):
            full = f"{parent_name}.{child_name}" if parent_name else child_name
            if _is_linear(child) and _matches(full, target_modules):
                setattr(parent, child_name, LoRALinear(child, r, lora_alpha, lora_dropout))
                replaced += 1
    if replaced == 0:
        raise ValueError(f"Target modules {target_modules} not found in the base model.")
    model._dalm_lora_config = {"r": r, "lora_alpha": lora_alpha, "lora_dropout": lora_dropout, "bias": "none",
                               "target_modules": list(target_modules), "peft_type": "LORA"}
    build_groups(model)
    return model


def lora_state_dict(model: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v for k, v in model.state_dict().items() if ".lora_A." in k or ".lora_B." in k}


def has_lora(model: nn.Module) -> bool:
    return any(isinstance(m, LoRALinear) for m in model.modules())


# ---- adapter I/O in peft's on-disk format ---------------------------------------------------------------------
# The reference saves adapters with peft's save_pretrained (dalm/training/utils/train_utils.py:16-45) and loads them
# with PeftModel.from_pretrained (train_utils.py:48-73, rag_e2e_base_model.py:113-134, dalm/eval/*).  peft's layout:
#   adapter_config.json         LoraConfig fields (peft_type, task_type, r, lora_alpha, target_modules, ...)
#   adapter_model.safetensors   keys "base_model.model.<module path>.lora_A.weight" / ".lora_B.weight"
#                               (the adapter name "default" is NOT part of the stored key)
# load_adapter also accepts peft's older adapter_model.bin and this package's round-1 files (raw module paths with
# ".default" kept) so existing checkpoints keep loading.
PEFT_PREFIX = "base_model.model."
ADAPTER_SAFETENSORS = "adapter_model.safetensors"


def _to_peft_key(k: str) -> str:
    return PEFT_PREFIX + k.replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")


def _from_peft_key(k: str) -> str:
    if k.startswith(PEFT_PREFIX):
        k = k[len(PEFT_PREFIX):]
    if ".lora_A.default." in k or ".lora_B.default." in k:
        return k
    return k.replace(".lora_A.", ".lora_A.default.").replace(".lora_B.", ".lora_B.default.")


def peft_adapter_config(model: nn.Module, task_type: str = None, base_model_name_or_path: str = None) -> Dict:
    cfg = dict(getattr(model, "_dalm_lora_config", {}))
    if task_type is None:  # the reference: CAUSAL_LM for generators, FEATURE_EXTRACTION for encoders (:145-160)
        task_type = "CAUSAL_LM" if hasattr(model, "lm_head") or hasattr(model, "get_output_embeddings") and \
            model.get_output_embeddings() is not None else "FEATURE_EXTRACTION"
    if base_model_name_or_path is None:
        base_model_name_or_path = getattr(getattr(model, "config", None), "_name_or_path", None) or None
    return {
        "alpha_pattern": {}, "auto_mapping": None, "base_model_name_or_path": base_model_name_or_path,
        "bias": cfg.get("bias", "none"), "fan_in_fan_out": False, "inference_mode": True, "init_lora_weights": True,
        "layers_pattern": None, "layers_to_transform": None, "lora_alpha": cfg.get("lora_alpha", 16),
        "lora_dropout": cfg.get("lora_dropout", 0.05), "modules_to_save": None, "peft_type": "LORA",
        "r": cfg.get("r", 8), "rank_pattern": {}, "revision": None,
        "target_modules": list(cfg.get("target_modules", [])), "task_type": task_type,
    }


def save_adapter(model: nn.Module, path: str, task_type: str = None, base_model_name_or_path: str = None) -> None:
    """peft-loadable adapter directory (PeftModel.from_pretrained(base, path) reads it)."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    sd = {_to_peft_key(k): v.detach().to("cpu").contiguous() for k, v in lora_state_dict(model).items()}
    save_file(sd, os.path.join(path, ADAPTER_SAFETENSORS), metadata={"format": "pt"})
    with open(os.path.join(path, ADAPTER_CONFIG), "w") as f:
        json.dump(peft_adapter_config(model, task_type, base_model_name_or_path), f, indent=2, sort_keys=True)


def load_adapter(model: nn.Module, path: str) -> nn.Module:
    with open(os.path.join(path, ADAPTER_CONFIG)) as f:
        cfg = json.load(f)
    if cfg.get("peft_type", "LORA") != "LORA":
        raise ValueError(f"unsupported peft_type {cfg.get('peft_type')!r} (LoRA adapters only)")
    targets = cfg["target_modules"]
    if isinstance(targets, str):
        targets = [targets]
    if not has_lora(model):
        inject_lora(model, list(targets), cfg.get("r", 8), cfg.get("lora_alpha", 16), cfg.get("lora_dropout", 0.05))
    st = os.path.join(path, ADAPTER_SAFETENSORS)
    if os.path.exists(st):
        from safetensors.torch import load_file

        raw = load_file(st)
    else:
        raw = torch.load(os.path.join(path, ADAPTER_WEIGHTS), map_location="cpu")
    sd = {_from_peft_key(k): v for k, v in raw.items()}
    want = set(lora_state_dict(model).keys())
    unexpected = sorted(set(sd) - want)
    missing = sorted(want - set(sd))
    if unexpected or missing:
        raise RuntimeError(f"adapter at {path} does not match the model: unexpected {unexpected[:3]}, missing {missing[:3]}")
    model.load_state_dict(sd, strict=False)
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
