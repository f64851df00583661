"""Speed-ups of the HF towers that keep the math: PyTorch-ROCm natives, and (round 4) two HIP kernels for the elementwise
chains of a Llama-family layer the step spent the most launches on (rotary embedding, SwiGLU: `tower_ops.py`).

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


_warned = set()


def _warn_once(key: str, msg: str) -> None:
    if key not in _warned:
        _warned.add(key)
        import warnings

        warnings.warn("dalm_amd.fastpath: " + msg)


def _close(a: torch.Tensor, b: torch.Tensor, ulps: float = 1.0) -> bool:
    """Equal up to `ulps` units in the last place of the tensors' dtype, relative to the largest magnitude (the patches round
    where transformers' chains round; what is left is the summation order of a mean or one fused multiply-add)."""
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}.get(a.dtype, 2.0 ** -20)
    scale = max(float(b.float().abs().max()), 1e-6)
    return float((a.float() - b.float()).abs().max()) <= ulps * eps * scale


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
        if not _norm_matches(mod):
            _warn_once("norm:" + type(mod).__name__, f"{type(mod).__name__}.forward is not w * x * rsqrt(mean(x^2) + eps) here: "
                       "transformers' own code stays in place")
            continue
        mod.forward = types.MethodType(_native_forward, mod)
        n += 1
    return n


def _norm_matches(mod: torch.nn.Module) -> bool:
    """Run-time guard: the module's OWN forward against the fused formula on a small random tensor, once per
    class, dtype and device type."""
    w = mod.weight
    key = ("norm", type(mod), w.dtype, w.device.type)
    if key not in _checked:
        x = torch.randn(4, 3, w.shape[0], generator=torch.Generator().manual_seed(0)).to(device=w.device, dtype=w.dtype)
        with torch.no_grad():
            _checked[key] = _close(_native_forward(mod, x), type(mod).forward(mod, x), 2.0)
    return _checked[key]


_checked: dict = {}


# ---------------------------------------------------------------------------
# rotary embedding with fewer launches (Llama-style "rotate_half" layout)
# ---------------------------------------------------------------------------
def _rope_roll(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):
    """Same result as transformers' apply_rotary_pos_emb: x*cos + rotate_half(x)*sin with
    rotate_half(x) = cat(-x2, x1).  Folding the sign into sin (a tiny [B,1,T,hd] tensor) turns
    rotate_half into a plain roll by hd/2, and addcmul fuses the second multiply with the add:
    3 launches forward / 4 backward per tensor instead of ~5 / ~8 (neg, cat and their slice backwards go)."""
    cos = cos.unsqueeze(unsqueeze_dim)
    sin = sin.unsqueeze(unsqueeze_dim)
    h = q.shape[-1] // 2
    sin_signed = torch.cat((-sin[..., :h], sin[..., h:]), dim=-1)
    q_embed = torch.addcmul(q * cos, torch.roll(q, h, dims=-1), sin_signed)
    k_embed = torch.addcmul(k * cos, torch.roll(k, h, dims=-1), sin_signed)
    return q_embed, k_embed


def _rope_hip(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):
    """transformers' apply_rotary_pos_emb as one HIP launch for q and k together (`dalm_rope_qk`, one more in the backward),
    rounding where the eager chain rounds: same values and gradients as transformers' own code (round 4; the roll + addcmul
    form above was 6 launches forward / 8 backward per layer and rounded once less than transformers does).
    DALM_ROPE_KERNEL=0 keeps the roll form; CPU tensors and layouts the kernel does not take use it as well."""
    from . import tower_ops

    if unsqueeze_dim == 1 and tower_ops.rope_supported(q, k, cos, sin):
        return tower_ops.rope_qk(q, k, cos, sin)
    return _rope_roll(q, k, cos, sin, position_ids, unsqueeze_dim)


def use_roll_rope(model: torch.nn.Module) -> bool:
    """Swap the module-level apply_rotary_pos_emb of the model's own modeling file (Llama and Falcon)."""
    if os.environ.get("DALM_FAST_ROPE", "1") == "0":
        return False
    fn = _rope_roll if os.environ.get("DALM_ROPE_KERNEL", "1") == "0" else _rope_hip
    import importlib

    mod_name = type(getattr(model, "base_model", model)).__module__
    # modeling files whose apply_rotary_pos_emb is the formula above, word for word (Falcon calls it only without alibi)
    if not mod_name.endswith(("modeling_llama", "modeling_falcon")):
        return False
    mod = importlib.import_module(mod_name)
    if not hasattr(mod, "apply_rotary_pos_emb"):
        return False
    if getattr(mod.apply_rotary_pos_emb, "__name__", "") not in ("_rope_roll", "_rope_hip"):
        mod._dalm_orig_apply_rotary_pos_emb = mod.apply_rotary_pos_emb
    # Run-time guard: the function being replaced must BE the formula the kernel implements.  The swap is
    # PROCESS-WIDE (transformers resolves apply_rotary_pos_emb through the modeling file's globals; there is no per-model
    # hook), so every model of this modeling file in the process takes the replacement.
    dev = next((p.device for p in model.parameters()), torch.device("cpu"))
    if not _rope_matches(mod._dalm_orig_apply_rotary_pos_emb, fn, dev):
        _warn_once("rope:" + mod_name, f"{mod_name}.apply_rotary_pos_emb is not x * cos + rotate_half(x) * sin here: "
                   "transformers' own code stays in place")
        mod.apply_rotary_pos_emb = mod._dalm_orig_apply_rotary_pos_emb
        return False
    mod.apply_rotary_pos_emb = fn
    return True


def _rope_matches(orig, repl, dev: torch.device) -> bool:
    """orig(q, k, cos, sin) against the replacement on small random tensors: on the CPU against the roll form in float32 (the
    same formula, one fused multiply-add apart), on a GPU additionally against the HIP kernel in bfloat16 - there the values
    must be EQUAL (tests/test_tower_ops_gpu.py asserts the same)."""
    key = ("rope", orig, repl, dev.type)
    if key in _checked:
        return _checked[key]
    g = torch.Generator().manual_seed(0)
    q, k = torch.randn(2, 4, 5, 16, generator=g), torch.randn(2, 2, 5, 16, generator=g)
    ang = torch.rand(2, 5, 8, generator=g) * 6.28
    cos, sin = torch.cat((ang.cos(), ang.cos()), -1), torch.cat((ang.sin(), ang.sin()), -1)
    ok = True
    try:
        with torch.no_grad():
            want = orig(q, k, cos, sin)
            got = _rope_roll(q, k, cos, sin)
            ok = all(_close(a, b, 8.0) for a, b in zip(got, want))
            if ok and dev.type == "cuda" and repl is _rope_hip:
                args = [t.to(dev, torch.bfloat16) for t in (q, k, cos, sin)]
                ok = all(torch.equal(a, b) for a, b in zip(_rope_hip(*args), orig(*args)))
    except Exception:
        ok = False
    _checked[key] = ok
    return ok


# ---------------------------------------------------------------------------
# SwiGLU MLP: silu(gate) * up as one launch per direction, no saved activation
# ---------------------------------------------------------------------------
def _swiglu_mlp_forward(self, x):
    from . import tower_ops

    from . import frozen_linear

    gate, up = frozen_linear.pair_forward(x, self.gate_proj, self.up_proj)   # frozen: one node, the two dx GEMMs accumulate
    if tower_ops.swiglu_supported(gate, up):
        return self.down_proj(tower_ops.swiglu(gate, up))
    return self.down_proj(self.act_fn(gate) * up)


def use_swiglu_kernel(model: torch.nn.Module) -> int:
    """Patch the `*MLP` modules that are exactly transformers' LlamaMLP formula - down(act(gate(x)) * up(x)) with
    act = SiLU - to evaluate silu(gate) * up through `dalm_swiglu_{fwd,bwd}`: 2 eager launches forward and 4 backward
    become 1 + 1, and the [tokens, intermediate] activation is recomputed instead of saved (101 MB per layer at cfg3).
    Same rounding points as the eager chain.  DALM_SWIGLU_KERNEL=0 disables; returns how many modules were patched."""
    if os.environ.get("DALM_SWIGLU_KERNEL", "1") == "0":
        return 0
    n = 0
    for mod in model.modules():
        if type(mod).__name__ not in ("LlamaMLP", "MistralMLP", "Qwen2MLP"):
            continue
        act = getattr(mod, "act_fn", None)
        if not all(hasattr(mod, a) for a in ("gate_proj", "up_proj", "down_proj")):
            continue
        if not (isinstance(act, torch.nn.SiLU) or type(act).__name__ in ("SiLUActivation", "SiLU")):
            continue
        if not _mlp_matches(mod):
            _warn_once("mlp:" + type(mod).__name__, f"{type(mod).__name__}.forward is not down(silu(gate(x)) * up(x)) here: "
                       "transformers' own code stays in place")
            continue
        mod.forward = types.MethodType(_swiglu_mlp_forward, mod)
        n += 1
    return n


def _mlp_matches(mod: torch.nn.Module) -> bool:
    """Run-time guard: the class's own forward against the patched one on a small random input, once per
    class, dtype and device type (on a GPU the patched forward runs the HIP kernel: <= 1 ulp of the output dtype)."""
    w = getattr(mod.gate_proj, "weight", None)
    if w is None:
        return False
    key = ("mlp", type(mod), w.dtype, w.device.type)
    if key not in _checked:
        x = torch.randn(2, 3, mod.gate_proj.in_features, generator=torch.Generator().manual_seed(0)).to(device=w.device, dtype=w.dtype)
        try:
            with torch.no_grad():
                _checked[key] = _close(_swiglu_mlp_forward(mod, x), type(mod).forward(mod, x), 2.0)
        except Exception:
            _checked[key] = False
    return _checked[key]


# ---------------------------------------------------------------------------
# Falcon multi-query attention: make the head split capturable
# ---------------------------------------------------------------------------
def _split_heads_sliced(self, fused_qkv: torch.Tensor):
    """Falcon-7B layout [.., num_heads + 2, head_dim]: queries, then ONE shared key head and ONE shared value head.
    transformers picks the two shared heads with python-list indices (`x[..., [-2], :]`), which builds an index
    tensor on the host and copies it to the device on every call - an operation a hipGraph capture refuses.  Plain
    slices select the same elements (as views) with no host work."""
    b, t, _ = fused_qkv.shape
    x = fused_qkv.view(b, t, self.num_heads + 2, self.head_dim)
    n = self.num_heads
    return x[..., :n, :], x[..., n:n + 1, :], x[..., n + 1:, :]


def use_capturable_falcon_heads(model: torch.nn.Module) -> int:
    """Patch FalconAttention modules of the 7B flavour (multi_query, old decoder architecture); returns the count.

    The module's `num_kv_heads` is NOT touched (round 5 set it to `num_heads` so that SDPA saw equal head counts,
    which also made every reader outside the training call - the KV cache of eval / generate, export, sharding - see 71 key /
    value heads).  The broadcast of the shared key / value head happens inside the patched TRAINING call only
    (`_falcon_attention_forward`, `layer_past is None`); `_dalm_expand_kv` merely says that call may do it."""
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "FalconAttention":
            continue
        if getattr(mod, "new_decoder_architecture", False) or not getattr(mod, "multi_query", False):
            continue
        mod._split_heads = types.MethodType(_split_heads_sliced, mod)
        # multi-query + SDPA with [B, 1, T, hd] keys falls back to torch's unfused "math" path - f32 scores, softmax, masks:
        # ~1.5 ms per layer at cfg5, 48 ms of a 230 ms step (profiles/r05cfg5_step_by_stream_before.txt).
        # DALM_FALCON_EXPAND_KV=0 keeps the [B, 1, T, hd] form.
        mod._dalm_expand_kv = (os.environ.get("DALM_FALCON_EXPAND_KV", "1") != "0" and getattr(mod, "num_kv_heads", None) == 1
                               and getattr(getattr(mod, "config", None), "_attn_implementation", None) == "sdpa")
        n += 1
    return n


# ---------------------------------------------------------------------------
# Falcon decoder layer (7B flavour: parallel attention, one LayerNorm, no dropout): LayerNorm, GELU and the residual adds
# ---------------------------------------------------------------------------
_FALCON_LAYER_PARAMS = ["self", "hidden_states", "alibi", "attention_mask", "position_ids", "layer_past", "use_cache",
                        "output_attentions", "position_embeddings", "kwargs"]


def _falcon_mlp_forward(self, x):
    from . import tower_ops

    h = self.dense_h_to_4h(x)
    if tower_ops.flat_bf16_supported(h):
        return self.dense_4h_to_h(tower_ops.gelu(h))
    return self.dense_4h_to_h(self.act(h))


def _falcon_layer_forward(self, hidden_states, alibi, attention_mask, position_ids=None, layer_past=None, use_cache=False,
                          output_attentions=False, position_embeddings=None, **kwargs):
    """transformers' FalconDecoderLayer.forward for `parallel_attn` without the new decoder architecture and without dropout:
        ln = input_layernorm(x);  out = x + (mlp(ln) + attention(ln))
    with the LayerNorm on `dalm_layer_norm_{fwd,bwd}` (under bf16 autocast the eager form is two up-casts, an f32 LayerNorm and
    two down-casts forward, and the same again backward) and the two adds on `dalm_add3`.  The backward kernel of the norm also
    adds the gradient that reaches x through the residual path."""
    from . import tower_ops

    ln = self.input_layernorm
    if not tower_ops.layer_norm_supported(hidden_states, ln.weight, ln.bias):
        return self._dalm_orig_forward(hidden_states, alibi, attention_mask, position_ids=position_ids, layer_past=layer_past,
                                       use_cache=use_cache, output_attentions=output_attentions,
                                       position_embeddings=position_embeddings, **kwargs)
    residual, normed = tower_ops.layer_norm_res(hidden_states, ln.weight, ln.bias, ln.eps)
    attn_out, attn_weights = self.self_attention(normed, layer_past=layer_past, attention_mask=attention_mask,
                                                 position_ids=position_ids, alibi=alibi, use_cache=use_cache,
                                                 output_attentions=output_attentions, position_embeddings=position_embeddings)
    mlp_out = self.mlp(normed)
    if tower_ops.flat_bf16_supported(mlp_out, attn_out, residual):
        return tower_ops.add3(mlp_out, attn_out, residual), attn_weights
    mlp_out = mlp_out + attn_out
    return residual + mlp_out, attn_weights


def use_falcon_layer_kernels(model: torch.nn.Module) -> int:
    """Patch FalconDecoderLayer / FalconMLP modules of the 7B flavour whose code is the one this file was written against
    (signature and source checked once per class).  DALM_FALCON_KERNELS=0 disables.  Returns how many layers were patched."""
    if os.environ.get("DALM_FALCON_KERNELS", "1") == "0":
        return 0
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "FalconDecoderLayer":
            continue
        cfg = getattr(mod, "config", None)
        if cfg is None or getattr(cfg, "new_decoder_architecture", True) or not getattr(cfg, "parallel_attn", False):
            continue
        if float(getattr(cfg, "hidden_dropout", 1.0)) != 0.0 or float(getattr(cfg, "attention_dropout", 1.0)) != 0.0:
            continue            # dropout_add with p > 0 draws from torch's generator: transformers' code stays
        ln = getattr(mod, "input_layernorm", None)
        if not isinstance(ln, torch.nn.LayerNorm) or ln.weight is None or not hasattr(mod, "self_attention") \
                or not hasattr(mod, "mlp"):
            continue
        if not _falcon_layer_matches(type(mod)):
            continue
        mod._dalm_orig_forward = mod.forward
        mod.forward = types.MethodType(_falcon_layer_forward, mod)
        mlp = mod.mlp
        if type(mlp).__name__ == "FalconMLP" and _is_erf_gelu(getattr(mlp, "act", None)) and _falcon_mlp_matches(type(mlp)):
            mlp.forward = types.MethodType(_falcon_mlp_forward, mlp)
        n += 1
    return n


def _is_erf_gelu(act) -> bool:
    """torch's exact GELU: nn.GELU() or transformers' GELUActivation around nn.functional.gelu (config.activation = "gelu")."""
    if isinstance(act, torch.nn.GELU):
        return getattr(act, "approximate", "none") == "none"
    return type(act).__name__ == "GELUActivation" and getattr(act, "act", None) is torch.nn.functional.gelu


def _falcon_layer_matches(cls) -> bool:
    """Run-time guard: the class's forward must have the signature and the statements `_falcon_layer_forward` restates."""
    key = ("falcon-layer", cls)
    if key not in _checked:
        try:
            import inspect

            params = list(inspect.signature(cls.forward).parameters)
            src = inspect.getsource(cls.forward)
            _checked[key] = (params == _FALCON_LAYER_PARAMS
                             and "attention_layernorm_out = self.input_layernorm(hidden_states)" in src
                             and "mlp_layernorm_out = attention_layernorm_out" in src
                             and "mlp_output = self.mlp(mlp_layernorm_out)" in src
                             and "mlp_output += attention_output" in src
                             and "output = dropout_add(mlp_output, residual, self.config.hidden_dropout, training=self.training)" in src
                             and "return output, attn_weights" in src)
        except Exception:
            _checked[key] = False
        if not _checked[key]:
            _warn_once("falcon-layer", "FalconDecoderLayer.forward is not the code this patch restates: transformers' own "
                       "code stays in place")
    return _checked[key]


def _falcon_mlp_matches(cls) -> bool:
    key = ("falcon-mlp", cls)
    if key not in _checked:
        try:
            import inspect

            src = inspect.getsource(cls.forward)
            _checked[key] = ("x = self.act(self.dense_h_to_4h(x))" in src and "x = self.dense_4h_to_h(x)" in src
                             and src.count("self.") == 3)
        except Exception:
            _checked[key] = False
        if not _checked[key]:
            _warn_once("falcon-mlp", "FalconMLP.forward is not dense_4h_to_h(act(dense_h_to_4h(x))) here: transformers' own "
                       "code stays in place")
    return _checked[key]


# ---------------------------------------------------------------------------
# BERT encoder layer: dropout + residual add + LayerNorm of BertSelfOutput / BertOutput in one launch each way
# ---------------------------------------------------------------------------
def _bert_output_forward(self, hidden_states: torch.Tensor, input_tensor: torch.Tensor) -> torch.Tensor:
    """transformers' BertSelfOutput.forward / BertOutput.forward (the same three statements):
        hidden_states = self.dense(hidden_states); hidden_states = self.dropout(hidden_states)
        hidden_states = self.LayerNorm(hidden_states + input_tensor)
    with the last two on `dalm_bert_add_norm_{fwd,bwd}` where the tensors are what bf16 autocast over a frozen base produces
    (bf16 dense output, f32 residual); everything else runs the statements as they are."""
    from . import bert_ops

    hidden_states = self.dense(hidden_states)
    if bert_ops.supported(hidden_states, input_tensor, self.LayerNorm):
        p = float(self.dropout.p) if self.training else 0.0
        self._dalm_calls = getattr(self, "_dalm_calls", 0) + 1
        salt = ((id(self) >> 4) << 12) ^ (self._dalm_calls & 0xFFF)     # module id above a host call counter (graph replays: the
        return bert_ops.add_norm(hidden_states, input_tensor, self.LayerNorm, p, salt)   # device seed word changes the masks)
    hidden_states = self.dropout(hidden_states)
    return self.LayerNorm(hidden_states + input_tensor)


def use_bert_layer_kernels(model: torch.nn.Module) -> int:
    """Patch BertSelfOutput / BertOutput modules whose forward is the three statements above (source checked once per class) and
    whose LayerNorm is frozen.  DALM_BERT_KERNELS=0 disables.  Returns how many modules were patched."""
    if os.environ.get("DALM_BERT_KERNELS", "1") == "0":
        return 0
    n = 0
    for mod in model.modules():
        if type(mod).__name__ not in ("BertSelfOutput", "BertOutput"):
            continue
        ln, drop = getattr(mod, "LayerNorm", None), getattr(mod, "dropout", None)
        if not isinstance(ln, torch.nn.LayerNorm) or not isinstance(drop, torch.nn.Dropout) or not hasattr(mod, "dense"):
            continue
        if not _bert_output_matches(type(mod)):
            continue
        mod.forward = types.MethodType(_bert_output_forward, mod)
        n += 1
    return n


def _bert_output_matches(cls) -> bool:
    key = ("bert-output", cls)
    if key not in _checked:
        try:
            import inspect

            params = list(inspect.signature(cls.forward).parameters)
            src = inspect.getsource(cls.forward)
            body = [ln.strip() for ln in src.splitlines()[1:] if ln.strip() and not ln.strip().startswith("#")]
            _checked[key] = (params == ["self", "hidden_states", "input_tensor"]
                             and body == ["hidden_states = self.dense(hidden_states)", "hidden_states = self.dropout(hidden_states)",
                                          "hidden_states = self.LayerNorm(hidden_states + input_tensor)", "return hidden_states"])
        except Exception:
            _checked[key] = False
        if not _checked[key]:
            _warn_once("bert-output:" + cls.__name__, f"{cls.__name__}.forward is not dense -> dropout -> LayerNorm(h + input) here: "
                       "transformers' own code stays in place")
    return _checked[key]


# ---------------------------------------------------------------------------
# Falcon attention (7B flavour, rotary, sdpa): the attention itself on dalm_attn_fwd / dalm_attn_bwd
# ---------------------------------------------------------------------------
_FALCON_ATTN_PARAMS = ["self", "hidden_states", "alibi", "attention_mask", "position_ids", "layer_past", "use_cache",
                       "output_attentions", "position_embeddings", "kwargs"]


def _falcon_attention_forward(self, hidden_states, alibi, attention_mask, position_ids=None, layer_past=None, use_cache=False,
                              output_attentions=False, position_embeddings=None, **kwargs):
    """transformers' FalconAttention.forward for the training call (rotary positions, no KV cache, "sdpa"): the same statements
    with F.scaled_dot_product_attention replaced by `attention.sdpa` (dalm_amd/csrc/attn.hip).  Every other call goes to
    transformers' own forward."""
    import importlib

    from . import attention

    orig = self._dalm_orig_attn_forward
    if (alibi is not None or layer_past is not None or output_attentions or position_embeddings is None
            or getattr(self.config, "_attn_implementation", None) != "sdpa" or os.environ.get("DALM_ATTN_KERNEL", "1") == "0"):
        return orig(hidden_states, alibi, attention_mask, position_ids=position_ids, layer_past=layer_past, use_cache=use_cache,
                    output_attentions=output_attentions, position_embeddings=position_embeddings, **kwargs)
    fused_qkv = self.query_key_value(hidden_states)
    num_kv_heads = self.num_heads if self.new_decoder_architecture else self.num_kv_heads
    query_layer, key_layer, value_layer = self._split_heads(fused_qkv)
    batch_size, query_length, _, _ = query_layer.shape
    cos, sin = position_embeddings
    is_causal = bool(self.is_causal and attention_mask is None and query_length > 1)
    rope_on = os.environ.get("DALM_ROPE_KERNEL", "1") != "0" and os.environ.get("DALM_FAST_ROPE", "1") != "0"
    mqa = bool(getattr(self, "_dalm_expand_kv", False) and num_kv_heads == 1 and key_layer.shape[2] == 1)
    if mqa and rope_on and os.environ.get("DALM_FALCON_MQA_VIEWS", "1") != "0":
        # multi-query, THIS call only (training, no KV cache): the ONE shared key / value head stays one head in memory.  The
        # rotation runs on the single key head, the attention kernels read key / value through stride-0 head views (every head
        # the same rows, served by the caches) - the 71 copies of K and V that the broadcast-then-reshape form materialises per
        # layer (2 x 93 us at cfg5, profiles/r06cfg5packed_step_by_stream.txt) never exist; dk / dv are summed over the heads
        # after the backward kernels, as autograd's expand backward would
        H, hd = self.num_heads, self.head_dim
        q4 = query_layer.transpose(1, 2).reshape(batch_size, H, query_length, hd)
        k1, v1 = key_layer.transpose(1, 2), value_layer.transpose(1, 2)                      # [b, 1, t, hd] views of fused_qkv
        kx, vx = k1.expand(batch_size, H, query_length, hd), v1.expand(batch_size, H, query_length, hd)
        seqs = attention.packed_of(attention_mask)
        ok = (attention.packed_supported(q4, kx, vx) if seqs is not None
              else attention.supported(q4, kx, vx, attention_mask, 0.0, is_causal, {}))
        if ok and attention.rope_fusable(q4, k1, cos, sin):
            attn_output = attention.rope_sdpa(q4, k1, v1, cos, sin, attention_mask, float(hd) ** -0.5,
                                              False if seqs is not None else is_causal)
            attn_output = attn_output.permute(0, 2, 1, 3).reshape(batch_size, query_length, H * hd)
            return self.dense(attn_output), None
    if mqa:
        # the one shared key / value head broadcast to the query heads (views; the reshape below materialises them) - equal head
        # counts for the kernels and for torch's fused SDPA
        key_layer = key_layer.expand(batch_size, query_length, self.num_heads, self.head_dim)
        value_layer = value_layer.expand(batch_size, query_length, self.num_heads, self.head_dim)
        num_kv_heads = self.num_heads
    query_layer = query_layer.transpose(1, 2).reshape(batch_size, self.num_heads, query_length, self.head_dim)
    key_layer = key_layer.transpose(1, 2).reshape(batch_size, num_kv_heads, query_length, self.head_dim)
    value_layer = value_layer.transpose(1, 2).reshape(batch_size, num_kv_heads, query_length, self.head_dim)
    seqs = attention.packed_of(attention_mask)
    if seqs is not None:
        # packed (un-padded) call (dalm_amd/packed.py): [1, H, n, hd], sequences from the descriptor
        if attention.packed_supported(query_layer, key_layer, value_layer) and attention.rope_fusable(query_layer, key_layer, cos, sin) \
                and rope_on:
            attn_output = attention.rope_sdpa(query_layer, key_layer, value_layer, cos, sin, attention_mask,
                                              float(self.head_dim) ** -0.5, False)
            attn_output = attn_output.permute(0, 2, 1, 3)
        else:
            rope = importlib.import_module(type(self).__module__).apply_rotary_pos_emb
            query_layer, key_layer = rope(query_layer, key_layer, cos, sin)
            attn_output, _ = attention._packed_attention(self, query_layer, key_layer, value_layer, attention_mask, seqs, 0.0,
                                                         float(self.head_dim) ** -0.5)            # [1, n, H, hd]
        attn_output = attn_output.reshape(batch_size, query_length, self.num_heads * self.head_dim)
        return self.dense(attn_output), None
    if (attention.supported(query_layer, key_layer, value_layer, attention_mask, 0.0, is_causal, {})
            and attention.rope_fusable(query_layer, key_layer, cos, sin) and rope_on):
        attn_output = attention.rope_sdpa(query_layer, key_layer, value_layer, cos, sin, attention_mask,
                                          float(self.head_dim) ** -0.5, is_causal)     # the rotation's backward: in dalm_attn_bwd
        attn_output = attn_output.view(batch_size, self.num_heads, query_length, self.head_dim)
        attn_output = attn_output.permute(0, 2, 1, 3)
        attn_output = attn_output.reshape(batch_size, query_length, self.num_heads * self.head_dim)
        return self.dense(attn_output), None
    rope = importlib.import_module(type(self).__module__).apply_rotary_pos_emb        # the modeling file's own (maybe swapped) one
    query_layer, key_layer = rope(query_layer, key_layer, cos, sin)
    if not attention.supported(query_layer, key_layer, value_layer, attention_mask, 0.0, is_causal, {}):
        attn_output = torch.nn.functional.scaled_dot_product_attention(query_layer, key_layer, value_layer, attn_mask=attention_mask,
                                                                       dropout_p=0.0, is_causal=is_causal)
    else:
        attn_output = attention.sdpa(query_layer, key_layer, value_layer, attention_mask, float(self.head_dim) ** -0.5, is_causal)
    attn_output = attn_output.view(batch_size, self.num_heads, query_length, self.head_dim)
    attn_output = attn_output.permute(0, 2, 1, 3)
    attn_output = attn_output.reshape(batch_size, query_length, self.num_heads * self.head_dim)
    return self.dense(attn_output), None


def use_falcon_attention_kernels(model: torch.nn.Module) -> int:
    """Patch FalconAttention modules (head width 64 or 128, no alibi) whose forward is the code `_falcon_attention_forward`
    restates (signature and statements checked once per class).  DALM_ATTN_KERNEL=0 disables.  Returns the count."""
    if os.environ.get("DALM_ATTN_KERNEL", "1") == "0":
        return 0
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "FalconAttention" or getattr(mod, "head_dim", None) not in (64, 128):
            continue
        if getattr(getattr(mod, "config", None), "alibi", True) or float(getattr(mod.config, "attention_dropout", 1.0)) != 0.0:
            continue
        if not _falcon_attention_matches(type(mod)):
            continue
        mod._dalm_orig_attn_forward = mod.forward
        mod.forward = types.MethodType(_falcon_attention_forward, mod)
        n += 1
    return n


def _falcon_attention_matches(cls) -> bool:
    key = ("falcon-attn", cls)
    if key not in _checked:
        try:
            import inspect

            params = list(inspect.signature(cls.forward).parameters)
            src = inspect.getsource(cls.forward)
            _checked[key] = (params == _FALCON_ATTN_PARAMS
                             and "fused_qkv = self.query_key_value(hidden_states)" in src
                             and "(query_layer, key_layer, value_layer) = self._split_heads(fused_qkv)" in src
                             and "query_layer, key_layer = apply_rotary_pos_emb(query_layer, key_layer, cos, sin)" in src
                             and "is_causal = self.is_causal and attention_mask is None and query_length > 1" in src
                             and "attn_output = attn_output.view(batch_size, self.num_heads, query_length, self.head_dim)" in src
                             and "attn_output = attn_output.permute(0, 2, 1, 3)" in src
                             and "attn_output = self.dense(attn_output)" in src
                             and "return attn_output, attention_scores" in src)
        except Exception:
            _checked[key] = False
        if not _checked[key]:
            _warn_once("falcon-attn", "FalconAttention.forward is not the code this patch restates: transformers' own code "
                       "stays in place")
    return _checked[key]


# ---------------------------------------------------------------------------
# Llama attention: rotary embedding + attention as ONE autograd node (the rotation's backward rides in dalm_attn_bwd's epilogues)
# ---------------------------------------------------------------------------
_LLAMA_ATTN_PARAMS = ["self", "hidden_states", "position_embeddings", "attention_mask", "past_key_values", "kwargs"]


def _llama_attention_forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
    """transformers' LlamaAttention.forward for the training call ("dalm_sdpa", no KV cache, no dropout): the same statements with
    apply_rotary_pos_emb + the attention call as `attention.rope_sdpa`.  Every other call goes to transformers' own forward."""
    from . import attention

    cfg = self.config
    if (past_key_values is not None or position_embeddings is None or getattr(cfg, "_attn_implementation", None) != attention.NAME
            or (self.training and float(getattr(self, "attention_dropout", 0.0)) != 0.0)
            or os.environ.get("DALM_ROPE_KERNEL", "1") == "0" or os.environ.get("DALM_FAST_ROPE", "1") == "0"
            or os.environ.get("DALM_ATTN_KERNEL", "1") == "0" or kwargs.get("output_attentions")):
        return self._dalm_orig_attn_forward(hidden_states, position_embeddings=position_embeddings, attention_mask=attention_mask,
                                            past_key_values=past_key_values, **kwargs)
    input_shape = hidden_states.shape[:-1]
    hidden_shape = (*input_shape, -1, self.head_dim)
    query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)
    cos, sin = position_embeddings
    causal = bool(query_states.shape[2] > 1 and attention_mask is None and getattr(self, "is_causal", True))
    if attention.packed_of(attention_mask) is not None:
        # packed (un-padded) call (dalm_amd/packed.py): [1, H, n, hd] views, sequences from the descriptor; same node
        fused = (key_states.shape == query_states.shape and attention.packed_supported(query_states, key_states, value_states)
                 and attention.rope_fusable(query_states, key_states, cos, sin))
    else:
        fused = (key_states.shape == query_states.shape
                 and attention.supported(query_states, key_states, value_states, attention_mask, 0.0, causal, {})
                 and attention.rope_fusable(query_states, key_states, cos, sin))
    if fused:
        attn_output = attention.rope_sdpa(query_states, key_states, value_states, cos, sin, attention_mask, float(self.scaling), causal)
        attn_output = attn_output.transpose(1, 2)
    else:                                            # grouped heads, CPU tensors, ...: transformers' own sequence from here on
        import importlib

        from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS

        rope = importlib.import_module(type(self).__module__).apply_rotary_pos_emb
        query_states, key_states = rope(query_states, key_states, cos, sin)
        attn_output, _ = ALL_ATTENTION_FUNCTIONS[cfg._attn_implementation](
            self, query_states, key_states, value_states, attention_mask, dropout=0.0, scaling=self.scaling, **kwargs)
    attn_output = attn_output.reshape(*input_shape, -1).contiguous()
    return self.o_proj(attn_output), None


def use_llama_attention_node(model: torch.nn.Module) -> int:
    """Patch LlamaAttention modules of a model that runs on "dalm_sdpa" (models/attention.py) and whose forward is the code
    `_llama_attention_forward` restates.  Returns the count."""
    from . import attention

    if getattr(getattr(model, "config", None), "_attn_implementation", None) != attention.NAME:
        return 0
    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "LlamaAttention" or not all(hasattr(mod, a) for a in ("q_proj", "k_proj", "v_proj", "o_proj", "scaling")):
            continue
        if not _llama_attention_matches(type(mod)):
            continue
        mod._dalm_orig_attn_forward = mod.forward
        mod.forward = types.MethodType(_llama_attention_forward, mod)
        n += 1
    return n


def _llama_attention_matches(cls) -> bool:
    key = ("llama-attn", cls)
    if key not in _checked:
        try:
            import inspect

            params = list(inspect.signature(cls.forward).parameters)
            src = inspect.getsource(cls.forward)
            _checked[key] = (params == _LLAMA_ATTN_PARAMS
                             and "query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)" in src
                             and "key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)" in src
                             and "value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)" in src
                             and "query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)" in src
                             and "dropout=0.0 if not self.training else self.attention_dropout" in src
                             and "scaling=self.scaling" in src
                             and "attn_output = attn_output.reshape(*input_shape, -1).contiguous()" in src
                             and "attn_output = self.o_proj(attn_output)" in src)
        except Exception:
            _checked[key] = False
        if not _checked[key]:
            _warn_once("llama-attn", "LlamaAttention.forward is not the code this patch restates: transformers' own code stays "
                       "in place")
    return _checked[key]


# ---------------------------------------------------------------------------
# decoder layer: residual add + RMSNorm in one launch each way
# ---------------------------------------------------------------------------
_LLAMA_LAYER_PARAMS = ["self", "hidden_states", "attention_mask", "position_ids", "past_key_values", "use_cache",
                       "position_embeddings", "kwargs"]


def _norm_eps(norm) -> float:
    return float(getattr(norm, "_dalm_eps", getattr(norm, "variance_epsilon", getattr(norm, "eps", 1e-6))))


def _llama_layer_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                         position_embeddings=None, **kwargs):
    """transformers' LlamaDecoderLayer.forward with its two (residual add, RMSNorm) pairs on `dalm_rms_norm_{fwd,bwd}`:
      input_layernorm         : the norm alone forward; its backward kernel also adds the residual-path gradient
      post_attention_layernorm: residual + attention output and the norm in one launch; one launch backward
    instead of add 18 us + norm 21 us forward and norm backward 60 us + gradient add 17 us per pair (cfg3).  Same values as
    LlamaRMSNorm up to the f32 summation order of mean(x^2)."""
    from . import tower_ops

    n1, n2 = self.input_layernorm, self.post_attention_layernorm
    if not (tower_ops.rms_norm_supported(hidden_states, n1.weight) and tower_ops.rms_norm_supported(hidden_states, n2.weight)):
        return self._dalm_orig_forward(hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                       past_key_values=past_key_values, use_cache=use_cache,
                                       position_embeddings=position_embeddings, **kwargs)
    residual, normed = tower_ops.add_rms_norm(hidden_states, None, n1.weight, _norm_eps(n1))
    attn_out, _ = self.self_attn(hidden_states=normed, attention_mask=attention_mask, position_ids=position_ids,
                                 past_key_values=past_key_values, use_cache=use_cache,
                                 position_embeddings=position_embeddings, **kwargs)
    residual, normed = tower_ops.add_rms_norm(residual, attn_out, n2.weight, _norm_eps(n2))
    return residual + self.mlp(normed)


def use_fused_residual_norm(model: torch.nn.Module) -> int:
    """Patch LlamaDecoderLayer modules whose forward has the signature this file was written against (transformers 5.x);
    DALM_NORM_KERNEL=0 disables.  Returns how many layers were patched."""
    if os.environ.get("DALM_NORM_KERNEL", "1") == "0":
        return 0
    import inspect

    n = 0
    for mod in model.modules():
        if type(mod).__name__ != "LlamaDecoderLayer":
            continue
        try:
            params = [p if p != "kwargs" else "kwargs" for p in inspect.signature(type(mod).forward).parameters]
        except (TypeError, ValueError):
            continue
        if params != _LLAMA_LAYER_PARAMS:
            continue
        ok = True
        for name in ("input_layernorm", "post_attention_layernorm"):
            nm = getattr(mod, name, None)
            if nm is None or not type(nm).__name__.endswith("RMSNorm") or getattr(nm, "weight", None) is None \
                    or nm.weight.dim() != 1 or "Gemma" in type(nm).__name__:
                ok = False
        if not ok or not all(hasattr(mod, a) for a in ("self_attn", "mlp")):
            continue
        mod._dalm_orig_forward = mod.forward
        mod.forward = types.MethodType(_llama_layer_forward, mod)
        n += 1
    return n
