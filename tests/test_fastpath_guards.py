"""Run-time guards of the tower patches: a patch goes in only where transformers' own function IS the formula
the replacement implements - checked numerically at patch time, on the CPU here (on a GPU the same guards compare against the
HIP kernels).  The reference reaches these functions through `self.generator_model(...)`
(dalm/models/rag_e2e_base_model.py:104-106)."""
import warnings

import pytest
import torch


@pytest.fixture()
def llama():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=100)
    return LlamaForCausalLM(cfg)


@pytest.fixture()
def clean():
    """Undo the process-wide rope swap and the guard cache around a test."""
    import transformers.models.llama.modeling_llama as ml

    from dalm_amd.models import fastpath

    saved = getattr(ml, "_dalm_orig_apply_rotary_pos_emb", ml.apply_rotary_pos_emb)
    fastpath._checked.clear()
    fastpath._warned.clear()
    yield ml, fastpath, saved
    ml.apply_rotary_pos_emb = saved
    ml._dalm_orig_apply_rotary_pos_emb = saved
    fastpath._checked.clear()


def test_patches_go_in_on_the_stock_functions_and_keep_the_values(llama, clean):
    ml, fastpath, saved = clean
    ml.apply_rotary_pos_emb = saved
    x = torch.randint(0, 100, (2, 9))
    with torch.no_grad():
        want = llama(x).logits
    assert fastpath.use_native_rms_norm(llama) == 5 and fastpath.use_swiglu_kernel(llama) == 2 and fastpath.use_roll_rope(llama)
    assert ml.apply_rotary_pos_emb.__name__ in ("_rope_hip", "_rope_roll")
    with torch.no_grad():
        got = llama(x).logits
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_a_different_rotary_function_is_refused(llama, clean):
    ml, fastpath, saved = clean

    def other(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):      # another model's convention: interleaved pairs
        a, b = saved(q, k, cos, sin)
        return a * 1.01, b

    ml.apply_rotary_pos_emb = other
    ml._dalm_orig_apply_rotary_pos_emb = other
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert fastpath.use_roll_rope(llama) is False
    assert ml.apply_rotary_pos_emb is other                              # transformers' code stays in place
    assert any("apply_rotary_pos_emb is not" in str(m.message) for m in w)


def test_a_different_mlp_or_norm_formula_is_refused(llama, clean):
    ml, fastpath, saved = clean
    mlp_cls = type(llama.model.layers[0].mlp)
    norm_cls = type(llama.model.norm)
    orig_mlp, orig_norm = mlp_cls.forward, norm_cls.forward
    try:
        mlp_cls.forward = lambda self, x: self.down_proj(torch.nn.functional.gelu(self.gate_proj(x)) * self.up_proj(x))
        norm_cls.forward = lambda self, x: self.weight * x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1.0)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert fastpath.use_swiglu_kernel(llama) == 0
            assert fastpath.use_native_rms_norm(llama) == 0
        assert sum("transformers' own code stays in place" in str(m.message) for m in w) == 2
        assert "forward" not in llama.model.layers[0].mlp.__dict__ and "forward" not in llama.model.norm.__dict__
    finally:
        mlp_cls.forward, norm_cls.forward = orig_mlp, orig_norm


def test_falcon_shared_kv_head_is_broadcast_for_sdpa_and_keeps_the_values():
    """Falcon-7B's multi-query attention (one key / value head): the patch hands scaled_dot_product_attention equal head counts
    (it otherwise runs its unfused math path) - same logits and gradients as transformers' broadcast form."""
    from transformers import FalconConfig, FalconForCausalLM

    from dalm_amd.models import fastpath

    torch.manual_seed(0)
    cfg = FalconConfig(num_hidden_layers=2, hidden_size=256, num_attention_heads=4, vocab_size=100)   # head width 64
    ref = FalconForCausalLM(cfg).train()
    new = FalconForCausalLM(cfg).train()
    new.load_state_dict(ref.state_dict())
    x = torch.randint(0, 100, (2, 9))
    am = torch.ones(2, 9, dtype=torch.long)
    am[0, :3] = 0                                                   # left padding
    assert fastpath.use_capturable_falcon_heads(new) == 2
    assert fastpath.use_falcon_attention_kernels(new) == 2          # the patched training call is where the broadcast happens
    att = new.transformer.h[0].self_attention
    # the module's own head count is untouched - readers outside the training call (KV cache, export) see 1
    assert att._dalm_expand_kv and att.num_kv_heads == 1 and att.num_heads == 4
    seen = []
    orig_sdpa = torch.nn.functional.scaled_dot_product_attention

    def spy(q, k, v, *a, **kw):
        seen.append((tuple(q.shape), tuple(k.shape)))
        return orig_sdpa(q, k, v, *a, **kw)

    torch.nn.functional.scaled_dot_product_attention = spy
    try:
        new(input_ids=x, attention_mask=am, use_cache=False)         # the trainers' call (models/rag_e2e_base_model.py forward)
    finally:
        torch.nn.functional.scaled_dot_product_attention = orig_sdpa
    assert seen and all(qs[1] == ks[1] == 4 for qs, ks in seen)      # equal head counts in the training call
    new.eval()
    with torch.no_grad():
        cache = new(input_ids=x, attention_mask=am, use_cache=True).past_key_values
        want_cache = ref.eval()(input_ids=x, attention_mask=am, use_cache=True).past_key_values
    k_new, k_ref = cache.layers[0].keys, want_cache.layers[0].keys
    assert k_new.shape == k_ref.shape and k_new.shape[1] == 1        # the KV cache holds ONE head, as transformers' own
    torch.testing.assert_close(k_new, k_ref, rtol=1e-5, atol=1e-5)
    ref.train(); new.train()
    outs = []
    for m in (ref, new):
        logits = m(input_ids=x, attention_mask=am, use_cache=False).logits
        logits.square().sum().backward()
        outs.append((logits.detach(), m.transformer.h[0].self_attention.query_key_value.weight.grad.clone()))
    torch.testing.assert_close(outs[1][0], outs[0][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(outs[1][1], outs[0][1], rtol=1e-4, atol=1e-5)


def test_falcon_layer_patch_goes_in_on_the_stock_code_and_is_refused_on_other_code():
    """FalconDecoderLayer / FalconMLP (7B flavour): the patch restates transformers' statements and checks them on the class's
    source.  On CPU tensors the patched layer hands over to transformers' forward (the kernels take GPU bf16 only): same values."""
    from transformers import FalconConfig, FalconForCausalLM

    from dalm_amd.models import fastpath

    fastpath._checked.clear()
    torch.manual_seed(0)
    cfg = FalconConfig(num_hidden_layers=2, hidden_size=128, num_attention_heads=4, vocab_size=100)
    m = FalconForCausalLM(cfg).eval()
    x = torch.randint(0, 100, (2, 9))
    with torch.no_grad():
        want = m(input_ids=x).logits
    assert fastpath.use_falcon_layer_kernels(m) == 2
    layer = m.transformer.h[0]
    assert layer.forward.__func__ is fastpath._falcon_layer_forward and layer.mlp.forward.__func__ is fastpath._falcon_mlp_forward
    with torch.no_grad():
        got = m(input_ids=x).logits
    torch.testing.assert_close(got, want, rtol=0, atol=0)

    # dropout in the residual adds, or the new decoder architecture: transformers' code stays
    for kw in ({"hidden_dropout": 0.1}, {"new_decoder_architecture": True, "num_kv_heads": 2}, {"parallel_attn": False}):
        other = FalconForCausalLM(FalconConfig(num_hidden_layers=1, hidden_size=128, num_attention_heads=4, vocab_size=100, **kw))
        assert fastpath.use_falcon_layer_kernels(other) == 0
        assert "forward" not in other.transformer.h[0].__dict__

    # another forward on the class: refused with a warning
    cls = type(layer)
    orig = cls.forward
    fastpath._checked.clear()
    fastpath._warned.clear()
    try:
        def forward(self, hidden_states, alibi, attention_mask, position_ids=None, layer_past=None, use_cache=False,
                    output_attentions=False, position_embeddings=None, **kwargs):
            return orig(self, hidden_states, alibi, attention_mask, position_ids, layer_past, use_cache, output_attentions,
                        position_embeddings, **kwargs)

        cls.forward = forward
        fresh = FalconForCausalLM(cfg)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert fastpath.use_falcon_layer_kernels(fresh) == 0
        assert any("FalconDecoderLayer.forward is not the code this patch restates" in str(m_.message) for m_ in w)
    finally:
        cls.forward = orig
        fastpath._checked.clear()


def test_dalm_sdpa_registration_and_cpu_delegation():
    """"dalm_sdpa" is a registered transformers attention implementation (models/attention.py).  On CPU tensors (and for anything
    else the HIP kernels do not take) it IS transformers' sdpa_attention_forward: same logits, same gradients.
    DALM_ATTN_KERNEL=0 leaves the model on "sdpa"."""
    import os

    from transformers import LlamaConfig, LlamaForCausalLM

    from dalm_amd.models import attention

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                      vocab_size=100)
    ref = LlamaForCausalLM(cfg)
    new = LlamaForCausalLM(cfg)
    new.load_state_dict(ref.state_dict())
    os.environ["DALM_ATTN_KERNEL"] = "0"
    try:
        assert attention.use_hip_attention_backward(new) is False and new.config._attn_implementation == "sdpa"
    finally:
        os.environ.pop("DALM_ATTN_KERNEL", None)
    assert attention.use_hip_attention_backward(new) is True and new.config._attn_implementation == "dalm_sdpa"
    x = torch.randint(0, 100, (2, 9))
    am = torch.ones(2, 9, dtype=torch.long)
    am[0, :3] = 0
    outs = []
    for m in (ref, new):
        logits = m(input_ids=x, attention_mask=am).logits
        logits.square().sum().backward()
        outs.append((logits.detach(), m.model.layers[0].self_attn.q_proj.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # head widths the kernels do not take: the model stays on "sdpa"
    other = LlamaForCausalLM(LlamaConfig(hidden_size=96, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                                         num_key_value_heads=2, vocab_size=50))
    assert attention.use_hip_attention_backward(other) is False and other.config._attn_implementation == "sdpa"
    # nothing the kernels take on the CPU
    q = torch.randn(1, 2, 8, 128, dtype=torch.bfloat16, requires_grad=True)
    assert attention.supported(q, q, q, None, 0.0, True, {}) is False


def test_falcon_attention_patch_is_refused_on_other_code():
    from transformers import FalconConfig, FalconForCausalLM

    from dalm_amd.models import fastpath

    fastpath._checked.clear()
    fastpath._warned.clear()
    cfg = FalconConfig(num_hidden_layers=1, hidden_size=128, num_attention_heads=2, vocab_size=100)
    m = FalconForCausalLM(cfg)
    assert fastpath.use_falcon_attention_kernels(m) == 1
    att = m.transformer.h[0].self_attention
    assert att.forward.__func__ is fastpath._falcon_attention_forward
    # alibi positions, attention dropout, other head widths: transformers' code stays
    for kw in ({"alibi": True}, {"attention_dropout": 0.1}, {"num_attention_heads": 4}):
        kw = {"num_attention_heads": 2, **kw}
        other = FalconForCausalLM(FalconConfig(num_hidden_layers=1, hidden_size=128, vocab_size=100, **kw))
        assert fastpath.use_falcon_attention_kernels(other) == 0
    cls = type(att)
    orig = cls.forward
    fastpath._checked.clear()
    try:
        def forward(self, hidden_states, alibi, attention_mask, position_ids=None, layer_past=None, use_cache=False,
                    output_attentions=False, position_embeddings=None, **kwargs):
            return orig(self, hidden_states, alibi, attention_mask, position_ids, layer_past, use_cache, output_attentions,
                        position_embeddings, **kwargs)

        cls.forward = forward
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert fastpath.use_falcon_attention_kernels(FalconForCausalLM(cfg)) == 0
        assert any("FalconAttention.forward is not the code this patch restates" in str(m_.message) for m_ in w)
    finally:
        cls.forward = orig
        fastpath._checked.clear()
