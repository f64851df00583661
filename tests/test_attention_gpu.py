"""`dalm_attn_bwd` / `dalm_attn_mask_bits` (dalm_amd/csrc/attn.hip) behind the "dalm_sdpa" attention implementation
(dalm_amd/models/attention.py) against torch.nn.functional.scaled_dot_product_attention as transformers'
sdpa_attention_forward calls it (the reference reaches it through self.generator_model(...),
dalm/models/rag_e2e_base_model.py:104-106, and differentiates it in loss.backward(), train_rage2e.py:466):

* forward (`dalm_attn_fwd`): against a float64 evaluation, no further from it than torch's kernel is (x 1.5 + 1e-3); with
  DALM_ATTN_FWD_KERNEL=0 (torch's kernel called for its log-sum-exp) EQUAL to F.scaled_dot_product_attention;
* backward: dq, dk, dv against a float64 evaluation of the same masked softmax attention, no further from it than
  torch's own bf16 backward is (x 1.5 + 1e-3), for HF's causal + left-padding masks, ragged lengths, rows with no live
  key, the pure causal case without a mask, arbitrary boolean masks, transposed ([B, T, H, hd] memory) and contiguous views;
* the mask bit words against a numpy packing;
* a Llama layer on "dalm_sdpa" against the same layer on "sdpa": logits and LoRA-free parameter gradients."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _hf_mask(B, T, starts, dev):
    col = torch.arange(T, device=dev)
    st = torch.tensor(starts, device=dev)
    return ((col[None, None, :] <= col[None, :, None]) & (col[None, None, :] >= st[:, None, None]))[:, None]


def _ref64(q, k, v, mask, causal, scale, go):
    q, k, v = [t.detach().double().requires_grad_(True) for t in (q, k, v)]
    s = (q @ k.transpose(-1, -2)) * scale
    T = s.shape[-1]
    live = torch.ones(T, T, dtype=torch.bool, device=s.device).tril() if causal else torch.ones(T, T, dtype=torch.bool, device=s.device)
    live = live[None, None] if mask is None else (mask & live)
    s = s.masked_fill(~live, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)                       # rows without a live key: zero output, zero gradient
    o = p @ v
    o.backward(go.double())
    return o, q.grad, k.grad, v.grad


def _run(fn, q, k, v, go):
    q, k, v = [t.detach().clone().requires_grad_(True) for t in (q, k, v)]
    o = fn(q, k, v)
    o.backward(go)
    return o.detach(), q.grad, k.grad, v.grad


CASES = [
    # B, H, T, starts (left padding per batch row; None = no mask, is_causal), layout, head width
    (2, 3, 256, [0, 37], "bthd", 128),
    (3, 2, 256, [0, 255, 128], "bhtd", 128),
    (2, 2, 200, [5, 150], "bthd", 128),
    (1, 4, 96, [0], "bthd", 128),
    (2, 2, 40, [3, 0], "bhtd", 128),
    (2, 3, 256, None, "bthd", 128),
    (1, 2, 333 // 8 * 8, None, "bthd", 128),
    (2, 2, 512, [100, 0], "bthd", 128),
    (2, 5, 256, [0, 37], "bhtd", 64),           # Falcon-7b's head width
    (3, 2, 256, [0, 255, 128], "bthd", 64),
    (2, 3, 200, [5, 150], "bhtd", 64),
    (2, 2, 40, [3, 0], "bthd", 64),
    (2, 3, 320, None, "bhtd", 64),
]


@pytest.mark.parametrize("B,H,T,starts,layout,hd", CASES)
def test_backward_vs_fp64_and_torch(dev, B, H, T, starts, layout, hd):
    from dalm_amd.models import attention

    g = torch.Generator().manual_seed(B * 1000 + T)
    def mk():
        if layout == "bthd":
            return (torch.randn(B, T, H, hd, generator=g) * 1.2).bfloat16().to(dev).transpose(1, 2)
        return (torch.randn(B, H, T, hd, generator=g) * 1.2).bfloat16().to(dev)
    q, k, v, go = mk(), mk(), mk(), mk()
    mask = None if starts is None else _hf_mask(B, T, starts, dev)
    causal = starts is None
    scale = hd ** -0.5
    assert attention.supported(q.requires_grad_(True), k, v, mask, 0.0, causal, {})

    ours = _run(lambda a, b, c: attention._SdpaHipBackward.apply(a, b, c, mask, scale, causal), q, k, v, go)
    theirs = _run(lambda a, b, c: torch.nn.functional.scaled_dot_product_attention(a, b, c, attn_mask=mask, is_causal=causal,
                                                                                    scale=scale), q, k, v, go)
    ref = _ref64(q, k, v, mask, causal, scale, go)
    assert torch.isfinite(ours[0]).all()
    assert _rel(ours[0], ref[0]) <= 1.5 * _rel(theirs[0], ref[0]) + 1e-3
    assert ours[0].transpose(1, 2).is_contiguous()           # [B, T, H, hd] memory: the caller's transpose(1, 2).contiguous() is free
    os.environ["DALM_ATTN_FWD_KERNEL"] = "0"
    try:
        lib = _run(lambda a, b, c: attention._SdpaHipBackward.apply(a, b, c, mask, scale, causal), q, k, v, go)
    finally:
        os.environ.pop("DALM_ATTN_FWD_KERNEL", None)
    assert torch.equal(lib[0], theirs[0])
    for a, r in zip(lib[1:], ref[1:]):
        assert _rel(a, r) < 2e-2
    for name, a, b, r in zip(("dq", "dk", "dv"), ours[1:], theirs[1:], ref[1:]):
        assert torch.isfinite(a).all(), name
        e_a, e_b = _rel(a, r), _rel(b, r)
        assert e_a <= 1.5 * e_b + 1e-3, (name, e_a, e_b)
        assert a.stride() == b.stride() or a.is_contiguous()
    if starts is not None:                                   # query rows in the padding have no live key: exactly zero dq
        for b_, st in enumerate(starts):
            if st > 0:
                assert float(ours[0][b_, :, :st].abs().max()) == 0.0
                assert float(ours[1][b_, :, :st].abs().max()) == 0.0
                assert float(ours[2][b_, :, :st].abs().max()) == 0.0 and float(ours[3][b_, :, :st].abs().max()) == 0.0


def test_arbitrary_boolean_mask(dev):
    from dalm_amd.models import attention

    B, H, T, hd = 2, 2, 128, 128
    g = torch.Generator().manual_seed(7)
    q, k, v, go = [(torch.randn(B, T, H, hd, generator=g)).bfloat16().to(dev).transpose(1, 2) for _ in range(4)]
    mask = (torch.rand(B, 1, T, T, generator=g) < 0.3).to(dev)
    mask[:, :, :, 0] = True                                  # every row keeps a key
    mask[0, 0, 64:96, :] = False
    mask[0, 0, 64:96, 5] = True                              # a block of rows with one live key, 32 x 32 tiles entirely dead
    scale = 0.11
    ours = _run(lambda a, b, c: attention._SdpaHipBackward.apply(a, b, c, mask, scale, False), q, k, v, go)
    theirs = _run(lambda a, b, c: torch.nn.functional.scaled_dot_product_attention(a, b, c, attn_mask=mask, scale=scale), q, k, v, go)
    ref = _ref64(q, k, v, mask, False, scale, go)
    for a, b, r in zip(ours, theirs, ref):
        assert _rel(a, r) <= 1.5 * _rel(b, r) + 1e-3


def test_mask_bit_words(dev):
    from dalm_amd import hip

    B, T = 3, 100
    W = (T + 31) // 32
    g = torch.Generator().manual_seed(1)
    mask = (torch.rand(B, 1, T, T, generator=g) < 0.5)
    mask[1] = False
    for causal in (0, 1):
        rows = torch.empty(B * 32 * W * W, dtype=torch.int32, device=dev)
        cols = torch.empty_like(rows)
        live = torch.empty(B * W * W, dtype=torch.uint8, device=dev)
        m = mask.to(dev)
        hip.call("dalm_attn_mask_bits", hip.ptr(m), B, T, m.stride(0), m.stride(2), causal, hip.ptr(rows), hip.ptr(cols), hip.ptr(live),
                 hip.stream())
        mm = mask[:, 0].numpy().copy()
        if causal:
            mm &= np.tril(np.ones((T, T), dtype=bool))[None]
        pad = np.zeros((B, 32 * W, 32 * W), dtype=bool)
        pad[:, :T, :T] = mm
        weights = (1 << np.arange(32, dtype=np.uint64))
        want_rows = (pad.reshape(B, 32 * W, W, 32).astype(np.uint64) * weights).sum(-1).astype(np.uint32)
        want_cols = (pad.transpose(0, 2, 1).reshape(B, 32 * W, W, 32).astype(np.uint64) * weights).sum(-1).astype(np.uint32)
        want_live = pad.reshape(B, W, 32, W, 32).any(axis=(2, 4)).astype(np.uint8)
        assert np.array_equal(rows.cpu().numpy().view(np.uint32).reshape(B, 32 * W, W), want_rows)
        assert np.array_equal(cols.cpu().numpy().view(np.uint32).reshape(B, 32 * W, W), want_cols)
        assert np.array_equal(live.cpu().numpy().reshape(B, W, W), want_live)


def test_llama_layer_on_dalm_sdpa_matches_sdpa(dev):
    from transformers import LlamaConfig, LlamaForCausalLM

    from dalm_amd.models import attention

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=300)
    ref = LlamaForCausalLM(cfg).to(dev).to(torch.bfloat16).train()
    new = copy.deepcopy(ref)
    assert attention.use_hip_attention_backward(new) and new.config._attn_implementation == "dalm_sdpa"
    assert ref.config._attn_implementation == "sdpa"
    B, T = 3, 64
    ids = torch.randint(0, 300, (B, T), device=dev)
    am = torch.ones(B, T, dtype=torch.long, device=dev)
    am[0, :20] = 0
    am[2, :63] = 0
    outs = []
    for m in (ref, new):
        logits = m(input_ids=ids, attention_mask=am).logits
        (logits.float() * am[..., None]).square().sum().backward()
        outs.append((logits.detach(), [p.grad.detach().clone() for p in m.parameters()]))
    live = am.bool()
    assert _rel(outs[1][0][live], outs[0][0][live]) < 1e-2
    for a, b in zip(outs[1][1], outs[0][1]):
        assert _rel(a, b) < 2e-2
    # no gradient wanted: transformers' own path, torch's kernel, the same values as the "sdpa" model
    with torch.no_grad():
        assert torch.equal(new(input_ids=ids, attention_mask=am).logits[live], ref(input_ids=ids, attention_mask=am).logits[live])


def test_falcon_attention_patch_matches_transformers(dev):
    """FalconAttention (multi-query, rotary, head width 64) with `attention.sdpa` in place of F.scaled_dot_product_attention:
    hidden states and the gradient reaching the embeddings against transformers' own forward."""
    from transformers import FalconConfig
    from transformers.models.falcon.modeling_falcon import FalconModel

    from dalm_amd.models import fastpath

    cfg = FalconConfig(vocab_size=300, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, multi_query=True,
                       parallel_attn=True, new_decoder_architecture=False, bias=False, alibi=False, hidden_dropout=0.0,
                       attention_dropout=0.0)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(1)
    ref = FalconModel(cfg).to(dev).to(torch.bfloat16).train()
    new = copy.deepcopy(ref)
    assert fastpath.use_capturable_falcon_heads(new) == 2 and fastpath.use_falcon_attention_kernels(new) == 2
    assert new.h[0].self_attention.forward.__func__ is fastpath._falcon_attention_forward
    B, T = 3, 96
    ids = torch.randint(0, 300, (B, T), device=dev)
    am = torch.ones(B, T, dtype=torch.long, device=dev)
    am[1, :30] = 0
    outs = []
    for m in (ref, new):
        for p in m.parameters():
            p.requires_grad_(False)
        e = m.word_embeddings(ids).detach().requires_grad_(True)
        h = m(inputs_embeds=e, attention_mask=am).last_hidden_state
        (h.float() * am[..., None] * torch.linspace(-1, 1, h.shape[-1], device=dev)).sum().backward()
        outs.append((h.detach(), e.grad.detach()))
    live = am.bool()
    assert _rel(outs[1][0][live], outs[0][0][live]) < 1e-2
    assert _rel(outs[1][1][live], outs[0][1][live]) < 2e-2


@pytest.mark.parametrize("hd,H,T,batch_tables", [(128, 3, 256, False), (64, 5, 200, False), (128, 2, 96, True)])
def test_rotation_backward_in_the_attention_epilogues_equals_the_two_node_form(dev, hd, H, T, batch_tables):
    """`rope_sdpa` (rotary embedding + attention as one autograd node; the rotation's backward applied in dalm_attn_bwd's
    epilogues) against `tower_ops.rope_qk` followed by `attention.sdpa` (two nodes, `dalm_rope_qk` launched for the backward):
    the same output and the SAME dq, dk, dv - the epilogue keeps that kernel's rounding points."""
    from dalm_amd.models import attention, tower_ops

    B = 2
    g = torch.Generator().manual_seed(hd + T)
    q, k, v, go = [(torch.randn(B, T, H, hd, generator=g)).bfloat16().to(dev).transpose(1, 2) for _ in range(4)]
    pos = torch.arange(T).float()[None, :, None] + (torch.tensor([0.0, 7.0])[:, None, None] if batch_tables else 0.0)
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.cat((pos * inv, pos * inv), -1)                       # [1 or B, T, hd], halves duplicated as transformers builds them
    cos, sin = ang.cos().bfloat16().to(dev), ang.sin().bfloat16().to(dev)
    mask = _hf_mask(B, T, [0, T // 3], dev)
    scale = hd ** -0.5
    assert attention.rope_fusable(q, k, cos, sin)

    def two_nodes(a, b, c):
        a2, b2 = tower_ops.rope_qk(a, b, cos, sin)
        return attention.sdpa(a2, b2, c, mask, scale, False)

    one = _run(lambda a, b, c: attention.rope_sdpa(a, b, c, cos, sin, mask, scale, False), q, k, v, go)
    two = _run(two_nodes, q, k, v, go)
    for name, a, b in zip(("out", "dq", "dk", "dv"), one, two):
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))
    # and against transformers' own rotary function differentiated by autograd, around torch's attention
    import transformers.models.llama.modeling_llama as ml

    rope = getattr(ml, "_dalm_orig_apply_rotary_pos_emb", ml.apply_rotary_pos_emb)
    eager = _run(lambda a, b, c: torch.nn.functional.scaled_dot_product_attention(*rope(a, b, cos, sin), c, attn_mask=mask, scale=scale),
                 q, k, v, go)
    for a, b in zip(one, eager):
        assert _rel(a, b) < 1.5e-2


def test_llama_attention_node_matches_the_unpatched_dalm_sdpa_model(dev):
    from transformers import LlamaConfig, LlamaForCausalLM

    from dalm_amd.models import attention, fastpath

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                      vocab_size=300)
    a = LlamaForCausalLM(cfg).to(dev).to(torch.bfloat16).train()
    b = copy.deepcopy(a)
    for m in (a, b):
        assert attention.use_hip_attention_backward(m) and fastpath.use_roll_rope(m)
    assert fastpath.use_llama_attention_node(b) == 2
    assert b.model.layers[0].self_attn.forward.__func__ is fastpath._llama_attention_forward
    B, T = 3, 64
    ids = torch.randint(0, 300, (B, T), device=dev)
    am = torch.ones(B, T, dtype=torch.long, device=dev)
    am[0, :20] = 0
    outs = []
    for m in (a, b):
        logits = m(input_ids=ids, attention_mask=am).logits
        (logits.float() * am[..., None]).square().sum().backward()
        outs.append((logits.detach(), [p.grad.detach().clone() for p in m.parameters()]))
    assert torch.equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[1][1], outs[0][1]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("B,H,T,hd,pad", [(3, 4, 128, 64, [128, 90, 17]), (2, 3, 50, 64, [50, 31]), (2, 2, 256, 128, [256, 200])])
def test_attention_dropout_mask_and_gradients(dev, B, H, T, hd, pad):
    """BERT's attention dropout inside the kernels (bidirectional padding mask, p = 0.1): the keep mask the forward used - read off
    its output with V = identity-like probes - equals oracle/attn_dropout.py bit for bit; out, dq, dk, dv against a float64
    evaluation of softmax -> (P o M) / (1 - p) -> P V with THAT mask; forward and backward therefore use the same bits."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import attn_dropout as AD

    from dalm_amd.models import attention, lora_ops

    p, salt = 0.1, 0x5A17
    g = torch.Generator().manual_seed(T + hd)
    q, k, v, go = [(torch.randn(B, T, H, hd, generator=g)).bfloat16().to(dev).transpose(1, 2) for _ in range(4)]
    col = torch.arange(T, device=dev)
    lens = torch.tensor(pad, device=dev)
    mask = (col[None, None, None, :] < lens[:, None, None, None]).expand(B, 1, T, T)          # HF's bidirectional padding mask
    scale = hd ** -0.5
    assert attention.supported(q.requires_grad_(True), k, v, mask, p, False, {})
    seed = int(lora_ops.dropout_seed(dev).item())
    keep = torch.from_numpy(AD.keep_mask(seed, salt, B, H, T, p)).to(dev)
    assert abs(float((~keep).float().mean()) - p) < 0.01

    ours = _run(lambda a, b, c: attention.sdpa(a, b, c, mask, scale, False, p, salt), q, k, v, go)
    q64, k64, v64 = [t.detach().double().requires_grad_(True) for t in (q, k, v)]
    s = (q64 @ k64.transpose(-1, -2)) * scale
    s = s.masked_fill(~mask, float("-inf"))
    pr = torch.softmax(s, -1) * keep.double() / (1.0 - p)
    o64 = pr @ v64
    o64.backward(go.double())
    for name, a, r in zip(("out", "dq", "dk", "dv"), ours, (o64, q64.grad, k64.grad, v64.grad)):
        assert torch.isfinite(a).all(), name
        assert _rel(a, r) < 1.2e-2, (name, _rel(a, r))
    # a different salt draws a different mask; p = 0 is the plain attention
    other = _run(lambda a, b, c: attention.sdpa(a, b, c, mask, scale, False, p, salt + 1), q, k, v, go)
    assert not torch.equal(other[0], ours[0])
    plain = _run(lambda a, b, c: attention.sdpa(a, b, c, mask, scale, False, 0.0, 0), q, k, v, go)
    theirs = _run(lambda a, b, c: torch.nn.functional.scaled_dot_product_attention(a, b, c, attn_mask=mask, scale=scale), q, k, v, go)
    for a, b in zip(plain, theirs):
        assert _rel(a, b) < 1.2e-2


def test_every_keep_bit_of_the_forward_equals_the_oracle(dev):
    """V = one-hot columns: out[b, h, i, d] = sum_j P_drop[i, j] [j == d] exposes every element of P o M / (1 - p) for T <= hd."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import attn_dropout as AD

    from dalm_amd.models import attention, lora_ops

    B, H, T, hd, p, salt = 2, 3, 64, 64, 0.25, 99
    q = torch.zeros(B, H, T, hd, dtype=torch.bfloat16, device=dev).requires_grad_(True)       # uniform probabilities 1 / T
    k = torch.zeros_like(q)
    v = torch.eye(T, hd, dtype=torch.bfloat16, device=dev).expand(B, H, T, hd).contiguous()
    out = attention.sdpa(q, k, v, None, 1.0, False, p, salt)
    got = out.detach().float() > 0
    want = torch.from_numpy(AD.keep_mask(int(lora_ops.dropout_seed(dev).item()), salt, B, H, T, p)).to(dev)
    assert torch.equal(got, want)
    assert torch.allclose(out.detach().float()[got], torch.tensor(1.0 / T / (1 - p), device=dev), rtol=1e-2)


def test_bert_layer_on_dalm_sdpa_trains_with_dropout(dev):
    """A BERT encoder (head width 64) on "dalm_sdpa" in training mode: runs through the kernels (attention dropout 0.1), matches the
    "sdpa" model exactly in eval mode through no_grad, and in training mode with dropout off."""
    from transformers import BertConfig, BertModel

    from dalm_amd.models import attention

    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, vocab_size=300,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref = BertModel(cfg).to(dev).train()                            # f32 parameters under bf16 autocast, as the trainers run it
    new = copy.deepcopy(ref)
    assert attention.use_hip_attention_backward(new) and new.config._attn_implementation == "dalm_sdpa"
    B, T = 4, 64
    ids = torch.randint(0, 300, (B, T), device=dev)
    am = torch.ones(B, T, dtype=torch.long, device=dev)
    am[1, 40:] = 0
    outs = []
    for m in (ref, new):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = m(input_ids=ids, attention_mask=am)[0]
        # (a sum of squares of a LayerNorm's output is a constant: weighted sum instead)
        (h.float() * am[..., None] * torch.linspace(-1, 1, h.shape[-1], device=dev)).sum().backward()
        # (the key bias has a zero gradient in exact arithmetic - softmax ignores a shift of every score of a row - so what
        # it holds is rounding noise: left out of the comparison)
        outs.append((h.detach(), {n: p_.grad.detach().clone() for n, p_ in m.named_parameters()
                                  if p_.grad is not None and not n.endswith("key.bias")}))
    live = am.bool()
    assert _rel(outs[1][0][live], outs[0][0][live]) < 1e-2
    for n in outs[0][1]:
        assert _rel(outs[1][1][n], outs[0][1][n]) < 3e-2, (n, _rel(outs[1][1][n], outs[0][1][n]))
    # with attention dropout: finite, different from the no-dropout output, deterministic for a fixed seed word and call count
    new.config.attention_probs_dropout_prob = 0.1
    for layer in new.encoder.layer:
        layer.attention.self.dropout.p = 0.1
        layer.attention.self._dalm_attn_calls = 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h1 = new(input_ids=ids, attention_mask=am)[0]
        for layer in new.encoder.layer:
            layer.attention.self._dalm_attn_calls = 0
        h2 = new(input_ids=ids, attention_mask=am)[0]
    assert torch.isfinite(h1).all() and torch.equal(h1, h2) and not torch.equal(h1[live], outs[1][0][live])
    h1.float().sum().backward()


@pytest.mark.parametrize("packed_mode", [False, True])
def test_multi_query_shared_head_through_stride0_views(dev, packed_mode):
    """Falcon-7B's multi-query attention: ONE key / value head.  `rope_sdpa` with a [B, 1, T, hd] key / value reads them through
    stride-0 head views (nothing broadcast in memory) and sums dk / dv over the heads - against the same call on keys / values
    broadcast and materialised first (what transformers' FalconAttention.forward builds), padded and packed layouts."""
    from dalm_amd import packed
    from dalm_amd.models import attention

    B, H, T, hd = 3, 7, 128, 64
    g = torch.Generator().manual_seed(17)
    lens = [128, 50, 90]
    m2 = (torch.arange(T).unsqueeze(0) >= (T - torch.tensor(lens)).unsqueeze(1)).long()
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    if packed_mode:
        rows, cu = packed.pack_plan(m2, shifted=True, multiple=64)
        _i, pos, mask, _v = packed.packed_inputs(torch.zeros(B, T, dtype=torch.long, device=dev), m2.to(dev), rows.to(dev), cu.to(dev), True)
        n, Bq = rows.numel(), 1
        ang = pos[0].float()[:, None] * inv[None, :]
        causal = False
    else:
        n, Bq = T, B
        mask = _hf_mask(B, T, [T - l for l in lens], dev)
        ang = torch.arange(T, device=dev).float()[:, None] * inv[None, :]
        causal = False
    cos = torch.cat((ang.cos(), ang.cos()), -1).to(torch.bfloat16)[None]
    sin = torch.cat((ang.sin(), ang.sin()), -1).to(torch.bfloat16)[None]
    fused = (0.7 * torch.randn(Bq, n, H + 2, hd, generator=g)).to(dev, torch.bfloat16)
    go = torch.randn(Bq, H, n, hd, generator=g).to(dev, torch.bfloat16)
    scale = hd ** -0.5

    def run(materialise):
        f = fused.clone().requires_grad_(True)
        q = f[..., :H, :].transpose(1, 2).reshape(Bq, H, n, hd)
        k1, v1 = f[..., H:H + 1, :].transpose(1, 2), f[..., H + 1:, :].transpose(1, 2)
        if materialise:
            k = k1.expand(Bq, H, n, hd).reshape(Bq, H, n, hd).contiguous()
            v = v1.expand(Bq, H, n, hd).reshape(Bq, H, n, hd).contiguous()
            assert attention.rope_fusable(q, k, cos, sin)
            out = attention.rope_sdpa(q, k, v, cos, sin, mask, scale, causal)
        else:
            assert attention.rope_fusable(q, k1, cos, sin) and k1.shape[1] == 1
            out = attention.rope_sdpa(q, k1, v1, cos, sin, mask, scale, causal)
        out.backward(go)
        return out.detach(), f.grad

    o_ref, g_ref = run(True)
    o_mqa, g_mqa = run(False)
    assert torch.equal(o_mqa, o_ref)                                   # the same kernel on the same values
    assert _rel(g_mqa[..., :H, :], g_ref[..., :H, :]) < 1e-6           # dq: identical launches
    assert _rel(g_mqa[..., H:, :], g_ref[..., H:, :]) < 4e-3           # dk, dv: per-head bf16 gradients summed over the heads
