"""`dalm_rope_qk` / `dalm_swiglu_*` (dalm_amd/csrc/tower.hip) against the eager chains they replace: transformers' own
`apply_rotary_pos_emb` and `LlamaMLP` (what the reference's generator runs, dalm/models/rag_e2e_base_model.py:104-106).
The kernels round where the eager ops round, so the rotary results must be EQUAL (forward and backward, bf16 and f32);
SwiGLU goes through exp(), where one f32 ulp of difference between two exp implementations may move a bf16 rounding: equal
up to one bf16 ulp on a vanishing fraction of elements, 1e-6 relative in f32."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _hf_rope():
    from transformers.models.llama import modeling_llama as m

    return getattr(m, "_dalm_orig_apply_rotary_pos_emb", m.apply_rotary_pos_emb)


def _cos_sin(B, T, hd, dtype, dev, g):
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
    if B % 2 == 0:       # transformers' default position_ids are [1, T]: one table broadcast over the batch
        B = 1
    pos = torch.arange(T).float()[None, :].expand(B, T) + torch.randint(0, 50, (B, 1), generator=g).float()
    fr = pos[..., None] * inv[None, None, :]
    emb = torch.cat((fr, fr), dim=-1)
    return emb.cos().to(dtype).to(dev), emb.sin().to(dtype).to(dev)


ROPE = [(18, 32, 32, 256, 128), (2, 4, 2, 7, 64), (3, 2, 2, 5, 8), (1, 3, 1, 9, 6), (2, 8, 8, 33, 96)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,Hq,Hk,T,hd", ROPE)
@pytest.mark.parametrize("layout", ["projection-view", "contiguous"])
def test_rope_kernel_equals_transformers_chain(dev, B, Hq, Hk, T, hd, dtype, layout):
    from dalm_amd.models import tower_ops

    g = torch.Generator().manual_seed(B * 100 + T + hd)
    cos, sin = _cos_sin(B, T, hd, dtype, dev, g)

    def make(H):
        x = torch.randn(B, T, H * hd, generator=g).to(dtype).to(dev)
        if layout == "projection-view":            # what LlamaAttention hands over: a transposed view of [B, T, H*hd]
            return x.requires_grad_(True), lambda t: t.view(B, T, H, hd).transpose(1, 2)
        x = x.view(B, T, H, hd).transpose(1, 2).contiguous()
        return x.requires_grad_(True), lambda t: t

    (q0, vq), (k0, vk) = make(Hq), make(Hk)
    q1, k1 = q0.detach().clone().requires_grad_(True), k0.detach().clone().requires_grad_(True)
    up_q = torch.randn(B, Hq, T, hd, generator=g).to(dtype).to(dev)
    up_k = torch.randn(B, T, Hk, hd, generator=g).to(dtype).to(dev).transpose(1, 2)      # a strided upstream gradient
    qe, ke = _hf_rope()(vq(q0), vk(k0), cos, sin)
    torch.autograd.backward([qe, ke], [up_q, up_k])
    assert tower_ops.rope_supported(vq(q1), vk(k1), cos, sin)
    qh, kh = tower_ops.rope_qk(vq(q1), vk(k1), cos, sin)
    torch.autograd.backward([qh, kh], [up_q, up_k])
    assert torch.equal(qh, qe) and torch.equal(kh, ke)
    assert torch.equal(q1.grad, q0.grad) and torch.equal(k1.grad, k0.grad)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(4608, 11008), (3, 7, 33), (1, 5), (2, 8192 + 3)])
def test_swiglu_kernel_vs_eager_chain(dev, shape, dtype):
    from dalm_amd.models import tower_ops

    g = torch.Generator().manual_seed(sum(shape))
    gate = (torch.randn(shape, generator=g) * 3).to(dtype).to(dev)
    up = torch.randn(shape, generator=g).to(dtype).to(dev)
    d_act = torch.randn(shape, generator=g).to(dtype).to(dev)
    g0, u0 = gate.clone().requires_grad_(True), up.clone().requires_grad_(True)
    g1, u1 = gate.clone().requires_grad_(True), up.clone().requires_grad_(True)
    ref = torch.nn.functional.silu(g0) * u0
    ref.backward(d_act)
    out = tower_ops.swiglu(g1, u1)
    out.backward(d_act)
    for name, a, b in (("act", out, ref), ("d_gate", g1.grad, g0.grad), ("d_up", u1.grad, u0.grad)):
        a, b = a.detach().float(), b.detach().float()
        if dtype == torch.float32:
            torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7, msg=lambda m: f"{name}: {m}")
            continue
        diff = (a - b).abs()
        ulp = b.abs().clamp_min(1e-30) * 2.0 ** -7                 # one bf16 ulp is at most 2^-7 of the value
        assert bool((diff <= ulp).all()), f"{name}: more than one bf16 ulp apart (max {float(diff.max())})"
        frac = float((diff > 0).float().mean())
        assert frac < 1e-3, f"{name}: {frac:.2e} of the elements differ from the eager chain"


def test_patched_llama_layer_matches_transformers(dev):
    """A 2-layer Llama (head_dim 128) with the rotary and SwiGLU kernels patched in against the unpatched module: logits and
    every parameter gradient, bf16 autocast and fp32."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from dalm_amd.models import fastpath

    cfg = LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, vocab_size=300, attention_dropout=0.0, pad_token_id=0)
    torch.manual_seed(0)
    ref = LlamaForCausalLM(cfg).to(dev)
    new = LlamaForCausalLM(cfg).to(dev)
    new.load_state_dict(ref.state_dict())
    assert fastpath.use_swiglu_kernel(new) == 2
    ids = torch.randint(1, 300, (3, 17), device=dev)
    mask = torch.ones_like(ids)
    mask[1, :5] = 0
    import transformers.models.llama.modeling_llama as m

    for autocast in (False, True):
        outs = []
        for model, patched in ((ref, False), (new, True)):
            if patched:
                assert fastpath.use_roll_rope(model)
            else:
                m.apply_rotary_pos_emb = getattr(m, "_dalm_orig_apply_rotary_pos_emb", m.apply_rotary_pos_emb)
            model.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                logits = model(input_ids=ids, attention_mask=mask).logits
            logits.float().square().mean().backward()
            outs.append((logits.detach().float(), {n: p.grad.detach().float() for n, p in model.named_parameters()}))
        (l0, g0), (l1, g1) = outs
        if autocast:
            # rotary is bit-identical; SwiGLU may move single bf16 roundings: a loose bound that a wrong formula cannot meet
            torch.testing.assert_close(l1, l0, rtol=2e-2, atol=2e-3)
        else:
            torch.testing.assert_close(l1, l0, rtol=1e-5, atol=1e-6)
        for n in g0:
            num = float((g1[n] - g0[n]).norm())
            den = float(g0[n].norm()) + 1e-12
            assert num / den < (2e-2 if autocast else 2e-5), f"{n}: relative gradient error {num / den:.2e}"
