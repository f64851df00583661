"""`dalm_bert_add_norm_{fwd,bwd}` (dalm_amd/csrc/bert.hip, models/bert_ops.py): dropout + residual add + LayerNorm of a BERT
encoder layer in one launch per direction, against the eager chain transformers' BertSelfOutput / BertOutput run under bf16
autocast (reference: self.retriever_model(...), dalm/models/rag_e2e_base_model.py:84-93):
    d = dropout(a)  (bf16);  s = d + res  (f32);  y = layer_norm(s)  (f32, autocast's f32 list);  consumers cast y to bf16.
* no dropout: y32 against the eager f32 result (summation order only), y16 == bf16(y32), gradients against autograd's;
* dropout: every keep bit equals oracle/lora_mask.py::keep_mask_v2, values / gradients equal the eager chain evaluated with THAT
  mask, the keep rate is 1 - p;
* a BERT layer with the patched modules against transformers' own layer (eval and train-without-dropout), incl. the bf16 twin that
  the consumer GEMMs take instead of a second cast."""
import copy
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _eager(a, res, w, b, eps, keep=None, p=0.0):
    """The autocast chain with an explicit keep mask (torch's bf16 dropout: x * mask * 1/(1-p) in f32, rounded to bf16)."""
    a = a.detach().clone().requires_grad_(True)
    res = res.detach().clone().requires_grad_(True)
    d = a if keep is None else (a.float() * keep.float() * (1.0 / (1.0 - p))).to(torch.bfloat16)
    s = d + res                                           # bf16 + f32 -> f32
    y = F.layer_norm(s, (s.shape[-1],), w.float(), b.float(), eps)
    return a, res, y


CASES = [(4608, 1024, torch.float32), (1672, 1024, torch.bfloat16), (19 * 50, 384, torch.float32), (7, 1024, torch.bfloat16),
         (33, 512, torch.float32), (5, 2048, torch.bfloat16), (64, 8, torch.float32)]


@pytest.mark.parametrize("R,D,wdt", CASES)
def test_add_norm_without_dropout_vs_eager_chain(dev, R, D, wdt):
    from dalm_amd.models import bert_ops

    g = torch.Generator().manual_seed(R + D)
    a = (torch.randn(R, D, generator=g) * 1.5).to(dev, torch.bfloat16)
    res = (torch.randn(R, D, generator=g) * 2.0).to(dev)
    ln = torch.nn.LayerNorm(D, eps=1e-12).to(dev)
    with torch.no_grad():
        ln.weight.copy_(1.0 + 0.2 * torch.randn(D, generator=g))
        ln.bias.copy_(0.1 * torch.randn(D, generator=g))
    ln = ln.to(wdt).requires_grad_(False)
    up32 = torch.randn(R, D, generator=g).to(dev)
    up16 = torch.randn(R, D, generator=g).to(dev, torch.bfloat16)
    a1, r1 = a.clone().requires_grad_(True), res.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert bert_ops.supported(a1, r1, ln)
        y32 = bert_ops.add_norm(a1, r1, ln, 0.0, 0)
    y16 = y32._dalm_bf16
    assert y32.dtype == torch.float32 and y16.dtype == torch.bfloat16
    assert torch.equal(y16, y32.detach().to(torch.bfloat16))
    torch.autograd.backward([y32, y16], [up32, up16])
    a0, r0, y0 = _eager(a, res, ln.weight, ln.bias, 1e-12)
    torch.autograd.backward([y0, y0.to(torch.bfloat16)], [up32, up16])     # the consumers' cast: its backward up-casts the bf16 gradient
    torch.testing.assert_close(y32.detach(), y0.detach(), rtol=3e-6, atol=3e-6)
    torch.testing.assert_close(r1.grad, r0.grad, rtol=2e-5, atol=2e-6)
    # d_a = bf16(d_res): equal up to a bf16 rounding flip where d_res differs in its last f32 bits (one bf16 ulp of the value; an
    # element that is the small difference of large terms carries the f32 summation-order error of those terms instead)
    diff = (a1.grad.float() - a0.grad.float()).abs()
    ref = a0.grad.float().abs()
    assert bool((diff <= ref * 2.0 ** -7 + 1e-5 * float(ref.max())).all())
    assert float((diff > 0).float().mean()) < 2e-3
    # only one of the two outputs used downstream
    a2, r2 = a.clone().requires_grad_(True), res.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        z = bert_ops.add_norm(a2, r2, ln, 0.0, 0)
    z._dalm_bf16.backward(up16)
    a3, r3, y3 = _eager(a, res, ln.weight, ln.bias, 1e-12)
    y3.to(torch.bfloat16).backward(up16)
    torch.testing.assert_close(r2.grad, r3.grad, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("R,D", [(1024, 1024), (300, 384)])
def test_add_norm_dropout_mask_is_the_oracles_and_values_follow_it(dev, R, D):
    import lora_mask as LM

    from dalm_amd.models import bert_ops, lora_ops

    p, salt = 0.1, 0x5A17
    g = torch.Generator().manual_seed(3)
    a = torch.randn(R, D, generator=g).to(dev, torch.bfloat16)
    res = torch.randn(R, D, generator=g).to(dev)
    ln = torch.nn.LayerNorm(D, eps=1e-12).to(dev).requires_grad_(False)
    lora_ops.advance_dropout_seed(dev)
    seed = int(lora_ops.dropout_seed(dev).item())
    a1, r1 = a.clone().requires_grad_(True), res.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y32, y16 = bert_ops._BertAddNorm.apply(a1, r1, ln.weight, ln.bias, 1e-12, p, salt)
    keep = torch.from_numpy(LM.keep_mask_v2(seed, salt, R, D, p)).to(dev)
    assert abs(float(keep.float().mean()) - (1 - p)) < 4 * np.sqrt(p * (1 - p) / (R * D))
    bits = y32.grad_fn.saved_tensors[5] if hasattr(y32.grad_fn, "saved_tensors") else None
    if bits is not None:
        want_bits = torch.from_numpy(LM.pack_bits(LM.keep_mask_v2(seed, salt, R, D, p))).to(dev)
        assert torch.equal(bits, want_bits)
    a0, r0, y0 = _eager(a, res, ln.weight, ln.bias, 1e-12, keep, p)
    up = torch.randn(R, D, generator=g).to(dev)
    y32.backward(up)
    y0.backward(up)
    torch.testing.assert_close(y32.detach(), y0.detach(), rtol=3e-6, atol=3e-6)
    torch.testing.assert_close(r1.grad, r0.grad, rtol=2e-5, atol=2e-6)
    assert bool((a1.grad[~keep] == 0).all())
    assert _rel(a1.grad, a0.grad) < 3e-3
    # another step: another mask
    lora_ops.advance_dropout_seed(dev)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y_b, _ = bert_ops._BertAddNorm.apply(a, res, ln.weight, ln.bias, 1e-12, p, salt)
    assert not torch.equal(y_b, y32)


def test_patched_bert_layers_match_transformers(dev):
    """A frozen BERT encoder (LoRA-style: nothing trainable but an input perturbation) with the patched output modules against
    the same encoder on transformers' code, under bf16 autocast, dropout off: hidden states and input gradients; the consumer
    GEMMs take the bf16 twin (no cast kernel) and produce the same values."""
    from transformers import BertConfig, BertModel

    from dalm_amd.models import bert_ops, fastpath, frozen_linear

    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=512, vocab_size=200,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ref = BertModel(cfg).to(dev).requires_grad_(False).train()
    new = copy.deepcopy(ref)
    frozen_linear.use_transposed_dgrad(new)
    assert fastpath.use_bert_layer_kernels(new) == 6
    ids = torch.randint(0, 200, (3, 40), device=dev)
    mask = torch.ones(3, 40, dtype=torch.long, device=dev)
    mask[1, 25:] = 0
    outs = []
    up = torch.randn(3, 40, 256, device=dev)        # (sum h^2 of a LayerNorm output is a constant: a random upstream gradient)
    for m in (ref, new):
        emb = m.embeddings.word_embeddings.weight
        emb.requires_grad_(True)
        emb.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = m(ids, mask)[0]
        assert h.dtype == torch.float32
        (h.float() * mask.unsqueeze(-1) * up).sum().backward()
        outs.append((h.detach(), emb.grad.clone()))
        emb.requires_grad_(False)
    live = mask.bool()
    assert _rel(outs[1][0][live], outs[0][0][live]) < 2e-3                # bf16 GEMM inputs on both sides; same rounding points
    assert _rel(outs[1][1], outs[0][1]) < 2e-2
    # the twin is what the next GEMM reads
    lin = new.encoder.layer[0].intermediate.dense
    x = outs[1][0].clone().requires_grad_(True)
    x._dalm_bf16 = (x.detach() * 0).to(torch.bfloat16).requires_grad_(True)   # a twin that differs from x: the GEMM must follow it
    x._dalm_bf16_version, x._dalm_bf16_tversion = x._version, x._dalm_bf16._version
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert bert_ops.twin(x) is x._dalm_bf16
        y = lin(x)
    assert float((y.float() - lin.bias.float()).abs().max()) < 1e-2       # W . 0 + b
    assert bert_ops.twin(x) is x                                           # outside autocast: no swap
    with torch.no_grad():
        x._dalm_bf16.add_(1.0)                                             # an in-place write: the twin is stale now
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert bert_ops.twin(x) is x
