"""Packed (un-padded) tower path, host logic and the torch fallback of its attention (no GPU): the token list, the labels, and
tiny Llama / BERT towers run on the packed rows against the same towers on the padded batch - the live tokens' states must be
the padded run's (reference semantics: dalm/training/utils/train_utils.py:134-136, dalm/models/rag_e2e_base_model.py:108-111)."""
import pytest
import torch

from dalm_amd import packed


def _masks(B, T, left, seed, holes=False):
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T                      # a row without padding
    if B > 2:
        lens[2] = 1                  # a single live token
    ar = torch.arange(T).unsqueeze(0)
    m = (ar >= (T - lens).unsqueeze(1)) if left else (ar < lens.unsqueeze(1))
    m = m.long()
    if holes:
        m[1, T // 2] = 0
    return m


@pytest.mark.parametrize("left", [True, False])
def test_pack_plan_lists_exactly_the_rows_that_matter(left):
    B, T = 5, 12
    m = _masks(B, T, left, 0)
    m[3] = 0                                                         # an all-padding row
    rows, cu = packed.pack_plan(m, shifted=True, multiple=8)
    n = int(cu[B])
    want = [(b, t) for b in range(B) for t in range(T) if m[b, t] or (t + 1 < T and m[b, t + 1])]
    assert [(int(r) // T, int(r) % T) for r in rows[:n]] == want
    assert rows.numel() % 8 == 0 and int(cu[B + 1]) == rows.numel() and (rows[n:] == -1).all()
    assert [int(c) for c in cu[1:B + 1] - cu[:B]] == [sum(1 for (b, _) in want if b == i) for i in range(B)]
    rows2, cu2 = packed.pack_plan(m, shifted=False, multiple=8)
    assert [int(r) for r in rows2[:int(cu2[B])]] == [int(i) for i in m.reshape(-1).nonzero().squeeze(1)]
    # labels: row (b, t) predicts ids[b, t + 1] with weight mask[b, t + 1]
    ids = torch.arange(B * T).view(B, T)
    y, w = packed.packed_labels(ids, m, rows)
    for i, (b, t) in enumerate(want):
        if t + 1 < T:
            assert int(y[i]) == int(ids[b, t + 1]) and int(w[i]) == int(m[b, t + 1])
        else:
            assert int(w[i]) == 0
    assert (w[n:] == 0).all()
    assert int(w.sum()) == int(m[:, 1:].sum())                      # every target token of the padded loss, once
    # a row multiple above the sequence length (query tower: T 50, multiple 256): the slack is cut into sequences of <= T rows
    mq = _masks(6, 10, False, 3)
    rq, cq = packed.pack_plan(mq, shifted=False, multiple=32)
    assert rq.numel() % 32 == 0 and cq.numel() == 6 + 1 + 4 and int(cq[-1]) == rq.numel()
    assert all(0 <= int(b - a) <= 10 for a, b in zip(cq[:-1], cq[1:]))
    r0, c0 = packed.pack_plan(torch.zeros(2, 10, dtype=torch.long), shifted=False, multiple=32)          # nothing live at all
    assert r0.numel() == 32 and (r0 == -1).all() and all(0 <= int(b - a) <= 10 for a, b in zip(c0[:-1], c0[1:]))


def _tiny_llama():
    from transformers import LlamaConfig, LlamaForCausalLM

    from dalm_amd.models import attention

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=97, max_position_embeddings=64)
    model = LlamaForCausalLM(cfg).eval()
    assert attention.register()
    model.config._attn_implementation = attention.NAME
    return model


@pytest.mark.parametrize("left,holes", [(True, False), (False, False), (True, True)])
def test_packed_llama_equals_padded_on_live_rows(left, holes):
    model = _tiny_llama()
    B, T = 4, 16
    mask = _masks(B, T, left, 1, holes)
    ids = torch.randint(3, 97, (B, T), generator=torch.Generator().manual_seed(2))
    rows, cu = packed.pack_plan(mask, shifted=True, multiple=8)
    assert packed.attention_is_packable(model)
    hp = packed.generator_hidden(model, ids, mask, rows, cu)                       # [n, H]
    model.config._attn_implementation = "sdpa"
    ref = model.base_model(input_ids=ids, attention_mask=mask, use_cache=False)[0].reshape(B * T, -1)
    y, w = packed.packed_labels(ids, mask, rows)
    carries_loss = w != 0
    got, want = hp[carries_loss], ref.index_select(0, rows.clamp_min(0))[carries_loss]
    assert carries_loss.sum() == mask[:, 1:].sum()
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-4), float((got - want).abs().max())
    assert torch.isfinite(hp).all()                                                # slack rows: finite, no NaN into weight grads


def test_packed_llama_gradients_equal_padded():
    from dalm_amd.models import attention

    model = _tiny_llama().train()
    B, T = 3, 12
    mask = _masks(B, T, True, 3)
    ids = torch.randint(3, 97, (B, T), generator=torch.Generator().manual_seed(4))
    rows, cu = packed.pack_plan(mask, shifted=True, multiple=8)
    y, w = packed.packed_labels(ids, mask, rows)
    head = model.get_output_embeddings().weight

    def loss_packed():
        h = packed.generator_hidden(model, ids, mask, rows, cu)
        lp = torch.log_softmax(h @ head.t(), dim=-1)
        return -(lp.gather(1, y.unsqueeze(1)).squeeze(1) * w).sum() / w.sum()

    def loss_padded():
        model.config._attn_implementation = "sdpa"
        lp = torch.log_softmax(model(input_ids=ids, attention_mask=mask, use_cache=False).logits[:, :-1], dim=-1)
        model.config._attn_implementation = attention.NAME
        m = mask[:, 1:].float()
        return -(lp.gather(2, ids[:, 1:].unsqueeze(2)).squeeze(2) * m).sum() / m.sum()

    grads = []
    for fn in (loss_packed, loss_padded):
        model.zero_grad()
        loss = fn()
        loss.backward()
        grads.append((float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert abs(grads[0][0] - grads[1][0]) <= 1e-5 * abs(grads[1][0])
    for n, g in grads[1][1].items():
        assert torch.allclose(grads[0][1][n], g, atol=1e-5, rtol=1e-3), n


def test_packed_bert_equals_padded_on_live_tokens():
    from transformers import BertConfig, BertModel

    from dalm_amd.models import attention

    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=101,
                     max_position_embeddings=64)
    model = BertModel(cfg).eval()
    assert attention.register()
    B, T = 4, 14
    mask = _masks(B, T, False, 5)
    ids = torch.randint(3, 101, (B, T), generator=torch.Generator().manual_seed(6))
    want = model(ids, mask)[0]
    model.config._attn_implementation = attention.NAME
    rows, cu = packed.pack_plan(mask, shifted=False, multiple=8)
    got = packed.retrieval_hidden(model, ids, mask, rows, cu)
    live = mask.bool()
    assert torch.allclose(got[live], want[live], atol=2e-5, rtol=1e-4)
    assert (got[~live] == 0).all()


def test_add_pack_plans_keys():
    b = {"generator_input_input_ids": torch.zeros(2, 8, dtype=torch.long), "generator_input_attention_mask": _masks(2, 8, True, 0),
         "retriever_query_input_ids": torch.zeros(2, 6, dtype=torch.long), "retriever_query_attention_mask": _masks(2, 6, False, 1)}
    out = packed.add_pack_plans(b, multiple=4)
    assert {"generator_pack_rows", "generator_pack_cu", "retriever_query_pack_rows", "retriever_query_pack_cu"} <= set(out)
    assert "retriever_passage_pack_rows" not in out
    assert out["generator_pack_cu"].dtype == torch.int32 and out["generator_pack_rows"].dtype == torch.int64


def test_sharded_batches_emit_pack_plans():
    from dalm_amd.training.common import ShardedBatches

    N, T = 10, 12
    data = {"generator_input_input_ids": torch.randint(0, 50, (N, T)), "generator_input_attention_mask": _masks(N, T, True, 7),
            "retriever_query_input_ids": torch.randint(0, 50, (N, 6)), "retriever_query_attention_mask": _masks(N, 6, False, 8)}
    sb = ShardedBatches(data, 4, 0, 1, 0, list(data), pack=dict(groups=packed.RAG_GROUPS, multiple={"generator": 8, "retriever_query": 4}))
    seen = 0
    for b in sb.epoch(0, torch.device("cpu")):
        seen += 1
        rows, cu = packed.pack_plan(b["generator_input_attention_mask"], True, 8)
        assert torch.equal(b["generator_pack_rows"], rows) and torch.equal(b["generator_pack_cu"], cu)
        assert b["retriever_query_pack_rows"].numel() % 4 == 0
        assert b["generator_pack_cu"].numel() == b["generator_input_input_ids"].shape[0] + 2
    assert seen == 3


def test_mixed_precision_default_follows_accelerate(monkeypatch):
    """Accelerator() without arguments (reference train_rage2e.py:276): ACCELERATE_MIXED_PRECISION, else "no"."""
    import inspect

    from dalm_amd.training import common
    from dalm_amd.training.rag_e2e.train_rage2e import parse_args, train_e2e
    from dalm_amd.training.retriever_only.train_retriever_only import train_retriever

    monkeypatch.delenv("ACCELERATE_MIXED_PRECISION", raising=False)
    assert common.resolve_mixed_precision(None) == "no"
    monkeypatch.setenv("ACCELERATE_MIXED_PRECISION", "bf16")
    assert common.resolve_mixed_precision(None) == "bf16"
    assert common.resolve_mixed_precision("no") == "no"
    monkeypatch.setenv("ACCELERATE_MIXED_PRECISION", "fp8")
    with pytest.raises(ValueError):
        common.resolve_mixed_precision(None)
    assert inspect.signature(train_e2e).parameters["mixed_precision"].default is None
    assert inspect.signature(train_retriever).parameters["mixed_precision"].default is None
    a = parse_args(["--dataset_path", "x.csv", "--retriever_name_or_path", "r", "--generator_name_or_path", "g"])
    assert a.mixed_precision is None and a.pack_tokens is False


def test_packed_falcon_equals_padded_on_live_rows():
    """Falcon-7B flavour (multi-query, parallel attention, rotary): stays on "sdpa", its patched attention forward reads the
    descriptor (fastpath._falcon_attention_forward)."""
    from transformers import FalconConfig, FalconForCausalLM

    from dalm_amd.models import fastpath

    torch.manual_seed(0)
    cfg = FalconConfig(num_hidden_layers=2, hidden_size=256, num_attention_heads=4, vocab_size=100)
    ref = FalconForCausalLM(cfg).eval()
    new = FalconForCausalLM(cfg).eval()
    new.load_state_dict(ref.state_dict())
    assert not packed.attention_is_packable(new)
    fastpath.use_capturable_falcon_heads(new)
    assert fastpath.use_falcon_attention_kernels(new) == 2 and packed.attention_is_packable(new)
    B, T = 4, 16
    mask = _masks(B, T, True, 9)
    ids = torch.randint(3, 100, (B, T), generator=torch.Generator().manual_seed(10))
    rows, cu = packed.pack_plan(mask, shifted=True, multiple=8)
    hp = packed.generator_hidden(new, ids, mask, rows, cu)
    want = ref.base_model(input_ids=ids, attention_mask=mask, use_cache=False)[0].reshape(B * T, -1)
    _y, w = packed.packed_labels(ids, mask, rows)
    sel = w != 0
    assert torch.allclose(hp[sel], want.index_select(0, rows.clamp_min(0))[sel], atol=2e-5, rtol=1e-4)
    assert torch.isfinite(hp).all()


def test_query_and_passage_in_one_encoder_call_equal_two_calls():
    from transformers import BertConfig, BertModel

    from dalm_amd.models import attention

    torch.manual_seed(0)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=101,
                     max_position_embeddings=64)
    model = BertModel(cfg).eval()
    assert attention.register()
    model.config._attn_implementation = attention.NAME
    B = 5
    mq, mp = _masks(B, 10, False, 11), _masks(B, 24, False, 12)
    iq = torch.randint(3, 101, (B, 10), generator=torch.Generator().manual_seed(13))
    ip = torch.randint(3, 101, (B, 24), generator=torch.Generator().manual_seed(14))
    rq, cq = packed.pack_plan(mq, False, 16)           # a multiple above the query length: several slack sequences
    rp, cp = packed.pack_plan(mp, False, 16)
    hq1 = packed.retrieval_hidden(model, iq, mq, rq, cq)
    hp1 = packed.retrieval_hidden(model, ip, mp, rp, cp)
    hp2, hq2 = packed.retrieval_hidden_pair(model, (ip, mp, rp, cp), (iq, mq, rq, cq))
    assert hq2.shape == hq1.shape and hp2.shape == hp1.shape
    assert torch.allclose(hq2, hq1, atol=2e-5, rtol=1e-4) and torch.allclose(hp2, hp1, atol=2e-5, rtol=1e-4)


def test_graphed_towers_key_distinguishes_packed_row_counts():
    from dalm_amd.training.graphed import GraphedTowers

    def batch(n_gen, with_pack=True):
        b = {"retriever_passage_input_ids": torch.zeros(2, 8), "retriever_query_input_ids": torch.zeros(2, 4),
             "generator_input_input_ids": torch.zeros(2, 16)}
        if with_pack:
            for k, n in (("retriever_passage", 8), ("retriever_query", 4), ("generator", n_gen)):
                b[f"{k}_pack_rows"] = torch.zeros(n, dtype=torch.long)
                b[f"{k}_pack_cu"] = torch.zeros(4, dtype=torch.int32)
        return b

    assert GraphedTowers.is_packed(batch(16)) and not GraphedTowers.is_packed(batch(16, False))
    assert GraphedTowers.key_of(batch(16)) != GraphedTowers.key_of(batch(24))          # another row count: another set of graphs
    assert GraphedTowers.key_of(batch(16)) != GraphedTowers.key_of(batch(16, False))   # padded graphs are their own set
    half = batch(16)
    del half["retriever_query_pack_rows"]
    assert not GraphedTowers.is_packed(half)
