"""Real-width, depth-1 towers + synthetic token batches for the step-parity checks (TEST INFRASTRUCTURE ONLY).

BASELINE.json's configs name bge-large (D = 1024), Llama-2-7b (4096 wide, V = 32000) and Falcon-7B (4544 wide,
V = 65024).  Their weights cannot be committed (and there is no network), so both sides of a parity check build
the SAME random-init modules from a CPU seed: one transformer layer at the true width, which keeps every
hot-path tensor shape of the step ([B,T,1024] token states, [B,256,V] logits, 4096/4544-deep lm_head) exactly
as in the named configuration ("depth-scaled model at full width").  A checksum of the
weights is recorded next to every golden number so a box whose CPU RNG differs is detected instead of compared.

Used by oracle/make_golden.py (which runs the REFERENCE's own classes on these modules) and by
tests/test_step_realwidth_gpu.py.  Nothing under dalm_amd/ imports this file.
"""
from __future__ import annotations

from typing import Dict

import torch

CASES = {
    # name: generator architecture, per-device batch, vocabulary, bench.py workload it mirrors
    "cfg3": {"generator": "llama", "B": 18, "V": 32000, "Tq": 50, "Tp": 128, "Tg": 256},
    "cfg5": {"generator": "falcon", "B": 18, "V": 65024, "Tq": 50, "Tp": 128, "Tg": 256},
    "cfg2": {"generator": None, "B": 150, "V": None, "Tq": 50, "Tp": 128, "Tg": None},
    # round 4 ("real-width parity is depth 1 ... nothing compares a depth > 1 step, where tower-side bf16 drift
    # would actually show"): cfg3 with TWO layers in both towers
    "cfg3_d2": {"generator": "llama", "B": 18, "V": 32000, "Tq": 50, "Tp": 128, "Tg": 256, "depth": 2},
}
SEED = 0


def build_retriever(depth: int = 1):
    """bge-large-en architecture (BERT 1024 / 16 heads / 4096, vocab 30522), dropout 0."""
    from transformers import BertConfig, BertModel

    return BertModel(BertConfig(hidden_size=1024, num_hidden_layers=depth, num_attention_heads=16, intermediate_size=4096,
                                vocab_size=30522, max_position_embeddings=512, hidden_dropout_prob=0.0,
                                attention_probs_dropout_prob=0.0))


def build_generator(kind: str, depth: int = 1):
    """Llama-2-7b / Falcon-7B architectures (HF config defaults == the published 7B shapes), dropout 0."""
    from transformers import FalconConfig, FalconForCausalLM, LlamaConfig, LlamaForCausalLM

    if kind == "llama":
        return LlamaForCausalLM(LlamaConfig(num_hidden_layers=depth, attention_dropout=0.0, pad_token_id=0))
    if kind == "falcon":
        return FalconForCausalLM(FalconConfig(num_hidden_layers=depth, hidden_dropout=0.0, attention_dropout=0.0))
    raise ValueError(kind)


def build_case(name: str, depth: int = None):
    """(retriever, generator-or-None) in fp32 on the CPU, from SEED: identical on every machine whose torch CPU RNG
    agrees (checked through `checksum`).  depth: layers per tower (default: the case's own, 1 unless it says otherwise)."""
    c = CASES[name]
    if depth is None:
        depth = c.get("depth", 1)
    torch.manual_seed(SEED)
    retriever = build_retriever(depth)
    generator = build_generator(c["generator"], depth) if c["generator"] else None
    return retriever, generator


def checksum(module) -> float:
    return float(sum(p.detach().double().abs().sum() for p in module.parameters()))


def synthetic_batch(name: str, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Token-id level synthetic (Passage, Query, Answer) rows, SURVEY.md section 8(d): right-padded retriever inputs with
    5-15 / 30-128 live tokens, left-padded generator rows with 60-256 live tokens, qlen = 0.8 * length."""
    c = CASES[name]
    B, Tq, Tp = c["B"], c["Tq"], c["Tp"]
    g = torch.Generator().manual_seed(1000 + seed)

    def right_mask(T, lo, hi):
        lens = torch.randint(lo, hi + 1, (B, 1), generator=g)
        return (torch.arange(T).unsqueeze(0) < lens).long()

    if c["generator"] is None:
        return {"query_input_ids": torch.randint(1000, 30522, (B, Tq), generator=g),
                "query_attention_mask": right_mask(Tq, 5, 15),
                "passage_input_ids": torch.randint(1000, 30522, (B, Tp), generator=g),
                "passage_attention_mask": right_mask(Tp, 30, Tp)}
    Tg, V = c["Tg"], c["V"]
    glen = torch.randint(60, Tg + 1, (B, 1), generator=g)
    return {
        "retriever_query_input_ids": torch.randint(1000, 30522, (B, Tq), generator=g),
        "retriever_query_attention_mask": right_mask(Tq, 5, 15),
        "retriever_passage_input_ids": torch.randint(1000, 30522, (B, Tp), generator=g),
        "retriever_passage_attention_mask": right_mask(Tp, 30, Tp),
        "generator_input_input_ids": torch.randint(1000, V, (B, Tg), generator=g),
        "generator_input_attention_mask": (torch.arange(Tg).unsqueeze(0) >= (Tg - glen)).long(),
        "query_passage_input_len": (glen.squeeze(1).float() * 0.8).long().clamp(min=1),
    }


def grad_norm(params) -> float:
    """Global L2 norm (fp64 accumulation) of the gradients of `params`."""
    tot = torch.zeros((), dtype=torch.float64)
    for p in params:
        if p.grad is not None:
            tot += (p.grad.detach().double() ** 2).sum().cpu()
    return float(tot.sqrt())
