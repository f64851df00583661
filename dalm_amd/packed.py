"""Packed (un-padded) token layout of the towers: run every row-wise kernel and every GEMM on the LIVE tokens only.

The reference pads every row to max_length (rag_e2e_dataloader_utils.py:47-52: Tq 50 / Tp 128 / Tg 256) and pushes the padding
through both towers; padding contributes exactly zero to its loss and to every gradient:

  * generator: `compute_marginalized_loss_from_logits` weights row (b, t) with attention_mask[b, t + 1]
    (dalm/training/utils/train_utils.py:134-136), a padded token is never attended as a key (HF's causal + padding mask) and its
    own output feeds nothing else;
  * retriever: `mean_pooling` multiplies the token states with the mask (dalm/models/rag_e2e_base_model.py:108-111), BERT's
    attention excludes padded keys.

So the same loss and the same gradients come out of the towers run on a [n_live, H] matrix of the tokens that matter - at
BASELINE.json's cfg3 batch 2.9 k instead of 4.6 k generator rows and ~20 % of the query tower's - provided that

  * every token keeps its ORIGINAL column as its position (rotary tables / BERT position embeddings: transformers numbers
    positions by column, padding included),
  * attention runs per sequence (`dalm_attn_*_packed`: cu_seqlens, no cross-sequence tiles), and
  * a token that is needed only as a QUERY stays in: with left padding, row t0 - 1 (the last padding position) predicts the
    first real token with weight attention_mask[b, t0] = 1.  It attends nothing (its keys are all padding: output 0, as the padded
    kernels and torch's memory-efficient kernels produce) and is marked key-dead.

Token set of a sequence: generator {t : mask[t] != 0 or mask[t + 1] != 0}, retriever {t : mask[t] != 0}; arbitrary masks (holes)
are fine - local order = column order, so causality is preserved.  The rows are listed on the HOST where the mask still is host
memory (data loader / batch staging; the count sets tensor shapes and reading it from the device would cost a sync per step):
`pack_plan` returns `rows` (flat index b T + t per packed row, -1 for the slack that rounds the count up to a multiple so that
only a few distinct shapes = hipGraphs occur) and `cu` (sequence starts; the slack forms one more sequence without live keys:
its rows are finite, carry no loss and therefore exactly zero gradients).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch

PACK_MULTIPLE = 128
# row multiples of the trainers' loaders and of bench.py --data-path packed: coarse on purpose - the packed row counts are part of
# a hipGraph's shape (one graph per combination seen) and of a GEMM's M (dalm_amd/tuning holds a tuned solution for every value
# these multiples can produce at the BASELINE lengths, tools/tune_packed.py)
ROW_MULTIPLES = {"generator": 256, "retriever_query": 256, "retriever_passage": 512, "query": 256, "passage": 512}


def pack_plan(attention_mask: torch.Tensor, shifted: bool, multiple: int = PACK_MULTIPLE) -> Tuple[torch.Tensor, torch.Tensor]:
    """HOST side.  attention_mask [B, T] (any integer / bool dtype) -> (rows int64 [n_pad], cu int32 [B + 1 + n_slack]).
    shifted=True (generator): a token also stays when the NEXT column is live (its row carries a label).
    The slack rows (n_pad - n < multiple) form n_slack = ceil(multiple / T) extra sequences of at most T rows each (one when
    multiple <= T): a sequence must fit the T rows of the attention layout, and the number of cu entries must not depend on the
    batch (it is a tensor shape)."""
    m = attention_mask.detach()
    if m.is_cuda:
        m = m.cpu()
    live = m != 0
    B, T = live.shape
    multiple = max(1, int(multiple))
    n_slack = max(1, -(-multiple // max(T, 1)))
    keep = live.clone()
    if shifted:
        keep[:, :-1] |= live[:, 1:]
    idx = keep.reshape(-1).nonzero().squeeze(1)
    n = int(idx.numel())
    n_pad = max(multiple, -(-n // multiple) * multiple)
    rows = torch.full((n_pad,), -1, dtype=torch.int64)
    rows[:n] = idx
    cu = torch.zeros(B + 1 + n_slack, dtype=torch.int32)
    cu[1:B + 1] = keep.sum(dim=1).cumsum(0).to(torch.int32)
    for j in range(n_slack):                 # slack sequences: T rows each until the slack is used up, the rest empty
        cu[B + 1 + j] = min(n_pad, n + (j + 1) * T)
    cu[B + n_slack] = n_pad
    return rows, cu


@dataclass
class PackedSeqs:
    """What the attention of one packed tower call reads (device side, built once per call, shared by every layer)."""

    cu: torch.Tensor                       # int32 [nseq + 1]
    nseq: int                              # B + the slack sequences (last)
    T: int                                 # rows of the mask-word / lse layout: the padded layout's T
    n: int                                 # packed rows
    key_live: torch.Tensor                 # uint8 [n]
    causal: bool
    rows_bits: Optional[torch.Tensor] = None
    cols_bits: Optional[torch.Tensor] = None
    live_tiles: Optional[torch.Tensor] = None
    stream: int = 0
    _pad: dict = field(default_factory=dict)

    def bits(self):
        """Mask words for `dalm_attn_*_packed` (GPU only; built on first use on the calling stream)."""
        from . import hip

        s = hip.stream()
        if self.rows_bits is None or self.stream != s:
            W = (self.T + 31) // 32
            dev = self.cu.device
            self.rows_bits = torch.empty(self.nseq * 32 * W * W, dtype=torch.int32, device=dev)
            self.cols_bits = torch.empty_like(self.rows_bits)
            self.live_tiles = torch.empty(self.nseq * W * W, dtype=torch.uint8, device=dev)
            hip.call("dalm_attn_mask_bits_packed", hip.ptr(self.key_live), hip.ptr(self.cu), self.nseq, self.T, int(self.causal),
                     hip.ptr(self.rows_bits), hip.ptr(self.cols_bits), hip.ptr(self.live_tiles), s)
            self.stream = s
        return self.rows_bits, self.cols_bits, self.live_tiles

    def padded_index(self):
        """(slot [n]: b T + local index of every packed row; gather [nseq T]: packed row of every slot, n for an empty one) -
        for the torch fallback of the attention (CPU tensors, fp32) only."""
        if not self._pad:
            dev = self.cu.device
            r = torch.arange(self.n, device=dev)
            b = torch.bucketize(r, self.cu[1:].to(torch.int64), right=True).clamp_max(self.nseq - 1)
            slot = b * self.T + (r - self.cu.to(torch.int64)[b])
            gather = torch.full((self.nseq * self.T,), self.n, dtype=torch.int64, device=dev)
            gather.scatter_(0, slot, r)
            self._pad = {"slot": slot, "gather": gather}
        return self._pad["slot"], self._pad["gather"]


def attach(desc: torch.Tensor, seqs: PackedSeqs) -> torch.Tensor:
    desc._dalm_packed = seqs
    return desc


def packed_of(mask) -> Optional[PackedSeqs]:
    return getattr(mask, "_dalm_packed", None) if mask is not None else None


def packed_inputs(input_ids: torch.Tensor, attention_mask: torch.Tensor, rows: torch.Tensor, cu: torch.Tensor, causal: bool):
    """Device side, no host sync: (ids [1, n], position_ids [1, n], mask descriptor, valid [n] bool).

    The descriptor is a 4-D tensor: transformers' mask builders return 4-D masks untouched (masking_utils
    `_preprocess_mask_arguments`), so it reaches every layer's attention call, where `packed_of` finds the sequences."""
    B, T = input_ids.shape
    valid = rows >= 0
    r = rows.clamp_min(0)
    ids_p = input_ids.reshape(-1).index_select(0, r)
    pos = torch.where(valid, r % T, torch.zeros_like(r))
    key_live = ((attention_mask.reshape(-1).index_select(0, r) != 0) & valid).to(torch.uint8)
    seqs = PackedSeqs(cu=cu.to(torch.int32), nseq=int(cu.numel()) - 1, T=int(T), n=int(rows.numel()), key_live=key_live, causal=causal)
    desc = attach(torch.ones((1, 1, 1, 1), dtype=torch.bool, device=input_ids.device), seqs)
    return ids_p.unsqueeze(0), pos.unsqueeze(0), desc, valid


def packed_labels(input_ids: torch.Tensor, attention_mask: torch.Tensor, rows: torch.Tensor):
    """Shifted labels of the packed generator rows: row (b, t) predicts ids[b, t + 1] with weight mask[b, t + 1]
    (train_utils.py:120-136); rows at t = T - 1 and slack rows weigh 0."""
    B, T = input_ids.shape
    valid = rows >= 0
    r = rows.clamp_min(0)
    has_next = valid & ((r % T) < T - 1)
    nxt = (r + 1).clamp_max(B * T - 1)
    y = input_ids.reshape(-1).index_select(0, nxt)
    m = attention_mask.reshape(-1).index_select(0, nxt) * has_next.to(attention_mask.dtype)
    return y, m


def attention_is_packable(model) -> bool:
    """The packed call hands transformers a 4-D DESCRIPTOR instead of a mask: only the "dalm_sdpa" attention implementation
    (models/attention.py) knows what it is - any other implementation would broadcast it as a mask."""
    from .models import attention

    cfg = getattr(model, "config", None)
    if cfg is None:
        return False
    if getattr(cfg, "_attn_implementation", None) == attention.NAME:
        return True
    # Falcon (7B flavour) stays on "sdpa": its attention modules are patched one by one (fastpath.use_falcon_attention_kernels),
    # and the patched forward is the one that reads the descriptor
    att = [m for m in model.modules() if type(m).__name__ == "FalconAttention"]
    return bool(att) and all(getattr(getattr(m.forward, "__func__", None), "__name__", "") == "_falcon_attention_forward" for m in att)


def generator_hidden(generator_model, input_ids, attention_mask, rows, cu):
    """Final (normed) hidden states [n, H] of the packed generator rows."""
    ids_p, pos, desc, _valid = packed_inputs(input_ids, attention_mask, rows, cu, causal=True)
    out = generator_model.base_model(input_ids=ids_p, attention_mask=desc, position_ids=pos, use_cache=False)[0]
    return out[0]


def retrieval_hidden(retriever_model, input_ids, attention_mask, rows, cu):
    """Token states of an encoder retriever on the packed rows, scattered back to the padded [B, T, D] layout (zeros at the
    padding, which the pooling kernel never reads: it multiplies with the mask)."""
    B, T = input_ids.shape
    ids_p, pos, desc, valid = packed_inputs(input_ids, attention_mask, rows, cu, causal=False)
    h = retriever_model(input_ids=ids_p, attention_mask=desc, position_ids=pos)[0][0]              # [n, D]
    dst = torch.where(valid, rows, torch.full_like(rows, B * T))                                     # slack -> a dump row
    full = h.new_zeros((B * T + 1, h.shape[-1])).index_copy(0, dst, h)
    return full[:B * T].view(B, T, -1)


def retrieval_hidden_pair(retriever_model, first, second):
    """BOTH retriever inputs of a step (queries and passages) through the encoder in ONE packed call: `first` / `second` are
    (input_ids, attention_mask, rows, cu).  The reference runs the encoder twice (rag_e2e_base_model.py:84-93 per call,
    train_rage2e.py:431-438); every op of an encoder layer is row-wise except attention, which runs per sequence here, so one
    call over the concatenated rows computes the same token states with half the kernel launches and larger GEMMs.
    Returns the two padded [B, T, D] tensors (zeros at the padding)."""
    parts = []
    off = 0
    for ids, mask, rows, cu in (first, second):
        ids_p, pos, desc, valid = packed_inputs(ids, mask, rows, cu, causal=False)
        sq = packed_of(desc)
        parts.append((ids_p, pos, sq, valid, rows, ids.shape, off))
        off += int(rows.numel())
    T = max(p[2].T for p in parts)
    cu_all = torch.cat([parts[0][2].cu] + [(p[2].cu[1:] + p[6]) for p in parts[1:]])
    seqs = PackedSeqs(cu=cu_all.to(torch.int32), nseq=int(cu_all.numel()) - 1, T=int(T), n=off,
                      key_live=torch.cat([p[2].key_live for p in parts]), causal=False)
    desc = attach(torch.ones((1, 1, 1, 1), dtype=torch.bool, device=cu_all.device), seqs)
    h = retriever_model(input_ids=torch.cat([p[0] for p in parts], dim=1), attention_mask=desc,
                        position_ids=torch.cat([p[1] for p in parts], dim=1))[0][0]                  # [n_first + n_second, D]
    outs = []
    for _ids_p, _pos, _sq, valid, rows, (B, Tn), o in parts:
        hp = h[o:o + rows.numel()]
        dst = torch.where(valid, rows, torch.full_like(rows, B * Tn))
        full = hp.new_zeros((B * Tn + 1, hp.shape[-1])).index_copy(0, dst, hp)
        outs.append(full[:B * Tn].view(B, Tn, -1))
    return outs[0], outs[1]


RAG_GROUPS = (("generator", "generator_input_input_ids", "generator_input_attention_mask", True),
              ("retriever_query", "retriever_query_input_ids", "retriever_query_attention_mask", False),
              ("retriever_passage", "retriever_passage_input_ids", "retriever_passage_attention_mask", False))
RETRIEVER_GROUPS = (("query", "query_input_ids", "query_attention_mask", False),
                    ("passage", "passage_input_ids", "passage_attention_mask", False))


def add_pack_plans(batch: dict, groups=RAG_GROUPS, multiple=None, device=None) -> dict:
    """HOST side (data loader / batch staging): add `<prefix>_pack_rows` / `<prefix>_pack_cu` for every tower input of a batch
    whose masks are host memory (a device mask is copied back: one sync - do this where batches are staged, not per step).
    The training steps take the packed path for every tower that finds its keys."""
    out = dict(batch)
    if multiple is None:
        multiple = ROW_MULTIPLES
    for prefix, _ids, mask_key, shifted in groups:
        if mask_key not in batch:
            continue
        rows, cu = pack_plan(batch[mask_key], shifted, multiple.get(prefix, PACK_MULTIPLE) if isinstance(multiple, dict) else int(multiple))
        dev = device if device is not None else batch[mask_key].device
        out[f"{prefix}_pack_rows"] = rows.to(dev)
        out[f"{prefix}_pack_cu"] = cu.to(dev)
    return out
