"""Autograd layer over the HIP loss kernels, single- and multi-GPU.

Fused entry points (never materialise S or the [B,Tg,V] log-probs):
    contrastive_loss(q_emb, p_emb, logit_scale)            retriever-only step
    rag_e2e_loss(q_emb, p_emb, logits, ids, mask, qlen, s)  RAG-end2end step
They reproduce, per step, dalm/training/rag_e2e/train_rage2e.py:441-467 and
dalm/training/retriever_only/train_retriever_only.py:369-374.

Multi-GPU: rank r owns rows [r*B_l, (r+1)*B_l) of Q, P and the logits.  Q and P
are all-gathered (RCCL over xGMI; tiny: B_l*D*4 bytes per rank), every rank runs
two row-problems on the matrix cores,
    rows    : S_r  = s * Q_r . P_all^T   -> lse_r[i], S_ii     (i in rank r)
    columns : S_r' = s * P_r . Q_all^T   -> lse_c[j]            (j in rank r)
and the closed-form backward needs only the all-gathered
(lse_r, lse_c, a) vectors - no autograd through a collective, no reduce-scatter.
The per-rank value is the rank's share L_r of the GLOBAL-batch loss
(sum_r L_r = loss of one process at batch B_g); parameter gradients must be
SUMMED over ranks (dalm_amd.sharded.allreduce_grads does that).
The reference itself never gathers negatives (DDP, local B_l x B_l only).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any, Optional

import torch

from .ops import default_ops


# ---------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------
class LocalComm:
    """world_size == 1: every collective is the identity."""

    world_size = 1
    rank = 0

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        return t


class TorchDistComm:
    """torch.distributed process group (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests)."""

    def __init__(self, group: Any = None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        t = t.contiguous()
        out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        self._dist.all_gather_into_tensor(out, t, group=self.group)
        return out

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t


class GatherHandle:
    """An all-gather of embeddings started early on a side stream (overlap with the other tower)."""

    def __init__(self, local: torch.Tensor, comm, side_stream: Optional["torch.cuda.Stream"] = None):
        self.local = local
        self.comm = comm
        self.result: Optional[torch.Tensor] = None
        self.event = None
        if isinstance(comm, LocalComm):
            self.result = local.detach()
            return
        if side_stream is not None and local.is_cuda:
            side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side_stream):
                self.result = comm.all_gather_rows(local.detach())
                self.event = torch.cuda.Event()
                self.event.record(side_stream)
            self.result.record_stream(torch.cuda.current_stream()) if hasattr(self.result, "record_stream") else None
        else:
            self.result = comm.all_gather_rows(local.detach())

    def wait(self) -> torch.Tensor:
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self.event = None
        assert self.result is not None
        return self.result


# ---------------------------------------------------------------------------
# K1
# ---------------------------------------------------------------------------
class _PoolL2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, mask, normalize, ops):
        emb, norm, inv_count = ops.pool_fwd(h, mask, normalize)
        ctx.save_for_backward(emb, norm, inv_count, mask)
        ctx.normalize, ctx.ops, ctx.T, ctx.h_dtype = normalize, ops, h.shape[1], h.dtype
        return emb

    @staticmethod
    def backward(ctx, d_emb):
        emb, norm, inv_count, mask = ctx.saved_tensors
        dh = ctx.ops.pool_bwd(d_emb, emb, norm, inv_count, mask, ctx.normalize, ctx.T, ctx.h_dtype)
        return dh, None, None, None


def pool_l2norm(token_embeddings: torch.Tensor, attention_mask: torch.Tensor, normalize: bool = True, ops=None):
    """mean_pooling (+ F.normalize(p=2, dim=1)) of rag_e2e_base_model.py:95-97,108-111 in one HIP pass."""
    return _PoolL2Norm.apply(token_embeddings, attention_mask, bool(normalize), ops or default_ops())


# ---------------------------------------------------------------------------
# shared forward/backward pieces
# ---------------------------------------------------------------------------
@dataclass
class _ConState:
    q: torch.Tensor
    p: torch.Tensor
    q_all: torch.Tensor
    p_all: torch.Tensor
    lse_r: torch.Tensor   # local rows
    lse_c: torch.Tensor   # local columns
    diag: torch.Tensor
    offset: int
    n_global: int
    # small-batch form: the saved S of the rows problem (one GPU: also serves the columns) / the columns problem
    S_rows: Optional[torch.Tensor] = None
    S_cols: Optional[torch.Tensor] = None


def _contrastive_forward(ops, comm, q, p, scale, q_all=None, p_all=None):
    q = q.detach().float().contiguous()
    p = p.detach().float().contiguous()
    if q.shape != p.shape or q.dim() != 2:
        raise ValueError(f"query/passage embeddings must both be [B,D], got {tuple(q.shape)} / {tuple(p.shape)}")
    b_l, D = q.shape
    offset = comm.rank * b_l
    if isinstance(comm, LocalComm) and ops.sim_small_supported(b_l, b_l, D):
        # one GPU at a real batch size: S is computed once, row AND column statistics come out of the same tiles
        S, lse_r, diag, lse_c = ops.sim_small_fwd(q, p, scale, 0, True)
        return _ConState(q, p, q, p, lse_r, lse_c, diag, 0, b_l, S_rows=S)
    if p_all is None:
        p_all = comm.all_gather_rows(p)
    if q_all is None:
        q_all = comm.all_gather_rows(q)
    n_g = comm.world_size * b_l
    S_rows = S_cols = None
    if ops.sim_small_supported(b_l, n_g, D):
        S_rows, lse_r, diag, _ = ops.sim_small_fwd(q, p_all, scale, offset, False)
        S_cols, lse_c, _, _ = ops.sim_small_fwd(p, q_all, scale, offset, False)
    else:
        lse_r, diag = ops.sim_rowstats(q, p_all, scale, offset)
        lse_c, _ = ops.sim_rowstats(p, q_all, scale, offset)
    return _ConState(q, p, q_all, p_all, lse_r, lse_c, diag, offset, n_g, S_rows=S_rows, S_cols=S_cols)


def _contrastive_backward(ops, comm, st: _ConState, scale, a_local, b_local):
    """a: row coefficients (this rank's queries), b: column coefficients (this rank's passages)."""
    if isinstance(comm, LocalComm):
        if st.S_rows is not None:  # one launch: dQ and dP from the saved S
            return ops.sim_small_bwd(st.S_rows, st.q, st.p, scale, 0, a_local, st.lse_r, b_local, st.lse_c, True, True)
        lse_r_all, lse_c_all, a_all, b_all = st.lse_r, st.lse_c, a_local, b_local
    else:
        packed = torch.stack([st.lse_r, st.lse_c, a_local, b_local], dim=1)  # [B_l,4] -> one all-gather
        allv = comm.all_gather_rows(packed)
        lse_r_all, lse_c_all, a_all, b_all = (allv[:, k].contiguous() for k in range(4))
    if st.S_rows is not None:
        dq, _ = ops.sim_small_bwd(st.S_rows, st.q, st.p_all, scale, st.offset, a_local, st.lse_r, b_all, lse_c_all, True, False)
    else:
        dq = ops.sim_grad(st.q, st.p_all, scale, st.offset, a_local, st.lse_r, b_all, lse_c_all)
    if st.S_cols is not None:
        dp, _ = ops.sim_small_bwd(st.S_cols, st.p, st.q_all, scale, st.offset, b_local, st.lse_c, a_all, lse_r_all, True, False)
    else:
        dp = ops.sim_grad(st.p, st.q_all, scale, st.offset, b_local, st.lse_c, a_all, lse_r_all)
    return dq, dp


# ---------------------------------------------------------------------------
# retriever-only: symmetric in-batch-negatives loss
# ---------------------------------------------------------------------------
class _Contrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, scale, ops, comm, q_gather, p_gather):
        st = _contrastive_forward(ops, comm, q, p, scale,
                                  q_gather.wait() if q_gather is not None else None,
                                  p_gather.wait() if p_gather is not None else None)
        loss, _ = ops.contrastive_finalize(st.lse_r, st.lse_c, st.diag, st.n_global)
        ctx.st, ctx.scale, ctx.ops, ctx.comm = st, scale, ops, comm
        ctx.in_dtypes = (q.dtype, p.dtype)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        st = ctx.st
        coef = (g.float() / (2.0 * st.n_global)).reshape(1).expand(st.q.shape[0]).contiguous()
        dq, dp = _contrastive_backward(ctx.ops, ctx.comm, st, ctx.scale, coef, coef)
        return dq.to(ctx.in_dtypes[0]), dp.to(ctx.in_dtypes[1]), None, None, None, None, None


def contrastive_loss(query_embs, passage_embs, logit_scale, *, comm=None, ops=None, q_gather=None, p_gather=None):
    """(get_nt_xent_loss(S) + get_nt_xent_loss(S.t())) / 2 with S = get_cosine_sim(q, p, scale),
    train_retriever_only.py:369-374, without materialising S.  With comm.world_size > 1 this is
    rank r's share of the global-batch loss."""
    return _Contrastive.apply(query_embs, passage_embs, float(logit_scale), ops or default_ops(),
                              comm or LocalComm(), q_gather, p_gather)


# ---------------------------------------------------------------------------
# RAG-end2end: contrastive + marginalised causal-LM cross-entropy
# ---------------------------------------------------------------------------
class _RagE2E(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, logits, ids, mask, qlen, scale, ops, comm, fuse_grad, inplace_grad, q_gather, p_gather, aux):
        st = _contrastive_forward(ops, comm, q, p, scale,
                                  q_gather.wait() if q_gather is not None else None,
                                  p_gather.wait() if p_gather is not None else None)
        stats, Nb, _Mb = ops.ce_prep(mask, qlen)
        if not isinstance(comm, LocalComm):
            comm.all_reduce_sum_(stats)  # stats[0] = M over the global batch (stats[1] becomes B_g)
        need_grad = logits.requires_grad and fuse_grad
        row_lse, row_nll, dlogits = ops.ce_fwd(logits.detach(), ids, mask, stats, need_grad, inplace_grad)
        out3, doc_lp = ops.rag_loss_finalize(row_nll, Nb, st.lse_r, st.lse_c, st.diag, st.n_global, stats)
        ctx.st, ctx.scale, ctx.ops, ctx.comm = st, scale, ops, comm
        ctx.in_dtypes = (q.dtype, p.dtype)
        ctx.fused = need_grad
        if need_grad:
            ctx.save_for_backward(stats, Nb, dlogits)
        else:
            ctx.save_for_backward(stats, Nb, logits.detach(), ids, mask, row_lse)
        if aux is not None:
            aux["contrastive"] = out3[1]
            aux["generator"] = out3[2]
            aux["doc_logprobs"] = doc_lp
            aux["num_target_tokens"] = stats[0]
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        ops, st = ctx.ops, ctx.st
        g = g.float().reshape(1)
        if ctx.fused:
            stats, Nb, dlogits = ctx.saved_tensors
            dlogits = ops.scale_inplace(dlogits, g)  # no-op launch when g == 1
        else:
            stats, Nb, logits, ids, mask, row_lse = ctx.saved_tensors
            dlogits = ops.ce_bwd(logits, ids, mask, stats, row_lse, g) if ctx.needs_input_grad[2] else None
        dq = dp = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # dL/dS = (a_i softmax_row - ...) + (b_j softmax_col - ...),  a_i = g (1/(2B) + N_i/M), b_j = g/(2B)
            b = (g / (2.0 * st.n_global)).expand(st.q.shape[0]).contiguous()
            a = b + g * Nb / stats[0]
            dq, dp = _contrastive_backward(ops, ctx.comm, st, ctx.scale, a, b)
            dq, dp = dq.to(ctx.in_dtypes[0]), dp.to(ctx.in_dtypes[1])
        return (dq, dp, dlogits) + (None,) * 11


def rag_e2e_loss(query_embs, passage_embs, generator_logits, input_ids, attention_mask, query_token_length,
                 logit_scale, *, comm=None, ops=None, fuse_grad=True, inplace_grad=False, q_gather=None,
                 p_gather=None, aux: Optional[dict] = None):
    """combined_loss of train_rage2e.py:441-467 in ~10 kernel launches:
       (CE_row(S) + CE_col(S))/2 + compute_marginalized_loss_from_logits(logits, ids, mask, S, qlen).

    fuse_grad    : write dL/dlogits in the same pass that reads the logits (2x instead of 3x bytes)
    inplace_grad : let that gradient overwrite the logits buffer (caller must not reuse the logits)
    aux          : optional dict receiving detached {"contrastive","generator","doc_logprobs",...}
    """
    return _RagE2E.apply(query_embs, passage_embs, generator_logits, input_ids, attention_mask,
                         query_token_length, float(logit_scale), ops or default_ops(), comm or LocalComm(),
                         bool(fuse_grad), bool(inplace_grad), q_gather, p_gather, aux)


# ---------------------------------------------------------------------------
# k retrieved contexts per sample (RAG-token marginalisation), end to end.  The reference marginalises over ONE context -
# the gold in-batch passage (train_utils.py:123-124) - and leaves more as a TODO (train_rage2e.py:461-462); this is the
# extension SURVEY section 8a leaves room for: forward + closed-form backward for the logits, the query and the k contexts.
# ---------------------------------------------------------------------------
class _RagTopK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, P, logits, ids, mask, qlen, scale, ops, aux):
        B, k, Tg, V = logits.shape
        if P.shape[:2] != (B, k) or ids.shape != (B, k, Tg) or mask.shape != (B, k, Tg) or qlen.shape != (B, k):
            raise ValueError(f"shapes: P {tuple(P.shape)}, logits {tuple(logits.shape)}, ids {tuple(ids.shape)}, "
                             f"mask {tuple(mask.shape)}, qlen {tuple(qlen.shape)}")
        qd, Pd = q.detach().float().contiguous(), P.detach().float().contiguous()
        scores, doc_lp = ops.doc_scores_topk_fwd(qd, Pd, scale)
        flat = logits.detach().reshape(B * k, Tg, V)
        ids_f, mask_f = ids.reshape(B * k, Tg), mask.reshape(B * k, Tg)
        stats, Nb_seq, _ = ops.ce_prep(mask_f, qlen.reshape(-1))
        stats = stats.clone()
        stats[0] = stats[0] / k                     # M = live rows per context set (every context repeats the answer)
        Nb = Nb_seq.reshape(B, k)[:, 0].contiguous()  # the answer has the same number of live rows under every context
        # first answer row of sequence (b,c): the python slice start of lp[qlen-1:] over the Tg-1 shifted rows
        cut = qlen.to(torch.int64) - 1
        cut = torch.where(cut < 0, torch.clamp(cut + (Tg - 1), min=0), cut)
        row_lse, row_nll, _ = ops.ce_fwd(flat, ids_f, mask_f, stats, False)
        out, w = ops.ce_finalize_topk(row_nll.reshape(B, k, Tg), cut, Nb, doc_lp, stats, want_weights=True)
        ctx.ops, ctx.scale, ctx.dims = ops, scale, (B, k, Tg, V)
        ctx.in_dtypes = (q.dtype, P.dtype)
        ctx.save_for_backward(qd, Pd, flat, ids_f, mask_f, stats, row_lse, w, doc_lp, cut, Nb)
        if aux is not None:
            aux["doc_scores"], aux["doc_logprobs"], aux["num_target_tokens"] = scores, doc_lp, stats[0]
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        ops = ctx.ops
        qd, Pd, flat, ids_f, mask_f, stats, row_lse, w, doc_lp, cut, Nb = ctx.saved_tensors
        B, k, Tg, V = ctx.dims
        g = g.float().reshape(1)
        dlogits = None
        if ctx.needs_input_grad[2]:
            dlogits = ops.ce_bwd_weighted(flat, ids_f, mask_f, stats, row_lse, g, w.reshape(-1)).reshape(B, k, Tg, V)
        dq = dP = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dq, dP, _ = ops.doc_scores_topk_bwd(qd, Pd, ctx.scale, doc_lp, w, cut, Nb, g)
            dq, dP = dq.to(ctx.in_dtypes[0]), dP.to(ctx.in_dtypes[1])
        return (dq, dP, dlogits) + (None,) * 6


def rag_e2e_loss_topk(query_embs, context_embs, generator_logits, input_ids, attention_mask, query_token_length,
                      logit_scale, *, ops=None, aux: Optional[dict] = None):
    """Generator loss marginalised over k retrieved contexts per query (RAG-token):

        L = -( sum_b [ 1/k sum_c sum_{t < cut_bc} m lp_bct + sum_j log sum_c p(c|q_b) p(y_bj | context c) ] ) / M
        p(c|q_b) = softmax_c(logit_scale * q_b . P[b,c]),   M = (live rows of all B k sequences) / k

    query_embs [B,D], context_embs [B,k,D], generator_logits [B,k,Tg,V] (sequence (b,c) = the prompt built with context c),
    input_ids / attention_mask [B,k,Tg], query_token_length [B,k] (un-truncated prompt length of every sequence; the answer
    must have the same number of live rows under every context).  Differentiable in the logits, the query and the
    contexts (closed forms, no log-probabilities in HBM).  The reference's loss is the k = 1 special case with the
    in-batch softmax of the gold passage as p(c|q) (`rag_e2e_loss`); this entry point is what its TODO asks for."""
    return _RagTopK.apply(query_embs, context_embs, generator_logits, input_ids, attention_mask, query_token_length,
                          float(logit_scale), ops or default_ops(), aux)


# ---------------------------------------------------------------------------
# lm_head + marginalised CE without ever holding the [B,Tg,V] logits.
# Samples are processed in chunks: logits_c = h_c W^T (hipBLASLt) -> the fused CE kernel turns the chunk into
# its own gradient in place -> dh_c = dlogits_c W (and dW += dlogits_c^T h_c when the head is trainable).
# A chunk (<= ~100 MB bf16) lives in the 256 MB Infinity Cache between the three passes, so the logits and
# their gradient stop costing HBM round trips and 2 x B*Tg*V elements of memory.
# ---------------------------------------------------------------------------
def gemm_wave_rows(V: int, cus: int = 256, tile: int = 256) -> int:
    """Row granularity at which a [rows, V] bf16 GEMM fills whole waves of the chip: hipBLASLt runs the lm_head shapes on
    256 x 256 macro tiles, one per CU per wave, so rows/256 * ceil(V/256) tiles should sit just under a multiple of 256
    CUs.  V = 32000 (125 column tiles): 512 rows = 250 tiles (0.98 of a wave; 768 rows = 375 tiles needs two waves at 0.73 -
    measured 828 vs 1245 TF/s); V = 65024 (254 column tiles): 256 rows."""
    ct = -(-V // tile)
    for k in range(1, 9):
        t = k * ct
        if t / (-(-t // cus) * cus) >= 0.9:
            return k * tile
    return tile


def live_row_index(attention_mask: torch.Tensor, multiple: int = 256) -> Optional[torch.Tensor]:
    """HOST-side list of the generator rows that carry loss: flat index b*Tg + t of every (b, t) with t < Tg-1 and
    attention_mask[b, t+1] != 0 (the shifted-label rows of compute_marginalized_loss_from_logits, reference
    dalm/training/utils/train_utils.py:113-138), padded with -1 to a multiple of `multiple` (use `gemm_wave_rows(V)`) so
    that the GEMMs run whole waves and only a few distinct lengths (= hipGraph shapes) occur.  None when no row is live (the degenerate NaN-loss batch keeps the
    uncompacted path and its NaN gradients) or when compaction would not drop anything.
    Computed where the mask still lives on the host (data loader / batch staging): the count sets tensor shapes, so
    reading it from the device would cost a sync per step."""
    m = attention_mask.detach()
    if m.is_cuda:
        m = m.cpu()
    B, Tg = m.shape
    live = torch.zeros((B, Tg), dtype=torch.bool)
    live[:, :-1] = m[:, 1:] != 0
    rows = live.reshape(-1).nonzero().squeeze(1)
    R = int(rows.numel())
    Rp = -(-R // multiple) * multiple
    if R == 0 or Rp >= B * Tg:
        return None
    out = torch.full((Rp,), -1, dtype=torch.int64)
    out[:R] = rows
    return out


def _row_chunks(rows: int, cap: int, unit: int):
    """Split `rows` into the fewest chunks of at most `cap` rows, sizes multiples of `unit` (the last takes the remainder)
    and as even as the unit allows, larger first: 3584 rows, cap 2048, unit 512 -> [2048, 1536]."""
    cap = max(unit, (cap // unit) * unit)
    units = -(-rows // unit)
    n = -(-units // (cap // unit))
    sizes = [(units // n + (1 if i < units % n else 0)) * unit for i in range(n)]
    sizes[-1] -= units * unit - rows
    return [z for z in sizes if z > 0]


def _use_two_contraction_kernels(ops, hc_all, w, dw, need_grad) -> bool:
    """DALM_LM_HEAD_TRAIN_KERNEL=2: the row-chunk path below with BOTH contractions on the library's own bf16 MFMA kernels
    (`dalm_lm_head_logits`, `dalm_lm_head_dhidden`) instead of hipBLASLt - the hand-written form of f1 WITHOUT the third
    contraction (round 5 recomputed the logits for the backward): frozen bf16 head, V a multiple of 64, hidden width a multiple of 64."""
    return (os.environ.get("DALM_LM_HEAD_TRAIN_KERNEL") == "2" and need_grad and dw is None and hc_all.is_cuda
            and hc_all.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
            and hasattr(ops, "lm_head_logits"))


def _lm_head_rows(ops, hc_all, w, ids_c, mask_c, stats, chunk_rows, dw, need_grad=True):
    """lm_head + marginalised CE + d(hidden) of a COMPACT list of rows: hc_all [Rp, H] hidden states, ids_c / mask_c [Rp] the
    label and the weight of each row (already shifted: row r predicts ids_c[r] with weight mask_c[r]).  The CE kernel sees each
    row chunk as one virtual sample [1, n+1, V] whose shifted labels are the chunk's labels (row n is the kernel's always-dead
    last position), so its code path and numerics are the uncompacted ones.  Returns (dh_c [Rp + 1, H] or None, nll_c [Rp + 1]):
    the extra last row is zero (the target of dead rows when a caller maps back to a padded layout)."""
    Rp, H = hc_all.shape
    V = w.shape[0]
    dh_c = torch.empty((Rp + 1, H), device=hc_all.device, dtype=hc_all.dtype) if need_grad else None
    nll_c = torch.empty((Rp + 1,), device=hc_all.device, dtype=torch.float32)
    if need_grad:
        dh_c[Rp].zero_()
    nll_c[Rp].zero_()
    zero1 = ids_c.new_zeros((1,))
    r1 = 0
    kernels2 = _use_two_contraction_kernels(ops, hc_all, w, dw, need_grad)
    wt = None
    if kernels2:
        from .models.frozen_linear import dgrad_weight

        wt = dgrad_weight(w, torch.bfloat16)                     # the head's transposed copy [K, V] (frozen: made once)
        if wt is None:
            wt = w.t().contiguous()
    if kernels2:
        # both contractions on the library's own kernels.  The logits GEMM and the CE run per row chunk (the chunk is still in the
        # Infinity Cache when the CE reads it); d(logits) of ALL chunks stays in one [Rp + 1, V] buffer and is contracted by
        # ONE dalm_lm_head_dhidden launch: its 256 x 256 output tiles number (Rp / 256) x (K / 256) - a 2048-row chunk alone gives
        # 128 tiles on a 256-CU part (measured 2.4 x the library path that way)
        sizes = _row_chunks(Rp, chunk_rows, gemm_wave_rows(V))
        big = torch.empty((Rp + 1, V), device=hc_all.device, dtype=hc_all.dtype)
        for n in sizes:
            # GEMM of chunk i, then its CE in place.  The CE kernel's virtual sample [1, n + 1, V] has ONE always-dead last row whose
            # gradient it zeroes: that is the first row of chunk i + 1 (not computed yet - its own GEMM writes it next) or, for the
            # last chunk, the extra row Rp
            r0, r1 = r1, r1 + n
            ops.lm_head_logits(hc_all[r0:r1], w, big[r0:r1])
            ids_v = torch.cat((zero1, ids_c[r0:r1])).view(1, n + 1)
            mask_v = torch.cat((zero1.to(mask_c.dtype), mask_c[r0:r1])).view(1, n + 1)
            _lse, nll_v, _dl = ops.ce_fwd(big[r0:r1 + 1].view(1, n + 1, V), ids_v, mask_v, stats, True, True)
            nll_c[r0:r1] = nll_v.reshape(-1)[:n]
        dh_c[:Rp] = ops.lm_head_dhidden(big[:Rp], wt)
        return dh_c, nll_c
    for n in _row_chunks(Rp, chunk_rows, gemm_wave_rows(V)):
        r0, r1 = r1, r1 + n
        buf = torch.empty((n + 1, V), device=hc_all.device, dtype=hc_all.dtype)
        if kernels2:
            ops.lm_head_logits(hc_all[r0:r1], w, buf)
        else:
            torch.mm(hc_all[r0:r1], w.t(), out=buf[:n])
        ids_v = torch.cat((zero1, ids_c[r0:r1])).view(1, n + 1)
        mask_v = torch.cat((zero1.to(mask_c.dtype), mask_c[r0:r1])).view(1, n + 1)
        _lse, nll_v, dl_v = ops.ce_fwd(buf.view(1, n + 1, V), ids_v, mask_v, stats, need_grad, need_grad)
        nll_c[r0:r1] = nll_v.reshape(-1)[:n]
        if not need_grad:
            continue
        dl2 = dl_v.view(n + 1, V)[:n]
        if kernels2:
            dh_c[r0:r1] = ops.lm_head_dhidden(dl2, wt)
            continue
        torch.mm(dl2, w, out=dh_c[r0:r1])
        if dw is not None:
            dw.addmm_(dl2.t().float(), hc_all[r0:r1].float())
    return dh_c, nll_c


def _lm_head_live_rows(ops, h, w, ids, mask, stats, live_rows, chunk_rows, dw, need_grad=True):
    """lm_head + marginalised CE + d(hidden) over the live rows only.  Padding rows (38 % of bench.py's cfg3 batch, and
    whatever padding='max_length' leaves in real data) carry no loss and a zero gradient, so both GEMMs and the CE pass
    skip them (`_lm_head_rows` on the gathered rows)."""
    B, Tg, H = h.shape
    R, Rp = B * Tg, live_rows.numel()
    valid = live_rows >= 0
    rows = live_rows.clamp_min(0)
    dst = torch.where(valid, rows, torch.full_like(rows, R))       # padding entries land in a dump row
    nxt = rows + 1                                                  # live rows have t < Tg-1: same sample
    ids_c = ids.reshape(-1).index_select(0, nxt)
    mask_c = mask.reshape(-1).index_select(0, nxt) * valid.to(mask.dtype)
    hc_all = h.reshape(R, H).index_select(0, rows)
    # results land in [Rp+1]-row buffers whose last row stays zero; the full-size outputs are then ONE gather each through
    # the inverse map (dead rows -> the zero row) instead of a zero fill plus a scatter (43 -> ~20 us at cfg3)
    dh_c, nll_c = _lm_head_rows(ops, hc_all, w, ids_c, mask_c, stats, chunk_rows, dw, need_grad)
    inv = torch.full((R + 1,), Rp, device=h.device, dtype=torch.int64)
    inv.scatter_(0, dst, torch.arange(Rp, device=h.device, dtype=torch.int64))   # padding entries land in inv[R] (unused)
    dh = dh_c.index_select(0, inv[:R]).view(B, Tg, H) if need_grad else None
    row_nll = nll_c.index_select(0, inv[:R])
    return dh, row_nll


# bytes of lm_head weight the fused kernel is preferred up to: 1.1 x the 256 MiB Infinity Cache.  Measured
# (profiles/history/r04_lm_head_rows_sweep.txt): Llama-2-7b's head (32000 x 4096 bf16 = 262 MB) - the kernel is faster than hipBLASLt's
# default-heuristic GEMM + the forward CE kernel at 10 of 15 row counts between 1024 and 4608 (0.84 ... 1.05, mean 0.97) and
# allocates no logits; Falcon-7B's head (65024 x 4544 = 591 MB, re-read from HBM once per band of row tiles) - 2-14 % slower
# at every row count.  (Both measured against hipBLASLt's DEFAULT heuristics: a data-dependent live-row count has no tuned solution.)
_LM_HEAD_KERNEL_MAX_WEIGHT_BYTES = int(1.1 * (256 << 20))


def _use_lm_head_kernel(ops, h, H, w=None) -> bool:
    """Evaluation (no gradient wanted) through the library's own bf16 MFMA kernel (`dalm_lm_head_lse_fwd`: lm_head +
    log-sum-exp + label gather in one kernel, no logits buffer at all).  Default since round 4 where it measured faster:
    lm_head weights that fit the Infinity Cache (see above) while the library runs on its default heuristics (no tuned
    solution table); DALM_LM_HEAD_KERNEL=1 / 0 forces it on / off."""
    import os

    if not (h.is_cuda and h.dtype == torch.bfloat16 and H % 64 == 0 and hasattr(ops, "lm_head_lse")):
        return False
    env = os.environ.get("DALM_LM_HEAD_KERNEL")
    if env is not None:
        return env == "1"
    if w is None or w.dtype != torch.bfloat16 or w.numel() * 2 > _LM_HEAD_KERNEL_MAX_WEIGHT_BYTES:
        return False
    # With the pre-tuned GEMM solution table replayed (dalm_amd.tuning - the trainers and bench.py switch it on) the library
    # path wins at the row counts the table holds (cfg3 live rows, chunks [2048, 1536]: 0.79 vs 0.84 ms,
    # profiles/history/r04_lm_head_eval_paths.txt): the kernel is the default where the library runs on its default heuristics.
    try:
        import torch.cuda.tunable as tunable

        if tunable.is_enabled():
            return False
    except Exception:
        pass
    return True


def _lm_head_nll_kernel(ops, h, w, ids, mask, live_rows):
    """row_nll [B*Tg] = mask-weighted NLL of the shifted labels, computed by the MFMA kernel over all rows or the live ones."""
    B, Tg, H = h.shape
    R = B * Tg
    nxt_ids = torch.cat((ids[:, 1:], ids[:, :1]), dim=1).reshape(-1)
    nxt_mask = torch.cat((mask[:, 1:], torch.zeros_like(mask[:, :1])), dim=1).reshape(-1)
    labels = torch.where(nxt_mask != 0, nxt_ids, torch.full_like(nxt_ids, -1))
    if live_rows is None:
        _lse, nll = ops.lm_head_lse(h.reshape(R, H), w, labels)
        return nll * nxt_mask.to(nll.dtype)
    valid = live_rows >= 0
    rows = live_rows.clamp_min(0)
    lab_c = torch.where(valid, labels.index_select(0, rows), torch.full_like(rows, -1))
    _lse, nll_c = ops.lm_head_lse(h.reshape(R, H).index_select(0, rows), w, lab_c)
    nll_c = nll_c * nxt_mask.index_select(0, rows).to(nll_c.dtype) * valid.to(nll_c.dtype)
    out = torch.zeros((R + 1,), device=h.device, dtype=torch.float32)
    return out.index_copy_(0, torch.where(valid, rows, torch.full_like(rows, R)), nll_c)[:R]


def _use_lm_head_train_kernel(ops, h, H, w, need_dw: bool) -> bool:
    """TRAINING through the library's own bf16 MFMA kernels end to end (round 5): forward `dalm_lm_head_lse_fwd`,
    backward `HipOps.lm_head_backward` (logits recomputed per vocabulary chunk; nothing of size [rows, V] is ever allocated -
    the workspace is two chunk-sized staging buffers, <= 160 MB).  Taken when the head is frozen (LoRA: every BASELINE
    configuration with a 7B generator) and bf16; a trainable head keeps the chunked library path below (it needs dW as well).
    DALM_LM_HEAD_TRAIN_KERNEL=0 keeps the library path, =1 insists (raises when the shapes do not fit)."""
    import os

    env = os.environ.get("DALM_LM_HEAD_TRAIN_KERNEL")
    ok = (h.is_cuda and h.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and H % 64 == 0 and not need_dw
          and hasattr(ops, "lm_head_backward"))
    if env == "1" and not ok:
        raise RuntimeError("DALM_LM_HEAD_TRAIN_KERNEL=1 needs bf16 hidden states, a frozen bf16 head and a hidden width that is a "
                           "multiple of 64")
    return ok and env not in ("0", "2")        # "2": the two-contraction form inside _lm_head_rows


def _lm_head_train_kernel(ops, h, w, ids, mask, stats, live_rows):
    """(d hidden [B, Tg, H], row_nll [B*Tg]) through the hand-written kernels; rows = the live ones when `live_rows` is given."""
    B, Tg, H = h.shape
    R = B * Tg
    nxt_ids = torch.cat((ids[:, 1:], ids[:, :1]), dim=1).reshape(-1)
    nxt_mask = torch.cat((mask[:, 1:], torch.zeros_like(mask[:, :1])), dim=1).reshape(-1)
    coef_all = nxt_mask.to(torch.float32) / stats[0]                         # m_bt / M  
    labels_all = torch.where(nxt_mask != 0, nxt_ids, torch.full_like(nxt_ids, -1))
    if live_rows is None:
        hc, labels, coef = h.reshape(R, H), labels_all, coef_all
    else:
        valid = live_rows >= 0
        rows = live_rows.clamp_min(0)
        hc = h.reshape(R, H).index_select(0, rows)
        labels = torch.where(valid, labels_all.index_select(0, rows), torch.full_like(rows, -1))
        coef = coef_all.index_select(0, rows) * valid.to(torch.float32)
    _lse, nll_c = ops.lm_head_lse(hc, w, labels)
    dh_c = ops.lm_head_backward(hc, w, labels, _lse, coef)
    nll_c = nll_c * (coef != 0).to(nll_c.dtype)
    if live_rows is None:
        return dh_c.view(B, Tg, H), nll_c
    Rp = live_rows.numel()
    inv = torch.full((R + 1,), Rp, device=h.device, dtype=torch.int64)
    dst = torch.where(valid, rows, torch.full_like(rows, R))
    inv.scatter_(0, dst, torch.arange(Rp, device=h.device, dtype=torch.int64))
    dh_pad = torch.cat((dh_c, dh_c.new_zeros((1, H))), dim=0)
    nll_pad = torch.cat((nll_c, nll_c.new_zeros((1,))), dim=0)
    return dh_pad.index_select(0, inv[:R]).view(B, Tg, H), nll_pad.index_select(0, inv[:R])


class _LMHeadRagE2E(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, hidden, weight, ids, mask, qlen, scale, ops, comm, chunk, q_gather, p_gather, aux, live_rows=None,
                packed=None):
        st = _contrastive_forward(ops, comm, q, p, scale,
                                  q_gather.wait() if q_gather is not None else None,
                                  p_gather.wait() if p_gather is not None else None)
        stats, Nb, _Mb = ops.ce_prep(mask, qlen)
        if not isinstance(comm, LocalComm):
            comm.all_reduce_sum_(stats)
        B, Tg = mask.shape
        H = hidden.shape[-1]
        h = hidden.detach()
        w = weight.detach().to(h.dtype)
        need_dw = weight.requires_grad
        dw = torch.zeros(w.shape, device=w.device, dtype=torch.float32) if need_dw else None
        # evaluation (torch.no_grad(), or nothing upstream wants a gradient): forward-only CE, no d(hidden) GEMM
        need_grad = need_dw or any(ctx.needs_input_grad[:3])
        if packed is not None:
            # hidden is [n, H]: the PACKED generator rows (dalm_amd/packed.py), packed = (labels [n], weights [n]) already
            # shifted.  Nothing maps back to a padded layout: the loss only sums row_nll, d(hidden) stays packed.
            ids_c, mask_c = packed
            if need_grad and _use_lm_head_train_kernel(ops, h, H, w, need_dw) and os.environ.get("DALM_LM_HEAD_TRAIN_KERNEL") == "1":
                labels = torch.where(mask_c != 0, ids_c, torch.full_like(ids_c, -1))
                coef = mask_c.to(torch.float32) / stats[0]
                _lse, row_nll = ops.lm_head_lse(h, w, labels)
                dh = ops.lm_head_backward(h, w, labels, _lse, coef)
                row_nll = row_nll * (coef != 0).to(row_nll.dtype)
            else:
                dh_c, nll_c = _lm_head_rows(ops, h, w, ids_c, mask_c, stats, chunk * Tg, dw, need_grad)
                dh = dh_c[:-1] if need_grad else None
                row_nll = nll_c[:-1]
        elif not need_grad and _use_lm_head_kernel(ops, h, H, w):
            dh, row_nll = None, _lm_head_nll_kernel(ops, h, w, ids, mask, live_rows)
        elif need_grad and _use_lm_head_train_kernel(ops, h, H, w, need_dw):
            dh, row_nll = _lm_head_train_kernel(ops, h, w, ids, mask, stats, live_rows)
        elif live_rows is None:
            dh = torch.empty_like(h) if need_grad else None
            row_nll = torch.empty((B * Tg,), device=h.device, dtype=torch.float32)
            for b0 in range(0, B, chunk):
                b1 = min(B, b0 + chunk)
                hc = h[b0:b1].reshape(-1, H)
                logits_c = (hc @ w.t()).view(b1 - b0, Tg, -1)
                _lse, nll_c, dl_c = ops.ce_fwd(logits_c, ids[b0:b1], mask[b0:b1], stats, need_grad, need_grad)
                row_nll[b0 * Tg:b1 * Tg] = nll_c
                if not need_grad:
                    continue
                dl2 = dl_c.view(-1, dl_c.shape[-1])
                torch.mm(dl2, w, out=dh[b0:b1].view(-1, H))
                if need_dw:
                    dw.addmm_(dl2.t().float(), hc.float())
        else:
            dh, row_nll = _lm_head_live_rows(ops, h, w, ids, mask, stats, live_rows, chunk * Tg, dw, need_grad)
        out3, doc_lp = ops.rag_loss_finalize(row_nll, Nb, st.lse_r, st.lse_c, st.diag, st.n_global, stats)
        ctx.st, ctx.scale, ctx.ops, ctx.comm = st, scale, ops, comm
        ctx.in_dtypes = (q.dtype, p.dtype, weight.dtype)
        ctx.save_for_backward(stats, Nb, dh, dw if need_dw else stats)
        ctx.need_dw = need_dw
        if aux is not None:
            aux["contrastive"], aux["generator"] = out3[1], out3[2]
            aux["doc_logprobs"], aux["num_target_tokens"] = doc_lp, stats[0]
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        ops, st = ctx.ops, ctx.st
        stats, Nb, dh, dw = ctx.saved_tensors
        g = g.float().reshape(1)
        dh = ops.scale_inplace(dh, g) if dh.is_cuda else dh * g.to(dh.dtype)
        dweight = (dw * g).to(ctx.in_dtypes[2]) if ctx.need_dw else None
        dq = dp = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            b = (g / (2.0 * st.n_global)).expand(st.q.shape[0]).contiguous()
            a = b + g * Nb / stats[0]
            dq, dp = _contrastive_backward(ops, ctx.comm, st, ctx.scale, a, b)
            dq, dp = dq.to(ctx.in_dtypes[0]), dp.to(ctx.in_dtypes[1])
        return (dq, dp, dh, dweight) + (None,) * 12


def rag_e2e_loss_packed(query_embs, passage_embs, hidden_rows, lm_head_weight, labels, weights, attention_mask,
                        query_token_length, logit_scale, *, comm=None, ops=None, q_gather=None, p_gather=None,
                        aux: Optional[dict] = None):
    """`rag_e2e_loss_from_hidden` for a generator that ran on the PACKED rows only (dalm_amd/packed.py): hidden_rows [n, H] are
    the final hidden states of the packed rows, labels / weights [n] what `packed.packed_labels` returns (row r predicts
    labels[r] with weight weights[r] = attention_mask[b, t + 1]); attention_mask [B, Tg] and query_token_length [B] are the
    batch's own (they give M and the per-sample answer-token counts N_b of train_utils.py:113-138)."""
    V = lm_head_weight.shape[0]
    Tg = attention_mask.shape[1]
    chunk_samples = max(1, (128 << 20) // max(Tg * V * hidden_rows.element_size(), 1))
    return _LMHeadRagE2E.apply(query_embs, passage_embs, hidden_rows, lm_head_weight, None, attention_mask,
                               query_token_length, float(logit_scale), ops or default_ops(), comm or LocalComm(),
                               int(chunk_samples), q_gather, p_gather, aux, None, (labels, weights))


def rag_e2e_loss_from_hidden(query_embs, passage_embs, hidden_states, lm_head_weight, input_ids, attention_mask,
                             query_token_length, logit_scale, *, comm=None, ops=None,
                             chunk_samples: Optional[int] = None, q_gather=None, p_gather=None, aux: Optional[dict] = None,
                             live_rows: Optional[torch.Tensor] = None):
    """Same value and gradients as `rag_e2e_loss(q, p, hidden @ W^T, ...)`, without materialising the logits:
    `hidden_states` [B,Tg,H] are the decoder's final (normed) states, `lm_head_weight` [V,H] (no bias).
    chunk_samples=None sizes a chunk's logits to ~100 MB so that it stays in the 256 MB Infinity Cache between its three
    passes (measured, tools/lm_head_bench.py: cfg3 6 samples 1.94 ms vs 2.63 ms materialised; cfg5 3 samples 5.18 vs 5.79).
    live_rows (from `live_row_index`, host side): only the rows that carry loss go through the two GEMMs and the CE."""
    if chunk_samples is None:
        Tg, V = hidden_states.shape[1], lm_head_weight.shape[0]
        # row chunks of the live-rows path may be a little larger: cfg3 [2048, 1536] rows 1.64 ms vs [1536, 1024, 1024] 1.82 ms
        budget = (128 << 20) if live_rows is not None else (100 << 20)
        chunk_samples = max(1, budget // max(Tg * V * hidden_states.element_size(), 1))
    return _LMHeadRagE2E.apply(query_embs, passage_embs, hidden_states, lm_head_weight, input_ids, attention_mask,
                               query_token_length, float(logit_scale), ops or default_ops(), comm or LocalComm(),
                               int(chunk_samples), q_gather, p_gather, aux, live_rows)


# ---------------------------------------------------------------------------
# drop-ins with the reference's signatures (materialised S / log-probs)
# ---------------------------------------------------------------------------
class _CosineSim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p, scale, ops):
        ctx.save_for_backward(q, p)
        ctx.scale, ctx.ops = scale, ops
        return ops.sim_matmul(q, p, scale)

    @staticmethod
    def backward(ctx, dS):
        q, p = ctx.saved_tensors
        ops, s = ctx.ops, ctx.scale
        dS = dS.float().contiguous()
        dq = ops.gemm(dS, p, s, False, False).to(q.dtype) if ctx.needs_input_grad[0] else None
        dp = ops.gemm(dS, q, s, True, False).to(p.dtype) if ctx.needs_input_grad[1] else None
        return dq, dp, None, None


class _NtXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, ops):
        loss, row_lse, S32 = ops.nt_xent_fwd(S)
        ctx.save_for_backward(S32, row_lse)
        ctx.ops, ctx.dtype = ops, S.dtype
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        S32, row_lse = ctx.saved_tensors
        return ctx.ops.nt_xent_bwd(S32, row_lse, g.float().reshape(1)).to(ctx.dtype), None


class _MargLossFromLogits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, ids, mask, scores, qlen, ops):
        doc_lp, s_lse, S32 = ops.doc_logprob_fwd(scores.detach())
        stats, Nb, _ = ops.ce_prep(mask, qlen)
        row_lse, row_nll, _ = ops.ce_fwd(logits.detach(), ids, mask, stats, False)
        out = ops.ce_finalize(row_nll, Nb, doc_lp, stats)
        ctx.save_for_backward(logits.detach(), ids, mask, stats, row_lse, Nb, S32, s_lse)
        ctx.ops, ctx.s_dtype = ops, scores.dtype
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, ids, mask, stats, row_lse, Nb, S32, s_lse = ctx.saved_tensors
        ops = ctx.ops
        g = g.float().reshape(1)
        dlogits = ops.ce_bwd(logits, ids, mask, stats, row_lse, g) if ctx.needs_input_grad[0] else None
        dS = None
        if ctx.needs_input_grad[3]:
            # L_gen = -(1/M) sum_b N_b doc_lp_b + ...  =>  dL/d doc_lp_b = -g N_b / M
            dS = ops.doc_logprob_bwd(S32, s_lse, -g * Nb / stats[0]).to(ctx.s_dtype)
        return dlogits, None, None, dS, None, None
