"""CPU oracle for the RAG-end2end / retriever-only loss path of arcee-ai/DALM.

TEST INFRASTRUCTURE ONLY.  Nothing under dalm_amd/ imports this module; it is
used by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg
as the checker / CPU baseline, never as the thing shipped or measured as the
product.

Pinning: the reference's own tests hold no vectors for this path
(tests/training/rag_e2e/test_*.py are `assert True`), so this restatement is
pinned against the reference CODE executed in the build container:
oracle/make_golden.py imports /root/reference (dalm.training.utils.train_utils,
dalm.models.rag_e2e_base_model, dalm.utils), runs it under autograd on seeded
inputs and commits inputs+outputs+gradients to tests/golden/*.npz;
tests/test_oracle_golden.py checks every function below against those files.

Two families:
  * `ref_*`     - op-for-op restatements of the reference's eager torch code
                  (same op sequence, so they double as the CPU-baseline "port").
  * `closed_*`  - the closed-form forward/backward that
                  the HIP kernels implement, written independently of autograd.
All functions are dtype-generic (run them in float64 for tight checks).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# ref_*: follow the reference line by line
# ---------------------------------------------------------------------------
def ref_mean_pooling(token_embeddings: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
    """dalm/models/rag_e2e_base_model.py:108-111 (== retriever_only_base_model.py:66-68)."""
    m = attention_mask.unsqueeze(-1).expand(token_embeddings.size()).to(token_embeddings.dtype)
    return torch.sum(token_embeddings * m, 1) / torch.clamp(m.sum(1), min=1e-9)


def ref_retrieval_embed(token_embeddings: torch.Tensor, attention_mask: torch.Tensor, normalize: bool = True):
    """pool + F.normalize(p=2, dim=1): rag_e2e_base_model.py:95-97."""
    e = ref_mean_pooling(token_embeddings, attention_mask)
    return F.normalize(e, p=2, dim=1) if normalize else e


def ref_eos_mask(mask: torch.Tensor, padding: str = "left") -> torch.Tensor:
    """dalm/utils.py:22-35."""
    new_mask = torch.zeros_like(mask)
    if padding == "right":
        ones = mask.sum(dim=1)
        new_mask[torch.arange(mask.size(0)), ones - 1] = 1
    else:
        new_mask[:, -1] = 1
    return new_mask


def ref_cosine_sim(q: torch.Tensor, p: torch.Tensor, logit_scale) -> torch.Tensor:
    """train_utils.py:76-77."""
    return torch.matmul(q, p.t()) * logit_scale


def ref_nt_xent(sim: torch.Tensor) -> torch.Tensor:
    """train_utils.py:80-88."""
    return F.cross_entropy(sim, torch.arange(len(sim), device=sim.device))


def ref_nll(log_probs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """train_utils.py:91-93."""
    return -torch.gather(log_probs, 2, labels.unsqueeze(2)).squeeze(-1)


def ref_marginalize_log_probs(lp: torch.Tensor, doc_lp: torch.Tensor, qlen) -> torch.Tensor:
    """train_utils.py:96-110 (python slice semantics on qlen-1 included)."""
    qlen = int(qlen)
    head = lp[: qlen - 1, :]
    tail = lp[qlen - 1:, :] + doc_lp
    return torch.cat([head, tail], dim=0)


def ref_marginalized_loss(logits, input_ids, attention_mask, scores, query_token_length) -> torch.Tensor:
    """compute_marginalized_loss_from_logits, train_utils.py:113-138."""
    lp = F.log_softmax(logits[:, :-1, :], dim=2).view(logits.shape[0], -1, logits.size(-1))
    doc = torch.log_softmax(scores, dim=1).diag().unsqueeze(-1).unsqueeze(-1)
    rows = [ref_marginalize_log_probs(a, b, c) for a, b, c in zip(lp, doc, query_token_length, strict=True)]
    marg = torch.stack(rows)
    loss = ref_nll(marg, input_ids[:, 1:])
    lt = loss * attention_mask[:, 1:]
    return lt.sum() / attention_mask[:, 1:].sum()


def ref_step_loss(q, p, logits, ids, mask, qlen, logit_scale) -> Dict[str, torch.Tensor]:
    """The loss part of the RAG-e2e step body, train_rage2e.py:441-467."""
    S = ref_cosine_sim(q, p, logit_scale)
    con = (ref_nt_xent(S) + ref_nt_xent(S.t())) / 2.0
    out = {"S": S, "contrastive": con}
    if logits is not None:
        gen = ref_marginalized_loss(logits, ids, mask, S, qlen)
        out["generator"] = gen
        out["loss"] = con + gen
    else:
        out["loss"] = con
    return out


# ---------------------------------------------------------------------------
# closed_*: what the kernels compute
# ---------------------------------------------------------------------------
def _cut_rows(qlen: torch.Tensor, T: int) -> torch.Tensor:
    """First shifted row that receives the doc term: python slice start of lp[qlen-1:] over T rows."""
    cut = qlen.to(torch.int64) - 1
    cut = torch.where(cut < 0, torch.clamp(cut + T, min=0), cut)
    return cut


def closed_pool(h: torch.Tensor, mask: torch.Tensor, normalize: bool = True):
    m = mask.to(h.dtype)
    cnt = torch.clamp(m.sum(1, keepdim=True), min=1e-9)
    u = (h * m.unsqueeze(-1)).sum(1) / cnt
    nrm = u.norm(dim=1, keepdim=True)
    e = u / torch.clamp(nrm, min=1e-12) if normalize else u
    return e, nrm.squeeze(1), (1.0 / cnt).squeeze(1)


def closed_pool_bwd(d_emb, emb, nrm, inv_cnt, mask, normalize: bool = True):
    if normalize:
        dot = (emb * d_emb).sum(1, keepdim=True)
        big = (nrm >= 1e-12).unsqueeze(1)
        du = torch.where(big, (d_emb - emb * dot) / nrm.unsqueeze(1).clamp(min=1e-300), d_emb / 1e-12)
    else:
        du = d_emb
    g = du * inv_cnt.unsqueeze(1)
    return mask.to(d_emb.dtype).unsqueeze(-1) * g.unsqueeze(1)


def closed_forward(q, p, logits, ids, mask, qlen, scale) -> Dict[str, torch.Tensor]:
    S = scale * (q @ p.t())
    B = S.shape[0]
    lse_r = torch.logsumexp(S, dim=1)
    lse_c = torch.logsumexp(S, dim=0)
    d = S.diag()
    con = 0.5 * ((lse_r - d).mean() + (lse_c - d).mean())
    out = {"S": S, "lse_r": lse_r, "lse_c": lse_c, "diag": d, "contrastive": con, "doc_lp": d - lse_r, "B": B}
    if logits is None:
        out["loss"] = con
        return out
    x = logits[:, :-1, :]
    T = x.shape[1]
    m = mask[:, 1:].to(x.dtype)
    y = ids[:, 1:]
    lse = torch.logsumexp(x, dim=2)
    xy = torch.gather(x, 2, y.unsqueeze(2)).squeeze(2)
    M = m.sum()
    cut = _cut_rows(qlen.reshape(-1), T)
    a = (torch.arange(T).unsqueeze(0) >= cut.unsqueeze(1)).to(x.dtype)
    Nb = (m * a).sum(1)
    gen = ((m * (lse - xy)).sum() - (Nb * out["doc_lp"]).sum()) / M
    out.update({"row_lse": lse, "M": M, "Nb": Nb, "Mb": m.sum(1), "generator": gen, "loss": con + gen})
    return out


def closed_backward(q, p, logits, ids, mask, qlen, scale, fwd: Optional[dict] = None, g: float = 1.0):
    """dL/dq, dL/dp, dL/dlogits from the closed form (no autograd)."""
    f = fwd or closed_forward(q, p, logits, ids, mask, qlen, scale)
    S, B = f["S"], f["B"]
    eye = torch.eye(B, dtype=S.dtype)
    soft_r = torch.exp(S - f["lse_r"].unsqueeze(1))
    soft_c = torch.exp(S - f["lse_c"].unsqueeze(0))
    dS = g * ((soft_r - eye) + (soft_c - eye)) / (2.0 * B)
    dlogits = None
    if logits is not None:
        dS = dS + g * (f["Nb"] / f["M"]).unsqueeze(1) * (soft_r - eye)
        x = logits[:, :-1, :]
        m = mask[:, 1:].to(x.dtype)
        soft = torch.softmax(x, dim=2)
        onehot = F.one_hot(ids[:, 1:], x.shape[2]).to(x.dtype)
        dl = g * (m / f["M"]).unsqueeze(2) * (soft - onehot)
        dlogits = torch.cat([dl, torch.zeros_like(logits[:, -1:, :])], dim=1)
    dq = scale * (dS @ p)
    dp = scale * (dS.t() @ q)
    return {"dS": dS, "dq": dq, "dp": dp, "dlogits": dlogits}


def closed_chunked(q, p, logits, ids, mask, qlen, scale, dlogits_got: Optional[torch.Tensor] = None) -> Dict:
    """closed_forward + closed_backward at BASELINE's FULL sizes (cfg3: 18x256x32000, cfg5: 18x256x65024) in fp64,
    one sample at a time so that the [B,Tg,V] fp64 log-probs / gradient never exist (a sample is ~130 MB at
    V = 65024).  `logits` may be fp32 or bf16 (up-cast exactly, as accelerate does for the reference).  If
    `dlogits_got` is given its error against the closed-form gradient is accumulated sample by sample:
    returns dlogits_err = ||got - ref|| / ||ref|| and dlogits_max_err = max|got - ref| / max|ref|."""
    dt = torch.float64
    q64, p64 = q.to(dt), p.to(dt)
    S = scale * (q64 @ p64.t())
    B = S.shape[0]
    lse_r, lse_c, d = torch.logsumexp(S, 1), torch.logsumexp(S, 0), S.diag()
    con = 0.5 * ((lse_r - d).mean() + (lse_c - d).mean())
    doc_lp = d - lse_r
    Tg = logits.shape[1]
    T = Tg - 1
    m_all = mask[:, 1:].to(dt)
    M = m_all.sum()
    cut = _cut_rows(qlen.reshape(-1), T)
    a = (torch.arange(T).unsqueeze(0) >= cut.unsqueeze(1)).to(dt)
    Nb = (m_all * a).sum(1)
    nll_sum = torch.zeros((), dtype=dt)
    err_sq = ref_sq = torch.zeros((), dtype=dt)
    max_err = max_ref = 0.0
    for b in range(B):
        x = logits[b, :-1, :].to(dt)
        y = ids[b, 1:]
        lse = torch.logsumexp(x, dim=1)
        xy = torch.gather(x, 1, y.unsqueeze(1)).squeeze(1)
        nll_sum = nll_sum + (m_all[b] * (lse - xy)).sum()
        if dlogits_got is not None:
            ref = torch.exp(x - lse.unsqueeze(1))
            ref[torch.arange(T), y] -= 1.0
            ref *= (m_all[b] / M).unsqueeze(1)
            got = dlogits_got[b].to(dt)
            diff = got[:-1] - ref
            err_sq = err_sq + (diff * diff).sum() + (got[-1] * got[-1]).sum()   # last position must be exactly 0
            ref_sq = ref_sq + (ref * ref).sum()
            max_err = max(max_err, float(diff.abs().max()), float(got[-1].abs().max()))
            max_ref = max(max_ref, float(ref.abs().max()))
    gen = (nll_sum - (Nb * doc_lp).sum()) / M
    eye = torch.eye(B, dtype=dt)
    soft_r, soft_c = torch.exp(S - lse_r.unsqueeze(1)), torch.exp(S - lse_c.unsqueeze(0))
    dS = ((soft_r - eye) + (soft_c - eye)) / (2.0 * B) + (Nb / M).unsqueeze(1) * (soft_r - eye)
    out = {"loss": con + gen, "contrastive": con, "generator": gen, "dq": scale * (dS @ p64), "dp": scale * (dS.t() @ q64),
           "M": M}
    if dlogits_got is not None:
        out["dlogits_err"] = float(torch.sqrt(err_sq / ref_sq))
        out["dlogits_max_err"] = max_err / max_ref
    return out


def closed_gen_loss_topk(label_lp: torch.Tensor, mask: torch.Tensor, cut: torch.Tensor, doc_lp: torch.Tensor) -> Dict:
    """Generator loss with k retrieved contexts per sample (RAG-token marginalisation; the reference, train_utils.py:113-138,
    is the k = 1 case and is what pins this function: at k = 1 it must equal `closed_forward`'s generator term).
    label_lp [B,k,T]: log-prob of the label at every shifted row of sequence (b,c); mask [B,k,T] (m_bt of that sequence);
    cut [B,k]: first answer row; doc_lp [B,k].  Answer token j of sample b sits at row cut_bc + j of sequence (b,c).
        L = -( sum_b [ 1/k sum_c sum_{t<cut_bc} m lp  +  sum_j log sum_c exp(doc_lp_bc + lp[b,c,cut_bc+j]) ] ) / M,
        M = (sum of all masks) / k.
    Written through PROBABILITIES (sum_c p(c) p(y|c)), not through logsumexp, so that it is an independent statement."""
    B, k, T = label_lp.shape
    dt = label_lp.dtype
    m = mask.to(dt)
    M = m.sum() / k
    tot = torch.zeros((), dtype=dt)
    nans = []
    for b in range(B):
        n_b = None
        for c in range(k):
            cu = int(cut[b, c])
            tot = tot + (m[b, c, :cu] * label_lp[b, c, :cu]).sum() / k
            n_c = int(m[b, c, cu:].sum())
            assert bool((m[b, c, cu:cu + n_c] == 1).all()), "answer rows must be contiguous and live"
            assert n_b is None or n_b == n_c, "the answer has the same length under every context"
            n_b = n_c
        nans.append(n_b)
        for j in range(n_b):
            p = sum(torch.exp(doc_lp[b, c]) * torch.exp(label_lp[b, c, int(cut[b, c]) + j]) for c in range(k))
            tot = tot + torch.log(p)
    return {"generator": -tot / M, "M": M, "Nb": torch.tensor(nans, dtype=dt)}


def ref_rag_topk_loss(q: torch.Tensor, P: torch.Tensor, logits: torch.Tensor, ids: torch.Tensor, mask: torch.Tensor,
                      qlen: torch.Tensor, scale: float) -> torch.Tensor:
    """The k-context generator loss stated through torch ops an autograd can differentiate (fp64 in the tests): document
    posteriors = softmax over the k contexts of scale * q.P, token log-probs = log_softmax of the logits gathered at the
    shifted labels (train_utils.py:121-131 per sequence), combined by `closed_gen_loss_topk` (probability form).
    q [B,D], P [B,k,D], logits [B,k,Tg,V], ids / mask [B,k,Tg], qlen [B,k].  TEST INFRASTRUCTURE."""
    B, k, Tg, V = logits.shape
    doc_lp = torch.log_softmax(scale * torch.einsum("bd,bkd->bk", q, P), dim=1)
    lp = torch.log_softmax(logits[:, :, :-1, :], dim=-1)
    label_lp = torch.gather(lp, 3, ids[:, :, 1:].unsqueeze(-1)).squeeze(-1)
    m = mask[:, :, 1:]
    cut = qlen.to(torch.int64) - 1
    cut = torch.where(cut < 0, torch.clamp(cut + (Tg - 1), min=0), cut)
    return closed_gen_loss_topk(label_lp, m, cut, doc_lp)["generator"]


# ---------------------------------------------------------------------------
# OracleOps: the dalm_amd.ops.HipOps interface on CPU tensors (float64 inside).
# Injected by tests/test_sharded_gloo.py to exercise the world_size>1 host logic
# without a GPU.  NOT importable from the product package.
# ---------------------------------------------------------------------------
class OracleOps:
    name = "oracle"
    dt = torch.float64

    def pool_fwd(self, h, mask, normalize):
        e, nrm, ic = closed_pool(h.to(self.dt), mask, normalize)
        return e.float(), nrm.float(), ic.float()

    def pool_bwd(self, d_emb, emb, norm, inv_count, mask, normalize, T, dtype):
        return closed_pool_bwd(d_emb.to(self.dt), emb.to(self.dt), norm.to(self.dt), inv_count.to(self.dt), mask,
                               normalize).to(dtype)

    def sim_rowstats(self, A, Bm, scale, diag_offset):
        S = scale * (A.to(self.dt) @ Bm.to(self.dt).t())
        idx = torch.arange(A.shape[0])
        return torch.logsumexp(S, 1).float(), S[idx, diag_offset + idx].float()

    def sim_grad(self, A, Bm, scale, diag_offset, row_coef, row_lse, col_coef, col_lse):
        A64, B64 = A.to(self.dt), Bm.to(self.dt)
        S = scale * (A64 @ B64.t())
        rc, cc = row_coef.to(self.dt).unsqueeze(1), col_coef.to(self.dt).unsqueeze(0)
        dS = rc * torch.exp(S - row_lse.to(self.dt).unsqueeze(1)) + cc * torch.exp(S - col_lse.to(self.dt).unsqueeze(0))
        idx = torch.arange(A.shape[0])
        dS[idx, diag_offset + idx] -= (rc.squeeze(1) + cc.squeeze(0)[diag_offset + idx])
        return (scale * (dS @ B64)).float()

    # small-batch form (mirrors HipOps.sim_small_*: same shape rule so the gloo tests take the same host paths)
    def sim_small_supported(self, m, n, D):
        return 0 < m <= 1024 and 0 < n <= 8192 and m * n <= (1 << 20)

    def sim_small_fwd(self, A, Bm, scale, diag_offset, want_cols):
        S = scale * (A.to(self.dt) @ Bm.to(self.dt).t())
        idx = torch.arange(A.shape[0])
        col = torch.logsumexp(S, 0).float() if want_cols else None
        return S.float(), torch.logsumexp(S, 1).float(), S[idx, diag_offset + idx].float(), col

    def sim_small_bwd(self, S, A, Bm, scale, diag_offset, row_coef, row_lse, col_coef, col_lse, want_dA=True, want_dB=True):
        A64, B64, S64 = A.to(self.dt), Bm.to(self.dt), S.to(self.dt)
        rc, cc = row_coef.to(self.dt).unsqueeze(1), col_coef.to(self.dt).unsqueeze(0)
        dS = rc * torch.exp(S64 - row_lse.to(self.dt).unsqueeze(1)) + cc * torch.exp(S64 - col_lse.to(self.dt).unsqueeze(0))
        idx = torch.arange(A.shape[0])
        dS[idx, diag_offset + idx] -= (rc.squeeze(1) + cc.squeeze(0)[diag_offset + idx])
        return ((scale * (dS @ B64)).float() if want_dA else None, (scale * (dS.t() @ A64)).float() if want_dB else None)

    def rag_loss_finalize(self, row_nll, Nb, row_lse, col_lse, diag, n_global, stats):
        con, doc_lp = self.contrastive_finalize(row_lse, col_lse, diag, n_global)
        gen = self.ce_finalize(row_nll, Nb, doc_lp, stats)
        return torch.cat([con + gen, con, gen]), doc_lp

    def contrastive_finalize(self, row_lse, col_lse, diag, n_global):
        r, c, d = row_lse.to(self.dt), col_lse.to(self.dt), diag.to(self.dt)
        out = 0.5 * ((r - d).sum() + (c - d).sum()) / n_global
        return out.reshape(1).float(), (d - r).float()

    def ce_prep(self, mask, qlen):
        m = mask[:, 1:].to(self.dt)
        T = m.shape[1]
        if qlen is None:
            Nb = torch.zeros(m.shape[0], dtype=self.dt)
        else:
            cut = _cut_rows(qlen.reshape(-1), T)
            a = (torch.arange(T).unsqueeze(0) >= cut.unsqueeze(1)).to(self.dt)
            Nb = (m * a).sum(1)
        stats = torch.tensor([m.sum(), float(m.shape[0])], dtype=torch.float32)
        return stats, Nb.float(), m.sum(1).float()

    def ce_fwd(self, logits, ids, mask, stats, want_grad, inplace=False):
        B, Tg, V = logits.shape
        x = logits[:, :-1, :].to(self.dt)
        m = mask[:, 1:].to(self.dt)
        y = ids[:, 1:]
        lse = torch.logsumexp(x, 2)
        xy = torch.gather(x, 2, y.unsqueeze(2)).squeeze(2)
        row_lse = torch.zeros(B, Tg, dtype=self.dt)
        row_nll = torch.zeros(B, Tg, dtype=self.dt)
        row_lse[:, :-1] = lse * (m != 0)
        row_nll[:, :-1] = m * (lse - xy)
        dl = None
        if want_grad:
            M = stats[0].to(self.dt)
            soft = torch.softmax(x, 2)
            onehot = F.one_hot(y, V).to(self.dt)
            d = (m / M).unsqueeze(2) * (soft - onehot)
            dl = torch.cat([d, torch.zeros(B, 1, V, dtype=self.dt)], 1).to(logits.dtype)
            if inplace:
                logits.copy_(dl)
                dl = logits
        return row_lse.reshape(-1).float(), row_nll.reshape(-1).float(), dl

    def ce_bwd(self, logits, ids, mask, stats, row_lse, gscale):
        _, _, dl = self.ce_fwd(logits, ids, mask, stats, True)
        return (dl.to(self.dt) * gscale.to(self.dt)).to(logits.dtype)

    def scale_inplace(self, x, gscale):
        return x.mul_(gscale.to(x.dtype))

    def ce_finalize(self, row_nll, Nb, doc_lp, stats):
        s = row_nll.to(self.dt).sum()
        if doc_lp is not None:
            s = s - (Nb.to(self.dt) * doc_lp.to(self.dt)).sum()
        return (s / stats[0].to(self.dt)).reshape(1).float()
