"""Loss functions of the training step, same names and signatures as the reference's
dalm/training/utils/train_utils.py:76-138, every one backed by hand-written gfx950 kernels
(libdalm_hip.so through dalm_amd.hip).  Tensors must live on an MI355X; CPU tensors raise.

For the training loop prefer the fused entry points in dalm_amd.fused
(`contrastive_loss`, `rag_e2e_loss`): they never materialise S or the log-probs.
"""
from __future__ import annotations

from typing import Dict

import torch

from ...fused import _CosineSim, _MargLossFromLogits, _NtXent
from ...ops import default_ops


def get_cosine_sim(query_embs: torch.Tensor, passage_embs: torch.Tensor, logit_scale: int) -> torch.Tensor:
    """matmul(q, p.T) * logit_scale (reference :76-77) on the f32 matrix cores."""
    return _CosineSim.apply(query_embs, passage_embs, float(logit_scale), default_ops())


def get_nt_xent_loss(sim_scores: torch.Tensor) -> torch.Tensor:
    """cross_entropy(sim, arange(n)) (reference :80-88); accepts the `.t()` view the trainer passes."""
    return _NtXent.apply(sim_scores, default_ops())


class _GatherNll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, labels):
        ctx.save_for_backward(labels)
        ctx.shape, ctx.dtype = log_probs.shape, log_probs.dtype
        return default_ops().gather_nll(log_probs, labels)

    @staticmethod
    def backward(ctx, g):
        (labels,) = ctx.saved_tensors
        d = torch.zeros(ctx.shape, device=g.device, dtype=ctx.dtype)
        d.scatter_(2, labels.unsqueeze(2), (-g).unsqueeze(2).to(ctx.dtype))
        return d, None


def get_nll(log_probs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """-gather(log_probs, 2, labels) (reference :91-93)."""
    return _GatherNll.apply(log_probs, labels)


class _MarginalizeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lp, doc_lp, qlen):
        T = lp.shape[0]
        if torch.is_tensor(qlen) and qlen.is_cuda:
            # the length stays on the device (the reference's loop hands over elements of a device tensor): python-slice
            # start of lp[qlen-1:] as tensor arithmetic, no .item() / int() round trip per sample
            cut = qlen.reshape(()).to(torch.int64) - 1
            cut = torch.where(cut < 0, torch.clamp(cut + T, min=0), cut).clamp(max=T)
            ctx.save_for_backward(cut)
            ctx.cut = None
        else:
            cut = int(qlen) - 1
            if cut < 0:
                cut = max(cut + T, 0)
            ctx.cut = min(cut, T)
            qlen = int(qlen)
        ctx.doc_shape = doc_lp.shape
        return default_ops().marginalize_rows(lp, doc_lp, qlen).to(lp.dtype)

    @staticmethod
    def backward(ctx, g):
        if ctx.cut is None:
            (cut,) = ctx.saved_tensors
            rows = (torch.arange(g.shape[0], device=g.device) >= cut).to(g.dtype)
            d_doc = (g.sum(dim=1) * rows).sum()
        else:
            d_doc = g[ctx.cut:].sum()
        return g, d_doc.reshape(ctx.doc_shape).to(g.dtype), None


def marginalize_log_probs(logprobs_logits: torch.Tensor, doc_logprobs: torch.Tensor,
                          query_token_length: torch.Tensor) -> torch.Tensor:
    """rows [qlen-1:] get `+ doc_logprobs` (reference :96-110), one streaming kernel instead of slice/add/cat."""
    return _MarginalizeRows.apply(logprobs_logits, doc_logprobs, query_token_length)


def compute_marginalized_loss_from_logits(logits: torch.Tensor, input_tensors: torch.Tensor,
                                          attention_mask: torch.Tensor, scores: torch.Tensor,
                                          query_token_length: torch.Tensor) -> torch.Tensor:
    """Reference :113-138 as one pass over the logits (online LSE + label gather + doc term + masked mean)."""
    return _MargLossFromLogits.apply(logits, input_tensors, attention_mask, scores, query_token_length, default_ops())


# ---- checkpoint hooks (reference :12-73): same directory layout -----------------------
def extract_sub_state_dict(full_state_dict: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in full_state_dict.items() if k.startswith(prefix)}


from .._hooks import load_model_hook, save_model_hook  # noqa: E402,F401
