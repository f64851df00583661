"""Exact inner-product top-k on the f32-MFMA similarity kernel (SURVEY.md section 8 f, rank 4).

The reference's evaluation (dalm/eval/utils.py:18-68) builds an approximate hnswlib index (space "ip",
M=100, efC=200) over the passage embeddings and queries it with ef=100.  For corpora up to a few million
passages the exact search is cheap on an MI355X: scores = Q . P_block^T on the matrix cores
(`dalm_sim_matmul`, exact f32), a running top-k merge per corpus block.  No index to build, no recall loss.
`ExactIndex` mirrors the two calls the reference makes (construct -> knn_query with hnswlib's
"distance = 1 - inner product" convention).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch

from .ops import default_ops


MAX_FUSED_BLOCK_BYTES = (1 << 31) - (1 << 20)


def _topk_materialised(ops, query_embs, blk, k):
    s = ops.sim_matmul(query_embs, blk, 1.0)                      # [Nq, blk] on the f32 MFMA kernel
    return torch.topk(s, min(k, s.shape[1]), dim=1)


def exact_topk(query_embs: torch.Tensor, corpus_embs: torch.Tensor, k: int, block: int = 262144, ops=None,
               fused: bool = True):
    """(scores [Nq,k], indices [Nq,k]) of the k largest inner products per query, exact, sorted descending.

    fused=True: `dalm_sim_topk` per corpus block - the [Nq, block] score matrix never exists (one pass of the
    streaming MFMA kernel leaves per-32-column maxima; a per-row threshold picks the ~k groups worth re-evaluating); rows with massive ties
    (device-side overflow flag) make that block fall back to the materialising search.  Blocks are merged with a
    running top-k."""
    ops = ops or default_ops()
    nq, nc = query_embs.shape[0], corpus_embs.shape[0]
    if k > nc:
        raise ValueError(f"k={k} exceeds the corpus size {nc}")
    D = corpus_embs.shape[1]
    block = max(1, min(block, MAX_FUSED_BLOCK_BYTES // (4 * ((D + 15) // 16 * 16)) - 128))
    best_s: Optional[torch.Tensor] = None
    best_i: Optional[torch.Tensor] = None
    for c0 in range(0, nc, block):
        blk = corpus_embs[c0:c0 + block]
        kk = min(k, blk.shape[0])
        if fused and ops.sim_topk_supported(D, kk):         # k <= 1024 and (D, k) within the refine kernel's LDS budget
            bs, bi, ovf = ops.sim_topk(query_embs, blk, kk)
            if int(ovf.item()) != 0:                              # eval path: one host sync per block is fine
                bs, bi = _topk_materialised(ops, query_embs, blk, kk)
        else:
            bs, bi = _topk_materialised(ops, query_embs, blk, kk)
        bi = bi + c0
        if best_s is None:
            best_s, best_i = bs, bi
        else:
            cs, ci = torch.cat([best_s, bs], 1), torch.cat([best_i, bi], 1)
            best_s, sel = torch.topk(cs, min(k, cs.shape[1]), dim=1)
            best_i = torch.gather(ci, 1, sel)
    assert best_s is not None and best_i is not None
    return best_s, best_i


class ExactIndex:
    """Drop-in for the hnswlib index of dalm/eval/utils.py: `knn_query` returns (labels, distances) with
    distance = 1 - inner product, like hnswlib's "ip" space."""

    def __init__(self, data: torch.Tensor):
        self.data = data

    def knn_query(self, query_embeddings: torch.Tensor, k: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
        s, i = exact_topk(query_embeddings.to(self.data.device), self.data, k)
        return i, 1.0 - s


def construct_search_index(dim: int, num_elements: int, data: torch.Tensor) -> ExactIndex:
    if data.shape != (num_elements, dim):
        raise ValueError(f"expected data of shape {(num_elements, dim)}, got {tuple(data.shape)}")
    return ExactIndex(data)


def get_nearest_neighbours(k: int, search_index: ExactIndex, query_embeddings: torch.Tensor,
                           ids_to_cat_dict: Dict[int, Any], threshold: float = 0.7) -> List[List[Tuple[Any, float]]]:
    labels, distances = search_index.knn_query(query_embeddings, k=k)
    out = []
    for lab, dist in zip(labels.tolist(), distances.tolist()):
        out.append([(ids_to_cat_dict[l], 1 - d) for l, d in zip(lab, dist) if (1 - d) >= threshold])
    return out


# ---------------------------------------------------------------------------
# retrieval quality metrics of the reference's eval drivers
# ---------------------------------------------------------------------------
def calculate_precision_recall(retrieved_items: List, correct_items: List) -> Tuple[float, float]:
    """dalm/eval/utils.py:68-82: set precision / recall of one query's retrieved list."""
    got, want = set(retrieved_items), set(correct_items)
    hit = len(got & want)
    return hit / len(got), hit / len(want)


def evaluate_retrieval(query_embs: torch.Tensor, corpus_embs: torch.Tensor, correct_idx: torch.Tensor, top_k: int = 10,
                       ops=None) -> Dict[str, float]:
    """recall / precision / hit-rate as dalm/eval/utils.py:225-272 + eval_retriever_only.py:105-178 compute them (one
    correct passage per query, `top_k` retrieved, threshold 0), on the exact fused top-k instead of the hnswlib index.
    `correct_idx[i]` is the corpus row of query i's gold passage."""
    _, idx = exact_topk(query_embs, corpus_embs, top_k, ops=ops)
    idx = idx.cpu()
    correct = correct_idx.cpu()
    precs, recs, hits = [], [], 0
    for i in range(idx.shape[0]):
        p, r = calculate_precision_recall(idx[i].tolist(), [int(correct[i])])
        precs.append(p)
        recs.append(r)
        hits += int(int(correct[i]) in idx[i].tolist())
    n = max(len(precs), 1)
    return {"recall": sum(recs) / n, "precision": sum(precs) / n, "hit_rate": hits / n, "top_k": top_k, "queries": n}
