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


def exact_topk(query_embs: torch.Tensor, corpus_embs: torch.Tensor, k: int, block: int = 131072, ops=None):
    """(scores [Nq,k], indices [Nq,k]) of the k largest inner products per query, exact, sorted descending."""
    ops = ops or default_ops()
    nq, nc = query_embs.shape[0], corpus_embs.shape[0]
    if k > nc:
        raise ValueError(f"k={k} exceeds the corpus size {nc}")
    best_s: Optional[torch.Tensor] = None
    best_i: Optional[torch.Tensor] = None
    for c0 in range(0, nc, block):
        blk = corpus_embs[c0:c0 + block]
        s = ops.sim_matmul(query_embs, blk, 1.0)                  # [Nq, blk] on the f32 MFMA kernel
        kk = min(k, s.shape[1])
        bs, bi = torch.topk(s, kk, dim=1)
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
