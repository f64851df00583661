"""Primitive device operations of the loss path, one method per C-ABI entry point.

`HipOps` is the product implementation: every method enqueues hand-written
gfx950 kernels from libdalm_hip.so on torch's current stream and returns freshly
allocated device tensors.  There is no CPU implementation here - CPU tensors
raise (see dalm_amd.hip.require_gpu).

The autograd layer (dalm_amd.fused) and the sharded-negatives logic
(dalm_amd.sharded) are written against this interface only, which is what lets
the world_size>1 host logic be exercised on CPU under gloo with a checker
backend injected by the tests.
"""
from __future__ import annotations

import os

from typing import Optional, Tuple

import torch

from . import hip


def lm_head_chunk_cols(R: int, V: int, K: int, budget_bytes: int = 160 << 20, cus: int = 256) -> int:
    """Vocabulary columns per chunk of `HipOps.lm_head_backward`: the chunk's two staging buffers (dlogits [R, c] and the transposed
    weight chunk [K, c], bf16) stay under `budget_bytes` (default 160 MB: they live in the 256 MB Infinity Cache between the three
    kernels that touch them), and the logits-recompute launch - ceil(R / 256) x (c / 256) tiles of 256 x 256 - fills whole rounds
    of the chip's 256 CUs as nearly as the budget allows."""
    mt = -(-R // 256)
    cap = max(256, (budget_bytes // (2 * (R + K))) // 256 * 256)
    best, best_eff = 256, 0.0
    for nt in range(1, cap // 256 + 1):
        tiles = mt * nt
        eff = tiles / (-(-tiles // cus) * cus)
        if eff >= best_eff - 1e-9:                      # the widest chunk among the best-filling ones
            best, best_eff = nt * 256, eff
    return min(best, -(-V // 256) * 256)


class HipOps:
    name = "hip"

    # ---- K1 ---------------------------------------------------------------
    def pool_fwd(self, h: torch.Tensor, mask: torch.Tensor, normalize: bool):
        """mean_pooling + F.normalize (rag_e2e_base_model.py:95-97,108-111)."""
        dev = hip.require_gpu(h, mask)
        if h.dim() != 3 or mask.shape != h.shape[:2]:
            raise ValueError(f"pool: expected h [B,T,D] and mask [B,T], got {tuple(h.shape)} / {tuple(mask.shape)}")
        h = h.contiguous()
        mask = hip.as_i64(mask)
        B, T, D = h.shape
        emb = torch.empty((B, D), device=dev, dtype=torch.float32)
        norm = torch.empty((B,), device=dev, dtype=torch.float32)
        inv_count = torch.empty((B,), device=dev, dtype=torch.float32)
        ws_bytes = hip.load().dalm_pool_l2norm_fwd_workspace_bytes(B, T, D, hip.dtype_code(h))
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32) if ws_bytes else None
        hip.call("dalm_pool_l2norm_fwd_ws", hip.ptr(h), hip.dtype_code(h), hip.ptr(mask), B, T, D, int(normalize),
                 hip.ptr(emb), hip.ptr(norm), hip.ptr(inv_count), hip.ptr(ws), ws_bytes, hip.stream())
        return emb, norm, inv_count

    def pool_bwd(self, d_emb, emb, norm, inv_count, mask, normalize: bool, T: int, dtype: torch.dtype):
        dev = hip.require_gpu(d_emb, emb, norm, inv_count, mask)
        d_emb = hip.as_f32c(d_emb)
        mask = hip.as_i64(mask)
        B, D = emb.shape
        dh = torch.empty((B, T, D), device=dev, dtype=dtype)
        hip.call("dalm_pool_l2norm_bwd", hip.ptr(d_emb), hip.ptr(emb), hip.ptr(norm), hip.ptr(inv_count),
                 hip.ptr(mask), B, T, D, int(normalize), hip.ptr(dh), hip.dtype_code(dh), hip.stream())
        return dh

    # ---- K2 materialising ---------------------------------------------------
    def sim_matmul(self, A: torch.Tensor, Bm: torch.Tensor, scale: float) -> torch.Tensor:
        """get_cosine_sim (train_utils.py:76-77): (A @ Bm.T) * scale."""
        dev = hip.require_gpu(A, Bm)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        if A.dim() != 2 or Bm.dim() != 2 or A.shape[1] != Bm.shape[1]:
            raise ValueError(f"sim: mat1 and mat2 shapes cannot be multiplied ({tuple(A.shape)} and {tuple(Bm.shape)}^T)")
        m, D = A.shape
        n = Bm.shape[0]
        S = torch.empty((m, n), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_matmul", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), hip.ptr(S), n, hip.stream())
        return S

    def gemm(self, A: torch.Tensor, Bm: torch.Tensor, alpha: float, transA: bool, transB: bool) -> torch.Tensor:
        dev = hip.require_gpu(A, Bm)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        M, K = (A.shape[1], A.shape[0]) if transA else A.shape
        N = Bm.shape[0] if transB else Bm.shape[1]
        Kb = Bm.shape[1] if transB else Bm.shape[0]
        if K != Kb:
            raise ValueError("gemm: inner dimensions differ")
        Cm = torch.empty((M, N), device=dev, dtype=torch.float32)
        hip.call("dalm_gemm_f32", int(transA), int(transB), M, N, K, float(alpha), hip.ptr(A), A.shape[1],
                 hip.ptr(Bm), Bm.shape[1], hip.ptr(Cm), N, hip.stream())
        return Cm

    # ---- K2-K4 fused ----------------------------------------------------------
    def sim_rowstats(self, A: torch.Tensor, Bm: torch.Tensor, scale: float, diag_offset: int):
        """row_lse[i] = logsumexp_j scale*A_i.B_j ; diag[i] = scale*A_i.B_{off+i}."""
        dev = hip.require_gpu(A, Bm)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        m, D = A.shape
        n = Bm.shape[0]
        lib = hip.load()
        ws_bytes = lib.dalm_sim_rowstats_workspace_bytes(m, n, D)
        ws = torch.empty((max(ws_bytes, 4) // 4,), device=dev, dtype=torch.float32)
        row_lse = torch.empty((m,), device=dev, dtype=torch.float32)
        diag = torch.empty((m,), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_rowstats", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset),
                 hip.ptr(row_lse), hip.ptr(diag), hip.ptr(ws), ws_bytes, hip.stream())
        return row_lse, diag

    def sim_rowstats_f32(self, A: torch.Tensor, Bm: torch.Tensor, scale: float, diag_offset: int):
        """`sim_rowstats` pinned to the exact-f32 MFMA kernels (what every shape took before round 4) - for A/B checks."""
        dev = hip.require_gpu(A, Bm)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        m, D = A.shape
        n = Bm.shape[0]
        lib = hip.load()
        ws_bytes = lib.dalm_sim_rowstats_workspace_bytes(m, n, D)
        ws = torch.empty((max(ws_bytes, 4) // 4,), device=dev, dtype=torch.float32)
        row_lse = torch.empty((m,), device=dev, dtype=torch.float32)
        diag = torch.empty((m,), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_rowstats_f32", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset),
                 hip.ptr(row_lse), hip.ptr(diag), hip.ptr(ws), ws_bytes, hip.stream())
        return row_lse, diag

    def sim_rowstats_bf16x3(self, A: torch.Tensor, Bm: torch.Tensor, scale: float, diag_offset: int):
        """The same statistics on the bf16 matrix cores at f32 accuracy (three bf16 thirds per operand, six products along
        K; see include/dalm_hip.h).  `sim_rowstats` routes here by itself for m, n >= 3072."""
        dev = hip.require_gpu(A, Bm)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        m, D = A.shape
        n = Bm.shape[0]
        lib = hip.load()
        if not lib.dalm_sim_rowstats_bf16x3_supported(m, n, D):
            raise ValueError(f"bf16x3 similarity does not support m={m}, n={n}, D={D} (D % 64 == 0, images below 4 GB)")
        ws_bytes = lib.dalm_sim_rowstats_bf16x3_workspace_bytes(m, n, D)
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        row_lse = torch.empty((m,), device=dev, dtype=torch.float32)
        diag = torch.empty((m,), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_rowstats_bf16x3", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset),
                 hip.ptr(row_lse), hip.ptr(diag), hip.ptr(ws), ws_bytes, hip.stream())
        return row_lse, diag

    def sim_grad(self, A, Bm, scale: float, diag_offset: int, row_coef, row_lse, col_coef, col_lse):
        """dA = scale * dS . Bm with the closed-form dS of include/dalm_hip.h."""
        dev = hip.require_gpu(A, Bm, row_coef, row_lse, col_coef, col_lse)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        row_coef, row_lse = hip.as_f32c(row_coef), hip.as_f32c(row_lse)
        col_coef, col_lse = hip.as_f32c(col_coef), hip.as_f32c(col_lse)
        m, D = A.shape
        n = Bm.shape[0]
        lib = hip.load()
        # large global batches: both contractions on the bf16 pipe at f32 accuracy (bf16x3).  From m, n >= 8192 it is the faster
        # form (profiles/r05_sim_grad_x3.txt: 16384^2 9.2 -> 7.1 ms; at 4096^2 its launches of 64-128 tiles do not fill the chip
        # and the f32-pipe flash kernel wins); DALM_SIM_GRAD_X3 = 0 keeps the flash kernel, = 1 forces x3
        x3 = os.environ.get("DALM_SIM_GRAD_X3")
        if x3 != "0" and lib.dalm_sim_grad_bf16x3_supported(m, n, D) and (x3 == "1" or (m >= 8192 and n >= 8192)):
            return self.sim_grad_bf16x3(A, Bm, scale, diag_offset, row_coef, row_lse, col_coef, col_lse)
        ws_bytes = lib.dalm_sim_grad_workspace_bytes(m, n, D)
        ws = torch.empty((max(ws_bytes, 4) // 4,), device=dev, dtype=torch.float32)
        dA = torch.empty((m, D), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_grad", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset),
                 hip.ptr(row_coef), hip.ptr(row_lse), hip.ptr(col_coef), hip.ptr(col_lse), hip.ptr(dA),
                 hip.ptr(ws), ws_bytes, hip.stream())
        return dA

    def sim_grad_bf16x3(self, A, Bm, scale: float, diag_offset: int, row_coef, row_lse, col_coef, col_lse):
        """`sim_grad` through `dalm_sim_grad_bf16x3` (S recomputed and dS . B contracted as bf16x3 on the lm_head core)."""
        dev = hip.require_gpu(A, Bm, row_coef, row_lse, col_coef, col_lse)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        row_coef, row_lse = hip.as_f32c(row_coef), hip.as_f32c(row_lse)
        col_coef, col_lse = hip.as_f32c(col_coef), hip.as_f32c(col_lse)
        m, D = A.shape
        n = Bm.shape[0]
        lib = hip.load()
        if not lib.dalm_sim_grad_bf16x3_supported(m, n, D):
            raise ValueError(f"bf16x3 similarity backward does not support m={m}, n={n}, D={D}")
        ws_bytes = lib.dalm_sim_grad_bf16x3_workspace_bytes(m, n, D)
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        dA = torch.empty((m, D), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_grad_bf16x3", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset), hip.ptr(row_coef),
                 hip.ptr(row_lse), hip.ptr(col_coef), hip.ptr(col_lse), hip.ptr(dA), hip.ptr(ws), ws_bytes, hip.stream())
        return dA

    # ---- K2-K4, small-batch form: S once, 2 launches forward / 1 launch backward ----------------
    def sim_small_supported(self, m: int, n: int, D: int) -> bool:
        return bool(hip.load().dalm_sim_small_supported(int(m), int(n), int(D)))

    # one launch (dalm_sim_small_fwd1) or two (dalm_sim_small_fwd): by shape, as measured (dalm_sim_small_fwd1_preferred,
    # profiles/history/r04_small_one_launch.txt); DALM_SMALL_FWD1 = 1 / 0 forces either
    small_one_launch = {"1": True, "0": False}.get(os.environ.get("DALM_SMALL_FWD1", ""), None)
    _tickets: dict = {}

    def _small_tickets(self, dev: torch.device, words: int) -> torch.Tensor:
        """Arrival tickets of the one-launch forward / backward: zeroed ONCE here, left zero by every call (the last arriver of
        a tile resets its ticket).  One buffer per (device, STREAM) (a buffer shared across streams would make the
        hand-off depend on every caller using one stream): calls that share a buffer are ordered by that stream.  Allocated on
        first use and kept for the life of the process.  The hand-off itself (write-through agent-scope payload stores, drained
        with s_waitcnt vmcnt(0), one agent-scope ticket, agent-scope payload loads) is the form the MI355X guide documents as
        valid on gfx950 ("inter-workgroup visibility"); tools/handoff_stress.py exercises it under uneven load."""
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        buf = HipOps._tickets.get(key)
        if buf is None or buf.numel() < words:
            # 64 Ki words cover every shape of the small path at D <= 8192 (forward: <= ~1.5 k tiles; sliced backward:
            # (rows / 32) x (D / 32) output tiles): no re-allocation - and so no buffer handed back to the allocator while a
            # launch on another stream might still hold it - in any configuration this package runs
            buf = torch.zeros((max(words, 1 << 16),), device=dev, dtype=torch.int32)
            HipOps._tickets[key] = buf
        return buf

    def sim_small_fwd(self, A: torch.Tensor, Bm: torch.Tensor, scale: float, diag_offset: int, want_cols: bool,
                      one_launch: Optional[bool] = None):
        """S = scale*A.B^T (saved), row_lse, diag and (want_cols) col_lse = logsumexp over rows."""
        dev = hip.require_gpu(A, Bm)
        A, Bm = hip.as_f32c(A), hip.as_f32c(Bm)
        m, D = A.shape
        n = Bm.shape[0]
        lib = hip.load()
        S = torch.empty((m, n), device=dev, dtype=torch.float32)
        row_lse = torch.empty((m,), device=dev, dtype=torch.float32)
        diag = torch.empty((m,), device=dev, dtype=torch.float32)
        col_lse = torch.empty((n,), device=dev, dtype=torch.float32) if want_cols else None
        if one_launch is None:
            one_launch = self.small_one_launch
        if one_launch is None:
            one_launch = bool(lib.dalm_sim_small_fwd1_preferred(m, n, D))
        if one_launch:
            ws_bytes = lib.dalm_sim_small_fwd1_workspace_bytes(m, n, D, int(want_cols))
            ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
            tickets = self._small_tickets(dev, lib.dalm_sim_small_fwd1_ticket_words(m, n))
            hip.call("dalm_sim_small_fwd1", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset), hip.ptr(S), n,
                     hip.ptr(row_lse), hip.ptr(diag), hip.ptr(col_lse), hip.ptr(ws), ws_bytes, hip.ptr(tickets), hip.stream())
            return S, row_lse, diag, col_lse
        ws_bytes = lib.dalm_sim_small_workspace_bytes(m, n, D, int(want_cols))
        ws = torch.empty((max(ws_bytes, 4) // 4,), device=dev, dtype=torch.float32)
        hip.call("dalm_sim_small_fwd", hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale), int(diag_offset), hip.ptr(S), n,
                 hip.ptr(row_lse), hip.ptr(diag), hip.ptr(col_lse), hip.ptr(ws), ws_bytes, hip.stream())
        return S, row_lse, diag, col_lse

    # sliced backward (one direction of a long contraction): slices summed in the launch (dalm_sim_small_bwd1) or by a
    # second kernel (dalm_sim_small_bwd_ws); DALM_SMALL_BWD1 = 1 / 0, default decided by measurement
    # (profiles/history/r04_small_one_launch.txt)
    small_bwd_one_launch = os.environ.get("DALM_SMALL_BWD1", "1") == "1"

    def sim_small_bwd(self, S, A, Bm, scale: float, diag_offset: int, row_coef, row_lse, col_coef, col_lse,
                      want_dA: bool = True, want_dB: bool = True, one_launch: Optional[bool] = None):
        """(dA, dB) = (scale dS.B, scale dS^T.A) from the saved S; the closed-form dS of include/dalm_hip.h."""
        dev = hip.require_gpu(S, A, Bm, row_coef, row_lse, col_coef, col_lse)
        A, Bm, S = hip.as_f32c(A), hip.as_f32c(Bm), hip.as_f32c(S)
        row_coef, row_lse = hip.as_f32c(row_coef), hip.as_f32c(row_lse)
        col_coef, col_lse = hip.as_f32c(col_coef), hip.as_f32c(col_lse)
        m, D = A.shape
        n = Bm.shape[0]
        dA = torch.empty((m, D), device=dev, dtype=torch.float32) if want_dA else None
        dB = torch.empty((n, D), device=dev, dtype=torch.float32) if want_dB else None
        lib = hip.load()
        if one_launch is None:
            one_launch = self.small_bwd_one_launch
        ws_bytes = lib.dalm_sim_small_bwd1_workspace_bytes(m, n, D, int(want_dA), int(want_dB)) if one_launch else 0
        if one_launch and ws_bytes:        # sliced: the last slice of every output tile adds the slices (one launch)
            ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32)
            tickets = self._small_tickets(dev, lib.dalm_sim_small_bwd1_ticket_words(m, n, D))
            hip.call("dalm_sim_small_bwd1", hip.ptr(S), S.shape[1], hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale),
                     int(diag_offset), hip.ptr(row_coef), hip.ptr(row_lse), hip.ptr(col_coef), hip.ptr(col_lse),
                     hip.ptr(dA), hip.ptr(dB), hip.ptr(ws), ws_bytes, hip.ptr(tickets), hip.stream())
            return dA, dB
        ws_bytes = lib.dalm_sim_small_bwd_workspace_bytes(m, n, D, int(want_dA), int(want_dB))
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32) if ws_bytes else None   # contraction slices
        hip.call("dalm_sim_small_bwd_ws", hip.ptr(S), S.shape[1], hip.ptr(A), hip.ptr(Bm), m, n, D, float(scale),
                 int(diag_offset), hip.ptr(row_coef), hip.ptr(row_lse), hip.ptr(col_coef), hip.ptr(col_lse),
                 hip.ptr(dA), hip.ptr(dB), hip.ptr(ws) if ws is not None else None, ws_bytes, hip.stream())
        return dA, dB

    def rag_loss_finalize(self, row_nll, Nb, row_lse, col_lse, diag, n_global: int, stats):
        """out = [L_con + L_gen, L_con, L_gen], doc_lp - the loss assembly of train_rage2e.py:443-467 in one launch."""
        dev = hip.require_gpu(row_nll, Nb, row_lse, col_lse, diag, stats)
        n_local = row_lse.shape[0]
        out = torch.empty((3,), device=dev, dtype=torch.float32)
        doc_lp = torch.empty((n_local,), device=dev, dtype=torch.float32)
        hip.call("dalm_rag_loss_finalize", hip.ptr(row_nll), row_nll.numel(), hip.ptr(Nb), hip.ptr(row_lse),
                 hip.ptr(col_lse), hip.ptr(diag), n_local, int(n_global), hip.ptr(stats), hip.ptr(out), hip.ptr(doc_lp),
                 hip.stream())
        return out, doc_lp

    def sim_topk_supported(self, D: int, k: int) -> bool:
        """Does (embedding width, k) fit the fused search?  (k <= 1024 and the refine kernel's LDS budget.)"""
        return bool(hip.load().dalm_sim_topk_supported(int(D), int(k)))

    def sim_topk(self, Q: torch.Tensor, Cm: torch.Tensor, k: int, scale: float = 1.0):
        """(values [m,k] f32, indices [m,k] int64, overflow [1] int32) - exact top-k of scale*Q.Cm^T per row without the
        score matrix; `overflow` non-zero means a row had too many ties for the candidate buffer or ended with fewer than k
        candidates (fall back to the materialising search)."""
        dev = hip.require_gpu(Q, Cm)
        Q, Cm = hip.as_f32c(Q), hip.as_f32c(Cm)
        m, D = Q.shape
        n = Cm.shape[0]
        ws_bytes = hip.load().dalm_sim_topk_workspace_bytes(m, n, D, int(k))
        if ws_bytes == 0:
            raise ValueError("sim_topk: corpus block too large for one call (n*D*4 must stay below 2^31)")
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32)
        val = torch.empty((m, k), device=dev, dtype=torch.float32)
        idx = torch.empty((m, k), device=dev, dtype=torch.int64)
        ovf = torch.empty((1,), device=dev, dtype=torch.int32)
        hip.call("dalm_sim_topk", hip.ptr(Q), hip.ptr(Cm), m, n, D, float(scale), int(k), hip.ptr(val), hip.ptr(idx),
                 hip.ptr(ovf), hip.ptr(ws), ws_bytes, hip.stream())
        return val, idx, ovf

    def lm_head_lse(self, hidden: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor):
        """(row_lse [R], row_nll [R]) of logits = hidden @ weight^T without the logits: hidden [R,K] bf16, weight [V,K] bf16
        (both contiguous, K % 64 == 0), labels [R] int64 with negative entries for rows that carry no loss (nll 0).
        Forward only (evaluation): one bf16 MFMA kernel + a merge of the per-tile (max, sum exp) partials."""
        dev = hip.require_gpu(hidden, weight, labels)
        if hidden.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
            raise TypeError("lm_head_lse needs bf16 hidden states and weights")
        hidden, weight = hidden.contiguous(), weight.contiguous()
        labels = hip.as_i64(labels).contiguous()
        R, K = hidden.shape
        V = weight.shape[0]
        if weight.shape[1] != K or labels.shape != (R,):
            raise ValueError(f"shapes: hidden {tuple(hidden.shape)}, weight {tuple(weight.shape)}, labels {tuple(labels.shape)}")
        ws_bytes = hip.load().dalm_lm_head_lse_workspace_bytes(R, V)
        ws = torch.empty((ws_bytes // 4,), device=dev, dtype=torch.float32)
        row_lse = torch.empty((R,), device=dev, dtype=torch.float32)
        row_nll = torch.empty((R,), device=dev, dtype=torch.float32)
        hip.call("dalm_lm_head_lse_fwd", hip.ptr(hidden), hip.ptr(weight), hip.ptr(labels), R, V, K, hip.ptr(row_lse),
                 hip.ptr(row_nll), hip.ptr(ws), ws_bytes, hip.stream())
        return row_lse, row_nll

    def lm_head_backward(self, hidden: torch.Tensor, weight: torch.Tensor, labels: torch.Tensor, row_lse: torch.Tensor,
                         coef: torch.Tensor, chunk_cols: Optional[int] = None) -> torch.Tensor:
        """d(hidden) [R, K] (bf16) of sum_r -coef_r log softmax(hidden_r W^T)[label_r], given row_lse from `lm_head_lse`: the
        logits are recomputed per vocabulary chunk (`dalm_lm_head_dlogits`), staged as bf16 in a chunk-sized workspace and
        contracted with the transposed chunk of W (`dalm_transpose_bf16`, `dalm_lm_head_dhidden`).  Workspace: about
        (R + K) * chunk_cols * 2 bytes + R * K * 4; the [R, V] logits never exist."""
        dev = hip.require_gpu(hidden, weight, labels, row_lse, coef)
        if hidden.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
            raise TypeError("lm_head_backward needs bf16 hidden states and weights")
        hidden, weight = hidden.contiguous(), weight.contiguous()
        labels, row_lse, coef = hip.as_i64(labels).contiguous(), hip.as_f32c(row_lse), hip.as_f32c(coef)
        R, K = hidden.shape
        V = weight.shape[0]
        if chunk_cols is None:
            chunk_cols = lm_head_chunk_cols(R, V, K)
        dh32 = torch.empty((R, K), device=dev, dtype=torch.float32)
        st = hip.stream()
        for c0 in range(0, V, chunk_cols):
            vc = min(chunk_cols, V - c0)
            pitch = -(-vc // 256) * 256
            dl = torch.empty((R, pitch), device=dev, dtype=torch.bfloat16)
            wt = torch.empty((K, pitch), device=dev, dtype=torch.bfloat16)
            wc = weight[c0:c0 + vc]
            hip.call("dalm_lm_head_dlogits", hip.ptr(hidden), hip.ptr(wc), hip.ptr(labels), hip.ptr(row_lse), hip.ptr(coef),
                     R, vc, K, c0, hip.ptr(dl), pitch, st)
            hip.call("dalm_transpose_bf16", hip.ptr(wc), vc, K, K, hip.ptr(wt), pitch, st)
            hip.call("dalm_lm_head_dhidden", hip.ptr(dl), hip.ptr(wt), R, pitch, K, hip.ptr(dh32), int(c0 > 0), st)
        dh = torch.empty((R, K), device=dev, dtype=torch.bfloat16)
        hip.call("dalm_f32_to_bf16", hip.ptr(dh32), hip.ptr(dh), R * K, st)
        return dh

    def lm_head_logits(self, hidden: torch.Tensor, weight: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """out[:R, :V] = hidden [R, K] . weight [V, K]^T in bf16 through the library's own MFMA main loop (`dalm_lm_head_logits`);
        `out` is a caller-provided [>= R, V] bf16 buffer with a contiguous last dimension (row pitch = its stride)."""
        hip.require_gpu(hidden, weight, out)
        if hidden.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or out.dtype != torch.bfloat16:
            raise TypeError("lm_head_logits needs bf16 tensors")
        hidden, weight = hidden.contiguous(), weight.contiguous()
        R, K = hidden.shape
        V = weight.shape[0]
        if out.stride(-1) != 1 or out.shape[0] < R or out.shape[1] < V:
            raise ValueError("lm_head_logits: out must be [>= R, >= V] with a contiguous last dimension")
        hip.call("dalm_lm_head_logits", hip.ptr(hidden), hip.ptr(weight), R, V, K, hip.ptr(out), out.stride(0), hip.stream())
        return out

    def lm_head_dhidden(self, dlogits: torch.Tensor, weight_t: torch.Tensor) -> torch.Tensor:
        """d(hidden) [R, K] bf16 = dlogits [R, V] . weight_t [K, V]^T (weight_t: the head's transposed copy), f32 accumulation,
        one rounding (`dalm_lm_head_dhidden` + `dalm_f32_to_bf16`).  dlogits rows must be contiguous with pitch V, V % 64 == 0."""
        dev = hip.require_gpu(dlogits, weight_t)
        R, V = dlogits.shape
        K = weight_t.shape[0]
        if dlogits.stride(0) != V or dlogits.stride(1) != 1 or not weight_t.is_contiguous() or weight_t.shape[1] != V or V % 64:
            raise ValueError("lm_head_dhidden: contiguous [R, V] d(logits), [K, V] transposed weight, V a multiple of 64")
        dh32 = torch.empty((R, K), device=dev, dtype=torch.float32)
        st = hip.stream()
        hip.call("dalm_lm_head_dhidden", hip.ptr(dlogits), hip.ptr(weight_t), R, V, K, hip.ptr(dh32), 0, st)
        dh = torch.empty((R, K), device=dev, dtype=torch.bfloat16)
        hip.call("dalm_f32_to_bf16", hip.ptr(dh32), hip.ptr(dh), R * K, st)
        return dh

    def contrastive_finalize(self, row_lse, col_lse, diag, n_global: int):
        dev = hip.require_gpu(row_lse, col_lse, diag)
        n_local = row_lse.shape[0]
        out = torch.empty((1,), device=dev, dtype=torch.float32)
        doc_lp = torch.empty((n_local,), device=dev, dtype=torch.float32)
        hip.call("dalm_contrastive_finalize", hip.ptr(row_lse), hip.ptr(col_lse), hip.ptr(diag), n_local,
                 int(n_global), hip.ptr(out), hip.ptr(doc_lp), hip.stream())
        return out, doc_lp

    # ---- K3 / K4 on a materialised S ---------------------------------------------
    def nt_xent_fwd(self, S: torch.Tensor):
        dev = hip.require_gpu(S)
        if S.dim() != 2 or S.shape[0] != S.shape[1]:
            raise ValueError("get_nt_xent_loss expects a square similarity matrix")
        if S.dtype != torch.float32:
            S = S.float()
        n = S.shape[0]
        loss = torch.empty((1,), device=dev, dtype=torch.float32)
        row_lse = torch.empty((n,), device=dev, dtype=torch.float32)
        hip.call("dalm_nt_xent_fwd", hip.ptr(S), n, S.stride(0), S.stride(1), hip.ptr(loss), hip.ptr(row_lse),
                 hip.stream())
        return loss, row_lse, S

    def nt_xent_bwd(self, S, row_lse, gscale):
        dev = hip.require_gpu(S, row_lse, gscale)
        n = S.shape[0]
        dS = torch.empty_strided(S.shape, S.stride(), device=dev, dtype=torch.float32)
        hip.call("dalm_nt_xent_bwd", hip.ptr(S), n, S.stride(0), S.stride(1), hip.ptr(row_lse), hip.ptr(gscale),
                 hip.ptr(dS), 0, hip.stream())
        return dS

    def doc_logprob_fwd(self, S: torch.Tensor):
        dev = hip.require_gpu(S)
        S = hip.as_f32c(S)
        n = S.shape[0]
        doc_lp = torch.empty((n,), device=dev, dtype=torch.float32)
        row_lse = torch.empty((n,), device=dev, dtype=torch.float32)
        hip.call("dalm_doc_logprob_fwd", hip.ptr(S), n, S.shape[1], hip.ptr(doc_lp), hip.ptr(row_lse), hip.stream())
        return doc_lp, row_lse, S

    def doc_logprob_bwd(self, S, row_lse, coef):
        dev = hip.require_gpu(S, row_lse, coef)
        n = S.shape[0]
        dS = torch.empty((n, S.shape[1]), device=dev, dtype=torch.float32)
        hip.call("dalm_doc_logprob_bwd", hip.ptr(S), n, S.shape[1], hip.ptr(row_lse), hip.ptr(hip.as_f32c(coef)),
                 hip.ptr(dS), S.shape[1], 0, hip.stream())
        return dS

    # ---- K5-K7 -------------------------------------------------------------------
    def ce_prep(self, mask: torch.Tensor, qlen: Optional[torch.Tensor]):
        """stats[0] = M = sum mask[:,1:]; Nb[b] = sum_t mask[b,t+1][t >= qlen_b-1]; Mb[b] = sum_t mask[b,t+1]."""
        dev = hip.require_gpu(mask)
        mask = hip.as_i64(mask)
        B, Tg = mask.shape
        if qlen is not None:
            qlen = hip.as_i64(qlen.reshape(-1))
            if qlen.shape[0] != B:
                # the reference zips with strict=True (train_utils.py:127-129)
                raise ValueError("zip() argument lengths differ: query_token_length vs batch")
        stats = torch.empty((2,), device=dev, dtype=torch.float32)
        Nb = torch.empty((B,), device=dev, dtype=torch.float32)
        Mb = torch.empty((B,), device=dev, dtype=torch.float32)
        hip.call("dalm_marg_ce_prep", hip.ptr(mask), hip.ptr(qlen), B, Tg, hip.ptr(stats), hip.ptr(Nb), hip.ptr(Mb),
                 hip.stream())
        return stats, Nb, Mb

    @staticmethod
    def _logits_view(logits: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
        if logits.dim() != 3:
            raise ValueError(f"logits must be [B,Tg,V], got {tuple(logits.shape)}")
        if logits.stride(2) != 1 or logits.stride(1) < logits.shape[2] or logits.stride(0) < logits.shape[1] * logits.stride(1):
            logits = logits.contiguous()
        return logits, logits.stride(0), logits.stride(1)

    def ce_fwd(self, logits, ids, mask, stats, want_grad: bool, inplace: bool = False, events=None):
        """One pass over the logits: row_lse, row_nll and (optionally) dlogits for upstream grad 1.
        events: optional (start, stop) torch.cuda.Event pair recorded on the launch stream immediately around the
        kernel launch - after the outputs have been allocated - for live roofline timing (bench.py)."""
        dev = hip.require_gpu(logits, ids, mask, stats)
        logits, sb, st = self._logits_view(logits)
        ids, mask = hip.as_i64(ids), hip.as_i64(mask)
        B, Tg, V = logits.shape
        if ids.shape != (B, Tg) or mask.shape != (B, Tg):
            raise ValueError(f"ids/mask must be [B,Tg]={B, Tg}, got {tuple(ids.shape)} / {tuple(mask.shape)}")
        row_lse = torch.empty((B * Tg,), device=dev, dtype=torch.float32)
        row_nll = torch.empty((B * Tg,), device=dev, dtype=torch.float32)
        dlogits = None
        if want_grad:
            dlogits = logits if inplace else torch.empty_strided(logits.shape, logits.stride(), device=dev, dtype=logits.dtype)
        if events is not None:
            events[0].record()
        hip.call("dalm_marg_ce_fwd", hip.ptr(logits), hip.dtype_code(logits), B, Tg, V, sb, st, hip.ptr(ids),
                 hip.ptr(mask), hip.ptr(stats), hip.ptr(row_lse), hip.ptr(row_nll), hip.ptr(dlogits), hip.stream())
        if events is not None:
            events[1].record()
        return row_lse, row_nll, dlogits

    def ce_bwd(self, logits, ids, mask, stats, row_lse, gscale):
        dev = hip.require_gpu(logits, ids, mask, stats, row_lse, gscale)
        logits, sb, st = self._logits_view(logits)
        ids, mask = hip.as_i64(ids), hip.as_i64(mask)
        B, Tg, V = logits.shape
        dlogits = torch.empty_strided(logits.shape, logits.stride(), device=dev, dtype=logits.dtype)
        hip.call("dalm_marg_ce_bwd", hip.ptr(logits), hip.dtype_code(logits), B, Tg, V, sb, st, hip.ptr(ids),
                 hip.ptr(mask), hip.ptr(stats), hip.ptr(row_lse), hip.ptr(gscale), hip.ptr(dlogits), hip.stream())
        return dlogits

    def ce_bwd_weighted(self, logits, ids, mask, stats, row_lse, gscale, row_weight):
        """dlogits = gscale * row_weight[row] * (softmax - onehot) on live rows (k retrieved contexts: the per-row weights of
        `ce_finalize_topk`); logits [S,Tg,V] with S = B*k sequences."""
        dev = hip.require_gpu(logits, ids, mask, stats, row_lse, gscale, row_weight)
        logits, sb, st = self._logits_view(logits)
        ids, mask = hip.as_i64(ids), hip.as_i64(mask)
        B, Tg, V = logits.shape
        row_weight = hip.as_f32c(row_weight)
        if row_weight.numel() != B * Tg:
            raise ValueError(f"row_weight must hold B*Tg = {B * Tg} values, got {row_weight.numel()}")
        dlogits = torch.empty_strided(logits.shape, logits.stride(), device=dev, dtype=logits.dtype)
        hip.call("dalm_marg_ce_bwd_weighted", hip.ptr(logits), hip.dtype_code(logits), B, Tg, V, sb, st, hip.ptr(ids),
                 hip.ptr(mask), hip.ptr(stats), hip.ptr(row_lse), hip.ptr(gscale), hip.ptr(row_weight), hip.ptr(dlogits),
                 hip.stream())
        return dlogits

    def doc_scores_topk_fwd(self, q: torch.Tensor, P: torch.Tensor, scale: float):
        """scores [B,k] = scale * q[b] . P[b,c] and doc_lp = log_softmax over the k contexts."""
        dev = hip.require_gpu(q, P)
        q, P = hip.as_f32c(q), hip.as_f32c(P)
        B, k, D = P.shape
        if q.shape != (B, D):
            raise ValueError(f"q must be [B,D] = {(B, D)}, got {tuple(q.shape)}")
        scores = torch.empty((B, k), device=dev, dtype=torch.float32)
        doc_lp = torch.empty((B, k), device=dev, dtype=torch.float32)
        hip.call("dalm_doc_scores_topk_fwd", hip.ptr(q), hip.ptr(P), B, k, D, float(scale), hip.ptr(scores), hip.ptr(doc_lp),
                 hip.stream())
        return scores, doc_lp

    def doc_scores_topk_bwd(self, q, P, scale: float, doc_lp, weights, cut, Nb, gscale):
        """(dq [B,D], dP [B,k,D], dscores [B,k]) of the k-context loss from the weights of `ce_finalize_topk`."""
        dev = hip.require_gpu(q, P, doc_lp, weights, cut, Nb, gscale)
        q, P = hip.as_f32c(q), hip.as_f32c(P)
        B, k, D = P.shape
        Tg = weights.shape[-1]
        doc_lp, weights, Nb = hip.as_f32c(doc_lp), hip.as_f32c(weights), hip.as_f32c(Nb)
        cut = hip.as_i64(cut).contiguous()
        dq = torch.empty((B, D), device=dev, dtype=torch.float32)
        dP = torch.empty((B, k, D), device=dev, dtype=torch.float32)
        ds = torch.empty((B, k), device=dev, dtype=torch.float32)
        hip.call("dalm_doc_scores_topk_bwd", hip.ptr(q), hip.ptr(P), B, k, D, Tg, float(scale), hip.ptr(doc_lp),
                 hip.ptr(weights), hip.ptr(cut), hip.ptr(Nb), hip.ptr(gscale), hip.ptr(dq), hip.ptr(dP), hip.ptr(ds), hip.stream())
        return dq, dP, ds

    def scale_inplace(self, x: torch.Tensor, gscale: torch.Tensor) -> torch.Tensor:
        hip.require_gpu(x, gscale)
        if not x.is_contiguous():
            raise ValueError("scale_inplace needs a contiguous tensor")
        hip.call("dalm_scale_inplace", hip.ptr(x), hip.dtype_code(x), x.numel(), hip.ptr(gscale), hip.stream())
        return x

    def ce_finalize(self, row_nll, Nb, doc_lp, stats):
        dev = hip.require_gpu(row_nll, stats)
        out = torch.empty((1,), device=dev, dtype=torch.float32)
        hip.call("dalm_marg_ce_finalize", hip.ptr(row_nll), row_nll.numel(), hip.ptr(Nb), hip.ptr(doc_lp),
                 Nb.shape[0], hip.ptr(stats), hip.ptr(out), hip.stream())
        return out

    def ce_finalize_topk(self, row_nll, cut, Nb, doc_lp, stats, want_weights: bool = False):
        """L_gen with k retrieved contexts per sample (`dalm_marg_ce_finalize_topk`): row_nll [B,k,Tg], cut [B,k] int64,
        Nb [B], doc_lp [B,k], stats[0] = M.  Returns (out [1], weights [B,k,Tg] or None).  k = 1 without weights is
        `ce_finalize` bit for bit."""
        dev = hip.require_gpu(row_nll, Nb, doc_lp, stats)
        B, k, Tg = row_nll.shape
        row_nll, doc_lp, Nb = hip.as_f32c(row_nll), hip.as_f32c(doc_lp), hip.as_f32c(Nb)
        cut = hip.as_i64(cut).contiguous()
        if doc_lp.shape != (B, k) or cut.shape != (B, k) or Nb.shape != (B,):
            raise ValueError(f"shapes: row_nll {tuple(row_nll.shape)}, cut {tuple(cut.shape)}, Nb {tuple(Nb.shape)}, doc_lp {tuple(doc_lp.shape)}")
        out = torch.empty((1,), device=dev, dtype=torch.float32)
        w = torch.empty((B, k, Tg), device=dev, dtype=torch.float32) if want_weights else None
        hip.call("dalm_marg_ce_finalize_topk", hip.ptr(row_nll), B, k, Tg, hip.ptr(cut), hip.ptr(Nb), hip.ptr(doc_lp),
                 hip.ptr(stats), hip.ptr(out), hip.ptr(w) if w is not None else None, hip.stream())
        return out, w

    # ---- get_nll / marginalize_log_probs drop-ins -----------------------------------
    def gather_nll(self, log_probs: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        dev = hip.require_gpu(log_probs, labels)
        lp = hip.as_f32c(log_probs)
        labels = hip.as_i64(labels)
        V = lp.shape[-1]
        R = lp.numel() // V
        if labels.numel() != R:
            raise RuntimeError("Size does not match at dimension 1 (gather index vs log_probs)")
        out = torch.empty(labels.shape, device=dev, dtype=torch.float32)
        hip.call("dalm_gather_nll", hip.ptr(lp), hip.ptr(labels), R, V, hip.ptr(out), hip.stream())
        return out

    def marginalize_rows(self, lp: torch.Tensor, doc_lp: torch.Tensor, qlen) -> torch.Tensor:
        """qlen: python int, or a device tensor (its first element is read by the kernel: no host sync)."""
        dev = hip.require_gpu(lp, doc_lp)
        lp = hip.as_f32c(lp)
        T, V = lp.shape
        out = torch.empty((T, V), device=dev, dtype=torch.float32)
        if torch.is_tensor(qlen):
            q = hip.as_i64(qlen.reshape(-1)[:1])
            hip.call("dalm_marginalize_rows_dev", hip.ptr(lp), T, V, hip.ptr(hip.as_f32c(doc_lp.reshape(-1)[:1])), hip.ptr(q),
                     hip.ptr(out), hip.stream())
            return out
        hip.call("dalm_marginalize_rows", hip.ptr(lp), T, V, hip.ptr(hip.as_f32c(doc_lp.reshape(-1)[:1])), int(qlen),
                 hip.ptr(out), hip.stream())
        return out


_DEFAULT = HipOps()


def default_ops() -> HipOps:
    return _DEFAULT
