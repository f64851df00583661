"""SURVEY 8 f1, the training direction: lm_head + marginalised cross-entropy + d(hidden) through the hand-written bf16 MFMA
kernels (`dalm_lm_head_lse_fwd`, `dalm_lm_head_dlogits`, `dalm_transpose_bf16`, `dalm_lm_head_dhidden`) - nothing of size
[rows, V] is ever allocated.  Stands in for `logits = lm_head(hidden)` + `compute_marginalized_loss_from_logits(...).backward()`
(dalm/models/rag_e2e_base_model.py:104-106, dalm/training/utils/train_utils.py:113-138) when the head is frozen (LoRA).

* the backward kernels against a float64 evaluation of  dh = sum_c coef (softmax - onehot) W  (ragged rows / vocabulary,
  several chunks, labels in every tile, rows without loss);
* `rag_e2e_loss_from_hidden` through the kernels against the fp64 oracle and against the chunked library path
  (`closed_chunked` of the repo: torch.mm + the CE kernel), all rows and live rows;
* the same at the full cfg3 (Llama-2-7b head, 32000 x 4096) and cfg5 (Falcon-7B head, 65024 x 4544) sizes."""
import os
import sys
from pathlib import Path

import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
sys.path.insert(0, str(Path(__file__).resolve().parent))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda:0")


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_transpose_bf16_is_exact(dev):
    from dalm_amd import hip

    for rows, cols, ld_dst in ((100, 64, 128), (4099, 512, 4160), (256, 4544, 256), (1, 64, 64)):
        src = torch.randn(rows, cols, generator=torch.Generator().manual_seed(rows)).bfloat16().to(dev)
        dst = torch.full((cols, ld_dst), 7.0, device=dev, dtype=torch.bfloat16)
        hip.call("dalm_transpose_bf16", hip.ptr(src), rows, cols, cols, hip.ptr(dst), ld_dst, hip.stream())
        assert torch.equal(dst[:, :rows], src.t())
        assert float(dst[:, rows:].abs().sum()) == 0.0


@pytest.mark.parametrize("R,V,K,chunk", [(128, 256, 64, None), (200, 1000, 128, 512), (1, 130, 64, None), (333, 4099, 512, 1024),
                                         (640, 32000, 1024, 8192), (300, 5000, 4544, 2048)])
def test_backward_kernels_vs_fp64(dev, R, V, K, chunk):
    from dalm_amd.ops import default_ops

    g = torch.Generator().manual_seed(R * 7 + V)
    h = (0.5 * torch.randn(R, K, generator=g)).to(torch.bfloat16)
    W = (0.5 * torch.randn(V, K, generator=g) / (K / 64) ** 0.5).to(torch.bfloat16)
    labels = torch.randint(0, V, (R,), generator=g)
    coef = torch.rand(R, generator=g) / R
    labels[::5] = -1
    coef[::5] = 0.0
    if R > 3:
        labels[1], labels[2], labels[3] = 0, V - 1, min(V - 1, 257)
    ops = default_ops()
    lse, _ = ops.lm_head_lse(h.to(dev), W.to(dev), labels.to(dev))
    dh = ops.lm_head_backward(h.to(dev), W.to(dev), labels.to(dev), lse, coef.to(dev), chunk_cols=chunk)
    x = h.double() @ W.double().t()
    sm = torch.softmax(x, dim=1)
    onehot = torch.zeros_like(sm)
    live = labels >= 0
    onehot[live, labels[live]] = 1.0
    dl = coef.double().unsqueeze(1) * (sm - onehot)
    ref = dl @ W.double()
    # the staged gradient is rounded to bf16 once (as the logits' gradient is on the reference's bf16 path); the same rounding
    # applied to the fp64 gradient bounds what that costs
    ref_b = dl.to(torch.bfloat16).double() @ W.double()
    cost = _rel(ref_b, ref)
    got = _rel(dh, ref)
    assert got < 6e-3 and got < 2.0 * cost + 2e-3, (got, cost)
    assert float(dh.cpu()[~live].abs().max() if (~live).any() else 0.0) == 0.0          # rows without loss: exactly zero
    again = ops.lm_head_backward(h.to(dev), W.to(dev), labels.to(dev), lse, coef.to(dev), chunk_cols=chunk)
    assert torch.equal(again, dh)                                                       # fixed summation order


def _batch(B, Tg, H, V, D, seed, dev, pad_left=True):
    g = torch.Generator().manual_seed(seed)
    glen = torch.randint(Tg // 4, Tg + 1, (B, 1), generator=g)
    glen[0] = Tg
    ar = torch.arange(Tg).unsqueeze(0)
    mask = ((ar >= (Tg - glen)) if pad_left else (ar < glen)).long()
    ids = torch.randint(0, V, (B, Tg), generator=g)
    qlen = (glen.squeeze(1).float() * 0.7).long().clamp(min=1)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
    p = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1)
    h = torch.randn(B, Tg, H, generator=g).bfloat16()
    W = (torch.randn(V, H, generator=g) / H ** 0.5 * 2.0).bfloat16()
    return [t.to(dev) for t in (q, p, h, W, ids, mask, qlen)]


def _run(q, p, h, W, ids, mask, qlen, live, kernel: bool):
    from dalm_amd.fused import rag_e2e_loss_from_hidden

    os.environ["DALM_LM_HEAD_TRAIN_KERNEL"] = kernel if isinstance(kernel, str) else ("1" if kernel else "0")
    try:
        qq, pp, hh = [t.clone().requires_grad_(True) for t in (q, p, h)]
        aux = {}
        loss = rag_e2e_loss_from_hidden(qq, pp, hh, W, ids, mask, qlen, 100.0, live_rows=live, aux=aux)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach(), qq.grad, pp.grad, hh.grad, aux
    finally:
        os.environ.pop("DALM_LM_HEAD_TRAIN_KERNEL", None)


@pytest.mark.parametrize("use_live", [False, True])
def test_training_through_the_kernels_vs_fp64_oracle_and_library_path(dev, use_live):
    import dalm_oracle as O

    from dalm_amd.fused import live_row_index

    B, Tg, H, V, D = 8, 96, 128, 3000, 64
    q, p, h, W, ids, mask, qlen = _batch(B, Tg, H, V, D, 11, dev)
    live = live_row_index(mask, multiple=64).to(dev) if use_live else None
    loss_k, dq_k, dp_k, dh_k, _ = _run(q, p, h, W, ids, mask, qlen, live, True)
    loss_l, dq_l, dp_l, dh_l, _ = _run(q, p, h, W, ids, mask, qlen, live, False)
    h64 = h.double().cpu().requires_grad_(True)
    q64, p64 = q.double().cpu().requires_grad_(True), p.double().cpu().requires_grad_(True)
    ref = O.ref_step_loss(q64, p64, h64 @ W.double().cpu().t(), ids.cpu(), mask.cpu(), qlen.cpu(), 100)
    ref["loss"].backward()
    # the kernels keep the logits in f32 (the library path rounds them to bf16 first): closer to fp64 than the library path
    assert abs(float(loss_k) - float(ref["loss"])) <= 2e-5 * abs(float(ref["loss"]))
    assert abs(float(loss_l) - float(ref["loss"])) <= 2e-3 * abs(float(ref["loss"]))
    for name, got, lib, want in (("dq", dq_k, dq_l, q64.grad), ("dp", dp_k, dp_l, p64.grad), ("dh", dh_k, dh_l, h64.grad)):
        e_k, e_l = _rel(got, want), _rel(lib, want)
        assert e_k < 6e-3 and e_k < 1.5 * e_l + 1e-3, (name, e_k, e_l)
    dead = (torch.cat((mask[:, 1:], torch.zeros_like(mask[:, :1])), 1) == 0)
    assert float(dh_k[dead].abs().max()) == 0.0


@pytest.mark.parametrize("cfg,V,H", [("cfg3", 32000, 4096), ("cfg5", 65024, 4544)])
def test_full_size_heads_match_the_chunked_library_path(dev, cfg, V, H):
    """B = 18, Tg = 256 (BASELINE configs 3 and 5): loss and d(hidden) of the kernel path against the repo's chunked library
    path (torch.mm + the fused CE kernel, what `--fuse-lm-head` ran through round 4), over the live rows as bench.py uses them;
    every element of d(hidden) is compared.  Written to profiles/ by tools/lm_head_train_bench.py with its times."""
    from dalm_amd.fused import gemm_wave_rows, live_row_index

    B, Tg, D = 18, 256, 1024
    q, p, h, W, ids, mask, qlen = _batch(B, Tg, H, V, D, 3 if cfg == "cfg3" else 5, dev)
    live = live_row_index(mask, multiple=gemm_wave_rows(V)).to(dev)
    loss_k, _, _, dh_k, _ = _run(q, p, h, W, ids, mask, qlen, live, True)
    loss_l, _, _, dh_l, _ = _run(q, p, h, W, ids, mask, qlen, live, False)
    assert abs(float(loss_k) - float(loss_l)) <= 1e-3 * abs(float(loss_l))
    # the two paths round different things to bf16 (the library: logits and their gradient; the kernels: the gradient only)
    assert _rel(dh_k, dh_l) < 8e-3
    assert float((dh_k.float() - dh_l.float()).abs().max()) <= 0.05 * float(dh_l.float().abs().max())
    dead = (torch.cat((mask[:, 1:], torch.zeros_like(mask[:, :1])), 1) == 0)
    assert float(dh_k[dead].abs().max()) == 0.0


@pytest.mark.parametrize("cfg,V,H", [("cfg3", 32000, 4096), ("cfg5", 65024, 4544), ("small", 3008, 128)])
def test_two_contraction_kernels_match_the_library_path(dev, cfg, V, H):
    """Round 6: the hand-written head WITHOUT the third contraction (DALM_LM_HEAD_TRAIN_KERNEL=2: `dalm_lm_head_logits` stores the
    bf16 logits of a row chunk, the fused CE kernel turns them into d(logits) in place, `dalm_lm_head_dhidden` contracts them) -
    the same algorithm as the library path (torch.mm x 2 around the same CE kernel), so the two agree to the summation order of
    the two GEMMs: loss and every element of d(hidden)."""
    from dalm_amd.fused import gemm_wave_rows, live_row_index

    B, Tg, D = (18, 256, 1024) if cfg != "small" else (6, 64, 64)
    q, p, h, W, ids, mask, qlen = _batch(B, Tg, H, V, D, 7, dev)
    live = live_row_index(mask, multiple=gemm_wave_rows(V) if cfg != "small" else 64).to(dev)
    loss_k, dq_k, dp_k, dh_k, _ = _run(q, p, h, W, ids, mask, qlen, live, "2")
    loss_l, dq_l, dp_l, dh_l, _ = _run(q, p, h, W, ids, mask, qlen, live, "0")
    assert abs(float(loss_k) - float(loss_l)) <= 2e-5 * abs(float(loss_l))
    assert _rel(dh_k, dh_l) < 4e-3                     # bf16 logits rounded from differently ordered f32 sums; dh rounded once
    assert _rel(dq_k, dq_l) < 1e-4 and _rel(dp_k, dp_l) < 1e-4
    dead = (torch.cat((mask[:, 1:], torch.zeros_like(mask[:, :1])), 1) == 0)
    assert float(dh_k[dead].abs().max()) == 0.0
