"""GPU parity: the HIP kernels (through the C ABI) vs the reference-pinned oracle.

Tolerances (north-star: losses / grad-norms within 1e-3 relative of the reference in fp32):
  fp32 : loss rel <= 1e-4, gradient norm-rel <= 1e-4, elementwise atol scaled by the tensor max
  bf16 : inputs are bf16-rounded ONCE and both sides see the same values, so the loss keeps the
         fp32 tolerance; a bf16 gradient OUTPUT carries 2^-9 relative rounding per element
         -> norm-rel <= 4e-3.
"""
import pytest
import torch

import dalm_oracle as O
from helpers import LOSS_CASES, POOL_CASES, load_npz, norm_rel_err, synth_batch

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4
GRAD_NORM_RTOL = 1e-4
BF16_GRAD_NORM_RTOL = 4e-3


@pytest.fixture(scope="module")
def dev():
    from dalm_amd import hip

    hip.load()  # fail loudly if libdalm_hip.so is absent: no fallback
    return torch.device("cuda:0")


def assert_grad_close(got, ref, rtol=GRAD_NORM_RTOL, name=""):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    nan_g, nan_r = torch.isnan(got), torch.isnan(ref)
    assert torch.equal(nan_g, nan_r), f"{name}: NaN pattern differs"
    got, ref = torch.nan_to_num(got), torch.nan_to_num(ref)
    scale = float(ref.abs().max())
    if scale == 0.0:  # analytically zero gradient (e.g. B == 1): allow rounding residue only
        assert float(got.abs().max()) <= 1e-5, name
        return
    assert norm_rel_err(got, ref) <= rtol, (name, norm_rel_err(got, ref))
    assert float((got - ref).abs().max()) <= 10 * rtol * scale, (name, float((got - ref).abs().max()), scale)


def assert_loss_close(got, ref, rtol=LOSS_RTOL):
    got, ref = float(got.detach()) if torch.is_tensor(got) else float(got), float(ref)
    if ref != ref:
        assert got != got
        return
    assert abs(got - ref) <= rtol * max(abs(ref), 1e-3), (got, ref)


# ---------------------------------------------------------------------------
# golden vectors from the reference
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("fuse_grad", [True, False])
@pytest.mark.parametrize("case", LOSS_CASES)
def test_fused_rag_e2e_loss_vs_reference_golden(dev, case, fuse_grad):
    from dalm_amd.fused import rag_e2e_loss

    z = load_npz(case)
    q = z["q"].float().to(dev).requires_grad_(True)
    p = z["p"].float().to(dev).requires_grad_(True)
    lg = z["logits"].float().to(dev).requires_grad_(True)
    aux = {}
    loss = rag_e2e_loss(q, p, lg, z["ids"].to(dev), z["mask"].to(dev), z["qlen"].to(dev), int(z["scale"]),
                        fuse_grad=fuse_grad, aux=aux)
    loss.backward()
    assert_loss_close(loss, z["ref64_loss"])
    assert_loss_close(aux["contrastive"], z["ref64_contrastive"])
    assert_loss_close(aux["generator"], z["ref64_generator"])
    assert_grad_close(q.grad, z["ref64_dq"], name="dq")
    assert_grad_close(p.grad, z["ref64_dp"], name="dp")
    assert_grad_close(lg.grad, z["ref64_dlogits"], name="dlogits")


@pytest.mark.parametrize("case", LOSS_CASES)
def test_dropin_functions_vs_reference_golden(dev, case):
    """The reference's own call sequence (train_rage2e.py:441-467) on the drop-in functions."""
    from dalm_amd.training.utils import train_utils as tu

    z = load_npz(case)
    q = z["q"].float().to(dev).requires_grad_(True)
    p = z["p"].float().to(dev).requires_grad_(True)
    lg = z["logits"].float().to(dev).requires_grad_(True)
    S = tu.get_cosine_sim(q, p, int(z["scale"]))
    loss_q = tu.get_nt_xent_loss(S)
    loss_p = tu.get_nt_xent_loss(S.t())
    con = (loss_q + loss_p) / 2.0
    gen = tu.compute_marginalized_loss_from_logits(lg, z["ids"].to(dev), z["mask"].to(dev), S, z["qlen"].to(dev))
    (con + gen).backward()
    assert_grad_close(S.detach(), z["ref64_S"], name="S")
    assert_loss_close(loss_q, z["ref64_loss_query"])
    assert_loss_close(loss_p, z["ref64_loss_passage"])
    assert_loss_close(gen, z["ref64_generator"])
    assert_grad_close(q.grad, z["ref64_dq"], name="dq")
    assert_grad_close(p.grad, z["ref64_dp"], name="dp")
    assert_grad_close(lg.grad, z["ref64_dlogits"], name="dlogits")


@pytest.mark.parametrize("case", LOSS_CASES)
def test_contrastive_only_vs_reference_golden(dev, case):
    from dalm_amd.fused import contrastive_loss

    z = load_npz(case)
    q = z["q"].float().to(dev).requires_grad_(True)
    p = z["p"].float().to(dev).requires_grad_(True)
    loss = contrastive_loss(q, p, int(z["scale"]))
    loss.backward()
    assert_loss_close(loss, z["ref64_con_only"])
    assert_grad_close(q.grad, z["ref64_con_only_dq"], name="dq")
    assert_grad_close(p.grad, z["ref64_con_only_dp"], name="dp")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", POOL_CASES)
def test_pool_l2norm_vs_reference_golden(dev, case, dtype):
    from dalm_amd.fused import pool_l2norm

    z = load_npz(case)
    normalize = bool(z["normalize"])
    h_in = z["h"].to(dtype)
    h = h_in.to(dev).requires_grad_(True)
    e = pool_l2norm(h, z["mask"].to(dev), normalize)
    (e * z["upstream"].float().to(dev)).sum().backward()
    if dtype == torch.float32:
        ref_e, ref_dh = z["ref64_emb"], z["ref64_dh"]
        tol = GRAD_NORM_RTOL
    else:  # same bf16-rounded inputs through the oracle
        hh = h_in.double().requires_grad_(True)
        ref = O.ref_retrieval_embed(hh, z["mask"], normalize)
        (ref * z["upstream"]).sum().backward()
        ref_e, ref_dh = ref.detach(), hh.grad
        tol = BF16_GRAD_NORM_RTOL
    assert e.dtype == torch.float32
    assert_grad_close(e.detach(), ref_e, GRAD_NORM_RTOL, "emb")
    assert h.grad.dtype == dtype
    assert_grad_close(h.grad, ref_dh, tol, "dh")


def test_pieces_get_nll_and_marginalize(dev):
    from dalm_amd.training.utils import train_utils as tu

    z = load_npz("pieces")
    lp = z["lp"].float().to(dev)
    torch.testing.assert_close(tu.get_nll(lp, z["labels"].to(dev)).cpu(), z["nll"].float(), rtol=0, atol=0)
    doc = torch.tensor([-1.25], device=dev)
    for ql in (1, 2, 4, 7, 8, 12):
        got = tu.marginalize_log_probs(lp[0], doc, torch.tensor(ql))
        torch.testing.assert_close(got.cpu(), z[f"marg_q{ql}"].float(), rtol=0, atol=1e-6)
        # the reference's loop hands over elements of a DEVICE tensor: same result, length never leaves the GPU,
        # and the doc-term gradient (sum of the upstream gradient over the rows that received it) matches autograd
        a, d = lp[0].clone().requires_grad_(True), doc.clone().requires_grad_(True)
        out = tu.marginalize_log_probs(a, d, torch.tensor([ql, 99], device=dev)[0])
        torch.testing.assert_close(out.detach().cpu(), z[f"marg_q{ql}"].float(), rtol=0, atol=1e-6)
        up = torch.randn_like(out)
        (out * up).sum().backward()
        a2, d2 = lp[0].clone().requires_grad_(True), doc.clone().requires_grad_(True)
        ref = torch.cat([a2[: ql - 1], a2[ql - 1:] + d2], 0)
        (ref * up).sum().backward()
        torch.testing.assert_close(a.grad, a2.grad)
        torch.testing.assert_close(d.grad, d2.grad, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------
# realistic shapes vs the closed-form oracle on the same seeded inputs
# ---------------------------------------------------------------------------
def _oracle_ce(logits, ids, mask, qlen, doc_lp):
    x = logits.double()
    q = torch.zeros(x.shape[0], 4, dtype=torch.float64)
    f = O.closed_forward(q, q, x, ids, mask, qlen, 1.0)
    M, Nb = f["M"], f["Nb"]
    m = mask[:, 1:].double()
    lse = torch.logsumexp(x[:, :-1], 2)
    xy = torch.gather(x[:, :-1], 2, ids[:, 1:].unsqueeze(2)).squeeze(2)
    gen = ((m * (lse - xy)).sum() - (Nb * doc_lp.double()).sum()) / M
    b = O.closed_backward(q, q, x, ids, mask, qlen, 1.0, f)
    return gen, b["dlogits"], f


CE_SHAPES = [
    # B, Tg, V, dtype, view
    (3, 40, 32000, torch.float32, "contig"),     # Llama-2 vocab: 512-thread register-resident rows
    (3, 40, 32000, torch.bfloat16, "contig"),
    (2, 24, 65024, torch.float32, "contig"),     # Falcon vocab fp32: 1024-thread rows
    (2, 24, 65024, torch.bfloat16, "contig"),    # Falcon vocab bf16 (cfg5)
    (2, 12, 131077, torch.float32, "contig"),    # streaming fallback, odd V (unaligned rows)
    (2, 12, 30522, torch.float32, "contig"),     # V % 4 != 0: unaligned row starts
    (2, 12, 30523, torch.bfloat16, "contig"),
    (3, 20, 4999, torch.float32, "vslice"),      # strided view (stride_t > V), unaligned base
    (3, 20, 5000, torch.bfloat16, "vslice"),
    (4, 16, 200, torch.float32, "contig"),
]


@pytest.mark.parametrize("B,Tg,V,dtype,view", CE_SHAPES)
def test_marg_ce_kernels_vs_oracle(dev, B, Tg, V, dtype, view):
    from dalm_amd.ops import default_ops

    ops = default_ops()
    _, _, logits, ids, mask, qlen = synth_batch(1234 + V, B, 8, Tg, V + (3 if view == "vslice" else 0), dtype=dtype,
                                                logit_gain=3.0, pad_side="left" if V % 2 else "right")
    if view == "vslice":
        ids = ids % V
    lg_dev_full = logits.to(dev)
    lg_dev = lg_dev_full[:, :, 1:V + 1] if view == "vslice" else lg_dev_full
    lg_cpu = (logits[:, :, 1:V + 1] if view == "vslice" else logits).float()
    doc_lp = -torch.rand(B)
    ref_gen, ref_dl, f = _oracle_ce(lg_cpu, ids, mask, qlen, doc_lp)

    stats, Nb, Mb = ops.ce_prep(mask.to(dev), qlen.to(dev))
    assert float(stats[0]) == float(f["M"])
    torch.testing.assert_close(Nb.cpu().double(), f["Nb"], rtol=0, atol=0)
    # fused forward + gradient
    row_lse, row_nll, dl = ops.ce_fwd(lg_dev, ids.to(dev), mask.to(dev), stats, True)
    gen = ops.ce_finalize(row_nll, Nb, doc_lp.to(dev), stats)
    assert_loss_close(gen, ref_gen)
    tol = GRAD_NORM_RTOL if dtype == torch.float32 else BF16_GRAD_NORM_RTOL
    assert dl.dtype == dtype
    assert_grad_close(dl, ref_dl, tol, "dlogits(fused)")
    # row_lse for unmasked rows
    got_lse = row_lse.reshape(B, Tg)[:, :-1].cpu().double()
    keep = mask[:, 1:] != 0
    torch.testing.assert_close(got_lse[keep], f["row_lse"][keep], rtol=1e-6, atol=1e-5)
    # forward-only + separate backward agree with the fused pass
    row_lse2, row_nll2, none = ops.ce_fwd(lg_dev, ids.to(dev), mask.to(dev), stats, False)
    assert none is None
    # (different kernels may serve the two modes - e.g. bf16 forward-only streams - so compare to rounding)
    torch.testing.assert_close(row_nll2, row_nll, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(row_lse2, row_lse, rtol=1e-5, atol=1e-5)
    g = torch.tensor([0.37], device=dev)
    dl2 = ops.ce_bwd(lg_dev, ids.to(dev), mask.to(dev), stats, row_lse, g)
    assert_grad_close(dl2, 0.37 * ref_dl, tol, "dlogits(bwd)")
    if view == "vslice":  # the padding columns of the parent buffer must be untouched
        assert torch.equal(lg_dev_full[:, :, 0], logits[:, :, 0].to(dev))
        assert torch.equal(lg_dev_full[:, :, V + 1:], logits[:, :, V + 1:].to(dev))


def test_marg_ce_inplace_gradient(dev):
    from dalm_amd.fused import rag_e2e_loss

    q, p, logits, ids, mask, qlen = synth_batch(5, 4, 64, 32, 32000)
    outs = []
    for inplace in (False, True):
        lg = logits.clone().to(dev).requires_grad_(True)
        lg_work = lg * 1.0  # non-leaf buffer that may be overwritten
        qq, pp = q.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
        loss = rag_e2e_loss(qq, pp, lg_work, ids.to(dev), mask.to(dev), qlen.to(dev), 100, inplace_grad=inplace)
        loss.backward()
        outs.append((loss.detach().clone(), lg.grad.clone(), qq.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])


def test_upstream_gradient_scaling(dev):
    from dalm_amd.fused import rag_e2e_loss

    q, p, logits, ids, mask, qlen = synth_batch(6, 5, 48, 20, 1000)
    grads = []
    for w in (1.0, 2.5):
        lg = logits.to(dev).requires_grad_(True)
        qq, pp = q.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
        (rag_e2e_loss(qq, pp, lg, ids.to(dev), mask.to(dev), qlen.to(dev), 100) * w).backward()
        grads.append((lg.grad, qq.grad, pp.grad))
    for a, b in zip(*grads):
        torch.testing.assert_close(b, a * 2.5, rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------------------
# similarity / GEMM on the f32 matrix cores
# ---------------------------------------------------------------------------
GEMM_CASES = [
    # M, N, K, transA, transB
    (18, 18, 1024, False, True),      # cfg3 S
    (150, 150, 1024, False, True),    # cfg2 S
    (19, 19, 384, False, True),       # cfg1 S (toy csv: 19 rows, bge-small)
    (150, 1024, 150, False, False),   # dQ = dS . P
    (150, 1024, 150, True, False),    # dP = dS^T . Q
    (37, 53, 29, False, False),       # ragged everything (scalar-load path)
    (37, 53, 29, True, True),
    (1536, 1536, 512, False, True),   # 128x128 tiles
    (1300, 1100, 260, True, False),   # 128x128 tiles, ragged edges
]


@pytest.mark.parametrize("M,N,K,ta,tb", GEMM_CASES)
def test_gemm_f32_mfma_vs_fp64(dev, M, N, K, ta, tb):
    from dalm_amd.ops import default_ops

    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = 0.5 * ((A.double().t() if ta else A.double()) @ (Bm.double().t() if tb else Bm.double()))
    got = default_ops().gemm(A.to(dev), Bm.to(dev), 0.5, ta, tb)
    # exact-f32 MFMA == an fmaf chain: error ~1e-7 * sum|a b| (asymmetric operands catch transposes)
    assert_grad_close(got, ref, 2e-6, "gemm")


@pytest.mark.parametrize("m,n,D,off", [(18, 18, 1024, 0), (150, 150, 1024, 0), (19, 19, 384, 0),
                                         (18, 144, 1024, 36), (150, 1200, 1024, 450), (1536, 1536, 256, 0),
                                         (700, 2304, 128, 1000)])
def test_sim_rowstats_and_grad_vs_fp64(dev, m, n, D, off):
    from dalm_amd.ops import default_ops

    ops = default_ops()
    g = torch.Generator().manual_seed(m + n + D)
    A = torch.nn.functional.normalize(torch.randn(m, D, generator=g), dim=1)
    Bm = torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=1)
    scale = 100.0
    S = scale * (A.double() @ Bm.double().t())
    idx = torch.arange(m)
    row_lse, diag = ops.sim_rowstats(A.to(dev), Bm.to(dev), scale, off)
    torch.testing.assert_close(row_lse.cpu().double(), torch.logsumexp(S, 1), rtol=1e-6, atol=2e-5)
    torch.testing.assert_close(diag.cpu().double(), S[idx, off + idx], rtol=1e-6, atol=2e-5)
    rc, cc = torch.rand(m, generator=g) / m, torch.rand(n, generator=g) / n
    cl = torch.logsumexp(S, 0) + 0.3
    rl = torch.logsumexp(S, 1)
    dS = rc.double().unsqueeze(1) * torch.exp(S - rl.unsqueeze(1)) + cc.double().unsqueeze(0) * torch.exp(S - cl.unsqueeze(0))
    dS[idx, off + idx] -= rc.double() + cc.double()[off + idx]
    ref = scale * (dS @ Bm.double())
    got = ops.sim_grad(A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    assert_grad_close(got, ref, 2e-4, "dA")


def test_bitwise_determinism(dev):
    """No float atomics anywhere: two runs are bitwise identical (doubles as a race check)."""
    from dalm_amd.fused import rag_e2e_loss

    q, p, logits, ids, mask, qlen = synth_batch(9, 18, 1024, 64, 32000)
    res = []
    for _ in range(2):
        lg = logits.to(dev).requires_grad_(True)
        qq, pp = q.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
        loss = rag_e2e_loss(qq, pp, lg, ids.to(dev), mask.to(dev), qlen.to(dev), 100)
        loss.backward()
        res.append((loss.detach().clone(), lg.grad.clone(), qq.grad.clone(), pp.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ---------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties (the fp64 oracle is too slow here)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("V,dtype", [(32000, torch.float32), (65024, torch.bfloat16)])
def test_full_size_properties_cfg3_cfg5(dev, V, dtype):
    from dalm_amd.fused import rag_e2e_loss

    B, Tg, D = 18, 256, 1024
    g = torch.Generator(device="cpu").manual_seed(3)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(dev).requires_grad_(True)
    p = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(dev).requires_grad_(True)
    logits = torch.randn(B, Tg, V, device=dev, dtype=dtype).requires_grad_(True)
    ids = torch.randint(0, V, (B, Tg), device=dev)
    lens = torch.randint(60, Tg + 1, (B,), device=dev)
    mask = (torch.arange(Tg, device=dev).unsqueeze(0) < lens.unsqueeze(1)).long()
    qlen = (lens.float() * 0.8).long()
    aux = {}
    loss = rag_e2e_loss(q, p, logits, ids, mask, qlen, 100, aux=aux)
    loss.backward()
    assert torch.isfinite(loss)
    dl = logits.grad.float()
    m = mask[:, 1:].bool()
    # softmax - onehot sums to zero over the vocabulary on every live row; dead rows and the last slot are 0
    rows = dl[:, :-1][m]
    tol = 1e-6 if dtype == torch.float32 else 2e-4
    assert float(rows.sum(1).abs().max()) <= tol
    assert float(dl[:, :-1][~m].abs().max()) == 0.0 and float(dl[:, -1].abs().max()) == 0.0
    # the label entry is the only negative-mass entry: sum of positive part == |label grad| == m/M (1 - p_y)
    M = float(aux["num_target_tokens"])
    assert M == float(mask[:, 1:].sum())
    lab = torch.gather(dl[:, :-1], 2, ids[:, 1:].unsqueeze(2)).squeeze(2)[m]
    assert float(lab.max()) <= 0.0 and float(lab.min()) >= -1.0 / M * (1 + 1e-2)
    # dS rows/cols: contrastive part sums to 0 along both axes => dq . 1-direction checks are covered by goldens;
    # here: gradient of a normalised-embedding loss is finite and non-trivial
    assert torch.isfinite(q.grad).all() and torch.isfinite(p.grad).all() and float(q.grad.abs().max()) > 0
    # loss is shift-invariant in the logits (log-softmax): add a per-row constant
    shift = (torch.randn(B, Tg, 1, device=dev) * 2).to(dtype)
    if dtype == torch.float32:
        loss2 = rag_e2e_loss(q.detach(), p.detach(), (logits.detach() + shift), ids, mask, qlen, 100)
        assert abs(float(loss2) - float(loss)) <= 2e-5 * abs(float(loss))


@pytest.mark.parametrize("dtype,train_head", [(torch.float32, False), (torch.float32, True), (torch.bfloat16, False)])
def test_lm_head_fused_loss_equals_materialised_logits(dev, dtype, train_head):
    """chunked lm_head + CE == lm_head then rag_e2e_loss (value and all gradients)."""
    from dalm_amd.fused import rag_e2e_loss, rag_e2e_loss_from_hidden

    B, Tg, H, V, D = 7, 24, 64, 1000, 32
    q, p, _, ids, mask, qlen = synth_batch(77, B, D, Tg, V, pad_side="left")
    g = torch.Generator().manual_seed(5)
    hidden = torch.randn(B, Tg, H, generator=g).to(dtype)
    W = (0.2 * torch.randn(V, H, generator=g)).to(dtype)
    res = {}
    for mode in ("ref", "fused"):
        qq, pp = q.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
        hh = hidden.to(dev).requires_grad_(True)
        ww = W.to(dev).requires_grad_(train_head)
        if mode == "ref":
            loss = rag_e2e_loss(qq, pp, hh @ ww.t(), ids.to(dev), mask.to(dev), qlen.to(dev), 100)
        else:
            loss = rag_e2e_loss_from_hidden(qq, pp, hh, ww, ids.to(dev), mask.to(dev), qlen.to(dev), 100, chunk_samples=3)
        (loss * 1.5).backward()
        res[mode] = (loss.detach(), qq.grad, pp.grad, hh.grad, ww.grad if train_head else None)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert abs(float(res["ref"][0]) - float(res["fused"][0])) <= 1e-5 * abs(float(res["ref"][0])) + 1e-6
    for a, b, name in zip(res["ref"][1:], res["fused"][1:], ("dq", "dp", "dhidden", "dW")):
        if a is None:
            continue
        assert norm_rel_err(b, a) <= tol, (name, norm_rel_err(b, a))


def test_exact_topk_matches_fp64_argsort(dev):
    """exact inner-product top-k on the similarity kernel (block merge included)."""
    from dalm_amd.retrieval import construct_search_index, exact_topk, get_nearest_neighbours

    g = torch.Generator().manual_seed(11)
    corpus = torch.nn.functional.normalize(torch.randn(5000, 96, generator=g), dim=1)
    queries = torch.nn.functional.normalize(corpus[:37] + 0.03 * torch.randn(37, 96, generator=g), dim=1)
    ref = queries.double() @ corpus.double().t()
    rs, ri = torch.topk(ref, 10, dim=1)
    s, i = exact_topk(queries.to(dev), corpus.to(dev), 10, block=1024)   # 5 blocks -> merges exercised
    assert torch.equal(i.cpu(), ri)
    torch.testing.assert_close(s.cpu().double(), rs, rtol=1e-5, atol=1e-6)
    idx = construct_search_index(96, 5000, corpus.to(dev))
    hits = get_nearest_neighbours(5, idx, queries.to(dev), {j: f"doc{j}" for j in range(5000)}, threshold=0.5)
    assert all(h and h[0][0] == f"doc{j}" for j, h in enumerate(hits))   # each query's source passage ranks first
