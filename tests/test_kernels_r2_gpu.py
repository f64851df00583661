"""GPU parity for the round-2 kernels, all through the C ABI, all against the fp64 oracle on the same seeded inputs:
small-batch contrastive path (S once: partial tiles -> statistics -> one-launch backward), flash-style similarity
backward (no dS panel), fused pool + L2-norm forward, one-launch loss assembly, and the FULL-size cfg3 / cfg5
configurations compared with the oracle itself (not just properties).

Tolerances as in test_hip_parity.py: fp32 loss rel <= 1e-4, gradient norm-rel <= 1e-4 (north-star 1e-3);
bf16 gradient outputs norm-rel <= 4e-3.
"""
import os

import pytest
import torch

import dalm_oracle as O
from helpers import norm_rel_err, synth_batch
from test_hip_parity import BF16_GRAD_NORM_RTOL, GRAD_NORM_RTOL, assert_grad_close, assert_loss_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from dalm_amd import hip

    hip.load()
    return torch.device("cuda:0")


def _problem(m, n, D, off, seed=None):
    g = torch.Generator().manual_seed(seed if seed is not None else m * 7 + n * 3 + D)
    A = torch.nn.functional.normalize(torch.randn(m, D, generator=g), dim=1)
    Bm = torch.nn.functional.normalize(torch.randn(n, D, generator=g), dim=1)
    # positives stand out (cos ~ 0.55 -> S_ii ~ 55 against ~N(0, 3) negatives), as after some training; the peaked
    # softmax makes the diagonal term (p_ii - 1) a cancellation, which is where fp32 (ours AND the reference's) loses
    # digits against this fp64 oracle - hence 5e-4 on the gradients below instead of the usual 1e-4
    Bm[off:off + m] = torch.nn.functional.normalize(A + 1.5 * Bm[off:off + m], dim=1)
    scale = 100.0
    S = scale * (A.double() @ Bm.double().t())
    rc, cc = torch.rand(m, generator=g) / m, torch.rand(n, generator=g) / n
    rl, cl = torch.logsumexp(S, 1), torch.logsumexp(S, 0) + 0.3   # any col_lse >= true one is a valid softmax scale
    idx = torch.arange(m)
    dS = rc.double().unsqueeze(1) * torch.exp(S - rl.unsqueeze(1)) + cc.double().unsqueeze(0) * torch.exp(S - cl.unsqueeze(0))
    dS[idx, off + idx] -= rc.double() + cc.double()[off + idx]
    return A, Bm, scale, S, rc, rl, cc, cl, dS


SMALL = [(18, 18, 1024, 0), (150, 150, 1024, 0), (19, 19, 384, 0), (1, 1, 64, 0), (18, 144, 1024, 36),
         (150, 1200, 1024, 450), (257, 300, 100, 20), (33, 65, 1030, 7), (600, 600, 768, 0), (1024, 1024, 32, 0)]


@pytest.mark.parametrize("m,n,D,off", SMALL + [(150, 1200, 1024, 0), (512, 512, 1024, 0), (70, 2100, 256, 1000), (1000, 1000, 64, 0)])
def test_small_forward_in_one_launch_equals_the_two_launch_form(dev, m, n, D, off):
    """dalm_sim_small_fwd1 (round 4): the statistics are finished inside the partial-tile kernel by the last-arriving
    workgroups.  S and diag: the SAME bits as the two-launch form (same slice order); row / column log-sum-exp: a different
    but fixed merge tree - equal to fp64 to the same bound and to the two-launch values to 2e-6; 30 repetitions bit-identical
    (the result must not depend on which workgroup arrives last); rows-only form; tickets left at zero."""
    from dalm_amd.ops import HipOps, default_ops

    ops = default_ops()
    A, Bm, scale, S, *_ = _problem(m, n, D, off)
    Ad, Bd = A.to(dev), Bm.to(dev)
    S2, r2, d2, c2 = ops.sim_small_fwd(Ad, Bd, scale, off, True, one_launch=False)
    S1, r1, d1, c1 = ops.sim_small_fwd(Ad, Bd, scale, off, True, one_launch=True)
    assert torch.equal(S1, S2) and torch.equal(d1, d2)
    torch.testing.assert_close(r1, r2, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(c1, c2, rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(r1.cpu().double(), torch.logsumexp(S, 1), rtol=1e-6, atol=3e-5)
    torch.testing.assert_close(c1.cpu().double(), torch.logsumexp(S, 0), rtol=1e-6, atol=3e-5)
    for _ in range(30):
        Sx, rx, dx, cx = ops.sim_small_fwd(Ad, Bd, scale, off, True, one_launch=True)
        assert torch.equal(Sx, S1) and torch.equal(rx, r1) and torch.equal(cx, c1) and torch.equal(dx, d1)
    Sr, rr, dr, none = ops.sim_small_fwd(Ad, Bd, scale, off, False, one_launch=True)
    assert none is None and torch.equal(Sr, S1) and torch.equal(rr, r1) and torch.equal(dr, d1)
    assert HipOps._tickets and all(int(buf.abs().sum()) == 0 for buf in HipOps._tickets.values())   # one buffer per (device, stream)


def test_small_forward_in_one_launch_under_uneven_load(dev):
    """The hand-off must not depend on timing: the same problem while other kernels keep the chip busy on a second stream
    (arrival orders change from run to run) - 40 repetitions, every word of every output identical."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    A, Bm, scale, S, *_ = _problem(150, 1200, 1024, 450)
    Ad, Bd = A.to(dev), Bm.to(dev)
    ref = ops.sim_small_fwd(Ad, Bd, scale, 450, False, one_launch=True)
    noise = torch.randn(4096, 4096, device=dev)
    side = torch.cuda.Stream()
    for i in range(40):
        with torch.cuda.stream(side):
            for _ in range(1 + i % 3):
                noise = torch.tanh(noise @ noise[:, : 4096] * 1e-3)
        got = ops.sim_small_fwd(Ad, Bd, scale, 450, False, one_launch=True)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
    torch.cuda.synchronize()
    torch.testing.assert_close(ref[1].cpu().double(), torch.logsumexp(S, 1), rtol=1e-6, atol=3e-5)


@pytest.mark.parametrize("m,n,D,off,want_a", [(150, 1200, 1024, 450, True), (150, 1200, 1024, 450, False), (18, 8000, 1024, 100, True),
                                               (70, 2100, 256, 1000, True), (33, 900, 1030, 7, True), (600, 600, 768, 0, True)])
def test_small_backward_sliced_in_one_launch_equals_the_two_launch_form(dev, m, n, D, off, want_a):
    """dalm_sim_small_bwd1 (round 4): the slices of a long contraction are added by the last slice of every output tile to
    arrive, in slice order - the SAME bits as small_slice_sum_kernel gives the two-launch form, 30 repetitions identical
    (also with another stream keeping the chip busy), tickets left at zero; unsliced shapes are untouched."""
    from dalm_amd.ops import HipOps, default_ops

    ops = default_ops()
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, D, off)
    Sg, *_ = ops.sim_small_fwd(A.to(dev), Bm.to(dev), scale, off, False)
    args = (Sg, A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    two = ops.sim_small_bwd(*args, want_a, not want_a, one_launch=False)[0 if want_a else 1]
    one = ops.sim_small_bwd(*args, want_a, not want_a, one_launch=True)[0 if want_a else 1]
    assert torch.equal(one, two)
    ref = scale * (dS @ Bm.double()) if want_a else scale * (dS.t() @ A.double())
    assert_grad_close(one, ref, 5e-4, "dA" if want_a else "dB")
    noise = torch.randn(2048, 2048, device=dev)
    side = torch.cuda.Stream()
    for i in range(30):
        if i % 2:
            with torch.cuda.stream(side):
                noise = torch.tanh(noise @ noise * 1e-3)
        again = ops.sim_small_bwd(*args, want_a, not want_a, one_launch=True)[0 if want_a else 1]
        assert torch.equal(again, one)
    torch.cuda.synchronize()
    assert HipOps._tickets and all(int(buf.abs().sum()) == 0 for buf in HipOps._tickets.values())   # one buffer per (device, stream)


@pytest.mark.parametrize("m,n,D,off", SMALL)
def test_small_path_vs_fp64(dev, m, n, D, off):
    from dalm_amd.ops import default_ops

    ops = default_ops()
    assert ops.sim_small_supported(m, n, D)
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, D, off)
    Sg, row_lse, diag, col_lse = ops.sim_small_fwd(A.to(dev), Bm.to(dev), scale, off, True)
    idx = torch.arange(m)
    torch.testing.assert_close(Sg.cpu().double(), S, rtol=1e-6, atol=3e-5)
    torch.testing.assert_close(row_lse.cpu().double(), rl, rtol=1e-6, atol=3e-5)
    torch.testing.assert_close(col_lse.cpu().double(), torch.logsumexp(S, 0), rtol=1e-6, atol=3e-5)
    torch.testing.assert_close(diag.cpu().double(), S[idx, off + idx], rtol=1e-6, atol=3e-5)
    # rows only (the W > 1 form): same row statistics, no column output
    S2, row2, diag2, none = ops.sim_small_fwd(A.to(dev), Bm.to(dev), scale, off, False)
    assert none is None and torch.equal(row2, row_lse) and torch.equal(diag2, diag) and torch.equal(S2, Sg)
    args = (Sg, A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    dA, dB = ops.sim_small_bwd(*args, True, True)
    assert_grad_close(dA, scale * (dS @ Bm.double()), 5e-4, "dA")
    assert_grad_close(dB, scale * (dS.t() @ A.double()), 5e-4, "dB")
    dA1, n1 = ops.sim_small_bwd(*args, True, False)
    n2, dB1 = ops.sim_small_bwd(*args, False, True)
    assert n1 is None and n2 is None
    from dalm_amd import hip

    for one, both, want_a in ((dA1, dA, 1), (dB1, dB, 0)):
        unsliced = hip.load().dalm_sim_small_bwd_workspace_bytes(m, n, D, want_a, 1 - want_a) == 0
        # the kernel's form (all fragments up front / rounds of 8 steps) follows the shortest contraction of the LAUNCH:
        # dir 0 contracts over n, dir 1 over m
        same_form = (min(m, n) > 64) == ((n if want_a else m) > 64)
        if unsliced and same_form:
            assert torch.equal(one, both)                 # same kernel, same order: same bits
        else:                                             # slices of a long contraction / the other form: fixed order each
            assert_grad_close(one, both.cpu().double(), 1e-5, "one direction vs both")
    again, _ = ops.sim_small_bwd(*args, True, False)
    assert torch.equal(again, dA1)                        # deterministic either way


def test_small_path_limits_and_errors(dev):
    from dalm_amd import hip
    from dalm_amd.ops import default_ops

    ops = default_ops()
    assert not ops.sim_small_supported(1025, 1025, 64) and not ops.sim_small_supported(2048, 2048, 1024)
    assert ops.sim_small_supported(128, 8192, 1024) and not ops.sim_small_supported(129, 8192, 1024)
    A = torch.randn(4, 8, device=dev)
    S = torch.empty(4, 4, device=dev)
    v = torch.empty(4, device=dev)
    with pytest.raises(ValueError, match="workspace too small"):
        hip.call("dalm_sim_small_fwd", hip.ptr(A), hip.ptr(A), 4, 4, 8, 1.0, 0, hip.ptr(S), 4, hip.ptr(v), hip.ptr(v),
                 hip.ptr(v), hip.ptr(S), 4, hip.stream())
    with pytest.raises(ValueError, match="diag_offset"):
        ops.sim_small_fwd(A, A[:2], 1.0, 0, True)


FLASH = [(1200, 1200, 1024, 0), (150, 1200, 1024, 450), (4096, 4096, 1024, 0), (700, 2304, 128, 1000),
         (1536, 1536, 256, 0), (333, 5000, 384, 77), (2100, 2100, 768, 0), (40, 130, 512, 3), (19, 19, 384, 0),
         (1, 1, 128, 0), (33, 257, 640, 100), (129, 129, 896, 0)]


@pytest.mark.parametrize("m,n,D,off", FLASH)
def test_flash_grad_vs_fp64(dev, m, n, D, off):
    """dalm_sim_grad without a dS panel: same closed form, every split / tail geometry (row tails, column tails,
    several column splits, one split, NT = 1..8)."""
    from dalm_amd import hip
    from dalm_amd.ops import default_ops

    ops = default_ops()
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, D, off)
    ws = hip.load().dalm_sim_grad_workspace_bytes(m, n, D)
    assert ws <= (m + n + 160 + 1024 * 32) * D * 4, ws   # operand copies + <= ~1k row blocks of partial outputs, never m*n
    got = ops.sim_grad(A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    assert_grad_close(got, scale * (dS @ Bm.double()), 5e-4, "dA")
    got2 = ops.sim_grad(A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    assert torch.equal(got, got2)   # deterministic


def test_flash_workspace_is_not_m_times_n():
    """k-major operand copies ((m' + n') * D floats - the size of the inputs) + per-split partial outputs (bounded by
    ~512 row blocks' worth); nothing proportional to m * n."""
    from dalm_amd import hip

    lib = hip.load()
    copies = lambda m, n, D: (-(-m // 32) * 32 + -(-n // 128) * 128) * D * 4
    assert lib.dalm_sim_grad_workspace_bytes(65536, 65536, 1024) == copies(65536, 65536, 1024)   # 2048 row blocks: no split
    assert lib.dalm_sim_grad_workspace_bytes(16384, 16384, 1024) == copies(16384, 16384, 1024)
    assert lib.dalm_sim_grad_workspace_bytes(4096, 4096, 1024) <= copies(4096, 4096, 1024) + 8 * 4096 * 1024 * 4
    assert lib.dalm_sim_grad_workspace_bytes(4096, 4096, 1000) > 4096 * 4096 * 4 - 1   # odd D keeps the panel form


POOL = [(18, 50, 1024), (18, 128, 1024), (150, 128, 1024), (19, 160, 384), (3, 17, 100), (5, 9, 5000),
        (4, 4096, 1024), (2, 1, 64), (130, 33, 96),
        # round 4: batches >= 512 take the 256-thread form of the forward (TPR 128 / 64 / 256 threads per row, two d-chunks)
        (600, 37, 1024), (512, 9, 384), (515, 5, 2056), (700, 3, 3000),
        # ... and their backward slices the tokens (B * d-chunks > 1024 workgroups)
        (1200, 60, 1024), (1100, 7, 96), (530, 21, 1504), (1030, 3, 4104), (2100, 2, 8200)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,T,D", POOL)
@pytest.mark.parametrize("normalize", [True, False])
def test_fused_pool_vs_oracle(dev, B, T, D, dtype, normalize):
    """One-launch pool + norm (and its token-sliced two-launch form at (4,4096,1024)): ragged masks incl. an
    all-padding sample, left and right padding, integer mask weights, odd / unaligned D."""
    from dalm_amd.fused import pool_l2norm

    g = torch.Generator().manual_seed(B * 1000 + T + D)
    h = torch.randn(B, T, D, generator=g).to(dtype)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    ar = torch.arange(T).unsqueeze(0)
    mask = (ar < lens.unsqueeze(1)).long()
    if B > 1:
        mask[1] = (ar[0] >= (T - lens[1])).long()      # left padding
    if B > 2:
        mask[2] = 0                                      # all padding -> u = 0, |u| = 0 (clamp branches)
    if B > 3:
        mask[3, : int(lens[3])] = 2                      # integer weights other than 0/1
    up = torch.randn(B, D, generator=g)
    hd = h.to(dev).requires_grad_(True)
    e = pool_l2norm(hd, mask.to(dev), normalize)
    (e * up.to(dev)).sum().backward()
    hh = h.double().requires_grad_(True)
    ref = O.ref_retrieval_embed(hh, mask, normalize)
    (ref * up).sum().backward()
    assert_grad_close(e.detach(), ref.detach(), GRAD_NORM_RTOL, "emb")
    assert_grad_close(hd.grad, hh.grad, GRAD_NORM_RTOL if dtype == torch.float32 else BF16_GRAD_NORM_RTOL, "dh")


def test_rag_loss_finalize_vs_oracle(dev):
    from dalm_amd.ops import default_ops

    ops, oo = default_ops(), O.OracleOps()
    g = torch.Generator().manual_seed(4)
    n_local, R = 37, 37 * 64
    row_nll = torch.rand(R, generator=g) * 9
    Nb = torch.randint(0, 50, (n_local,), generator=g).float()
    lse_r, lse_c = torch.randn(n_local, generator=g) + 5, torch.randn(n_local, generator=g) + 5
    diag = torch.randn(n_local, generator=g)
    stats = torch.tensor([1234.0, float(n_local)])
    out, doc = ops.rag_loss_finalize(*(t.to(dev) for t in (row_nll, Nb, lse_r, lse_c, diag)), 74, stats.to(dev))
    ref, rdoc = oo.rag_loss_finalize(row_nll, Nb, lse_r, lse_c, diag, 74, stats)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-6, atol=1e-6)
    torch.testing.assert_close(doc.cpu(), rdoc, rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------------------
# FULL BASELINE sizes against the oracle ("property-tested, not oracle-compared")
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,V,dtype,inplace", [("cfg3", 32000, torch.float32, False), ("cfg5", 65024, torch.bfloat16, False),
                                                   ("cfg3-inplace", 32000, torch.float32, True),
                                                   ("cfg5-inplace", 65024, torch.bfloat16, True)])
def test_full_size_vs_oracle(dev, name, V, dtype, inplace):
    """cfg3: 18 x 256 x 32000 fp32 logits; cfg5: 18 x 256 x 65024 bf16 logits (the 1024-thread CE rows), D = 1024,
    left padding, ragged lengths: loss, dq, dp and EVERY element of dlogits vs closed_chunked (fp64, sample by sample)."""
    from dalm_amd.fused import rag_e2e_loss

    B, Tg, D = 18, 256, 1024
    q, p, logits, ids, mask, qlen = synth_batch(31, B, D, Tg, V, pad_side="left", dtype=dtype, logit_gain=2.0)
    qd, pd = q.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
    ld = logits.to(dev).requires_grad_(True)
    work = ld * 1.0 if inplace else ld          # a non-leaf buffer the kernel may overwrite with its own gradient
    aux = {}
    loss = rag_e2e_loss(qd, pd, work, ids.to(dev), mask.to(dev), qlen.to(dev), 100, inplace_grad=inplace, aux=aux)
    loss.backward()
    ref = O.closed_chunked(q, p, logits, ids, mask, qlen, 100.0, dlogits_got=ld.grad.cpu())
    assert_loss_close(loss, ref["loss"])
    assert_loss_close(aux["contrastive"], ref["contrastive"])
    assert_loss_close(aux["generator"], ref["generator"])
    assert float(aux["num_target_tokens"]) == float(ref["M"])
    assert_grad_close(qd.grad, ref["dq"], name="dq")
    assert_grad_close(pd.grad, ref["dp"], name="dp")
    tol = GRAD_NORM_RTOL if dtype == torch.float32 else BF16_GRAD_NORM_RTOL
    assert ref["dlogits_err"] <= tol, ref["dlogits_err"]
    assert ref["dlogits_max_err"] <= 10 * tol, ref["dlogits_max_err"]


def test_stream_ce_kernel_inplace_large_vocab(dev):
    """V > 65536 takes marg_ce_stream_kernel; with the gradient written over the logits the label logit
    must be read before any store of the row.  131077-wide rows, in place, vs the oracle; repeated to catch the race."""
    from dalm_amd.fused import rag_e2e_loss

    B, Tg, V, D = 3, 24, 131077, 64
    q, p, logits, ids, mask, qlen = synth_batch(8, B, D, Tg, V, pad_side="right", logit_gain=3.0)
    for _ in range(3):
        ld = logits.to(dev).requires_grad_(True)
        loss = rag_e2e_loss(q.to(dev), p.to(dev), ld * 1.0, ids.to(dev), mask.to(dev), qlen.to(dev), 100, inplace_grad=True)
        loss.backward()
        ref = O.closed_chunked(q, p, logits, ids, mask, qlen, 100.0, dlogits_got=ld.grad.cpu())
        assert_loss_close(loss, ref["loss"])
        assert ref["dlogits_err"] <= GRAD_NORM_RTOL


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lm_head_fused_vs_fp64_oracle(dev, dtype):
    """rag_e2e_loss_from_hidden (logits never materialised) against the fp64 oracle itself - round 1 only compared
    it with the materialising HIP path."""
    from dalm_amd.fused import rag_e2e_loss_from_hidden

    B, Tg, H, V, D = 7, 24, 64, 1000, 32
    q, p, _, ids, mask, qlen = synth_batch(77, B, D, Tg, V, pad_side="left")
    g = torch.Generator().manual_seed(5)
    hidden = torch.randn(B, Tg, H, generator=g).to(dtype)
    W = (0.2 * torch.randn(V, H, generator=g)).to(dtype)
    qq, pp = q.to(dev).requires_grad_(True), p.to(dev).requires_grad_(True)
    hh, ww = hidden.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    loss = rag_e2e_loss_from_hidden(qq, pp, hh, ww, ids.to(dev), mask.to(dev), qlen.to(dev), 100, chunk_samples=3)
    loss.backward()
    h64, w64 = hidden.double().requires_grad_(True), W.double().requires_grad_(True)
    q64, p64 = q.double().requires_grad_(True), p.double().requires_grad_(True)
    logits64 = h64 @ w64.t()
    if dtype == torch.bfloat16:   # the product path rounds the logits to bf16 once (the lm_head's output dtype)
        logits64 = logits64 + (logits64.detach().to(torch.bfloat16).double() - logits64.detach())
    ref = O.ref_step_loss(q64, p64, logits64, ids, mask, qlen, 100)
    ref["loss"].backward()
    assert_loss_close(loss, ref["loss"], 1e-4 if dtype == torch.float32 else 2e-3)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    for got, want, name in ((qq.grad, q64.grad, "dq"), (pp.grad, p64.grad, "dp"), (hh.grad, h64.grad, "dh"),
                            (ww.grad, w64.grad, "dW")):
        assert norm_rel_err(got, want) <= tol, (name, norm_rel_err(got, want))


STREAM = [(1200, 1200, 1024, 0), (600, 900, 100, 17), (4096, 4096, 384, 0), (4101, 4300, 64, 100), (513, 513, 72, 0),
          (150, 4000, 1024, 1000), (1199, 1203, 200, 3), (2048, 2048, 128, 0), (1500, 1530, 72, 5)]


@pytest.mark.parametrize("m,n,D,off", STREAM)
def test_streaming_rowstats_vs_fp64(dev, m, n, D, off):
    """dalm_sim_rowstats in its LDS-free streaming form (m*n >= 512^2): any D (zero-padded k-major copies), row and
    column tails, several column splits, both row-tile variants (two row tiles per wave from 4096 rows), the 2- and 4-way
    K-sliced workgroups of mid-size problems (1200^2, 1199 x 1203: 2 slices; 1500 x 1530: 4), sharded diag offsets.  atol 2e-4 on |S| ~ 55: the f32 dot products of 1024 terms carry ~1e-4 absolute error there (3e-6 relative)."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    A, Bm, scale, S, *_ = _problem(m, n, D, off)
    row_lse, diag = ops.sim_rowstats(A.to(dev), Bm.to(dev), scale, off)
    idx = torch.arange(m)
    torch.testing.assert_close(row_lse.cpu().double(), torch.logsumexp(S, 1), rtol=1e-6, atol=2e-4)
    torch.testing.assert_close(diag.cpu().double(), S[idx, off + idx], rtol=1e-6, atol=2e-4)
    r2, d2 = ops.sim_rowstats(A.to(dev), Bm.to(dev), scale, off)
    assert torch.equal(r2, row_lse) and torch.equal(d2, diag)


@pytest.mark.parametrize("nq,nc,D,k", [(37, 5000, 96, 10), (300, 70000, 384, 10), (5, 40, 64, 7), (129, 3001, 1024, 1),
                                        (64, 2000, 100, 100)])
def test_fused_topk_vs_fp64(dev, nq, nc, D, k):
    """dalm_sim_topk: exact top-k without the score matrix == fp64 brute force (indices identical, values to f32)."""
    from dalm_amd.ops import default_ops
    from dalm_amd.retrieval import exact_topk

    g = torch.Generator().manual_seed(nq + nc)
    corpus = torch.nn.functional.normalize(torch.randn(nc, D, generator=g), dim=1)
    queries = torch.nn.functional.normalize(corpus[:nq] + 0.05 * torch.randn(nq, D, generator=g), dim=1)
    ref = queries.double() @ corpus.double().t()
    rs, ri = torch.topk(ref, k, dim=1)
    val, idx, ovf = default_ops().sim_topk(queries.to(dev), corpus.to(dev), k)
    if k * 32 > nc and nc > 8 * k + 64:
        # fewer 32-column groups than k: no useful threshold exists, every score is a candidate and the buffer
        # overflows - flagged, and exact_topk (below) takes the materialising route for that block
        assert int(ovf) != 0
    else:
        assert int(ovf) == 0
        # identical unless two scores are closer than f32 resolution: compare through the fp64 scores of the picks
        picked = torch.gather(ref, 1, idx.cpu())
        torch.testing.assert_close(picked, rs, rtol=0, atol=2e-6)
        torch.testing.assert_close(val.cpu().double(), rs, rtol=1e-5, atol=2e-6)
        assert (idx.cpu()[:, 0] == ri[:, 0]).float().mean() > 0.99
    s2, i2 = exact_topk(queries.to(dev), corpus.to(dev), k, block=1500)      # several blocks -> merges exercised
    torch.testing.assert_close(torch.gather(ref, 1, i2.cpu()), rs, rtol=0, atol=2e-6)


@pytest.mark.parametrize("nq,nc,D,k,scale", [(300, 70000, 384, 10, 1.0), (129, 3001, 1024, 1, 1.0), (5, 40, 64, 7, 1.0),
                                              (260, 20000, 256, 20, 100.0), (64, 6000, 256, 8, 1.0)])
def test_fused_topk_with_the_bf16x3_first_pass(dev, monkeypatch, nq, nc, D, k, scale):
    """Round 4: the first pass of dalm_sim_topk (one maximum per row and 32 corpus columns) on the bf16 matrix cores
    (DALM_TOPK_BF16X3=1 forces it at these small sizes; it is automatic from 1024 tiles of 256 x 256).  The candidates are
    still recomputed in f32, so the result is the exact top-k: picks identical to fp64 brute force, values to f32 - for
    unit-norm embeddings, for scale 100, and for UN-normalised embeddings whose scores sit near zero (the threshold's slack
    has to cover the first pass's absolute error there)."""
    from dalm_amd.ops import default_ops

    monkeypatch.setenv("DALM_TOPK_BF16X3", "1")
    g = torch.Generator().manual_seed(nq + nc)
    if nc == 6000:       # scores near zero, large norms
        corpus = torch.randn(nc, D, generator=g)
        queries = torch.randn(nq, D, generator=g) * 0.05
    else:
        corpus = torch.nn.functional.normalize(torch.randn(nc, D, generator=g), dim=1)
        queries = torch.nn.functional.normalize(corpus[:nq] + 0.05 * torch.randn(nq, D, generator=g), dim=1)
    ref = scale * (queries.double() @ corpus.double().t())
    rs, ri = torch.topk(ref, k, dim=1)
    val, idx, ovf = default_ops().sim_topk(queries.to(dev), corpus.to(dev), k, scale)
    assert int(ovf) == 0
    picked = torch.gather(ref, 1, idx.cpu())
    tol = 2e-6 * max(1.0, float(ref.abs().max()))
    torch.testing.assert_close(picked, rs, rtol=0, atol=tol)
    torch.testing.assert_close(val.cpu().double(), rs, rtol=1e-5, atol=tol)
    assert (idx.cpu()[:, 0] == ri[:, 0]).float().mean() > 0.99
    monkeypatch.setenv("DALM_TOPK_BF16X3", "0")          # and the f32 first pass picks the same passages
    v0, i0, o0 = default_ops().sim_topk(queries.to(dev), corpus.to(dev), k, scale)
    assert int(o0) == 0 and torch.equal(i0, idx)
    torch.testing.assert_close(v0, val, rtol=0, atol=0)   # values come from the same f32 refine pass either way


def test_fused_topk_ties_overflow_falls_back(dev):
    from dalm_amd.ops import default_ops
    from dalm_amd.retrieval import exact_topk

    corpus = torch.ones(4000, 64) / 8.0          # every score identical: every row overflows the candidate buffer
    q = torch.ones(3, 64) / 8.0
    val, idx, ovf = default_ops().sim_topk(q.to(dev), corpus.to(dev), 5)
    assert int(ovf) != 0
    s, i = exact_topk(q.to(dev), corpus.to(dev), 5)
    torch.testing.assert_close(s.cpu(), torch.ones(3, 5), rtol=1e-6, atol=1e-6)


def test_fused_topk_scores_near_zero_and_lds_gate(dev):
    """(a) scores near 0 - a purely relative threshold slack vanishes there; the threshold now carries an absolute
    term and a row with fewer than k surviving candidates reports overflow instead of emitting (-inf, -1) slots;
    (b) (D, k) beyond the refine kernel's LDS budget is refused BEFORE anything is enqueued and `exact_topk` falls back."""
    from dalm_amd import hip
    from dalm_amd.ops import default_ops
    from dalm_amd.retrieval import exact_topk

    ops = default_ops()
    g = torch.Generator().manual_seed(11)
    nq, nc, D, k = 64, 6000, 256, 8
    corpus = torch.randn(nc, D, generator=g)
    corpus = corpus - corpus.mean(0, keepdim=True)
    queries = torch.randn(nq, D, generator=g)
    queries = queries - (queries @ corpus.t()).mean(1, keepdim=True) * 0      # scores are sums of +- products around 0
    corpus, queries = 1e-3 * torch.nn.functional.normalize(corpus, dim=1), torch.nn.functional.normalize(queries, dim=1)
    ref = queries.double() @ corpus.double().t()
    rs, _ = torch.topk(ref, k, dim=1)
    val, idx, ovf = ops.sim_topk(queries.to(dev), corpus.to(dev), k)
    if int(ovf) == 0:
        assert int((idx < 0).sum()) == 0 and bool(torch.isfinite(val).all())
        torch.testing.assert_close(torch.gather(ref, 1, idx.cpu()), rs, rtol=0, atol=1e-9)
    s2, i2 = exact_topk(queries.to(dev), corpus.to(dev), k)
    assert int((i2 < 0).sum()) == 0
    torch.testing.assert_close(torch.gather(ref, 1, i2.cpu()), rs, rtol=0, atol=1e-9)
    # (b) D = 1024: k = 700 needs 1024 + 3 * 5664 floats of LDS > 15360
    assert ops.sim_topk_supported(1024, 10) and ops.sim_topk_supported(384, 500)
    assert not ops.sim_topk_supported(1024, 700) and not ops.sim_topk_supported(64, 1025)
    big_q = torch.nn.functional.normalize(torch.randn(4, 1024, generator=g), dim=1)
    big_c = torch.nn.functional.normalize(torch.randn(3000, 1024, generator=g), dim=1)
    with pytest.raises(ValueError, match="LDS"):
        ops.sim_topk(big_q.to(dev), big_c.to(dev), 700)
    torch.cuda.synchronize()
    s3, i3 = exact_topk(big_q.to(dev), big_c.to(dev), 700)                     # materialising route, no error
    r3, _ = torch.topk(big_q.double() @ big_c.double().t(), 700, dim=1)
    torch.testing.assert_close(torch.gather(big_q.double() @ big_c.double().t(), 1, i3.cpu()), r3, rtol=0, atol=2e-6)


def test_recall_hit_rate_on_synthetic_corpus(dev):
    """The eval quality metrics of the reference (recall / precision / hit-rate, dalm/eval/utils.py:225-272) on a
    synthetic corpus: queries are noisy copies of their gold passages; vs a brute-force fp64 evaluation."""
    from dalm_amd.retrieval import evaluate_retrieval

    g = torch.Generator().manual_seed(3)
    nc, nq, D, k = 20000, 500, 384, 10
    corpus = torch.nn.functional.normalize(torch.randn(nc, D, generator=g), dim=1)
    gold = torch.randperm(nc, generator=g)[:nq]
    noise = torch.linspace(0.05, 3.0, nq).unsqueeze(1)          # from trivially easy to hopeless
    queries = torch.nn.functional.normalize(corpus[gold] + noise * torch.randn(nq, D, generator=g) / D ** 0.5 * 4, dim=1)
    res = evaluate_retrieval(queries.to(dev), corpus.to(dev), gold, top_k=k)
    ref_idx = torch.topk(queries.double() @ corpus.double().t(), k, dim=1).indices
    hit = (ref_idx == gold.unsqueeze(1)).any(1).double()
    assert abs(res["hit_rate"] - float(hit.mean())) <= 2.0 / nq
    assert abs(res["recall"] - float(hit.mean())) <= 2.0 / nq           # one gold passage: recall == hit
    assert abs(res["precision"] - float(hit.mean()) / k) <= 2.0 / nq / k
    assert 0.2 < res["hit_rate"] < 1.0


def test_native_rccl_communicator_one_rank(dev, tmp_path, monkeypatch):
    """dalm_comm_*: the library's own RCCL binding (dlopen'd at run time).  One rank on this box: unique id, init, an
    all-gather and a SUM all-reduce on the communicator's side stream with event ordering against torch's stream, and
    the sharded loss through it == the single-process loss.  (>= 2 GPUs: tests/test_sharded_two_ranks_gpu.py.)"""
    from dalm_amd.comm import NativeRcclComm
    from dalm_amd.fused import rag_e2e_loss

    monkeypatch.setenv("DALM_COMM_ID_FILE", str(tmp_path / "id"))
    comm = NativeRcclComm(rank=0, world_size=1, device=0)
    try:
        x = torch.randn(7, 33, device=dev)
        y = comm.all_gather_rows(x * 2.0)              # producer kernel on torch's stream, gather on the comm stream
        z = y + 1.0                                    # consumer on torch's stream again
        torch.testing.assert_close(z, x * 2.0 + 1.0)
        g = torch.arange(1000, device=dev, dtype=torch.float32)
        comm.all_reduce_sum_(g)
        torch.testing.assert_close(g, torch.arange(1000, device=dev, dtype=torch.float32))
        q, p, logits, ids, mask, qlen = synth_batch(2, 6, 64, 16, 500)
        args = [t.to(dev) for t in (q, p, logits, ids, mask, qlen)]
        a = [args[0].clone().requires_grad_(True), args[1].clone().requires_grad_(True), args[2].clone().requires_grad_(True)]
        b = [args[0].clone().requires_grad_(True), args[1].clone().requires_grad_(True), args[2].clone().requires_grad_(True)]
        la = rag_e2e_loss(a[0], a[1], a[2], *args[3:], 100)
        lb = rag_e2e_loss(b[0], b[1], b[2], *args[3:], 100, comm=comm)     # the W > 1 code path (two row problems, stats exchange)
        la.backward(); lb.backward()
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la))
        for u, v in zip(a, b):
            assert norm_rel_err(v.grad, u.grad) <= 1e-5
        with pytest.raises(TypeError):
            comm.all_reduce_sum_(torch.zeros(3, device=dev, dtype=torch.bfloat16))
    finally:
        comm.close()


# ---------------------------------------------------------------------------
# seeded shape fuzz: every geometry decision of the round-2 kernels (tile tails, split counts, K padding, row-tile
# variants, token slicing) is driven by the shapes, so a fixed pseudo-random sample of shapes runs through each family
# ---------------------------------------------------------------------------
def _fuzz_shapes(seed, count, gen):
    import os
    import random

    # DALM_FUZZ_SEED / DALM_FUZZ_SCALE: other samples / more of them for an occasional soak run (defaults: the fixed sample)
    rnd = random.Random(seed + 1000 * int(os.environ.get("DALM_FUZZ_SEED", "0")))
    return [gen(rnd) for _ in range(count * int(os.environ.get("DALM_FUZZ_SCALE", "1")))]


SMALL_FUZZ = _fuzz_shapes(11, 14, lambda r: (lambda m, n: (m, n, r.choice([32, 64, 100, 384, 520, 1024]), r.randint(0, n - m)))(
    *sorted((r.randint(1, 300), r.randint(1, 900)))))
BIG_FUZZ = _fuzz_shapes(12, 8, lambda r: (lambda m, n: (m, n, r.choice([128, 256, 384, 640, 1024]), r.randint(0, n - m)))(
    *sorted((r.randint(520, 1800), r.randint(700, 3000)))))


@pytest.mark.parametrize("m,n,D,off", SMALL_FUZZ)
def test_fuzz_small_path(dev, m, n, D, off):
    from dalm_amd.ops import default_ops

    ops = default_ops()
    if not ops.sim_small_supported(m, n, D):
        pytest.skip("outside the small path")
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, D, off, seed=m * 131 + n)
    Sg, row_lse, diag, col_lse = ops.sim_small_fwd(A.to(dev), Bm.to(dev), scale, off, True)
    torch.testing.assert_close(row_lse.cpu().double(), rl, rtol=1e-6, atol=2e-4)
    torch.testing.assert_close(col_lse.cpu().double(), torch.logsumexp(S, 0), rtol=1e-6, atol=2e-4)
    dA, dB = ops.sim_small_bwd(Sg, A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev),
                               cl.float().to(dev), True, True)
    assert_grad_close(dA, scale * (dS @ Bm.double()), 5e-4, "dA")
    assert_grad_close(dB, scale * (dS.t() @ A.double()), 5e-4, "dB")


@pytest.mark.parametrize("m,n,D,off", BIG_FUZZ)
def test_fuzz_streaming_forward_and_flash_backward(dev, m, n, D, off):
    from dalm_amd.ops import default_ops

    ops = default_ops()
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, D, off, seed=m * 17 + n)
    row_lse, diag = ops.sim_rowstats(A.to(dev), Bm.to(dev), scale, off)
    torch.testing.assert_close(row_lse.cpu().double(), rl, rtol=1e-6, atol=2e-4)
    torch.testing.assert_close(diag.cpu().double(), S[torch.arange(m), off + torch.arange(m)], rtol=1e-6, atol=2e-4)
    got = ops.sim_grad(A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    assert_grad_close(got, scale * (dS @ Bm.double()), 5e-4, "dA")


POOL_FUZZ = _fuzz_shapes(13, 10, lambda r: (r.randint(1, 40), r.randint(1, 300), r.choice([64, 96, 384, 768, 1024, 1500, 4100])))


@pytest.mark.parametrize("B,T,D", POOL_FUZZ)
def test_fuzz_fused_pool(dev, B, T, D):
    from dalm_amd.fused import pool_l2norm

    g = torch.Generator().manual_seed(B * 7919 + T * 31 + D)
    for dtype in (torch.float32, torch.bfloat16):
        h = torch.randn(B, T, D, generator=g).to(dtype)
        lens = torch.randint(0, T + 1, (B,), generator=g)           # zero-length (all padding) rows included
        mask = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).long()
        hd = h.to(dev).requires_grad_(True)
        e = pool_l2norm(hd, mask.to(dev), True)
        up = torch.randn(B, D, generator=g)
        (e * up.to(dev)).sum().backward()
        hh = h.double().requires_grad_(True)
        ref = O.ref_retrieval_embed(hh, mask, True)
        (ref * up).sum().backward()
        assert_grad_close(e.detach(), ref.detach(), GRAD_NORM_RTOL, "emb")
        assert_grad_close(hd.grad, hh.grad, GRAD_NORM_RTOL if dtype == torch.float32 else BF16_GRAD_NORM_RTOL, "dh")


TOPK_FUZZ = _fuzz_shapes(14, 6, lambda r: (r.randint(1, 200), r.randint(400, 9000), r.choice([64, 100, 384, 1024]), r.randint(1, 12)))


@pytest.mark.parametrize("nq,nc,D,k", TOPK_FUZZ)
def test_fuzz_fused_topk(dev, nq, nc, D, k):
    from dalm_amd.retrieval import exact_topk

    g = torch.Generator().manual_seed(nq * 101 + nc)
    corpus = torch.nn.functional.normalize(torch.randn(nc, D, generator=g), dim=1)
    queries = torch.nn.functional.normalize(torch.randn(nq, D, generator=g), dim=1)
    ref = queries.double() @ corpus.double().t()
    rs, _ = torch.topk(ref, k, dim=1)
    s, i = exact_topk(queries.to(dev), corpus.to(dev), k)
    torch.testing.assert_close(torch.gather(ref, 1, i.cpu()), rs, rtol=0, atol=2e-6)
    assert bool((i.cpu()[:, 1:] != i.cpu()[:, :-1]).all()) if k > 1 else True     # no duplicates in a row


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_lm_head_over_live_rows_equals_all_rows(dtype):
    """Row compaction of the fused lm_head path (several row chunks, a ragged last chunk, -1 padding entries): the loss and
    every gradient equal the all-rows path - dead rows only ever contributed zeros."""
    from dalm_amd.fused import live_row_index, rag_e2e_loss_from_hidden

    dev = torch.device("cuda:0")
    B, Tg, H, V, D = 8, 128, 64, 1000, 32
    g = torch.Generator().manual_seed(3)
    glen = torch.randint(20, Tg + 1, (B, 1), generator=g)
    glen[2] = Tg                                       # one sample without padding
    mask = (torch.arange(Tg).unsqueeze(0) >= (Tg - glen)).long()
    ids = torch.randint(0, V, (B, Tg), generator=g)
    qlen = (glen.squeeze(1).float() * 0.7).long().clamp(min=1)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(dev)
    p = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(dev)
    h = torch.randn(B, Tg, H, generator=g).to(dev, dtype)
    W = (0.2 * torch.randn(V, H, generator=g)).to(dev, dtype)
    live = live_row_index(mask, multiple=32)
    assert live is not None and int((live < 0).sum()) > 0 and live.numel() > 512   # >= 3 chunks of <= 256 rows
    res = []
    for rows in (None, live.to(dev)):
        qq, pp, hh, ww = [t.clone().requires_grad_(True) for t in (q, p, h, W)]
        loss = rag_e2e_loss_from_hidden(qq, pp, hh, ww, ids.to(dev), mask.to(dev), qlen.to(dev), 100.0, chunk_samples=2,
                                        live_rows=rows)
        loss.backward()
        res.append([loss.detach()] + [t.grad.float() for t in (qq, pp, hh, ww)])
    tol = 1e-5 if dtype == torch.float32 else 2e-2   # bf16: the dW accumulation order changes with the chunking
    for a, b in zip(*res):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(a.abs().max())), (a - b).abs().max()
    dead = (torch.cat((mask[:, 1:], torch.zeros(B, 1, dtype=mask.dtype)), 1) == 0).to(dev)
    assert float(res[1][3][dead].abs().max()) == 0.0   # d(hidden) of rows without loss is exactly zero
    import os

    with torch.no_grad():   # evaluation: forward-only CE kernels, no d(hidden) GEMM, same value
        # "1": through the bf16 MFMA kernel (logits never stored); f32 inputs keep the library path; None: the default
        # (round 4: the kernel whenever the lm_head weight fits the Infinity Cache - it does here)
        for kern in ("0", "1", None):
            os.environ.pop("DALM_LM_HEAD_KERNEL", None)
            if kern is not None:
                os.environ["DALM_LM_HEAD_KERNEL"] = kern
            try:
                for rows in (None, live.to(dev)):
                    ev = rag_e2e_loss_from_hidden(q, p, h, W, ids.to(dev), mask.to(dev), qlen.to(dev), 100.0, chunk_samples=2,
                                                  live_rows=rows)
                    assert abs(float(ev) - float(res[0][0])) <= (1e-5 if dtype == torch.float32 else 2e-3) * abs(float(res[0][0]))
            finally:
                os.environ.pop("DALM_LM_HEAD_KERNEL", None)


@pytest.mark.parametrize("R,V,K", [(128, 256, 64), (200, 1000, 128), (1, 130, 64), (333, 4099, 512), (640, 32000, 1024),
                                   (1100, 32000, 256)])   # the last one takes the 256-row tiles
def test_lm_head_lse_kernel_matches_the_product_in_fp64(dev, R, V, K):
    """dalm_lm_head_lse_fwd (bf16 MFMA, logits never stored): log-sum-exp and label NLL of hidden @ W^T against the fp64
    product of the SAME bf16 values - ragged row / vocabulary tails, rows without loss, a label in every 64-column strip."""
    from dalm_amd.ops import default_ops

    g = torch.Generator().manual_seed(R * 7 + V)
    h = (0.5 * torch.randn(R, K, generator=g)).to(torch.bfloat16)
    W = (0.5 * torch.randn(V, K, generator=g)).to(torch.bfloat16)
    labels = torch.randint(0, V, (R,), generator=g)
    labels[::5] = -1
    if R > 3:
        labels[1], labels[2], labels[3] = 0, V - 1, min(V - 1, 64)
    lse, nll = default_ops().lm_head_lse(h.to(dev), W.to(dev), labels.to(dev))
    x = h.double() @ W.double().t()
    ref_lse = torch.logsumexp(x, dim=1)
    ref_nll = torch.where(labels >= 0, ref_lse - x.gather(1, labels.clamp_min(0).unsqueeze(1)).squeeze(1), torch.zeros(()).double())
    # f32 accumulation of exact bf16 products: 1e-5 relative on |x| <= ~K/4
    tol = 2e-5 * max(1.0, float(x.abs().max()))
    assert float((lse.cpu().double() - ref_lse).abs().max()) <= tol
    assert float((nll.cpu().double() - ref_nll).abs().max()) <= tol


def test_marg_ce_finalize_with_a_context_extent(dev):
    """`dalm_marg_ce_finalize_topk` - k = 1 (the reference's case) is `dalm_marg_ce_finalize` bit for bit;
    k = 3 equals the fp64 statement -log sum_c p(c|q) p(y|c) per answer token (oracle closed_gen_loss_topk, pinned to the
    reference at k = 1 by tests/test_oracle_golden.py), and the returned per-row weights are its gradient."""
    import dalm_oracle as O

    from dalm_amd.ops import default_ops

    ops = default_ops()
    g = torch.Generator().manual_seed(21)
    B, k, Tg = 5, 3, 40
    T = Tg - 1
    label_lp = -torch.rand(B, k, T, generator=g, dtype=torch.float64) * 6.0
    n_ans = torch.randint(2, 9, (B,), generator=g)
    cut = torch.randint(5, 20, (B, k), generator=g)
    mask = torch.zeros(B, k, T, dtype=torch.int64)
    for b in range(B):
        for c in range(k):
            start = int(torch.randint(0, 4, (1,), generator=g))        # left padding of this sequence
            mask[b, c, start:int(cut[b, c]) + int(n_ans[b])] = 1
    doc_lp = torch.log_softmax(torch.randn(B, k, generator=g, dtype=torch.float64), dim=1)
    ref = O.closed_gen_loss_topk(label_lp.clone().requires_grad_(True), mask, cut, doc_lp)
    lp_req = label_lp.clone().requires_grad_(True)
    ref = O.closed_gen_loss_topk(lp_req, mask, cut, doc_lp)
    ref["generator"].backward()
    # what the CE kernel leaves: row_nll[b,c,t] = m (lse - x_y) = -m lp, one trailing row (t = Tg-1) that is always 0
    row_nll = torch.zeros(B, k, Tg)
    row_nll[:, :, :T] = (-label_lp * mask).float()
    stats = torch.tensor([float(mask.sum()) / k, float(B), 0.0, 0.0])
    out, w = ops.ce_finalize_topk(row_nll.to(dev), cut.to(dev), n_ans.float().to(dev), doc_lp.float().to(dev), stats.to(dev),
                                  want_weights=True)
    assert abs(float(out) - float(ref["generator"])) <= 2e-6 * abs(float(ref["generator"]))
    # weights = -dL/d(label log-prob) on live rows
    want_w = (-lp_req.grad * mask)
    torch.testing.assert_close(w.cpu()[:, :, :T].double() * mask, want_w, rtol=2e-5, atol=1e-8)
    # k = 1: the very same bits as the k-less entry point
    r1 = row_nll[:, :1].contiguous().to(dev)
    s1 = torch.tensor([float(mask[:, 0].sum()), float(B), 0.0, 0.0]).to(dev)
    a, _ = ops.ce_finalize_topk(r1, cut[:, :1].contiguous().to(dev), n_ans.float().to(dev), doc_lp[:, :1].float().contiguous().to(dev), s1)
    b_ = ops.ce_finalize(r1.reshape(-1), n_ans.float().to(dev), doc_lp[:, 0].float().contiguous().to(dev), s1)
    assert torch.equal(a, b_)
    ref1 = O.closed_gen_loss_topk(label_lp[:, :1], mask[:, :1], cut[:, :1], doc_lp[:, :1])
    assert abs(float(a) - float(ref1["generator"])) <= 2e-6 * abs(float(ref1["generator"]))


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the similarity row statistics on the bf16 matrix cores at f32 accuracy (three bf16 thirds per operand)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,D,off", [(256, 256, 64, 0), (300, 517, 128, 100), (37, 1000, 384, 500), (1024, 1536, 1024, 256),
                                       (4096, 4096, 1024, 0), (513, 4352, 1024, 1111)])
def test_bf16x3_rowstats_vs_fp64(dev, m, n, D, off):
    """dalm_sim_rowstats_bf16x3 against fp64 on the `_problem` generator (positives at S ~ 55, negatives ~ N(0, 3)): S
    itself (through diag) within 2e-6 relative of |S| - VERDICT r3 item 4's bound, 1.1e-4 absolute at 55; the f32 MFMA
    kernel is allowed 2e-4 there - and the row log-sum-exp to the same absolute bound; ragged row / column tails, all six
    K segments at D = 64 (one K tile per segment) ... 1024 (16 per segment), sharded diagonal offsets; deterministic."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    A, Bm, scale, S, *_ = _problem(m, n, D, off)
    row_lse, diag = ops.sim_rowstats_bf16x3(A.to(dev), Bm.to(dev), scale, off)
    idx = torch.arange(m)
    ref_d, ref_l = S[idx, off + idx], torch.logsumexp(S, 1)
    err_d = (diag.cpu().double() - ref_d).abs().max().item()
    err_l = (row_lse.cpu().double() - ref_l).abs().max().item()
    assert err_d <= 2e-6 * ref_d.abs().max().item() + 2e-5, (err_d, ref_d.abs().max().item())
    assert err_l <= 2e-6 * ref_l.abs().max().item() + 2e-5, (err_l, ref_l.abs().max().item())
    r2, d2 = ops.sim_rowstats_bf16x3(A.to(dev), Bm.to(dev), scale, off)
    assert torch.equal(r2, row_lse) and torch.equal(d2, diag)
    # no worse than the exact-f32 MFMA kernel on the same inputs (where that kernel takes the shape)
    f_l, f_d = ops.sim_rowstats_f32(A.to(dev), Bm.to(dev), scale, off)
    err_f = (f_d.cpu().double() - ref_d).abs().max().item()
    assert err_d <= max(2.0 * err_f, 3e-5), (err_d, err_f)


def test_bf16x3_scores_off_the_diagonal_vs_fp64(dev):
    """Every element of S, not only the diagonal: with diag_offset = c the 'label' of row i is column c + i, so sweeping c
    reads a full wrapped diagonal per call - 40 random offsets of a 300 x 340 problem cover ~12 000 distinct scores,
    including scores near zero (relative error there is bounded by the ABSOLUTE error, |scale| * 2e-6)."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    m, n, D = 300, 340, 256
    A, Bm, scale, S, *_ = _problem(m, n, D, 0)
    idx = torch.arange(m)
    worst = 0.0
    for c in range(0, 41):
        _, diag = ops.sim_rowstats_bf16x3(A.to(dev), Bm.to(dev), scale, c)
        worst = max(worst, (diag.cpu().double() - S[idx, c + idx]).abs().max().item())
    assert worst <= scale * 2e-6, worst


def test_large_rowstats_route_to_bf16x3_and_keep_the_loss(dev):
    """dalm_sim_rowstats sends m, n >= 4096 to the bf16x3 form: same statistics as the explicit entry point (bit for bit),
    and the contrastive loss built from them agrees with fp64.  Embeddings WITHOUT stand-out positives (an untrained
    retriever: loss ~ log n, every score matters) and the `_problem` ones (loss ~ 0: the terms lse - diag cancel to ~1e-7,
    so that case is held to the ABSOLUTE error of a score, 2e-6 |S|)."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    m = n = 4096
    idx = torch.arange(m)
    for kind in ("untrained", "trained"):
        if kind == "trained":
            A, Bm, scale, S, *_ = _problem(m, n, 1024, 0, seed=5)
        else:
            g = torch.Generator().manual_seed(6)
            A = torch.nn.functional.normalize(torch.randn(m, 1024, generator=g), dim=1)
            Bm = torch.nn.functional.normalize(torch.randn(n, 1024, generator=g), dim=1)
            scale = 100.0
            S = scale * (A.double() @ Bm.double().t())
        r1, d1 = ops.sim_rowstats(A.to(dev), Bm.to(dev), scale, 0)
        r2, d2 = ops.sim_rowstats_bf16x3(A.to(dev), Bm.to(dev), scale, 0)
        assert torch.equal(r1, r2) and torch.equal(d1, d2)
        c1, _ = ops.sim_rowstats(Bm.to(dev), A.to(dev), scale, 0)
        ref = 0.5 * ((torch.logsumexp(S, 1) - S[idx, idx]).mean() + (torch.logsumexp(S, 0) - S[idx, idx]).mean())
        got = 0.5 * ((r1.double().cpu() - d1.double().cpu()).mean() + (c1.double().cpu() - d1.double().cpu()).mean())
        tol = 1e-6 * abs(float(ref)) + 2e-6 * float(S.abs().max()) * (1.0 if kind == "trained" else 0.0)
        assert abs(float(got) - float(ref)) <= tol + 1e-7, (kind, float(got), float(ref))


# ---------------------------------------------------------------------------------------------------------------------
# round 5: the similarity BACKWARD on the bf16 matrix cores (dalm_sim_grad_bf16x3)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,D,off", [(256, 256, 64, 0), (300, 517, 128, 100), (37, 1000, 384, 500), (1024, 1536, 1024, 256),
                                       (4096, 4096, 1024, 0), (513, 4352, 1024, 1111), (700, 5000, 256, 2049)])
def test_bf16x3_sim_grad_vs_fp64(dev, m, n, D, off):
    """dA = scale * dS . B on the `_problem` generator (peaked softmax: the diagonal term is a cancellation) against fp64:
    within 2e-6 of the largest |dA| entry-wise, no worse than the f32-pipe flash kernel; ragged rows and
    columns, several 2048-column blocks (accumulation across blocks), sharded diagonal offsets; deterministic."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, D, off)
    ref = scale * (dS @ Bm.double())
    args = (A.to(dev), Bm.to(dev), scale, off, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    got = ops.sim_grad_bf16x3(*args)
    # the row / column statistics enter in f32 (rl, cl rounded): the same inputs the f32 kernel gets - compare both to fp64
    err = float((got.cpu().double() - ref).abs().max())
    big = float(ref.abs().max())
    monkey = os.environ.get("DALM_SIM_GRAD_X3")
    os.environ["DALM_SIM_GRAD_X3"] = "0"
    try:
        f32 = ops.sim_grad(*args)
    finally:
        if monkey is None:
            os.environ.pop("DALM_SIM_GRAD_X3", None)
        else:
            os.environ["DALM_SIM_GRAD_X3"] = monkey
    err_f = float((f32.cpu().double() - ref).abs().max())
    assert err <= max(2.0 * err_f, 2e-6 * big), (err, err_f, big)
    assert_grad_close(got, ref, 5e-4, "dA")
    assert torch.equal(ops.sim_grad_bf16x3(*args), got)


def test_large_sim_grad_routes_to_bf16x3(dev):
    from dalm_amd.ops import default_ops

    ops = default_ops()
    m = n = 8192
    A, Bm, scale, S, rc, rl, cc, cl, dS = _problem(m, n, 1024, 0, seed=9)
    args = (A.to(dev), Bm.to(dev), scale, 0, rc.to(dev), rl.float().to(dev), cc.to(dev), cl.float().to(dev))
    assert torch.equal(ops.sim_grad(*args), ops.sim_grad_bf16x3(*args))
    assert_grad_close(ops.sim_grad(*args), scale * (dS @ Bm.double()), 5e-4, "dA")
