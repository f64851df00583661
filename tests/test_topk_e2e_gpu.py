"""k retrieved contexts per sample, END TO END: `dalm_amd.fused.rag_e2e_loss_topk` - document posteriors
over the k contexts (dalm_doc_scores_topk_fwd), the CE forward over the B*k sequences, the RAG-token reduction with its
per-row weights (dalm_marg_ce_finalize_topk), the WEIGHTED CE backward (dalm_marg_ce_bwd_weighted) and the closed-form
backward for the query and the contexts (dalm_doc_scores_topk_bwd) - against fp64 AUTOGRAD through the probability form
(oracle.ref_rag_topk_loss -> closed_gen_loss_topk: sum_c p(c|q) p(y|c), not a logsumexp restatement of the kernels).

The reference is k = 1 (train_utils.py:123-124; TODO at train_rage2e.py:461-462); at k = 1 closed_gen_loss_topk is pinned
to the reference goldens by tests/test_oracle_golden.py, and nothing on the k = 1 product path changed."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, k, Tg, V, D, seed, left_pad=False):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(B, D, generator=g, dtype=torch.float64), dim=1)
    P = torch.nn.functional.normalize(torch.randn(B, k, D, generator=g, dtype=torch.float64) + 0.5 * q.unsqueeze(1), dim=2)
    logits = 2.0 * torch.randn(B, k, Tg, V, generator=g, dtype=torch.float64)
    ids = torch.randint(0, V, (B, k, Tg), generator=g)
    n_ans = torch.randint(2, max(3, Tg // 4), (B,), generator=g)
    mask = torch.zeros(B, k, Tg, dtype=torch.int64)
    qlen = torch.zeros(B, k, dtype=torch.int64)
    for b in range(B):
        for c in range(k):
            plen = int(torch.randint(3, Tg - int(n_ans[b]) - 1, (1,), generator=g))   # prompt tokens of sequence (b,c)
            if left_pad:        # the sequence ends in the last column; qlen counts from the start of the row (absolute index)
                total = plen + int(n_ans[b])
                mask[b, c, Tg - total:] = 1
                qlen[b, c] = Tg - int(n_ans[b])
            else:
                mask[b, c, :plen + int(n_ans[b])] = 1
                qlen[b, c] = plen
    return q, P, logits, ids, mask, qlen


@pytest.mark.parametrize("B,k,Tg,V,D,left_pad", [(4, 3, 24, 200, 64, False), (3, 3, 40, 1003, 128, True), (2, 5, 32, 32000, 1024, False),
                                                  (5, 1, 24, 200, 64, False)])
def test_rag_topk_loss_and_every_gradient_vs_fp64_autograd(B, k, Tg, V, D, left_pad):
    import dalm_oracle as O

    from dalm_amd.fused import rag_e2e_loss_topk

    dev = torch.device("cuda:0")
    q, P, logits, ids, mask, qlen = _case(B, k, Tg, V, D, seed=B * 100 + k)
    scale = 20.0
    q64, P64, l64 = (t.clone().requires_grad_(True) for t in (q, P, logits))
    ref = O.ref_rag_topk_loss(q64, P64, l64, ids, mask, qlen, scale)
    ref.backward()
    qd, Pd, ld = (t.float().to(dev).requires_grad_(True) for t in (q, P, logits))
    aux = {}
    loss = rag_e2e_loss_topk(qd, Pd, ld, ids.to(dev), mask.to(dev), qlen.to(dev), scale, aux=aux)
    (3.0 * loss).backward()                       # an upstream gradient != 1 goes through gscale
    assert abs(float(loss) - float(ref)) <= 2e-5 * abs(float(ref)), (float(loss), float(ref))

    def nrel(got, want):
        want = want.double()
        return float((got.double().cpu() - want).norm() / (want.norm() + 1e-30))

    assert nrel(ld.grad, 3.0 * l64.grad) <= 1e-4
    assert nrel(qd.grad, 3.0 * q64.grad) <= 2e-4
    assert nrel(Pd.grad, 3.0 * P64.grad) <= 2e-4
    # rows without loss get exact zeros; softmax - onehot rows sum to ~0
    dead = (mask[:, :, 1:] == 0)
    assert float(ld.grad[:, :, :-1][dead.to(dev)].abs().max()) == 0.0 and float(ld.grad[:, :, -1].abs().max()) == 0.0
    torch.testing.assert_close(aux["doc_logprobs"].cpu().double(),
                               torch.log_softmax(scale * torch.einsum("bd,bkd->bk", q, P), dim=1), rtol=1e-5, atol=1e-5)


def test_rag_topk_loss_bf16_logits():
    """bf16 logits (what the generator emits under autocast): both sides see the same rounded values; the gradient comes
    back in bf16 (2^-9 per element)."""
    import dalm_oracle as O

    from dalm_amd.fused import rag_e2e_loss_topk

    dev = torch.device("cuda:0")
    q, P, logits, ids, mask, qlen = _case(3, 3, 32, 5000, 256, seed=9)
    lb = logits.to(torch.bfloat16)
    l64 = lb.double().clone().requires_grad_(True)
    ref = O.ref_rag_topk_loss(q, P, l64, ids, mask, qlen, 20.0)
    ref.backward()
    ld = lb.to(dev).requires_grad_(True)
    loss = rag_e2e_loss_topk(q.float().to(dev), P.float().to(dev), ld, ids.to(dev), mask.to(dev), qlen.to(dev), 20.0)
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 2e-5 * abs(float(ref))
    assert float((ld.grad.double().cpu() - l64.grad).norm() / l64.grad.norm()) <= 4e-3


def test_finalize_topk_weights_are_deterministic_across_waves():
    """the prompt-row loop and the answer-row loop of ce_finalize_topk_kernel write the same weight entries from
    DIFFERENT waves when cut + j and j land in different 64-thread groups (Tg = 700, cut ~ 300..600) - without a barrier
    between them an answer weight could be zeroed after it was written.  200 repetitions must agree bit for bit with each
    other and with the fp64 weights."""
    from dalm_amd.ops import default_ops

    ops = default_ops()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, k, Tg = 4, 3, 700
    T = Tg - 1
    label_lp = -torch.rand(B, k, T, generator=g, dtype=torch.float64) * 6.0
    n_ans = torch.randint(40, 90, (B,), generator=g)
    cut = torch.randint(300, 600, (B, k), generator=g)
    mask = torch.zeros(B, k, T, dtype=torch.int64)
    for b in range(B):
        for c in range(k):
            mask[b, c, :int(cut[b, c]) + int(n_ans[b])] = 1
    doc_lp = torch.log_softmax(torch.randn(B, k, generator=g, dtype=torch.float64), dim=1)
    M = float(mask.sum()) / k
    want = torch.zeros(B, k, Tg, dtype=torch.float64)
    for b in range(B):
        for c in range(k):
            want[b, c, :int(cut[b, c])] = 1.0 / k / M
        for j in range(int(n_ans[b])):
            v = torch.stack([doc_lp[b, c] + label_lp[b, c, int(cut[b, c]) + j] for c in range(k)])
            post = torch.softmax(v, 0)
            for c in range(k):
                want[b, c, int(cut[b, c]) + j] = post[c] / M
    row_nll = torch.zeros(B, k, Tg)
    row_nll[:, :, :T] = (-label_lp * mask).float()
    stats = torch.tensor([M, float(B)])
    args = (row_nll.to(dev), cut.to(dev), n_ans.float().to(dev), doc_lp.float().to(dev), stats.to(dev))
    first_out, first_w = ops.ce_finalize_topk(*args, want_weights=True)
    torch.testing.assert_close(first_w.cpu().double(), want, rtol=3e-5, atol=1e-9)
    for _ in range(200):
        out, w = ops.ce_finalize_topk(*args, want_weights=True)
        assert torch.equal(w, first_w) and torch.equal(out, first_out)
    # an answer row outside every sequence (inconsistent Nb) contributes nothing instead of NaN
    bad_n = n_ans.float().clone()
    bad_n[0] = Tg + 50
    out, w = ops.ce_finalize_topk(args[0], args[1], bad_n.to(dev), args[3], args[4], want_weights=True)
    assert torch.isfinite(out).all() and torch.isfinite(w).all()
