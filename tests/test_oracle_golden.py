"""Pin the CPU oracle to the reference: every oracle function vs the golden vectors that
oracle/make_golden.py dumped from the reference's own code."""
import numpy as np
import pytest
import torch

import dalm_oracle as O
from helpers import LOSS_CASES, POOL_CASES, load_npz


def test_golden_files_present():
    assert len(LOSS_CASES) >= 10 and len(POOL_CASES) >= 4


@pytest.mark.parametrize("case", LOSS_CASES)
def test_ref_restatement_matches_reference_fp64(case):
    z = load_npz(case)
    q = z["q"].clone().requires_grad_(True)
    p = z["p"].clone().requires_grad_(True)
    lg = z["logits"].clone().requires_grad_(True)
    out = O.ref_step_loss(q, p, lg, z["ids"], z["mask"], z["qlen"], int(z["scale"]))
    out["loss"].backward()
    for k in ("S", "contrastive", "generator", "loss"):
        torch.testing.assert_close(out[k].detach(), z[f"ref64_{k}"], rtol=1e-12, atol=1e-12, equal_nan=True)
    torch.testing.assert_close(q.grad, z["ref64_dq"], rtol=1e-11, atol=1e-12, equal_nan=True)
    torch.testing.assert_close(p.grad, z["ref64_dp"], rtol=1e-11, atol=1e-12, equal_nan=True)
    torch.testing.assert_close(lg.grad, z["ref64_dlogits"], rtol=1e-11, atol=1e-13, equal_nan=True)


@pytest.mark.parametrize("case", LOSS_CASES)
def test_closed_form_matches_reference_fp64(case):
    z = load_npz(case)
    s = float(z["scale"])
    f = O.closed_forward(z["q"], z["p"], z["logits"], z["ids"], z["mask"], z["qlen"], s)
    for k in ("contrastive", "generator", "loss"):
        torch.testing.assert_close(f[k], z[f"ref64_{k}"], rtol=1e-11, atol=1e-11, equal_nan=True)
    b = O.closed_backward(z["q"], z["p"], z["logits"], z["ids"], z["mask"], z["qlen"], s, f)
    torch.testing.assert_close(b["dS"], z["ref64_dS"], rtol=1e-10, atol=1e-13)
    torch.testing.assert_close(b["dq"], z["ref64_dq"], rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(b["dp"], z["ref64_dp"], rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(b["dlogits"], z["ref64_dlogits"], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("case", LOSS_CASES)
def test_contrastive_only_closed_form(case):
    z = load_npz(case)
    s = float(z["scale"])
    f = O.closed_forward(z["q"], z["p"], None, None, None, None, s)
    torch.testing.assert_close(f["loss"], z["ref64_con_only"], rtol=1e-11, atol=1e-11)
    b = O.closed_backward(z["q"], z["p"], None, None, None, None, s, f)
    torch.testing.assert_close(b["dq"], z["ref64_con_only_dq"], rtol=1e-10, atol=1e-11)
    torch.testing.assert_close(b["dp"], z["ref64_con_only_dp"], rtol=1e-10, atol=1e-11)


def test_fp32_reference_within_tolerance_of_fp64():
    """How far the reference's own fp32 run is from fp64: sets the scale of the 1e-3 target."""
    for case in LOSS_CASES:
        z = load_npz(case)
        a, b = float(z["ref32_loss"]), float(z["ref64_loss"])
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (case, a, b)


def test_pieces_nll_and_marginalize():
    z = load_npz("pieces")
    torch.testing.assert_close(O.ref_nll(z["lp"], z["labels"]), z["nll"], rtol=0, atol=0)
    doc = torch.tensor([-1.25], dtype=torch.float64)
    for ql in (1, 2, 4, 7, 8, 12):
        torch.testing.assert_close(O.ref_marginalize_log_probs(z["lp"][0], doc, ql), z[f"marg_q{ql}"], rtol=0, atol=0)
        # closed-form row rule used by the kernels
        T = z["lp"].shape[1]
        cut = int(O._cut_rows(torch.tensor([ql]), T)[0])
        exp = z["lp"][0].clone()
        exp[cut:] += doc
        torch.testing.assert_close(exp, z[f"marg_q{ql}"], rtol=0, atol=0)


@pytest.mark.parametrize("case", POOL_CASES)
def test_pool_oracle_matches_reference(case):
    z = load_npz(case)
    normalize = bool(z["normalize"])
    h = z["h"].clone().requires_grad_(True)
    e = O.ref_retrieval_embed(h, z["mask"], normalize)
    (e * z["upstream"]).sum().backward()
    torch.testing.assert_close(e.detach(), z["ref64_emb"], rtol=1e-12, atol=1e-12)
    torch.testing.assert_close(h.grad, z["ref64_dh"], rtol=1e-11, atol=1e-12)
    e2, nrm, ic = O.closed_pool(z["h"], z["mask"], normalize)
    torch.testing.assert_close(e2, z["ref64_emb"], rtol=1e-11, atol=1e-12)
    dh = O.closed_pool_bwd(z["upstream"], e2, nrm, ic, z["mask"], normalize)
    torch.testing.assert_close(dh, z["ref64_dh"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(O.ref_eos_mask(z["mask"]), z["eos_mask_left"], rtol=0, atol=0)


def test_oracle_ops_consistent_with_closed_form():
    """OracleOps (the checker backend injected into the sharded-loss tests) == closed form."""
    z = load_npz("loss_dead_row")
    ops = O.OracleOps()
    q, p, s = z["q"].float(), z["p"].float(), float(z["scale"])
    lse_r, diag = ops.sim_rowstats(q, p, s, 0)
    lse_c, _ = ops.sim_rowstats(p, q, s, 0)
    con, doc = ops.contrastive_finalize(lse_r, lse_c, diag, q.shape[0])
    stats, Nb, Mb = ops.ce_prep(z["mask"], z["qlen"])
    rl, rn, dl = ops.ce_fwd(z["logits"].float(), z["ids"], z["mask"], stats, True)
    gen = ops.ce_finalize(rn, Nb, doc, stats)
    assert abs(float(con) - float(z["ref64_contrastive"])) < 1e-4 * abs(float(z["ref64_contrastive"]))
    assert abs(float(gen) - float(z["ref64_generator"])) < 1e-4 * abs(float(z["ref64_generator"]))
    torch.testing.assert_close(dl.double(), z["ref64_dlogits"], rtol=1e-4, atol=1e-7)


def test_closed_chunked_equals_closed_form_and_reference_golden():
    """closed_chunked (the full-size, sample-at-a-time oracle used by the GPU tests) == closed_forward/backward,
    and == the reference's own outputs on the golden cases."""
    import dalm_oracle as O
    from helpers import LOSS_CASES, load_npz

    for case in LOSS_CASES:
        z = load_npz(case)
        if float(z["mask"][:, 1:].sum()) == 0:
            continue
        c = O.closed_chunked(z["q"], z["p"], z["logits"], z["ids"], z["mask"], z["qlen"], float(z["scale"]),
                             dlogits_got=z["ref64_dlogits"])
        assert abs(float(c["loss"]) - float(z["ref64_loss"])) <= 1e-10 * max(1.0, abs(float(z["ref64_loss"]))), case
        torch.testing.assert_close(c["dq"], z["ref64_dq"], rtol=1e-9, atol=1e-11)
        torch.testing.assert_close(c["dp"], z["ref64_dp"], rtol=1e-9, atol=1e-11)
        assert c["dlogits_err"] <= 1e-10 and c["dlogits_max_err"] <= 1e-10, (case, c["dlogits_err"])


def test_topk_marginalisation_reduces_to_the_reference_at_k1():
    """`closed_gen_loss_topk` (k retrieved contexts, used to check dalm_marg_ce_finalize_topk) at k = 1 on a reference golden:
    it must reproduce the generator loss the REFERENCE computed (train_utils.py:113-138)."""
    import dalm_oracle as O
    from helpers import load_npz

    for case in ("loss_base_right_pad", "loss_left_pad", "loss_qlen_one", "loss_partial_batch_3"):
        z = load_npz(case)
        logits, ids, mask, qlen = z["logits"].double(), z["ids"], z["mask"], z["qlen"]
        S = z["ref64_S"].double()
        T = logits.shape[1] - 1
        lp = torch.log_softmax(logits[:, :-1], dim=2).gather(2, ids[:, 1:].unsqueeze(2)).squeeze(2)
        cut = O._cut_rows(qlen.reshape(-1), T).clamp(max=T)
        doc = torch.log_softmax(S, dim=1).diag()
        m = mask[:, 1:]
        if not all(bool((m[b, int(cut[b]):int(cut[b]) + int(m[b, int(cut[b]):].sum())] == 1).all()) for b in range(len(cut))):
            continue      # (left-padded rows whose cut falls into the padding: not the layout this helper states)
        got = O.closed_gen_loss_topk(lp.unsqueeze(1), m.unsqueeze(1), cut.unsqueeze(1), doc.unsqueeze(1))
        assert abs(float(got["generator"]) - float(z["ref64_generator"])) <= 1e-12 * abs(float(z["ref64_generator"]))


def test_lora_mask_oracle_is_a_deterministic_well_mixed_function():
    """oracle/lora_mask.py (the numpy restatement the GPU tests pin the LoRA dropout kernels to): same inputs -> same mask;
    seed word, salt and position each change it; the keep rate is 1 - p to binomial accuracy; known bits for one input (so an
    edit of the restatement cannot go unnoticed when no GPU is around)."""
    import lora_mask as O

    m = O.keep_mask(0x0123456789ABCDEF, 42, 256, 1024, 0.05)
    assert m.shape == (256, 1024) and np.array_equal(m, O.keep_mask(0x0123456789ABCDEF, 42, 256, 1024, 0.05))
    n = m.size
    assert abs(m.mean() - 0.95) < 4 * (0.05 * 0.95 / n) ** 0.5 + 1e-4
    for other in (O.keep_mask(0x0123456789ABCDEE, 42, 256, 1024, 0.05), O.keep_mask(0x0123456789ABCDEF, 43, 256, 1024, 0.05),
                  O.keep_mask(0x1123456789ABCDEF, 42, 256, 1024, 0.05)):
        agree = (other == m).mean()                       # two independent masks agree on 0.95^2 + 0.05^2 = 0.905 of the elements
        assert 0.89 < agree < 0.92, agree
    rows = m.mean(1)
    assert rows.min() > 0.9 and m.mean(0).min() > 0.85           # no dead rows / columns
    half = O.keep_mask(7, 7, 64, 64, 0.5)
    assert abs(half.mean() - 0.5) < 0.04
    assert O.threshold(0.05) == 3277 and O.threshold(0.5) == 32768 and O.threshold(0.0) == 0
    assert O.keep_mask(1, 2, 2, 8, 0.0).all()
    assert int(np.packbits(O.keep_mask(0xDEADBEEF, 5, 1, 64, 0.5)).astype(np.uint64).sum()) == KNOWN_LORA_MASK_BYTESUM


KNOWN_LORA_MASK_BYTESUM = 888


def test_attention_dropout_oracle_known_answers():
    """oracle/attn_dropout.py (the numpy restatement the GPU tests pin the attention kernels' keep mask to): known answers of the
    restatement itself, so that an edit of the oracle cannot silently move with an edit of the kernels; drop rate; salt and seed
    both change the mask; one hash serves the pair (c even, c + 1)."""
    import attn_dropout as A

    m = A.keep_mask(0x0123456789ABCDEF, 0x5A17, 2, 3, 8, 0.1)
    assert m.shape == (2, 3, 8, 8) and int(m.sum()) == 345
    assert m.reshape(-1)[:40].astype(int).tolist() == [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1,
                                                       1, 1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1]
    m2 = A.keep_mask(0x0123456789ABCDEF, 0x5A17, 1, 2, 64, 0.5)
    assert int(m2.sum()) == 4157 and np.packbits(m2.reshape(-1)[:64]).tolist() == [240, 106, 249, 173, 59, 187, 238, 140]
    big = A.keep_mask(7, 1, 4, 4, 128, 0.1)
    assert abs(1.0 - big.mean() - 0.1) < 3e-3
    assert not np.array_equal(big, A.keep_mask(7, 2, 4, 4, 128, 0.1)) and not np.array_equal(big, A.keep_mask(8, 1, 4, 4, 128, 0.1))
    assert A.keep_mask(7, 1, 1, 1, 8, 0.0).all()
