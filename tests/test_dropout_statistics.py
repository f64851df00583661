"""Statistical properties of the two in-kernel dropout generators.

The LoRA dropout (dalm_amd/csrc/lora2.hip, p = 0.05: peft's lora_dropout, reference dalm/models/rag_e2e_base_model.py:145-160)
and the attention dropout (dalm_amd/csrc/attn.hip, p = 0.1: BERT's attention_probs_dropout_prob) have no reference random
stream to match; `oracle/lora_mask.py` / `oracle/attn_dropout.py` restate the generators and the GPU tests pin EVERY bit of the
kernels to them (tests/test_lora2_gpu.py, tests/test_attention_gpu.py).  Bit-equality with our own restatement is a regression
test, not evidence that the generator is a fair coin: a biased or correlated hash would train differently from torch's dropout
and pass all of them.  These tests put the restated generators (= the kernels' masks, bit for bit) through what dropout needs:

  * keep rate 1 - p within binomial bounds per head / per layer (module salt) / per step (seed advance);
  * masks of different layers, steps, calls and heads are uncorrelated, and so are neighbouring elements of one mask (the LoRA
    generator draws 8 elements from ONE hash through an xorshift chain, both generators take 2 elements from one 32-bit word).

Deterministic (fixed seeds).  Bounds: no cell further than 4.5 sigma from its expectation and at most 1.5 % of the cells beyond
3 sigma (0.27 % expected) - a statement about the generator, not about one lucky seed.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))

GOLD = 0x9E3779B97F4A7C15                 # lora_ops.advance_dropout_seed: the seed word += this, once per step
SEED0 = 0x243F6A8885A308D3


def _salt(uid: int, call: int) -> int:     # models/lora.py::next_mask_key, models/attention.py: module id above a 12-bit call counter
    return ((uid << 12) ^ (call & 0xFFF)) & 0xFFFFFFFF


def _check_cells(z: np.ndarray, what: str):
    z = np.abs(np.asarray(z, dtype=np.float64)).ravel()
    assert z.max() <= 4.5, f"{what}: a cell {z.max():.2f} sigma from its expectation"
    frac = float((z > 3.0).mean())
    assert frac <= 0.015, f"{what}: {100 * frac:.2f} % of {z.size} cells beyond 3 sigma"


def _corr(a: np.ndarray, b: np.ndarray) -> float:
    a = a.astype(np.float64).ravel() - a.mean()
    b = b.astype(np.float64).ravel() - b.mean()
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


def test_lora_mask_keep_rate_per_layer_and_step():
    import lora_mask as LM

    p, rows, cols = 0.05, 512, 1024
    n = rows * cols
    sigma = np.sqrt(p * (1 - p) / n)
    z = []
    for step in range(6):
        seed = (SEED0 + step * GOLD) & ((1 << 64) - 1)
        for layer in range(32):
            for proj in range(2):                       # q_proj / v_proj: their own module ids
                m = LM.keep_mask_v2(seed, _salt(1000 + 2 * layer + proj, 1), rows, cols, p)
                z.append((m.mean() - (1 - p)) / sigma)
    _check_cells(z, "LoRA keep rate per (step, layer, projection)")
    # the exact threshold: round(p * 65536) / 65536, not p - the bias of the 16-bit comparison is below 1e-5
    assert abs(LM.threshold(p) / 65536.0 - p) < 1e-5


def test_lora_mask_rows_and_columns_are_unbiased():
    """Per activation ROW (a token never loses systematically more features) and per COLUMN (a feature is not dropped for
    systematically more tokens): the chunk hash is keyed by the flat index, so both marginals must be binomial."""
    import lora_mask as LM

    p, rows, cols = 0.05, 4096, 1024
    m = LM.keep_mask_v2(SEED0, _salt(77, 3), rows, cols, p)
    _check_cells((m.mean(axis=1) - (1 - p)) / np.sqrt(p * (1 - p) / cols), "LoRA keep rate per row")
    _check_cells((m.mean(axis=0) - (1 - p)) / np.sqrt(p * (1 - p) / rows), "LoRA keep rate per column")


def test_lora_masks_are_uncorrelated_across_layers_steps_calls_and_neighbours():
    import lora_mask as LM

    p, rows, cols = 0.05, 1024, 1024
    n = rows * cols
    base = LM.keep_mask_v2(SEED0, _salt(1000, 1), rows, cols, p)
    others = {"next layer": LM.keep_mask_v2(SEED0, _salt(1001, 1), rows, cols, p),
              "same layer, next call": LM.keep_mask_v2(SEED0, _salt(1000, 2), rows, cols, p),
              "next step": LM.keep_mask_v2((SEED0 + GOLD) & ((1 << 64) - 1), _salt(1000, 1), rows, cols, p),
              "two steps on": LM.keep_mask_v2((SEED0 + 2 * GOLD) & ((1 << 64) - 1), _salt(1000, 1), rows, cols, p)}
    for name, m in others.items():
        assert abs(_corr(base, m)) <= 4.0 / np.sqrt(n), (name, _corr(base, m))
        assert not np.array_equal(base, m)
    flat = base.ravel()
    for lag in (1, 2, 3, 4, 5, 6, 7, 8, cols, cols + 1):        # inside a chunk of 8 (one hash, xorshift chain), across chunks, rows
        assert abs(_corr(flat[:-lag], flat[lag:])) <= 4.0 / np.sqrt(n), (lag, _corr(flat[:-lag], flat[lag:]))
    # the two 16-bit fields of one word and the four words of one chunk: joint drop frequency = p^2 within bounds
    drop = ~flat.reshape(-1, 8)
    for i in range(8):
        for j in range(i + 1, 8):
            both = (drop[:, i] & drop[:, j]).mean()
            s = np.sqrt(p * p * (1 - p * p) / drop.shape[0])
            assert abs(both - p * p) <= 4.5 * s, (i, j, both)


def test_attention_dropout_keep_rate_per_head_layer_and_step():
    import attn_dropout as AD

    p, B, H, T = 0.1, 3, 16, 128
    sigma = np.sqrt(p * (1 - p) / (T * T))
    z = []
    for step in range(3):
        seed = (SEED0 + step * GOLD) & ((1 << 64) - 1)
        for layer in range(24):
            m = AD.keep_mask(seed, _salt(500 + layer, 1), B, H, T, p)
            z.append((m.reshape(B * H, -1).mean(axis=1) - (1 - p)) / sigma)
    _check_cells(np.concatenate(z), "attention keep rate per (step, layer, batch row, head)")
    # per query row of one head: no row loses systematically more keys
    m = AD.keep_mask(SEED0, _salt(500, 1), 2, 4, 256, p)
    _check_cells((m.mean(axis=-1) - (1 - p)) / np.sqrt(p * (1 - p) / 256), "attention keep rate per query row")
    _check_cells((m.mean(axis=-2) - (1 - p)) / np.sqrt(p * (1 - p) / 256), "attention keep rate per key column")


def test_attention_masks_are_uncorrelated_across_heads_layers_steps_and_neighbours():
    import attn_dropout as AD

    p, B, H, T = 0.1, 2, 8, 128
    m = AD.keep_mask(SEED0, _salt(500, 1), B, H, T, p)
    n = T * T
    heads = m.reshape(B * H, n)
    c = np.corrcoef(heads.astype(np.float64))
    off = c[~np.eye(B * H, dtype=bool)]
    assert np.abs(off).max() <= 4.5 / np.sqrt(n), np.abs(off).max()          # 240 pairs of heads
    tot = B * H * n
    for name, o in {"next layer": AD.keep_mask(SEED0, _salt(501, 1), B, H, T, p),
                    "same layer, next call": AD.keep_mask(SEED0, _salt(500, 2), B, H, T, p),
                    "next step": AD.keep_mask((SEED0 + GOLD) & ((1 << 64) - 1), _salt(500, 1), B, H, T, p)}.items():
        assert abs(_corr(m, o)) <= 4.0 / np.sqrt(tot), (name, _corr(m, o))
    flat = m.ravel()
    for lag in (1, 2, 3, T, T + 1, n):           # the pair sharing one hash word, neighbours, the next query row, the next head
        assert abs(_corr(flat[:-lag], flat[lag:])) <= 4.0 / np.sqrt(tot), (lag, _corr(flat[:-lag], flat[lag:]))
    pairs = ~flat.reshape(-1, 2)
    both = (pairs[:, 0] & pairs[:, 1]).mean()
    assert abs(both - p * p) <= 4.5 * np.sqrt(p * p * (1 - p * p) / pairs.shape[0]), both


def test_lora_and_attention_generators_do_not_share_masks():
    """Both derive their keys from the same device seed word; a LoRA mask and an attention mask of the same salt must still be
    unrelated (different index -> hash maps)."""
    import attn_dropout as AD
    import lora_mask as LM

    p = 0.1
    a = AD.keep_mask(SEED0, 1234, 1, 4, 128, p).ravel()
    l = LM.keep_mask_v2(SEED0, 1234, 4 * 128, 128, p).ravel()
    assert abs(_corr(a, l)) <= 4.0 / np.sqrt(a.size)
