"""TEST INFRASTRUCTURE ONLY (never imported by dalm_amd/): numpy restatement of the keep mask of the attention dropout that
`dalm_attn_fwd` / `dalm_attn_bwd` regenerate in every kernel (dalm_amd/csrc/attn.hip: `attn_drop`, `drop_pair`).

The reference reaches attention dropout through transformers' BertSelfAttention (`dropout=self.dropout.p` in training mode, p =
attention_probs_dropout_prob = 0.1 for bge-large; dalm/models/rag_e2e_base_model.py:84-93 calls the retriever).  torch draws that mask
from its own philox stream inside the attention kernel; there is no reference stream to match (torch's differs between devices and
kernels) - what is pinned is that EVERY keep bit the three kernels use equals this restatement, and that forward and backward use
the same bits (tests/test_attention_gpu.py).  Parity unpinned against the reference's stream, by construction.

Element (b, h, i, j) of a [B, H, T, T] probability tensor has index c = ((b H + h) T + i) T + j (mod 2^32); T is even, so c and j
have the same parity.  One 32-bit hash per PAIR (c even, c + 1): the low 16 bits decide element c, the high 16 bits element c + 1;
an element is kept when its field >= round(p * 65536)."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def keep_mask(seed_word: int, salt: int, B: int, H: int, T: int, p: float) -> np.ndarray:
    """bool [B, H, T, T]: True where the probability is kept (and scaled by 1 / (1 - p))."""
    assert T % 2 == 0
    s = int(seed_word) & 0xFFFFFFFFFFFFFFFF
    salt = int(salt) & 0xFFFFFFFF
    a = _mix32(np.array([(s & 0xFFFFFFFF) ^ ((salt * 0x9E3779B9) & 0xFFFFFFFF)], dtype=np.uint64))[0]
    b = _mix32(np.array([((s >> 32) + salt + 0x85EBCA6B) & 0xFFFFFFFF], dtype=np.uint64))[0] | np.uint64(1)
    thresh = np.uint64(int(p * 65536.0 + 0.5))
    c = (np.arange(B * H * T * T, dtype=np.uint64) & M32)
    h = _mix32((((c >> np.uint64(1)) ^ a) + b) & M32)
    field = np.where((c & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return (field >= thresh).reshape(B, H, T, T)
