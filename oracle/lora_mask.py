"""CPU restatement (numpy, integer arithmetic) of the LoRA dropout mask of dalm_amd/csrc/lora.hip - TEST INFRASTRUCTURE,
never imported by dalm_amd.

The reference's adapters drop activations with torch's dropout (peft `lora_dropout=0.05`, dalm/models/rag_e2e_base_model.py:
145-160); which elements fall is an implementation detail of the random stream (torch's Philox stream differs between
devices and versions, so there is no reference mask to match).  The HIP kernels never store a mask: three of them regenerate
it from (seed word, salt, element index).  This file restates that function bit for bit so that the tests can pin the
kernels' mask to something written independently of them:

    key.a = mix(lo32(seed) ^ (salt * 0x9E3779B9))         key.b = mix(hi32(seed) + salt + 0x85EBCA6B) | 1
    h(i)  = mix(mix(i ^ key.a) + key.b)                   for the pair index i = flat_element_index >> 1
    element 2i   is kept iff  (h(i) & 0xFFFF) >= thr      element 2i+1 iff (h(i) >> 16) >= thr,   thr = round(p * 65536)
    mix(x): x ^= x >> 16; x *= 0x7FEB352D; x ^= x >> 15; x *= 0x846CA68B; x ^= x >> 16        (all modulo 2^32)

Round 5 (dalm_amd/csrc/lora2.hip, bf16 activations): mask v2 - the same keys and 16-bit threshold test; ONE two-multiply hash
per chunk of 8 elements, its four words chained by xorshift32; the forward kernel computes it ONCE and stores it as bits (byte
(row, c) bit e = element 8 c + e survives), the backward kernels read the bits:

    w_0(c) = mix2(c ^ key.a, key.b)      for the chunk index c = flat_element_index >> 3
    w_{q+1} = xorshift32(w_q):  w ^= w << 13; w ^= w >> 17; w ^= w << 5
    element 8c + 2q is kept iff (w_q & 0xFFFF) >= thr,  element 8c + 2q + 1 iff (w_q >> 16) >= thr
    mix2(x, b): x ^= x >> 16; x *= 0x7FEB352D; x ^= x >> 15; x += b; x *= 0x846CA68B; x ^= x >> 16   (all modulo 2^32)

PARITY: not a reference algorithm (see above) - what is pinned is kernel == this restatement for every element, the keep rate,
and that the three kernels agree with each other (tests/test_lora_ops_gpu.py).
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def threshold(p: float) -> int:
    return int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))


def keep_mask(seed: int, salt: int, rows: int, cols: int, p: float) -> np.ndarray:
    """bool [rows, cols]: True where the element survives.  `seed`: the 64-bit seed word (any Python int, taken modulo 2^64),
    `salt`: 32 bits, cols a multiple of 2."""
    seed &= (1 << 64) - 1
    salt &= 0xFFFFFFFF
    a = _mix(np.array([(seed & 0xFFFFFFFF) ^ ((salt * 0x9E3779B9) & 0xFFFFFFFF)], dtype=np.uint64))[0]
    b = _mix(np.array([((seed >> 32) + salt + 0x85EBCA6B) & 0xFFFFFFFF], dtype=np.uint64))[0] | np.uint64(1)
    n = rows * cols
    pair = (np.arange(0, n, 2, dtype=np.uint64) & M32) >> np.uint64(1)          # the flat index wraps modulo 2^32 like the kernel's
    h = _mix((_mix(pair ^ a) + b) & M32)
    thr = np.uint64(threshold(p))
    out = np.empty(n, dtype=bool)
    out[0::2] = (h & np.uint64(0xFFFF)) >= thr
    out[1::2] = (h >> np.uint64(16)) >= thr
    return out.reshape(rows, cols)


def _mix2(x: np.ndarray, b) -> np.ndarray:
    x = x.astype(np.uint64) & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x + np.uint64(b)) & M32
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def keep_mask_v2(seed: int, salt: int, rows: int, cols: int, p: float) -> np.ndarray:
    """bool [rows, cols], mask v2 (dalm_lora2_rowdot)."""
    seed &= (1 << 64) - 1
    salt &= 0xFFFFFFFF
    a = _mix(np.array([(seed & 0xFFFFFFFF) ^ ((salt * 0x9E3779B9) & 0xFFFFFFFF)], dtype=np.uint64))[0]
    b = _mix(np.array([((seed >> 32) + salt + 0x85EBCA6B) & 0xFFFFFFFF], dtype=np.uint64))[0] | np.uint64(1)
    n = rows * cols
    chunk = (np.arange(0, n, 8, dtype=np.uint64) & M32) >> np.uint64(3)
    w = _mix2(chunk ^ a, b)
    thr = np.uint64(threshold(p))
    out = np.empty(n, dtype=bool)
    for q in range(4):
        if q:
            w ^= (w << np.uint64(13)) & M32
            w ^= w >> np.uint64(17)
            w ^= (w << np.uint64(5)) & M32
        out[2 * q::8] = (w & np.uint64(0xFFFF)) >= thr
        out[2 * q + 1::8] = (w >> np.uint64(16)) >= thr
    return out.reshape(rows, cols)


def pack_bits(mask: np.ndarray) -> np.ndarray:
    """bool [rows, cols] -> uint8 [rows, cols / 8], bit e of byte c = mask[row, 8 c + e] (the layout dalm_lora2_rowdot writes)."""
    return np.packbits(mask.astype(np.uint8), axis=1, bitorder="little")
