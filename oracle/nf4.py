"""CPU restatement of bitsandbytes' blockwise NF4 quantisation - TEST INFRASTRUCTURE, never imported by dalm_amd.

What the reference delegates to a third-party dependency: `BitsAndBytesConfig(load_in_4bit=True,
bnb_4bit_quant_type="nf4", bnb_4bit_compute_dtype=torch.bfloat16)` (dalm/models/rag_e2e_base_model.py:137-142,
dalm/models/retriever_only_base_model.py:84-90).  `bitsandbytes` is listed unpinned in the reference's pyproject.toml
(:28), is not vendored under /root/reference and is not installed in this image, so this file restates the PUBLISHED
algorithm: the 16 NF4 levels of QLoRA (Dettmers et al. 2023, appendix E) and bitsandbytes' kQuantizeBlockwise /
kDequantizeBlockwise for DATA_TYPE = NF4 (csrc/kernels.cu: blocksize 64, absmax per block, x * (1 / absmax) walked down
a decision tree of midpoints with `>`, element 2j in the high nibble of byte j, dequantised as level * absmax).

PARITY UNPINNED against bitsandbytes itself (no copy of it is reachable here): what the tests pin is (1) the published
level table and its defining properties (symmetric end points, exact zero, quantiles of N(0,1) as the paper constructs
them), (2) the HIP kernels bit-for-bit against this restatement, (3) the round-trip error bound of the format.
"""
from __future__ import annotations

import numpy as np

BLOCK = 64
LEVELS = np.array([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                   -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                   0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                   0.7229568362236023, 1.0], dtype=np.float32)
# the tree's thresholds as bitsandbytes spells them (midpoints of neighbouring levels)
MIDPOINTS = np.array([-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544,
                      -0.23460740596055984, -0.13791173323988914, -0.045525018125772476, 0.03979014977812767,
                      0.1202552504837513, 0.2035212516784668, 0.2920137718319893, 0.3893125355243683,
                      0.5016634166240692, 0.6427869200706482, 0.8614784181118011], dtype=np.float32)


def quantize(w: np.ndarray):
    """flattened float32 weights -> (packed uint8 [ceil(n/2)], absmax float32 [ceil(n/64)])."""
    flat = np.asarray(w, dtype=np.float32).reshape(-1)
    n = flat.size
    nb = (n + BLOCK - 1) // BLOCK
    pad = np.zeros(nb * BLOCK, dtype=np.float32)
    pad[:n] = flat
    blocks = pad.reshape(nb, BLOCK)
    absmax = np.abs(blocks).max(axis=1).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(absmax > 0, np.float32(1.0) / absmax, np.float32(0.0)).astype(np.float32)
    x = (blocks * inv[:, None]).astype(np.float32).reshape(-1)
    # number of thresholds strictly below x == the leaf the `>` tree reaches
    idx = (x[:, None] > MIDPOINTS[None, :]).sum(axis=1).astype(np.uint8)
    idx = idx[: n + (n & 1)]
    packed = ((idx[0::2] << 4) | idx[1::2]).astype(np.uint8)
    return packed, absmax


def dequantize(packed: np.ndarray, absmax: np.ndarray, n: int) -> np.ndarray:
    """-> float32 [n] (level * absmax in float32; a bf16 consumer rounds this to nearest-even)."""
    idx = np.empty(packed.size * 2, dtype=np.uint8)
    idx[0::2] = packed >> 4
    idx[1::2] = packed & 15
    idx = idx[:n]
    scale = np.repeat(np.asarray(absmax, dtype=np.float32), BLOCK)[:n]
    return (LEVELS[idx] * scale).astype(np.float32)


def roundtrip(w: np.ndarray) -> np.ndarray:
    p, a = quantize(w)
    return dequantize(p, a, np.asarray(w).size).reshape(np.asarray(w).shape)
