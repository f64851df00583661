"""The bf16 MFMA lm_head + log-sum-exp kernel (dalm_lm_head_lse_fwd, logits never stored) against the library path
(hipBLASLt GEMM + the forward-only CE kernel) at the cfg3 / cfg5 live-row shapes.  GPU time inside a hipGraph.
    python tools/lm_head_kernel_bench.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from kernel_bench import time_graph  # noqa: E402

from dalm_amd.ops import default_ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ops = default_ops()
    for name, R, K, V in (("cfg3 Llama-2-7b live rows", 3584, 4096, 32000), ("cfg5 Falcon-7B live rows", 3072, 4544, 65024),
                          ("cfg3 all rows", 4608, 4096, 32000)):
        g = torch.Generator().manual_seed(0)
        h = torch.randn(R, K, generator=g).to(dev, torch.bfloat16)
        W = (0.02 * torch.randn(V, K, generator=g)).to(dev, torch.bfloat16)
        labels = torch.randint(0, V, (R,), generator=g).to(dev)
        ids = torch.cat((labels[:1] * 0, labels)).view(1, R + 1)
        mask = torch.ones_like(ids)
        stats = torch.tensor([float(R), 0, 0, 0], device=dev)
        buf = torch.empty((R + 1, V), device=dev, dtype=torch.bfloat16)

        def library():
            torch.mm(h, W.t(), out=buf[:R])
            return ops.ce_fwd(buf.view(1, R + 1, V), ids, mask, stats, False)

        def kernel():
            return ops.lm_head_lse(h, W, labels)

        lse_l = library()[0][:R]
        lse_k = kernel()[0]
        err = float((lse_l - lse_k).abs().max())
        fl = 2.0 * R * K * V
        tl, _ = time_graph(library, reps=5, replays=5)
        tk, _ = time_graph(kernel, reps=5, replays=5)
        print(f"{name}: R={R} K={K} V={V}  ({fl / 1e12:.2f} TFLOP)")
        print(f"   hipBLASLt GEMM + forward CE kernel   {tl * 1e6:8.1f} us   {fl / tl / 1e12:7.1f} TF/s")
        print(f"   dalm_lm_head_lse_fwd (no logits)      {tk * 1e6:8.1f} us   {fl / tk / 1e12:7.1f} TF/s = {fl / tk / 2.5e15:.3f} of the 2.5 PF bf16 "
              f"MFMA peak   max |lse diff| vs the library path {err:.2e}")


if __name__ == "__main__":
    main()
