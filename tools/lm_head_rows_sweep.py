"""lm_head forward + log-sum-exp at ARBITRARY row counts: the library path (torch.mm through hipBLASLt's default heuristics +
the forward CE kernel; the [rows, V] logits are materialised) against dalm_lm_head_lse_fwd (one bf16 MFMA kernel, no
logits), GPU time inside a hipGraph.  The live-row count of an evaluation batch is data-dependent, so no pre-tuned GEMM
solution exists for it.
    python tools/lm_head_rows_sweep.py [--vocab 32000 --hidden 4096] [--rows 1024,1536,...]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from kernel_bench import time_graph  # noqa: E402

from dalm_amd.ops import default_ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--rows", default="1024,1536,2048,2304,2560,2816,3000,3072,3328,3584,3840,4000,4096,4352,4608")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ops = default_ops()
    V, K = a.vocab, a.hidden
    g = torch.Generator().manual_seed(0)
    W = (0.02 * torch.randn(V, K, generator=g)).to(dev, torch.bfloat16)
    print(f"V={V} K={K}: rows | library (GEMM + CE fwd) us | kernel us | kernel / library")
    wins = 0
    rows_list = [int(x) for x in a.rows.split(",")]
    for R in rows_list:
        h = torch.randn(R, K, generator=g).to(dev, torch.bfloat16)
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

        tl, _ = time_graph(library, reps=5, replays=5)
        tk, _ = time_graph(kernel, reps=5, replays=5)
        wins += tk < tl
        print(f"   {R:5d} | {tl*1e6:9.1f} | {tk*1e6:9.1f} | {tk/tl:5.3f}", flush=True)
        del buf, h
    print(f"kernel faster at {wins} of {len(rows_list)} row counts")


if __name__ == "__main__":
    main()
