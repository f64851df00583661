"""How the two lm_head GEMMs (hipBLASLt, bf16) scale with the number of rows M on MI355X:
    GEMM1  logits[M, V]  = h[M, H] @ W[V, H]^T          GEMM2  dh[M, H] = dlogits[M, V] @ W[V, H]
GPU-side time (hipGraph replay), so that chunk sizes for `dalm_amd.fused._lm_head_live_rows` are picked from measurements.
    python tools/lm_head_gemm_sweep.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from kernel_bench import time_graph  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for name, H, V in (("Llama-2-7b", 4096, 32000), ("Falcon-7B", 4544, 65024)):
        W = (0.02 * torch.randn(V, H)).to(dev, torch.bfloat16)
        Wt = W.t().contiguous()   # [H, V]: GEMM2 with a k-contiguous B operand (the layout GEMM1 already has)
        print(f"{name} H={H} V={V}:  rows   GEMM1 us  TF/s    GEMM2 us  TF/s   both per 256 rows (us)   GEMM2 through W^T copy us  TF/s")
        for M in (256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2560, 3072, 3584, 4096, 4608):
            h = torch.randn(M, H).to(dev, torch.bfloat16)
            dl = torch.randn(M, V).to(dev, torch.bfloat16)
            out1 = torch.empty(M, V, device=dev, dtype=torch.bfloat16)
            out2 = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
            t1, _ = time_graph(lambda: torch.mm(h, W.t(), out=out1), reps=5, replays=5)
            t2, _ = time_graph(lambda: torch.mm(dl, W, out=out2), reps=5, replays=5)
            t3, _ = time_graph(lambda: torch.mm(dl, Wt.t(), out=out2), reps=5, replays=5)
            fl = 2.0 * M * H * V
            print(f"{M:24d} {t1 * 1e6:9.1f} {fl / t1 / 1e12:6.0f} {t2 * 1e6:10.1f} {fl / t2 / 1e12:6.0f} "
                  f"{(t1 + t2) * 1e6 / (M / 256):10.1f} {t3 * 1e6:24.1f} {fl / t3 / 1e12:6.0f}")


if __name__ == "__main__":
    main()
