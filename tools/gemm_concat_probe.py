"""Frozen projections that read the same input as ONE GEMM (W_q|W_k|W_v -> [3N, K], W_gate|W_up -> [2N, K]) against the separate
GEMMs the step issues today, forward (TN) and backward (dx through the transposed copies: three accumulating GEMMs vs one with
the contraction over 3N), at the padded row count of cfg3 (4608) and at packed row counts (dalm_amd/packed.py).
hipGraph replay timing; `--tune` lets TunableOp search solutions for the shapes first (written to gpurun_out/gemm_concat_tuned.csv).

    python tools/gemm_concat_probe.py [--tune] [--rows 4608,3072,2944]"""
import argparse
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gemm_layout_probe import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--rows", default="4608,3072,2944")
    args = ap.parse_args()
    import torch.cuda.tunable as tunable

    from dalm_amd.tuning import enable_tuned_gemms

    print("tuned table loaded:", enable_tuned_gemms())
    if args.tune:
        tunable.enable(True)
        tunable.tuning_enable(True)
        tunable.set_max_tuning_duration(400)
        tunable.set_max_tuning_iterations(20)
        Path("gpurun_out").mkdir(exist_ok=True)
        tunable.set_filename("gpurun_out/gemm_concat_tuned.csv", insert_device_ordinal=False)
    dev = torch.device("cuda:0")
    K = 4096
    for R in [int(r) for r in args.rows.split(",")]:
        x = torch.randn(R, K, device=dev, dtype=torch.bfloat16)
        for name, N, parts in (("q|k|v", 4096, 3), ("gate|up", 11008, 2)):
            Ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(parts)]
            Wcat = torch.cat(Ws, 0)
            Wts = [w.t().contiguous() for w in Ws]
            WcatT = Wcat.t().contiguous()                        # [K, parts N]
            gs = [torch.randn(R, N, device=dev, dtype=torch.bfloat16) for _ in range(parts)]
            gcat = torch.cat(gs, 1)
            fl = 2.0 * R * N * K * parts

            def fwd_sep():
                return [F.linear(x, w) for w in Ws]

            def fwd_cat():
                return F.linear(x, Wcat)

            def bwd_sep():
                dx = F.linear(gs[0], Wts[0])
                for g, wt in zip(gs[1:], Wts[1:]):
                    dx.addmm_(g, wt.t())
                return dx

            def bwd_cat():
                return F.linear(gcat, WcatT)

            for label, fn in (("forward  separate", fwd_sep), ("forward  one GEMM", fwd_cat), ("backward separate (accumulating)", bwd_sep),
                              ("backward one GEMM", bwd_cat)):
                t = timed(fn)
                print(f"rows {R:5d}  {name:8s} N={N:5d} x{parts}  {label:34s} {t:8.1f} us  {fl / t / 1e6:7.1f} TF/s", flush=True)
        # the single projections for scale
        for name, N, Kk in (("o_proj", 4096, 4096), ("down_proj", 4096, 11008)):
            xx = torch.randn(R, Kk, device=dev, dtype=torch.bfloat16)
            W = torch.randn(N, Kk, device=dev, dtype=torch.bfloat16) * 0.02
            t = timed(lambda: F.linear(xx, W))
            print(f"rows {R:5d}  {name:8s} N={N:5d} K={Kk:5d}  forward {t:8.1f} us  {2.0 * R * N * Kk / t / 1e6:7.1f} TF/s", flush=True)
    if args.tune:
        tunable.write_file()


if __name__ == "__main__":
    main()
