"""SURVEY 8(f) rank 1, measured: lm_head + marginalised CE at the cfg3 / cfg5 shapes, forward + backward to dh,
(a) logits materialised (hipBLASLt GEMM -> fused CE kernel in place -> hipBLASLt GEMM back) vs
(b) `rag_e2e_loss_from_hidden` (sample chunks, the [B,Tg,V] logits never exist),
(c) the same over the live rows only (`live_row_index`: padding rows skip both GEMMs and the CE).
    python tools/lm_head_bench.py
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from kernel_bench import time_fn  # noqa: E402

from dalm_amd.fused import gemm_wave_rows, live_row_index, rag_e2e_loss, rag_e2e_loss_from_hidden  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    if "--tuned" in sys.argv:   # replay dalm_amd/tuning/tunableop_gfx950.csv, as bench.py and the trainers do
        from dalm_amd.tuning import enable_tuned_gemms
        print("tuned GEMM table loaded:", enable_tuned_gemms())
    B, Tg, D = 18, 256, 1024
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]     # e.g. --only=cfg3 (rocprofv3 runs)
    arms = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--arms=")]     # substring filter on the arm label
    for name, H, V in (("cfg3 Llama-2-7b", 4096, 32000), ("cfg5 Falcon-7B", 4544, 65024)):
        if only and not any(name.startswith(o) for o in only):
            continue
        g = torch.Generator(device="cpu").manual_seed(0)
        q = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(dev)
        p = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(dev)
        hidden = torch.randn(B, Tg, H, generator=g).to(dev, torch.bfloat16)
        W = (0.02 * torch.randn(V, H, generator=g)).to(dev, torch.bfloat16)
        ids = torch.randint(0, V, (B, Tg), generator=g).to(dev)
        lens = torch.randint(60, Tg + 1, (B,), generator=g)
        mask = (torch.arange(Tg).unsqueeze(0) >= (Tg - lens).unsqueeze(1)).long().to(dev)
        qlen = (lens.float() * 0.8).long().to(dev)

        def materialised():
            h = hidden.detach().requires_grad_(True)
            loss = rag_e2e_loss(q, p, h @ W.t(), ids, mask, qlen, 100, inplace_grad=True)
            loss.backward()
            return h.grad

        live = live_row_index(mask, gemm_wave_rows(V))
        n_live = int((live >= 0).sum())
        live = live.to(dev)

        def chunked(chunk, rows=None):
            def f():
                h = hidden.detach().requires_grad_(True)
                loss = rag_e2e_loss_from_hidden(q, p, h, W, ids, mask, qlen, 100, chunk_samples=chunk, live_rows=rows)
                loss.backward()
                return h.grad
            return f

        def evaluate(rows, fused=True):
            def f():
                with torch.no_grad():
                    if fused:
                        return rag_e2e_loss_from_hidden(q, p, hidden, W, ids, mask, qlen, 100, live_rows=rows)
                    return rag_e2e_loss(q, p, hidden @ W.t(), ids, mask, qlen, 100)
            return f

        flops = 2 * 2.0 * B * Tg * H * V          # two GEMMs (logits, dh)
        print(name, f"B={B} Tg={Tg} H={H} V={V} bf16: 2 GEMMs = {flops / 1e12:.2f} TFLOP; "
                    f"{n_live} of {B * Tg} rows carry loss, padded to {live.numel()}")
        ref = materialised()
        for label, fn in (("materialised logits", materialised), ("chunked, 6 samples", chunked(6)),
                          ("chunked, 3 samples", chunked(3)), ("chunked, 18 samples (one chunk)", chunked(18)),
                          ("live rows, chunks <= 1536 rows", chunked(6, live)), ("live rows, chunks <= 2048 rows", chunked(8, live)),
                          ("live rows, chunks <= 1024 rows", chunked(4, live))):
            if arms and not any(a in label for a in arms):
                continue
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            out = fn()
            peak = (torch.cuda.max_memory_allocated() - base) / 1e6
            err = float((out.float() - ref.float()).norm() / ref.float().norm())
            med, _ = time_fn(fn, iters=10, warmup=3)
            print(f"   {label:34s} {med * 1e3:7.3f} ms   {flops / med / 1e12:7.1f} TF/s on the GEMM flops   "
                  f"peak extra memory {peak:8.1f} MB   dh rel diff vs materialised {err:.2e}")
        if not arms or any("eval" in a for a in arms):
            import os

            for kern, tag in (("0", "library GEMM + forward CE"), ("1", "dalm_lm_head_lse_fwd kernel (no logits)"), (None, "DEFAULT")):
                os.environ.pop("DALM_LM_HEAD_KERNEL", None)
                if kern is not None:
                    os.environ["DALM_LM_HEAD_KERNEL"] = kern
                lv = [float(evaluate(None, False)()), float(evaluate(None)()), float(evaluate(live)())]
                for label, fn in (("eval (no_grad): materialised logits", evaluate(None, False)),
                                  ("eval (no_grad): sample chunks", evaluate(None)), ("eval (no_grad): live rows", evaluate(live))):
                    if "materialised" in label and kern != "0":
                        continue          # the materialised form never takes the kernel
                    torch.cuda.reset_peak_memory_stats()
                    base = torch.cuda.memory_allocated()
                    fn()
                    peak = (torch.cuda.max_memory_allocated() - base) / 1e6
                    med, _ = time_fn(fn, iters=10, warmup=3)
                    print(f"   {label:38s} [{tag:40s}] {med * 1e3:7.3f} ms   peak extra memory {peak:7.1f} MB   "
                          f"(loss values {lv[0]:.5f} / {lv[1]:.5f} / {lv[2]:.5f})")
            os.environ.pop("DALM_LM_HEAD_KERNEL", None)


if __name__ == "__main__":
    main()
