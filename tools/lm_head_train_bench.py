"""SURVEY 8 f1, training direction: lm_head + marginalised CE + d(hidden) over the live rows of a BASELINE batch (B = 18,
Tg = 256) - the hand-written kernel path (`dalm_lm_head_lse_fwd` + `dalm_lm_head_dlogits` / `dalm_transpose_bf16` /
`dalm_lm_head_dhidden`; nothing of size [rows, V] allocated) next to the chunked library path (torch.mm + the fused CE kernel).

    python tools/lm_head_train_bench.py [--json out.json]

Reports per path: ms per call (hipGraph replay), peak extra memory of the call, and the agreement of the two paths."""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--tuned", action="store_true", help="replay the pre-tuned GEMM solution table for the library path (what bench.py does)")
    a = ap.parse_args()
    from test_lm_head_backward_gpu import _batch

    from dalm_amd.fused import gemm_wave_rows, live_row_index, rag_e2e_loss_from_hidden

    if a.tuned:
        from dalm_amd import tuning

        tuning.enable_tuned_gemms()
    dev = torch.device("cuda:0")
    out = {}
    for cfg, V, H in (("cfg3 (Llama-2-7b head)", 32000, 4096), ("cfg5 (Falcon-7B head)", 65024, 4544)):
        q, p, h, W, ids, mask, qlen = _batch(18, 256, H, V, 1024, 3, dev)
        live = live_row_index(mask, multiple=gemm_wave_rows(V)).to(dev)
        res = {}
        for name, env in (("kernels", "1"), ("library (torch.mm + CE kernel, chunked)", "0")):
            os.environ["DALM_LM_HEAD_TRAIN_KERNEL"] = env

            def call():
                hh = h.clone().requires_grad_(True)
                loss = rag_e2e_loss_from_hidden(q, p, hh, W, ids, mask, qlen, 100.0, live_rows=live)
                loss.backward()
                return loss.detach(), hh.grad

            for _ in range(3):
                loss, dh = call()
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            loss, dh = call()
            torch.cuda.synchronize()
            peak = torch.cuda.max_memory_allocated() - base
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(5):
                    call()
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            res[name] = {"ms": e0.elapsed_time(e1) / 20, "peak_extra_MB": peak / 1e6, "loss": float(loss), "dh": dh.float()}
        os.environ.pop("DALM_LM_HEAD_TRAIN_KERNEL", None)
        k, l = res["kernels"], res["library (torch.mm + CE kernel, chunked)"]
        rel = float((k["dh"] - l["dh"]).norm() / l["dh"].norm())
        rows = int(live.numel())
        flop_lib = 2 * 2 * rows * V * H
        line = {"live_rows_padded": rows, "V": V, "H": H,
                "kernels_ms": round(k["ms"], 3), "library_ms": round(l["ms"], 3), "ratio": round(k["ms"] / l["ms"], 3),
                "kernels_peak_extra_MB": round(k["peak_extra_MB"], 1), "library_peak_extra_MB": round(l["peak_extra_MB"], 1),
                "logits_tensor_MB_never_allocated": round(18 * 256 * V * 2 / 1e6, 1),
                "kernel_path_TFLOPs_3_contractions": round(1.5 * flop_lib / (k["ms"] * 1e-3) / 1e12, 1),
                "library_path_TFLOPs_2_contractions": round(flop_lib / (l["ms"] * 1e-3) / 1e12, 1),
                "loss_kernels": k["loss"], "loss_library": l["loss"], "dh_rel_diff": rel}
        out[cfg] = line
        print(cfg, json.dumps(line), flush=True)
    if a.json:
        Path(a.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
