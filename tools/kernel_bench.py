"""Per-kernel timings (HIP events on the launch stream) against the gfx950 rooflines.

    python tools/kernel_bench.py [--quick]

Algorithmic bytes / flops follow SURVEY.md section 8(d): dense definitions (all rows counted).
HBM peak 8 TB/s (spec; ~6.3 TB/s achievable), f32 MFMA peak 157.3 TFLOP/s.
"""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dalm_amd.ops import default_ops  # noqa: E402

HBM_PEAK = 8.0e12
MFMA_F32_PEAK = 157.3e12


def time_fn(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
    return ts[len(ts) // 2], ts[0]


def time_graph(fn, reps=20, replays=10):
    """GPU-side time per call for latency-bound ops: `reps` calls captured in one hipGraph, replayed; the host
    (python + ctypes + allocator, ~10 us per launch) is out of the measurement.  Returns (median, best) seconds."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(replays):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3 / reps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def bench_ce(B, Tg, V, dtype, full_mask=True):
    ops = default_ops()
    dev = torch.device("cuda:0")
    logits = torch.randn(B, Tg, V, device=dev, dtype=dtype)
    ids = torch.randint(0, V, (B, Tg), device=dev)
    if full_mask:
        mask = torch.ones(B, Tg, dtype=torch.int64, device=dev)
    else:
        lens = torch.randint(60, Tg + 1, (B,), device=dev)
        mask = (torch.arange(Tg, device=dev).unsqueeze(0) < lens.unsqueeze(1)).long()
    qlen = torch.full((B,), int(0.8 * Tg), device=dev)
    stats, Nb, Mb = ops.ce_prep(mask, qlen)
    el = logits.element_size()
    R = B * (Tg - 1)
    out = {}
    med, best = time_fn(lambda: ops.ce_fwd(logits, ids, mask, stats, False))
    out["fwd"] = {"s": med, "best_s": best, "GBps": R * V * el / med / 1e9, "frac": R * V * el / med / HBM_PEAK}
    buf = torch.empty_like(logits)
    med, best = time_fn(lambda: ops.ce_fwd(logits, ids, mask, stats, True))
    out["fwd+grad(fused)"] = {"s": med, "best_s": best, "GBps": 2 * R * V * el / med / 1e9, "frac": 2 * R * V * el / med / HBM_PEAK}
    # what the training step runs: the gradient overwrites the logits (footprint = one buffer, not two)
    scratch = logits.clone()
    med, best = time_fn(lambda: ops.ce_fwd(scratch, ids, mask, stats, True, inplace=True))
    out["fwd+grad(in place)"] = {"s": med, "best_s": best, "GBps": 2 * R * V * el / med / 1e9, "frac": 2 * R * V * el / med / HBM_PEAK}
    del scratch
    row_lse, _, _ = ops.ce_fwd(logits, ids, mask, stats, False)
    g = torch.ones(1, device=dev)
    med, best = time_fn(lambda: ops.ce_bwd(logits, ids, mask, stats, row_lse, g))
    out["bwd(separate)"] = {"s": med, "best_s": best, "GBps": 2 * R * V * el / med / 1e9, "frac": 2 * R * V * el / med / HBM_PEAK}
    # a plain device copy of the same bytes, for calibration of the achievable ceiling
    med, best = time_fn(lambda: buf.copy_(logits))
    out["torch_copy(ref)"] = {"s": med, "GBps": 2 * B * Tg * V * el / med / 1e9}
    return out


def bench_sim(m, n, D):
    ops = default_ops()
    dev = torch.device("cuda:0")
    A = torch.nn.functional.normalize(torch.randn(m, D, device=dev), dim=1)
    Bm = torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1)
    out = {}
    # mid sizes are a handful of launches of tens of microseconds: besides the eager number (HIP events around each call,
    # comparable with earlier rounds) report the GPU time of the call inside a hipGraph - how the training step runs it
    graphed = 256 * 256 <= m * n <= 4096 * 4096
    med, best = time_fn(lambda: ops.sim_rowstats(A, Bm, 100.0, 0), iters=10, warmup=3)
    # the default entry point routes m, n >= 3072 to the bf16x3 form (6 x the flops on the bf16 pipe): its roofline fraction is
    # the flops ACTUALLY ISSUED over the bf16 peak - never an f32-equivalent rate over the f32 peak (that read 1.19 / 1.40 in
    # profiles/history/r04_kernel_bench.txt)
    on_bf16 = m >= 3072 and n >= 3072 and D % 64 == 0
    flops = (12.0 if on_bf16 else 2.0) * m * n * D
    peak = 2.5e15 if on_bf16 else MFMA_F32_PEAK
    out["rowstats"] = {"s": med, "TFLOPs_f32_equivalent": 2.0 * m * n * D / med / 1e12, "pipe": "bf16 x3" if on_bf16 else "f32",
                       "TFLOPs_issued": flops / med / 1e12, "frac": flops / med / peak}
    if graphed:
        gm, _ = time_graph(lambda: ops.sim_rowstats(A, Bm, 100.0, 0), reps=10, replays=7)
        out["rowstats"].update({"graph_s": gm, "graph_frac": flops / gm / peak})
    if m >= 1024 and n >= 1024:
        # round 4: the same statistics on the bf16 matrix cores (three bf16 thirds per operand, 6 D deep); "TFLOPs" stays the
        # f32-EQUIVALENT rate 2 m n D / t (the kernel executes 6 x that on the bf16 pipe); sim_rowstats itself routes
        # m, n >= 4096 there, so the exact-f32 kernel is timed through its pinned entry point
        med, _ = time_fn(lambda: ops.sim_rowstats_bf16x3(A, Bm, 100.0, 0), iters=10, warmup=3)
        # priced on the flops ISSUED against the bf16 peak only: an f32-equivalent rate over the f32 peak reads above 1 here
        # (1.21 / 1.43 in earlier files) and says nothing about the kernel
        out["rowstats_bf16x3"] = {"s": med, "TFLOPs_f32_equivalent": 2.0 * m * n * D / med / 1e12, "pipe": "bf16 x3",
                                  "TFLOPs_issued": 12.0 * m * n * D / med / 1e12, "frac": 12.0 * m * n * D / med / 2.5e15}
        med, _ = time_fn(lambda: ops.sim_rowstats_f32(A, Bm, 100.0, 0), iters=10, warmup=3)
        out["rowstats_f32"] = {"s": med, "TFLOPs": 2.0 * m * n * D / med / 1e12, "frac": 2.0 * m * n * D / med / MFMA_F32_PEAK}
    if m * n <= 20000 * 20000:
        rl, _ = ops.sim_rowstats(A, Bm, 100.0, 0)
        cl = torch.zeros(n, device=dev) + 5.0
        rc = torch.full((m,), 1.0 / m, device=dev)
        cc = torch.full((n,), 1.0 / n, device=dev)
        med, best = time_fn(lambda: ops.sim_grad(A, Bm, 100.0, 0, rc, rl, cc, cl), iters=10, warmup=3)
        # sim_grad routes m, n >= 8192 to the bf16x3 form (HipOps.sim_grad): 6 x the flops on the bf16 pipe - priced against
        # the bf16 peak on the flops ISSUED, as the forward above (an f32-equivalent rate over the f32 peak reads > 1 there)
        import os

        g_x3 = os.environ.get("DALM_SIM_GRAD_X3") != "0" and m >= 8192 and n >= 8192 and D % 64 == 0
        g_fl = (24.0 if g_x3 else 4.0) * m * n * D
        g_peak = 2.5e15 if g_x3 else MFMA_F32_PEAK
        out["grad"] = {"s": med, "TFLOPs_f32_equivalent": 4.0 * m * n * D / med / 1e12, "pipe": "bf16 x3" if g_x3 else "f32",
                       "TFLOPs_issued": g_fl / med / 1e12, "frac": g_fl / med / g_peak}
        if graphed:
            gm, _ = time_graph(lambda: ops.sim_grad(A, Bm, 100.0, 0, rc, rl, cc, cl), reps=10, replays=7)
            out["grad"].update({"graph_s": gm, "graph_frac": g_fl / gm / g_peak})
    return out


def bench_small(m, n, D, want_cols=True):
    """Small-batch contrastive path: S once (2 launches), backward 1 launch; latency-bound, reported in us."""
    ops = default_ops()
    dev = torch.device("cuda:0")
    A = torch.nn.functional.normalize(torch.randn(m, D, device=dev), dim=1)
    Bm = torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1)
    out = {}
    med, best = time_graph(lambda: ops.sim_small_fwd(A, Bm, 100.0, 0, want_cols, one_launch=False))
    out["fwd(partial+stats)"] = {"us": med * 1e6, "best_us": best * 1e6, "TFLOPs": 2.0 * m * n * D / med / 1e12}
    ops.sim_small_fwd(A, Bm, 100.0, 0, want_cols, one_launch=True)        # tickets allocated outside the capture
    med, best = time_graph(lambda: ops.sim_small_fwd(A, Bm, 100.0, 0, want_cols, one_launch=True))
    out["fwd(one launch)"] = {"us": med * 1e6, "best_us": best * 1e6, "TFLOPs": 2.0 * m * n * D / med / 1e12}
    S, rl, _, cl = ops.sim_small_fwd(A, Bm, 100.0, 0, True)
    rc = torch.full((m,), 1.0 / m, device=dev)
    cc = torch.full((n,), 1.0 / n, device=dev)
    med, best = time_graph(lambda: ops.sim_small_bwd(S, A, Bm, 100.0, 0, rc, rl, cc, cl, True, want_cols, one_launch=False))
    nd = 2 if want_cols else 1
    out["bwd(dQ,dP)" if want_cols else "bwd(dQ)"] = {"us": med * 1e6, "best_us": best * 1e6,
                                                      "TFLOPs": nd * 2.0 * m * n * D / med / 1e12}
    if not want_cols:       # one direction of a long contraction: the slices summed inside the launch (round 4)
        ops.sim_small_bwd(S, A, Bm, 100.0, 0, rc, rl, cc, cl, True, False, one_launch=True)
        med, best = time_graph(lambda: ops.sim_small_bwd(S, A, Bm, 100.0, 0, rc, rl, cc, cl, True, False, one_launch=True))
        out["bwd(dQ, one launch)"] = {"us": med * 1e6, "best_us": best * 1e6, "TFLOPs": 2.0 * m * n * D / med / 1e12}
    return out


def bench_pool(B, T, D, dtype):
    ops = default_ops()
    dev = torch.device("cuda:0")
    h = torch.randn(B, T, D, device=dev, dtype=dtype)
    mask = torch.ones(B, T, dtype=torch.int64, device=dev)
    el = h.element_size()
    out = {}
    timer = time_graph if B * T * D * el < (256 << 20) else time_fn   # small shapes are host-bound when launched eagerly
    med, _ = timer(lambda: ops.pool_fwd(h, mask, True))
    out["fwd"] = {"s": med, "GBps": B * T * D * el / med / 1e9, "frac": B * T * D * el / med / HBM_PEAK}
    emb, norm, ic = ops.pool_fwd(h, mask, True)
    de = torch.randn_like(emb)
    med, _ = timer(lambda: ops.pool_bwd(de, emb, norm, ic, mask, True, T, dtype))
    out["bwd"] = {"s": med, "GBps": B * T * D * el / med / 1e9, "frac": B * T * D * el / med / HBM_PEAK}
    return out


def bench_nf4(rows, cols, dtype):
    """`use_bnb` storage: dequantise one [rows, cols] weight (Llama-2-7b's largest Linear is 11008 x 4096)."""
    from dalm_amd.models import nf4

    dev = torch.device("cuda:0")
    w = (torch.randn(rows, cols, device=dev) * 0.02).to(dtype)
    n = rows * cols
    el = w.element_size()
    out = {}
    med, _ = time_fn(lambda: nf4.quantize(w))
    b = n * el + n / 2 + n / 16
    out["quantize"] = {"s": med, "GBps": b / med / 1e9, "frac": b / med / HBM_PEAK}
    p, a = nf4.quantize(w)
    med, _ = time_fn(lambda: nf4.dequantize(p, a, (rows, cols), dtype))
    out["dequantize"] = {"s": med, "GBps": b / med / 1e9, "frac": b / med / HBM_PEAK}
    # as the step runs it: inside a hipGraph (no host time), a DIFFERENT weight every call (8 layers' worth, so the packed
    # input of a call was not just read by the previous one)
    ws = [nf4.quantize((torch.randn(rows, cols, device=dev) * 0.02).to(dtype)) for _ in range(8)]

    def sweep():
        for pk, am in ws:
            nf4.dequantize(pk, am, (rows, cols), dtype)

    med, _ = time_graph(sweep, reps=4)
    med /= len(ws)
    out["dequantize (graph, 8 weights in turn)"] = {"s": med, "GBps": b / med / 1e9, "frac": b / med / HBM_PEAK}
    return out


def bench_tower(B, T, H, hd, inter, dtype):
    """The two elementwise chains of a Llama-family generator layer: dalm_rope_qk / dalm_swiglu_* against the eager chains
    they replace (transformers' apply_rotary_pos_emb as the step ran it before - roll + addcmul - and silu(gate) * up)."""
    from dalm_amd.models import fastpath, tower_ops

    dev = torch.device("cuda:0")
    el = torch.empty((), dtype=dtype).element_size()
    q = torch.randn(B, T, H * hd, device=dev, dtype=dtype).view(B, T, H, hd).transpose(1, 2)
    k = torch.randn(B, T, H * hd, device=dev, dtype=dtype).view(B, T, H, hd).transpose(1, 2)
    cos, sin = torch.rand(B, T, hd, device=dev, dtype=dtype), torch.rand(B, T, hd, device=dev, dtype=dtype)
    out = {}
    b = 4 * q.numel() * el + 2 * cos.numel() * el
    med, _ = time_graph(lambda: tower_ops._rope_launch(q, k, cos, sin, False))
    out["rope fwd (1 launch)"] = {"s": med, "GBps": b / med / 1e9, "frac": b / med / HBM_PEAK}
    med, _ = time_graph(lambda: tower_ops._rope_launch(q, k, cos, sin, True))
    out["rope bwd (1 launch)"] = {"s": med, "GBps": b / med / 1e9, "frac": b / med / HBM_PEAK}
    med, _ = time_graph(lambda: fastpath._rope_roll(q, k, cos, sin))
    out["rope fwd eager (roll+addcmul)"] = {"s": med, "GBps": b / med / 1e9, "frac": b / med / HBM_PEAK}
    n = B * T * inter
    gate, up, da = (torch.randn(B * T, inter, device=dev, dtype=dtype) for _ in range(3))
    act, dg, du = torch.empty_like(gate), torch.empty_like(gate), torch.empty_like(gate)
    from dalm_amd import hip

    code = hip.dtype_code(gate)
    med, _ = time_graph(lambda: hip.call("dalm_swiglu_fwd", hip.ptr(gate), hip.ptr(up), hip.ptr(act), code, n, hip.stream()))
    out["swiglu fwd (1 launch)"] = {"s": med, "GBps": 3 * n * el / med / 1e9, "frac": 3 * n * el / med / HBM_PEAK}
    med, _ = time_graph(lambda: hip.call("dalm_swiglu_bwd", hip.ptr(da), hip.ptr(gate), hip.ptr(up), hip.ptr(dg), hip.ptr(du),
                                         code, n, hip.stream()))
    out["swiglu bwd (1 launch)"] = {"s": med, "GBps": 5 * n * el / med / 1e9, "frac": 5 * n * el / med / HBM_PEAK}
    med, _ = time_graph(lambda: torch.nn.functional.silu(gate) * up)
    out["swiglu fwd eager (2 launches)"] = {"s": med, "GBps": 3 * n * el / med / 1e9, "frac": 3 * n * el / med / HBM_PEAK}
    # the same on the two halves of ONE [rows, 2 C] GEMM output (gate | up as one forward GEMM, models/frozen_linear.py)
    R, Cc = B * T, inter
    both = torch.randn(R, 2 * Cc, device=dev, dtype=dtype)
    g2, u2 = both[:, :Cc], both[:, Cc:]
    med, _ = time_graph(lambda: hip.call("dalm_swiglu_fwd_2d", hip.ptr(g2), hip.ptr(u2), hip.ptr(act), code, R, Cc, 2 * Cc, 2 * Cc, Cc,
                                         hip.stream()))
    out["swiglu fwd, halves of one buffer"] = {"s": med, "GBps": 3 * n * el / med / 1e9, "frac": 3 * n * el / med / HBM_PEAK}
    med, _ = time_graph(lambda: hip.call("dalm_swiglu_bwd_2d", hip.ptr(da), hip.ptr(g2), hip.ptr(u2), hip.ptr(dg), hip.ptr(du), code,
                                         R, Cc, Cc, 2 * Cc, 2 * Cc, Cc, Cc, hip.stream()))
    out["swiglu bwd, halves of one buffer"] = {"s": med, "GBps": 5 * n * el / med / 1e9, "frac": 5 * n * el / med / HBM_PEAK}
    return out


def print_results(res) -> None:
    for k, v in res.items():
        print(k)
        for kk, vv in v.items():
            # values are numbers except for labels such as "pipe": "bf16 x3" (formatting every value with :.4g crashed the
            # whole run on the first similarity entry in round 5, before the JSON was written)
            print("   %-18s" % kk, "  ".join(f"{a}={b:.4g}" if isinstance(b, (int, float)) else f"{a}={b}" for a, b in vv.items()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--vocab", type=int, default=0,
                    help="with --only ce: run the B = 18 shapes of this vocabulary size only (PMC passes: one process per V, "
                         "because the counter CSV cannot tell two launches of one instantiation with the same grid apart)")
    ap.add_argument("--sizes", default="", help="with --only sim: square sizes to run instead of the standard list, e.g. 1200,1536")
    args = ap.parse_args()
    res = {}
    if args.only in ("", "ce"):
        for name, B, V, dt in (("ce cfg3 B18 Tg256 V32000 f32", 18, 32000, torch.float32),
                               ("ce cfg3 B18 Tg256 V32000 bf16", 18, 32000, torch.bfloat16),
                               ("ce cfg5 B18 Tg256 V65024 bf16", 18, 65024, torch.bfloat16),
                               ("ce cfg5 B18 Tg256 V65024 f32", 18, 65024, torch.float32),
                               ("ce B144 Tg256 V32000 f32 (8x)", 144, 32000, torch.float32)):
            if args.vocab and (V != args.vocab or B != 18):
                continue
            res[name] = bench_ce(B, 256, V, dt)
    if args.only in ("", "sim"):
        sizes = [(18, 18), (150, 150), (1200, 1200), (4096, 4096), (16384, 16384)]
        if not args.quick:
            sizes.append((65536, 65536))
        if args.sizes:
            sizes = [(int(x), int(x)) for x in args.sizes.split(",")]
        for m, n in sizes:
            res[f"sim {m}x{n} D1024"] = bench_sim(m, n, 1024)
        # per-rank blocks of the sharded negatives at W = 8: cfg3 (18 x 144) and cfg2 (150 x 1200)
        for m, n in ([] if args.sizes else [(18, 144), (150, 1200)]):
            res[f"sim sharded {m}x{n} D1024"] = bench_sim(m, n, 1024)
    if args.only in ("", "small"):
        for m, n in [(18, 18), (150, 150), (512, 512)]:
            res[f"small {m}x{n} D1024 (one GPU: rows+cols)"] = bench_small(m, n, 1024, True)
        for m, n in [(18, 144), (150, 1200)]:
            res[f"small sharded {m}x{n} D1024 (rows only)"] = bench_small(m, n, 1024, False)
    if args.only in ("", "pool"):
        res["pool cfg3 q B18 T50 D1024 bf16"] = bench_pool(18, 50, 1024, torch.bfloat16)
        res["pool cfg3 p B18 T128 D1024 bf16"] = bench_pool(18, 128, 1024, torch.bfloat16)
        res["pool cfg2 p B150 T128 D1024 bf16"] = bench_pool(150, 128, 1024, torch.bfloat16)
        res["pool cfg2 q B150 T50 D1024 f32"] = bench_pool(150, 50, 1024, torch.float32)
        res["pool cfg2 p B150 T128 D1024 f32"] = bench_pool(150, 128, 1024, torch.float32)
        res["pool cfg3 p B18 T128 D1024 f32"] = bench_pool(18, 128, 1024, torch.float32)
        res["pool B1200 T128 D1024 bf16"] = bench_pool(1200, 128, 1024, torch.bfloat16)
    if args.only in ("", "nf4"):
        res["nf4 11008x4096 -> bf16"] = bench_nf4(11008, 4096, torch.bfloat16)
        res["nf4 4096x4096 -> bf16"] = bench_nf4(4096, 4096, torch.bfloat16)
        res["nf4 11008x4096 -> f32"] = bench_nf4(11008, 4096, torch.float32)
    if args.only in ("", "tower"):
        res["tower cfg3 layer: rope [18,32,256,128] + swiglu [4608,11008] bf16"] = bench_tower(18, 256, 32, 128, 11008, torch.bfloat16)
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/kernel_bench.json").write_text(json.dumps(res, indent=1))
    print_results(res)


if __name__ == "__main__":
    main()
