"""Per-kernel times of the LoRA branch at the shapes of the BASELINE configs (bf16): the round-4 kernels (`dalm_lora_*`, one
projection per launch, mask recomputed) next to the round-5 ones (`dalm_lora2_*`, stacked projections, mask bits).

    python tools/lora_bench.py [--rows 4608 --cols 4096] [--sets 8]

Every timed call cycles through `--sets` different activation buffers (8 x 37.7 MB > the 256 MB Infinity Cache) so that a
kernel is not timed on data its own previous launch left on die.  Times are HIP-event averages over 3 replays of a hipGraph holding `--iters` launches."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from dalm_amd.models import lora_ops as L  # noqa: E402


def timed(fn, sets, iters, warmup=3):
    """us per call: `iters` calls captured into ONE hipGraph and replayed (a python call of these ops costs 20-40 us of host
    time - launched eagerly the loop would measure the host, as the first version of this tool did)."""
    for i in range(warmup):
        fn(i % sets)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(0)                                          # per-stream buffers (tickets) exist before the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        for i in range(iters):
            fn(i % sets)
    graph.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        graph.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (3 * iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4608)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--sets", type=int, default=8)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--p", type=float, default=0.05)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="substring of the line names to run")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    R, C, S, p, rank = a.rows, a.cols, a.sets, a.p, 8
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(R, C, generator=g).bfloat16().to(dev) for _ in range(S)]
    ys = [torch.randn(R, C, generator=g).bfloat16().to(dev) for _ in range(S)]
    A0, A1 = (torch.randn(rank, C, generator=g).to(dev) / C ** 0.5 for _ in range(2))
    B0 = torch.randn(C, rank, generator=g).to(dev)                     # [N, r] contiguous: what round 4 read
    z0, z1 = (torch.randn(R, rank, generator=g).to(dev) for _ in range(2))
    seed = L.dropout_seed(dev)
    _, bits = L.rowdot2([xs[0]], [A0, A1], rank, 1.0, p, [1, 2], 2)
    mb = R * C * 2 / 1e6
    keep = 1.0 / (1.0 - p)
    rows = []

    def line(name, us, bytes_mb, note=""):
        tbs = bytes_mb / us
        rows.append({"kernel": name, "us": round(us, 2), "MB": round(bytes_mb, 1), "TB/s": round(tbs, 2), "frac_of_8TBs": round(tbs / 8, 3),
                     "note": note})
        print(f"{name:58s} {us:8.2f} us  {bytes_mb:7.1f} MB  {tbs:5.2f} TB/s  {tbs / 8:5.3f} of 8 TB/s  {note}", flush=True)

    def run(name, fn, bytes_mb, note=""):
        if a.only is None or a.only in name:
            line(name, timed(fn, S, a.iters), bytes_mb, note)

    print(f"# [{R}, {C}] bf16, rank 8, p = {p}, {S} buffer sets, {a.iters} launches per line")
    # ---- forward: z = dropout(x) A^T for q and v ----
    run("r4 rowdot (x, A) dropout, ONE projection", lambda i: L._rowdot(xs[i], A0, True, rank, keep, p, seed, 1), mb)
    run("r5 rowdot2 mode 1 dropout (+bits), ONE projection", lambda i: L.rowdot2([xs[i]], [A0], rank, keep, p, [1], 1), mb * (1 + 1 / 16))
    run("r5 rowdot2 mode 2 dropout (+bits), q AND v, one x pass", lambda i: L.rowdot2([xs[i]], [A0, A1], rank, keep, p, [1, 2], 2), mb * (1 + 2 / 16), "r4 needs 2 launches")
    run("r5 rowdot2 mode 3 dropout (+bits), q AND v, SAME x in both slots", lambda i: L.rowdot2([xs[i], xs[i]], [A0, A1], rank, keep, p, [1, 2], 3), mb * (1 + 2 / 16), "x read twice, the second time from the caches")
    run("r5 rowdot2 mode 2 no dropout, q AND v", lambda i: L.rowdot2([xs[i]], [A0, A1], rank, 1.0, 0.0, [0, 0], 2), mb)
    # ---- backward: dz = s g B ----
    run("r4 rowdot (g, B [N,r]), ONE projection", lambda i: L._rowdot(ys[i], B0, False, rank, 2.0, 0.0, None, 0), mb)
    run("r5 rowdot2 mode 3 (g_q, g_v; B^T [r,N]), TWO projections", lambda i: L.rowdot2([ys[i], ys[(i + 1) % S]], [A0, A1], rank, 2.0, 0.0, [0, 0], 3), 2 * mb)
    # ---- out += s z B^T ----
    run("r4 rankupd forward, ONE projection", lambda i: L._rankupd_(ys[i], z0, B0, True, rank, 2.0, 0.0, None, 0), 2 * mb)
    run("r5 rankupd2 mode 3 forward, TWO projections", lambda i: L.rankupd2_([ys[i], ys[(i + 1) % S]], [z0, z1], [A0, A1], None, rank, 2.0, 3), 4 * mb)
    run("r5 rankupd2 mode 1 forward, ONE projection", lambda i: L.rankupd2_([ys[i]], [z0], [A0], None, rank, 2.0, 1), 2 * mb)
    # ---- dx += mask (dz A) ----
    run("r4 rankupd backward (mask hashed), ONE projection", lambda i: L._rankupd_(xs[i], z0, A0, False, rank, keep, p, seed, 1), 2 * mb)
    run("r5 rankupd2 mode 2 backward (mask bits), q AND v on one dx", lambda i: L.rankupd2_([xs[i]], [z0, z1], [A0, A1], bits, rank, keep, 2), 2 * mb * (1 + 1 / 16), "r4 needs 2 launches")
    # ---- dB / dA ----
    run("r4 colacc dB (2 launches), ONE projection", lambda i: L._colacc(ys[i], z0, rank, 2.0, 0.0, None, 0, False), mb)
    run("r5 colacc2 mode 3 dB, TWO projections, one launch", lambda i: L.colacc2([ys[i], ys[(i + 1) % S]], [z0, z1], None, rank, 2.0, 3), 2 * mb)
    run("r4 colacc dA (mask hashed, 2 launches), ONE projection", lambda i: L._colacc(xs[i], z0, rank, keep, p, seed, 1, True), mb)
    run("r5 colacc2 mode 2 dA (mask bits), q AND v, one x pass", lambda i: L.colacc2([xs[i]], [z0, z1], bits, rank, keep, 2), mb * (1 + 2 / 16), "r4 needs 4 launches")
    run("r5 colacc2 mode 1 dA (mask bits), ONE projection", lambda i: L.colacc2([xs[i]], [z0], [bits[0]], rank, keep, 1), mb * (1 + 1 / 16))
    # ---- a plain read of the same bytes for scale ----
    run("torch sum over the activation (a read-only pass)", lambda i: xs[i].sum(dtype=torch.float32), mb)
    if a.json:
        Path(a.json).write_text(json.dumps({"rows": R, "cols": C, "p": p, "lines": rows}, indent=1))


if __name__ == "__main__":
    main()
