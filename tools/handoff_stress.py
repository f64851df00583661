"""Stress the in-launch hand-offs (arrival tickets + write-through payloads: one-launch small forward / sliced backward,
streaming statistics merge): the same problem N times while a second stream keeps the chip busy, every word of every
output compared with the first run.  Arrival orders change from repetition to repetition; the results must not.
    python tools/handoff_stress.py [--reps 400]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from dalm_amd.ops import default_ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=400)
    a = ap.parse_args()
    ops = default_ops()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    total = 0
    for (m, n, D, off) in [(150, 1200, 1024, 450), (512, 512, 1024, 0), (600, 600, 768, 0), (1000, 1000, 64, 0), (70, 2100, 256, 1000)]:
        A = torch.nn.functional.normalize(torch.randn(m, D, device=dev), dim=1)
        B = torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1)
        ref = ops.sim_small_fwd(A, B, 100.0, off, True, one_launch=True)
        rc, cc = torch.rand(m, device=dev) / m, torch.rand(n, device=dev) / n
        refb = ops.sim_small_bwd(ref[0], A, B, 100.0, off, rc, ref[1], cc, ref[3], True, False, one_launch=True)[0]
        noise, side, bad = torch.randn(3072, 3072, device=dev), torch.cuda.Stream(), 0
        for i in range(a.reps):
            if i % 3 == 0:
                with torch.cuda.stream(side):
                    noise = torch.tanh(noise @ noise * 1e-3)
            got = ops.sim_small_fwd(A, B, 100.0, off, True, one_launch=True)
            gb = ops.sim_small_bwd(ref[0], A, B, 100.0, off, rc, ref[1], cc, ref[3], True, False, one_launch=True)[0]
            bad += not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[3], ref[3]) and torch.equal(gb, refb))
        torch.cuda.synchronize()
        total += bad
        print(f"small path {m} x {n} x {D}: forward + sliced backward in one launch each, mismatches in {a.reps} repetitions: {bad}")
    A = torch.nn.functional.normalize(torch.randn(1200, 1024, device=dev), dim=1)
    B = torch.nn.functional.normalize(torch.randn(1200, 1024, device=dev), dim=1)
    ref, bad = ops.sim_rowstats(A, B, 100.0, 0), 0
    for _ in range(a.reps):
        got = ops.sim_rowstats(A, B, 100.0, 0)
        bad += not (torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]))
    total += bad
    print(f"streaming statistics 1200^2 (merge inside the launch): mismatches in {a.reps} repetitions: {bad}")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
