"""dalm_sim_grad at D = 1024: the f32-pipe flash kernel next to the bf16x3 form (both contractions on the bf16 matrix cores at f32
accuracy).  f32-equivalent TFLOP/s = 4 m n D / t; bf16 TFLOP/s issued = 24 m n D / t.    python tools/sim_grad_x3_bench.py [sizes...]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dalm_amd.ops import default_ops  # noqa: E402


def timed(fn, it=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    ops = default_ops()
    dev = torch.device("cuda:0")
    sizes = [int(x) for x in sys.argv[1:]] or [4096, 8192, 16384]
    print(f"# DALM_X3_GRAD_BLOCK = {os.environ.get('DALM_X3_GRAD_BLOCK', 'adaptive: 2048 ... 8192 by the size of the dS image (default)')}")
    for m in sizes:
        n, D = m, 1024
        A = torch.nn.functional.normalize(torch.randn(m, D, device=dev), dim=1)
        B = torch.nn.functional.normalize(torch.randn(n, D, device=dev), dim=1)
        rl, _ = ops.sim_rowstats(A, B, 100.0, 0)
        cl = torch.zeros(n, device=dev) + 5.0
        rc = torch.full((m,), 1.0 / m, device=dev)
        cc = torch.full((n,), 1.0 / n, device=dev)
        os.environ["DALM_SIM_GRAD_X3"] = "0"
        t0 = timed(lambda: ops.sim_grad(A, B, 100.0, 0, rc, rl, cc, cl))
        os.environ["DALM_SIM_GRAD_X3"] = "1"
        t1 = timed(lambda: ops.sim_grad(A, B, 100.0, 0, rc, rl, cc, cl))
        os.environ.pop("DALM_SIM_GRAD_X3")
        fl = 4.0 * m * n * D
        print(f"m = n = {m:6d}  f32 pipe {t0:8.3f} ms {fl / t0 / 1e9:7.1f} TF   bf16x3 {t1:8.3f} ms {fl / t1 / 1e9:7.1f} TF f32-equivalent, "
              f"{6 * fl / t1 / 1e9:7.1f} TF issued = {6 * fl / t1 / 1e9 / 2500:5.3f} of the bf16 peak", flush=True)


if __name__ == "__main__":
    main()
