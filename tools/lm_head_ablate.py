"""Ablations of the round-3 lm_head kernel (measurement only: DALM_LM_HEAD_ABL builds skip parts of the loop, results are
garbage unless 0).  One process per variant because the library reads the variable once.
    python tools/lm_head_ablate.py [variant ...]
"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kernel_bench import time_graph
from dalm_amd.ops import default_ops
dev = torch.device("cuda:0"); ops = default_ops()
out = []
for R, K, V in ((3584, 4096, 32000), (3072, 4544, 65024)):
    g = torch.Generator().manual_seed(0)
    h = torch.randn(R, K, generator=g).to(dev, torch.bfloat16)
    W = (0.02 * torch.randn(V, K, generator=g)).to(dev, torch.bfloat16)
    labels = torch.randint(0, V, (R,), generator=g).to(dev)
    t, tb = time_graph(lambda: ops.lm_head_lse(h, W, labels), reps=5, replays=8)
    out.append(f"{t*1e6:7.1f} us {2.0*R*K*V/t/1e12:6.1f} TF/s")
print("   |   ".join(out))
''' % (str(ROOT), str(ROOT / "tools"))

NAMES = {0: "shipped kernel (load pieces 8/8/0/0)", 1: "no loads in the loop", 2: "no fragment reads in the loop",
         35: "MFMA only (no loads, reads, barrier)", 64: "full loop, no epilogue", 99: "MFMA only, no epilogue",
         100: "load pieces 6/5/5/0", 101: "load pieces 4/4/4/4", 102: "load pieces 12/4/0/0", 103: "load pieces 0/8/8/0"}
print(f"{'variant':44s} cfg3 live rows 3584 x 32000 x 4096   |   cfg5 live rows 3072 x 65024 x 4544")
for abl in [int(a) for a in sys.argv[1:]] or sorted(NAMES):
    env = dict(os.environ, DALM_LM_HEAD_ABL=str(abl))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"ABL={abl:3d} {NAMES.get(abl, ''):36s} {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
