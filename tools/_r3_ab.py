"""Interleaved A/B of lm_head kernel variants (one process per variant and round; 3 rounds)."""
import os, subprocess, sys
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
    out.append(f"{t*1e6:7.1f} us {2.0*R*K*V/t/1e12:6.1f} TF")
print("   ".join(out))
''' % (str(ROOT), str(ROOT / "tools"))
variants = sys.argv[1:]
for rnd in range(3):
    for v in variants:
        env = dict(os.environ, DALM_LM_HEAD_PIECES=v)
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"round {rnd} PIECES={v:>3s}: {o.stdout.strip() or o.stderr.strip()[-200:]}", flush=True)
