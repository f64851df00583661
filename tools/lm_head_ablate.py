"""Ablations of the round-3 lm_head kernel (measurement only: DALM_LM_HEAD_ABL builds skip parts of the loop, results are
garbage).  One process per variant because the library reads the variable once.
    python tools/lm_head_ablate.py
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
R, K, V = 3584, 4096, 32000
g = torch.Generator().manual_seed(0)
h = torch.randn(R, K, generator=g).to(dev, torch.bfloat16)
W = (0.02 * torch.randn(V, K, generator=g)).to(dev, torch.bfloat16)
labels = torch.randint(0, V, (R,), generator=g).to(dev)
t, tb = time_graph(lambda: ops.lm_head_lse(h, W, labels), reps=5, replays=5)
fl = 2.0 * R * K * V
print(f"{t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s (best {fl/tb/1e12:7.1f})")
''' % (str(ROOT), str(ROOT / "tools"))

NAMES = {0: "full kernel", 1: "no loads in the loop", 2: "no fragment reads", 3: "no loads, no reads (MFMA + barriers)",
         4: "no stagger", 8: "no s_setprio", 16: "no MFMA (loads + reads + barriers)", 17: "no MFMA, no loads (reads + barriers)",
         18: "no MFMA, no reads (loads + barriers)", 19: "barriers only"}
NAMES4 = {20: "8-wave ring NA=2", 21: "8-wave ring NA=4", 22: "8-wave ring NA=0", 211: "8w ring: no loads", 212: "8w ring: no reads",
          274: "8w ring: no epilogue", 275: "8w ring: MFMA only, no epilogue", 10: "ring kernel NA=4", 111: "ring: no loads in the loop",
          112: "ring: no fragment reads", 174: "ring: no epilogue", 0: "4-wave kernel", 101: "4w: no loads in the loop", 102: "4w: no fragment reads", 103: "4w: MFMA + barrier only",
          135: "4w: MFMA only (no barrier)", 164: "4w: full loop, no epilogue", 199: "4w: MFMA only, no epilogue"}
if os.environ.get("DALM_LM_HEAD_GEN", "4") == "4":
    for var in [int(a) for a in sys.argv[1:]] or sorted(NAMES4):
        env = dict(os.environ, DALM_LM_HEAD_PIECES=str(var))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(f"PIECES={var:3d} {NAMES4.get(var, ''):45s} {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
    sys.exit(0)
for abl in [int(a) for a in sys.argv[1:]] or sorted(NAMES):
    env = dict(os.environ, DALM_LM_HEAD_ABL=str(abl))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"ABL={abl:2d} {NAMES.get(abl, ''):45s} {out.stdout.strip() or out.stderr.strip()[-300:]}", flush=True)
