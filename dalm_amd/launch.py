"""Torchrun-free single-node launcher: one process per GPU, rendezvous over 127.0.0.1.

The reference gets its process-per-GPU launch from `accelerate launch`
(dalm/training/rag_e2e/train_rage2e.py:276,416-418); here `spawn_ranks` starts N copies of a command with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, which is everything
`dalm_amd.sharded.init_distributed` (torch.distributed, backend nccl == RCCL over xGMI) reads.

    python -m dalm_amd.launch --nproc 8 -m dalm_amd.training.rag_e2e.train_rage2e --dataset_path ...
    python bench.py --gpus 8          # bench.py calls spawn_ranks on itself when no RANK is set

Rank 0 inherits stdout (one JSON line from bench.py stays one JSON line); the other ranks' stdout is folded
into stderr.  The first failing rank takes the job down: the remaining ranks are terminated by PID and
the launcher exits with that rank's code.
"""
from __future__ import annotations

import argparse
import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence

_LAUNCH_T0 = time.time()


def in_distributed_env(env: Optional[Dict[str, str]] = None) -> bool:
    """True when a launcher (torchrun, this module) has already set the rank environment."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def visible_gpus() -> int:
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    # the host driver supports only dmabuf IPC: without this RCCL's cross-process buffer sharing fails
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (the native communicator's rendezvous runs over a TCPStore on MASTER_PORT since round 4: no id file under /tmp)
    return env


def spawn_ranks(cmd: Sequence[str], nproc: int, *, require_gpus: bool = True, port: Optional[int] = None,
                extra_env: Optional[Dict[str, str]] = None, poll_s: float = 0.2, term_grace_s: float = 10.0) -> int:
    """Run `cmd` nproc times (rank r gets LOCAL_RANK=r -> cuda:r) and return the job's exit code."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    if require_gpus:
        have = visible_gpus()
        if have < nproc:
            sys.stderr.write(f"[dalm_amd.launch] {nproc} ranks requested but only {have} GPU(s) are visible on this "
                             f"node; one process drives one GPU\n")
            return 2
    port = port or free_port()
    procs: List[subprocess.Popen] = []
    for r in range(nproc):
        env = rank_env(r, nproc, port)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(cmd), env=env, stdout=None if r == 0 else sys.stderr))
    code = 0
    live = set(range(nproc))
    kill_at: Optional[float] = None      # set at the first failure: ranks still alive after the grace period get SIGKILL
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and code == 0:
                code = rc
                sys.stderr.write(f"[dalm_amd.launch] rank {r} exited with code {rc}; stopping the other ranks\n")
                for o in live:
                    procs[o].terminate()
                kill_at = time.time() + term_grace_s
        if live and kill_at is not None and time.time() >= kill_at:
            # a rank stuck inside an RCCL collective or a HIP call never sees SIGTERM: escalate inside the wait loop
            for o in sorted(live):
                sys.stderr.write(f"[dalm_amd.launch] rank {o} ignored SIGTERM for {term_grace_s:.0f} s; killing it\n")
                procs[o].kill()
            kill_at = time.time() + term_grace_s
        if live:
            time.sleep(poll_s)
    return code


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(prog="python -m dalm_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--nproc", type=int, required=True, help="ranks = GPUs of this node to use")
    ap.add_argument("--cpu", action="store_true", help="do not require GPUs (gloo runs)")
    ap.add_argument("-m", dest="module", default=None, help="run a module (python -m) instead of a script")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="script (or module arguments) and its arguments")
    a = ap.parse_args(argv)
    rest = [x for x in a.rest if x != "--"] if a.rest[:1] == ["--"] else a.rest
    cmd = [sys.executable] + (["-m", a.module] if a.module else []) + rest
    if len(cmd) == 1:
        ap.error("nothing to run")
    return spawn_ranks(cmd, a.nproc, require_gpus=not a.cpu)


if __name__ == "__main__":
    raise SystemExit(main())
