"""Build libdalm_hip.so (gfx950) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the repo
snapshot; `python -m dalm_amd._build` (or `__graft_entry__.build()`) rebuilds it
whenever a source is newer than the library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libdalm_hip.so"
SOURCES = ["lib.hip", "ce.hip", "sim.hip", "sim_small.hip", "pool.hip", "comm.hip", "lmhead.hip", "nf4.hip", "tower.hip", "falcon.hip", "attn.hip", "lora.hip", "lora2.hip"]
HEADERS = [CSRC / "common.hpp", CSRC / "lora_common.hpp", CSRC.parent.parent / "include" / "dalm_hip.h"]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return LIB
    cc = hipcc_path()
    objs = []
    newest_header = max(h.stat().st_mtime for h in HEADERS)
    for src in SOURCES:
        obj = CSRC / (Path(src).stem + ".o")
        objs.append(str(obj))
        # an object newer than its source and every header is reused (a one-file edit recompiles one file, not eleven)
        if not force and obj.exists() and obj.stat().st_mtime > max((CSRC / src).stat().st_mtime, newest_header):
            continue
        cmd = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print("[dalm_amd build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", str(LIB)]
    if verbose:
        print("[dalm_amd build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
