"""Build libdalm_hip.so (gfx950) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the repo
snapshot; `python -m dalm_amd._build` (or `__graft_entry__.build()`) rebuilds it
whenever a source changed.

Object reuse is keyed on CONTENT, not on mtimes: next to every `x.o` sits `x.o.stamp` holding a hash of
(compiler version, full compile command, the source's bytes, every header's bytes); the library carries
`libdalm_hip.so.stamp` = hash of the object stamps + link command.  A tree shipped by rsync / a snapshot
(coarse or reset mtimes), another ARCH / flag set or an upgraded hipcc therefore never links stale
objects, and an unchanged tree never rebuilds.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libdalm_hip.so"
SOURCES = ["lib.hip", "ce.hip", "sim.hip", "sim_small.hip", "pool.hip", "comm.hip", "lmhead.hip", "nf4.hip", "tower.hip", "falcon.hip", "attn.hip", "lora.hip", "lora2.hip", "bert.hip"]
HEADERS = [CSRC / "common.hpp", CSRC / "lora_common.hpp", CSRC.parent.parent / "include" / "dalm_hip.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


_CC_VERSION: dict = {}


def _cc_version(cc: str) -> str:
    if cc not in _CC_VERSION:
        try:
            _CC_VERSION[cc] = subprocess.run([cc, "--version"], capture_output=True, text=True, check=True).stdout
        except Exception:  # no compiler on this box (GPU box without hipcc on PATH is still allowed to LOAD a shipped .so)
            _CC_VERSION[cc] = "unknown"
    return _CC_VERSION[cc]


def _headers_digest() -> bytes:
    h = hashlib.sha256()
    for p in HEADERS:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.digest()


def _compile_cmd(cc: str, src: str) -> list:
    return [cc, f"--offload-arch={ARCH}", *FLAGS, "-c", str(CSRC / src), "-o", str(CSRC / (Path(src).stem + ".o"))]


def _obj_stamp(cc: str, src: str, hdr: bytes) -> str:
    h = hashlib.sha256()
    h.update(_cc_version(cc).encode())
    # the command without absolute paths: a tree copied elsewhere (the GPU box) keeps its stamps valid
    h.update(" ".join([f"--offload-arch={ARCH}", *FLAGS, src]).encode())
    h.update((CSRC / src).read_bytes())
    h.update(hdr)
    return h.hexdigest()


def _lib_stamp(stamps: list) -> str:
    return hashlib.sha256(("|".join(stamps) + "|-shared -fPIC -ldl").encode()).hexdigest()


def _read(p: Path) -> str:
    try:
        return p.read_text().strip()
    except OSError:
        return ""


def _wanted_stamps():
    cc = hipcc_path()
    hdr = _headers_digest()
    return cc, [_obj_stamp(cc, s, hdr) for s in SOURCES]


def needs_build() -> bool:
    if not LIB.exists():
        return True
    try:
        _, stamps = _wanted_stamps()
    except RuntimeError:
        return False  # no compiler here: the shipped library is what there is
    return _read(Path(str(LIB) + ".stamp")) != _lib_stamp(stamps)


def build(force: bool = False, verbose: bool = True, jobs: int | None = None) -> Path:
    if not force and not needs_build():
        return LIB
    cc, stamps = _wanted_stamps()
    todo = []
    for src, stamp in zip(SOURCES, stamps):
        obj = CSRC / (Path(src).stem + ".o")
        if force or not obj.exists() or _read(Path(str(obj) + ".stamp")) != stamp:
            todo.append((src, stamp))

    def one(item):
        src, stamp = item
        cmd = _compile_cmd(cc, src)
        if verbose:
            print("[dalm_amd build]", " ".join(cmd), flush=True)
        st = Path(cmd[-1] + ".stamp")
        st.unlink(missing_ok=True)
        subprocess.run(cmd, check=True)
        st.write_text(stamp + "\n")

    jobs = jobs or int(os.environ.get("DALM_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        list(ex.map(one, todo))
    objs = [str(CSRC / (Path(s).stem + ".o")) for s in SOURCES]
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-ldl", "-o", str(LIB)]
    if verbose:
        print("[dalm_amd build]", " ".join(cmd), flush=True)
    Path(str(LIB) + ".stamp").unlink(missing_ok=True)
    subprocess.run(cmd, check=True)
    Path(str(LIB) + ".stamp").write_text(_lib_stamp(stamps) + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
