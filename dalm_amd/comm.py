"""`NativeRcclComm`: the communicator interface of dalm_amd.fused (all_gather_rows / all_reduce_sum_) on the
library's own RCCL binding (`dalm_comm_*` in include/dalm_hip.h) instead of torch.distributed.

`DALM_NATIVE_COMM=1` (trainers / bench) or direct construction; the torch.distributed(nccl) path stays the default only
because it is the one that has run on hardware with more than one rank (no multi-GPU box was available to rounds 1-3).  Bootstrap without
torch.distributed: rank 0 asks RCCL for the 128-byte unique id and publishes it through a file next to the
rendezvous port (`DALM_COMM_ID_FILE`, which `dalm_amd.launch` sets to a path unique to the launch; under other launchers
`/tmp/dalm_comm_<MASTER_PORT>.id` - remove a stale one after a crashed job), the other ranks poll for it.

Stream contract: a collective is enqueued on torch's CURRENT stream (`dalm_comm_*_on`): stream-ordered like any kernel
launch - no library-owned stream, no events, capturable into a hipGraph with the rest of the step.  Overlap is the caller's
choice of stream (`GatherHandle` runs the call under a side torch stream; the gradient buckets use their communication
stream).  Round 2 routed every collective through a stream owned by the communicator and measured 214 ms instead of 180 ms
per step on one GPU: one more HIP stream shifts the stream -> hardware-queue mapping of the whole process.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Optional

import torch

from . import hip


def _id_path() -> str:
    return os.environ.get("DALM_COMM_ID_FILE", f"/tmp/dalm_comm_{os.environ.get('MASTER_PORT', '0')}.id")


class NativeRcclComm:
    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None, device: Optional[int] = None,
                 unique_id: Optional[bytes] = None, timeout_s: float = 120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        lib = hip.load()
        if unique_id is None:
            unique_id = self._bootstrap(lib, timeout_s)
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of an ncclUniqueId")
        self._id = C.create_string_buffer(unique_id, 128)
        handle = C.c_void_p()
        hip.call("dalm_comm_init", C.byref(handle), self._id, self.rank, self.world_size, self.device)
        self._h = handle
        torch.cuda.set_device(self.device)

    def _bootstrap(self, lib, timeout_s: float) -> bytes:
        path = _id_path()
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            hip.call("dalm_comm_unique_id", buf)
            tmp = path + f".{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(buf.raw)
            os.replace(tmp, path)            # atomic publish
            return buf.raw
        t0 = time.time()
        while time.time() - t0 < timeout_s:
            try:
                with open(path, "rb") as f:
                    data = f.read()
                if len(data) == 128:
                    return data
            except FileNotFoundError:
                pass
            time.sleep(0.05)
        raise TimeoutError(f"rank {self.rank}: no RCCL unique id at {path} after {timeout_s:.0f} s")

    # ---- communicator interface used by dalm_amd.fused / dalm_amd.sharded ----
    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        hip.require_gpu(t)
        t = t.contiguous()
        out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        hip.call("dalm_comm_allgather_on", self._h, hip.ptr(t), hip.ptr(out), t.numel() * t.element_size(), hip.stream())
        return out

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        hip.require_gpu(t)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise TypeError("NativeRcclComm.all_reduce_sum_ needs a contiguous float32 tensor")
        hip.call("dalm_comm_allreduce_sum_f32_on", self._h, hip.ptr(t), t.numel(), hip.stream())
        return t

    def close(self) -> None:
        if getattr(self, "_h", None) is not None:
            hip.call("dalm_comm_destroy", self._h)
            self._h = None
            if self.rank == 0:
                try:
                    os.remove(_id_path())
                except OSError:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
