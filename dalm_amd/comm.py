"""`NativeRcclComm`: the communicator interface of dalm_amd.fused (all_gather_rows / all_reduce_sum_) on the
library's own RCCL binding (`dalm_comm_*` in include/dalm_hip.h) instead of torch.distributed.

Opt-in at W > 1 (`DALM_NATIVE_COMM=1` / `=auto`, see `dalm_amd.sharded.init_distributed`): torch.distributed(nccl) is the default until
a run with two or more real ranks has been recorded.
Bootstrap without a torch.distributed process group: rank 0 asks RCCL for the 128-byte unique id and publishes it
  * through the launcher's file when `DALM_COMM_ID_FILE` is set (`dalm_amd.launch` sets a path unique to the launch), else
  * through a `torch.distributed.TCPStore` on MASTER_ADDR:MASTER_PORT - the store torchrun's agent already hosts
    (TORCHELASTIC_USE_AGENT_STORE) or one rank 0 hosts itself; the key carries the restart count and a per-process
    generation number, so nothing stale can be read (a file under /tmp keyed by the port could survive a crashed job).

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


class Rendezvous:
    """What the ranks have to tell each other BEFORE a communicator exists, over a `torch.distributed.TCPStore` on
    MASTER_ADDR:MASTER_PORT (no process group): the RCCL unique id, and whether every rank's communicator came up - so that
    a fallback to torch.distributed is taken by ALL ranks or by none.  Under torchrun the agent already hosts the store
    (TORCHELASTIC_USE_AGENT_STORE); otherwise rank 0 hosts it and `release()` gives the port back (the fallback's
    init_process_group binds it again).  Keys carry the restart count and a per-process generation number."""

    _generation = 0          # rendezvous objects are constructed in the same order on every rank

    def __init__(self, rank: int, world_size: int, timeout_s: float = 120.0):
        from datetime import timedelta

        from torch.distributed import TCPStore

        Rendezvous._generation += 1
        self.rank, self.world_size = rank, world_size
        self.prefix = f"dalm_comm/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}/{Rendezvous._generation}"
        addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29531"))
        self.agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
        self.store = TCPStore(addr, port, world_size, is_master=(rank == 0 and not self.agent),
                              timeout=timedelta(seconds=timeout_s), wait_for_workers=False)

    def exchange(self, make_payload) -> bytes:
        """Rank 0 publishes `make_payload()` (or the fact that it failed: the others then raise at once instead of waiting
        for the timeout), everybody returns the payload."""
        key = self.prefix + "/uid"
        if self.rank == 0:
            try:
                data = bytes(make_payload())
            except Exception as e:
                self.publish_failure(e)
                raise
            self.store.set(key, b"OK:" + data)
            self.published = True
            return data
        got = bytes(self.store.get(key))             # blocks until rank 0 has published (or the timeout raises)
        if not got.startswith(b"OK:"):
            raise RuntimeError(f"rank 0 could not create the RCCL unique id: {got[7:].decode(errors='replace')}")
        return got[3:]

    published = False

    def publish_failure(self, err: BaseException) -> None:
        """Rank 0 could not produce the payload (or failed before it got there): tell the ranks blocked in `exchange` now
        instead of letting them run into the timeout.  No-op once something has been published."""
        if self.rank == 0 and not self.published:
            self.store.set(self.prefix + "/uid", b"FAILED:" + repr(err).encode()[:200])
            self.published = True

    def agree(self, ok: bool) -> bool:
        """True iff EVERY rank reports ok.  Every rank calls this exactly once.  A rank whose verdict does not arrive within
        the store's timeout counts as failed (the caller then falls back - as will that rank, if it is still alive)."""
        self.store.set(f"{self.prefix}/status/{self.rank}", b"1" if ok else b"0")
        all_ok = True
        for r in range(self.world_size):
            try:
                all_ok &= bytes(self.store.get(f"{self.prefix}/status/{r}")) == b"1"
            except Exception:
                all_ok = False
        self.store.add(self.prefix + "/read", 1)
        if self.rank == 0 and not self.agent:
            # this rank HOSTS the store: do not walk away (a short-lived process would take the server with it) before every
            # rank has read the verdict - a rank that found the store gone used to count that as "failed" and fall back alone
            t0 = time.time()
            while self.store.add(self.prefix + "/read", 0) < self.world_size and time.time() - t0 < 30.0:
                time.sleep(0.005)
        return all_ok

    def release(self) -> None:
        """Rank 0 (when it hosts the store): wait until every rank has read the verdict, then close the server."""
        if self.store is None:
            return
        if self.rank == 0 and not self.agent:
            t0 = time.time()
            while self.store.add(self.prefix + "/read", 0) < self.world_size and time.time() - t0 < 60.0:
                time.sleep(0.01)
        self.store = None
        import gc

        gc.collect()


class NativeRcclComm:
    def __init__(self, rank: Optional[int] = None, world_size: Optional[int] = None, device: Optional[int] = None,
                 unique_id: Optional[bytes] = None, timeout_s: float = 120.0, rendezvous: Optional[Rendezvous] = None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
        self.device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        self.rendezvous = rendezvous
        hip.load()
        if unique_id is None:
            unique_id = self._bootstrap(timeout_s)
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of an ncclUniqueId")
        self._id = C.create_string_buffer(unique_id, 128)
        handle = C.c_void_p()
        hip.call("dalm_comm_init", C.byref(handle), self._id, self.rank, self.world_size, self.device)
        self._h = handle
        torch.cuda.set_device(self.device)

    @staticmethod
    def _new_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        hip.call("dalm_comm_unique_id", buf)
        return buf.raw

    def _bootstrap(self, timeout_s: float) -> bytes:
        if self.rendezvous is None and os.environ.get("DALM_COMM_ID_FILE"):
            return self._bootstrap_file(timeout_s)         # explicit file rendezvous (no agreement channel)
        if self.rendezvous is None:
            self.rendezvous = Rendezvous(self.rank, self.world_size, timeout_s)
        return self.rendezvous.exchange(self._new_unique_id)

    def _bootstrap_file(self, timeout_s: float) -> bytes:
        path = _id_path()
        if self.rank == 0:
            data = self._new_unique_id()
            tmp = path + f".{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(data)
            os.replace(tmp, path)            # atomic publish
            return data
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

    def self_test(self) -> None:
        """One all-reduce and one all-gather with known answers (raises on a wrong result): run once after construction so
        that a broken binding is found before the first training step, while a fallback is still possible."""
        dev = torch.device("cuda", self.device)
        ones = torch.ones(4, device=dev)
        self.all_reduce_sum_(ones)
        got = self.all_gather_rows(torch.full((1, 2), float(self.rank), device=dev))
        torch.cuda.current_stream().synchronize()
        if not torch.equal(ones.cpu(), torch.full((4,), float(self.world_size))):
            raise RuntimeError(f"native all-reduce self-test failed: {ones.tolist()} (world size {self.world_size})")
        want = torch.arange(self.world_size, dtype=torch.float32).unsqueeze(1).expand(-1, 2)
        if not torch.equal(got.cpu(), want):
            raise RuntimeError(f"native all-gather self-test failed: {got.tolist()}")

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
            if self.rank == 0 and os.environ.get("DALM_COMM_ID_FILE"):
                try:
                    os.remove(_id_path())
                except OSError:
                    pass
            if self.rendezvous is not None:        # rank 0 may be hosting the store: give MASTER_PORT back
                try:
                    self.rendezvous.release()
                except Exception:
                    pass
                self.rendezvous = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
