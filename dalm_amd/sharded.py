"""Multi-GPU plumbing for the sharded in-batch negatives (one process per GPU, RCCL over xGMI).

What crosses GPUs per step (W ranks, B_l rows each, D = embedding width):
  * all-gather of passage embeddings  [B_l, D] f32   - started right after the passage tower on a
    side HIP stream so it overlaps the query tower's forward (GatherHandle)
  * all-gather of query embeddings    [B_l, D] f32
  * all-reduce of the token count M   1 float
  * all-gather of (lse_r, lse_c, a, b) [B_l, 4] f32   in the backward
  * all-reduce (SUM) of the trainable-parameter gradients, one flat bucket (LoRA: ~21 MB)
Nothing else: the backward is closed-form, so no autograd graph spans a collective and no
reduce-scatter of dP is needed (each rank recomputes its own column block on the matrix cores).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch

from .fused import GatherHandle, LocalComm, TorchDistComm  # noqa: F401


def init_distributed(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment; returns (comm, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # DALM_FORCE_DIST=1 runs the RCCL code path even with one rank (used to smoke-test the collectives,
    # side-stream gathers and the gradient bucket on a single-GPU box)
    if world == 1 and os.environ.get("DALM_FORCE_DIST", "0") != "1":
        dev = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        return LocalComm(), dev
    import torch.distributed as dist

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        backend = backend or "nccl"  # == RCCL on ROCm
    else:
        dev = torch.device("cpu")
        backend = backend or "gloo"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend=backend, device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    return TorchDistComm(), dev


def allreduce_grads(params: Iterable[torch.nn.Parameter], comm) -> None:
    """SUM the gradients over ranks through one flat bucket (rank losses are shares of the
    global-batch loss, so summing reproduces the single-process gradient)."""
    if isinstance(comm, LocalComm):
        return
    ps: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
    comm.all_reduce_sum_(flat)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


class GradBucket:
    """All trainable gradients as views of ONE flat fp32 buffer (what DDP calls gradient_as_bucket_view):
    backward accumulates straight into the bucket, the all-reduce runs on the bucket, the fused optimizer
    reads the views - no per-step flatten / ~400 tiny copy-back launches.  Grads are zeroed, never set to None."""

    def __init__(self, params: Iterable[torch.nn.Parameter], comm):
        self.comm = comm
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev = self.params[0].device
        if any(p.dtype != torch.float32 or p.device != dev for p in self.params):
            raise ValueError("GradBucket needs fp32 trainable parameters on one device")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self) -> None:
        self.flat.zero_()
        off = 0
        for p in self.params:  # re-attach in case something replaced .grad (e.g. set_to_none)
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def all_reduce(self) -> None:
        if not isinstance(self.comm, LocalComm):
            self.comm.all_reduce_sum_(self.flat)


def barrier(comm) -> None:
    if not isinstance(comm, LocalComm):
        import torch.distributed as dist

        dist.barrier()
