"""The step's compute streams, created (and used once) BEFORE any communicator exists.

HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues when they first submit work.  The step runs the generator on the current
stream and the retriever towers on a side stream; with a live RCCL communicator (its own streams submit first) the two could end up
on ONE hardware queue and serialise: 140.9 against 118.5 ms per cfg3 step with one rank and the runtime's 4 queues (tools/queue_ab.sh).
Claiming the queues for the two compute streams first makes the mapping independent of what the communicator creates afterwards."""
from __future__ import annotations

from typing import Dict, Optional

import torch

_tower: Dict[int, "torch.cuda.Stream"] = {}


def tower_stream(device: Optional[int] = None) -> "torch.cuda.Stream":
    """The side stream of the retriever towers of `device` (one per device and process)."""
    idx = torch.cuda.current_device() if device is None else int(device)
    s = _tower.get(idx)
    if s is None:
        s = _tower[idx] = torch.cuda.Stream(device=idx)
    return s


def claim_compute_queues(device: Optional[int] = None) -> None:
    """One tiny launch on the current stream and on the tower stream, then a device sync: both have their hardware queue."""
    if not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None else int(device)
    with torch.cuda.device(idx):
        a = torch.zeros(64, device="cuda")
        a.add_(1.0)
        s = tower_stream(idx)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            b = torch.zeros(64, device="cuda")
            b.add_(1.0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
