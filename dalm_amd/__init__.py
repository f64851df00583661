"""dalm_amd: the RAG-end2end training-step loss path of arcee-ai/DALM as hand-written HIP for MI355X (gfx950)."""
import logging
import os

__version__ = "0.2.0"

logging.getLogger(__name__).addHandler(logging.NullHandler())


def configure_hw_queues(world_size=None) -> str:
    """Opt-in hardware-queue setting for processes that keep an RCCL communicator alive.

    HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The step runs the retriever towers and
    the generator on two streams; measured on MI355X with ONE rank and a live RCCL process group, RCCL's own streams
    shifted the mapping so that the two compute streams shared a queue (214 ms/step at 4 queues, 180 ms at 3; without
    a process group 181 ms at 3 or 4).  That is the only configuration the evidence covers, so nothing is changed
    at import any more: entry points (bench.py, the trainers) call this before the HIP runtime starts, it applies 3
    only when a process group will exist (world_size > 1 or DALM_FORCE_DIST=1), an explicit GPU_MAX_HW_QUEUES always
    wins, and DALM_HW_QUEUES=<n> forces a value (0 = leave the runtime default).  Returns what was decided."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return "user:" + os.environ["GPU_MAX_HW_QUEUES"]
    forced = os.environ.get("DALM_HW_QUEUES")
    if forced is not None:
        if forced not in ("", "0"):
            os.environ["GPU_MAX_HW_QUEUES"] = forced
            return "DALM_HW_QUEUES:" + forced
        return "runtime-default"
    if world_size is None:
        world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size > 1 or os.environ.get("DALM_FORCE_DIST", "0") == "1":
        os.environ["GPU_MAX_HW_QUEUES"] = "3"
        return "rccl-alive:3"
    return "runtime-default"
