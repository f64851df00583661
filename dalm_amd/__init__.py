"""dalm_amd: the RAG-end2end training-step loss path of arcee-ai/DALM as hand-written HIP for MI355X (gfx950)."""
import logging
import os

# HIP maps streams onto a small set of hardware queues.  The step runs the retriever towers and the generator
# on two streams; with the default of 4 queues and a live RCCL process group (its own streams) the two landed on
# one queue and the overlap was lost (214 ms/step); with 3 queues they do not (180 ms), and the single-GPU step is
# unchanged (181 ms).  Measured on MI355X / ROCm 7.0 runtime; must be set before the HIP runtime initialises,
# an explicit user setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")

__version__ = "0.1.0"

logging.getLogger(__name__).addHandler(logging.NullHandler())
