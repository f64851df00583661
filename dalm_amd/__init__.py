"""dalm_amd: the RAG-end2end training-step loss path of arcee-ai/DALM as hand-written HIP for MI355X (gfx950)."""
import logging

__version__ = "0.1.0"

logging.getLogger(__name__).addHandler(logging.NullHandler())
