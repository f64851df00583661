"""Pre-tuned GEMM solution table for the transformer towers (PyTorch TunableOp, replay only).

The towers run on PyTorch-ROCm; their GEMMs go to hipBLASLt / rocBLAS.  `tunableop_gfx950.csv` holds the
fastest library solution per GEMM shape of the cfg3 workload (bge-large + Llama-2-7b, batch 18), found once
on an MI355X with `tools/tune_gemms.sh` (~3 min).  `enable_tuned_gemms()` loads it with tuning switched OFF:
shapes in the table use the recorded solution, every other shape uses the library default; nothing is tuned
at run time.  The table carries validator rows (torch / HIP / hipBLASLt / rocBLAS versions, gfx arch); on a
mismatch TunableOp ignores it and everything falls back to the defaults.
Measured on cfg3 (same box, hipGraph replay): 233 ms/step -> 203 ms/step.
"""
from __future__ import annotations

import os
import tempfile
from pathlib import Path

TABLE = Path(__file__).resolve().parent / "tunableop_gfx950.csv"


def enable_tuned_gemms(table: os.PathLike | str | None = None) -> bool:
    """Returns True if the table was loaded.  Never raises: tuning tables are an optimisation only."""
    if os.environ.get("DALM_TUNED_GEMMS", "1") == "0":
        return False
    try:
        import torch
        import torch.cuda.tunable as tunable

        if not torch.cuda.is_available():
            return False
        path = str(table or TABLE)
        if not os.path.exists(path):
            return False
        tunable.enable(True)
        tunable.tuning_enable(False)          # replay only
        ok = tunable.read_file(path)
        # anything TunableOp writes at exit goes to a scratch file, never into the package
        scratch = os.path.join(tempfile.gettempdir(), f"dalm_tunableop_{os.getpid()}.csv")
        tunable.set_filename(scratch, insert_device_ordinal=False)
        return bool(ok) if ok is not None else True
    except Exception:
        return False
