"""bench.f1_head_paths on a depth-1 cfg3 model, alone (for rocprofv3 kernel traces of the three lm_head + CE training paths)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from dalm_amd.tuning import enable_tuned_gemms  # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
gen = sys.argv[1] if len(sys.argv) > 1 else "llama-2-7b"
model = bench.build_models(dev, torch.bfloat16, 1, 1, generator=gen)
b = bench.synthetic_batch(dev, 100, V=bench.GENERATORS[gen][1])
r = bench.f1_head_paths(dev, b, model)
r.pop("note", None)
print(r)
