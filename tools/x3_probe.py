"""The bf16x3 similarity row statistics alone (split + 256x256x64 bf16 MFMA kernel + merge), a few launches per size, for
rocprofv3 passes (kernel-trace durations, MfmaUtil, FETCH_SIZE / WRITE_SIZE):
    python tools/x3_probe.py 4096 16384
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from dalm_amd.ops import default_ops  # noqa: E402

ops = default_ops()
dev = torch.device("cuda:0")
for size in [int(x) for x in sys.argv[1:]] or [4096, 16384]:
    A = torch.nn.functional.normalize(torch.randn(size, 1024, device=dev), dim=1)
    B = torch.nn.functional.normalize(torch.randn(size, 1024, device=dev), dim=1)
    for _ in range(4):
        ops.sim_rowstats_bf16x3(A, B, 100.0, 0)
    torch.cuda.synchronize()
print("done")
