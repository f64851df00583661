"""The dominant loss kernel of bench.py (marginalised CE, fused forward + gradient) launched alone at the bench's exact
shapes and masks, for the PMC passes bench.py runs around it (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE cannot share a pass
and perturb timing, so they never run inside the timed region).
    python tools/ce_traffic_probe.py [--workload cfg3|cfg5] [--dtype bf16|fp32]
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

import bench  # noqa: E402
from dalm_amd.ops import default_ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    V = bench.GENERATORS["falcon-7b" if a.workload == "cfg5" else "llama-2-7b"][1]
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    ops = default_ops()
    g = torch.Generator(device="cpu").manual_seed(0)
    for i in range(4):                                   # the four batches bench.py cycles through (rank 0)
        b = bench.synthetic_batch(dev, 100 + i, V=V)
        ids, mask = b["generator_input_input_ids"], b["generator_input_attention_mask"]
        logits = torch.randn(ids.shape[0], ids.shape[1], V, generator=g).to(dev, dt)
        stats, _, _ = ops.ce_prep(mask, b["query_passage_input_len"])
        for _ in range(3):
            ops.ce_fwd(logits.clone(), ids, mask, stats, True, True)
    torch.cuda.synchronize()
    print("probe done")


if __name__ == "__main__":
    main()
