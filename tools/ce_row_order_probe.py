"""does the ORDER in which workgroups take rows, or the cache policy of the zero fill, move the fused
CE kernel (marginalised CE, forward + gradient in one pass, bf16, in place) at the bench's masks?

37 % of the rows of a bench batch are padding: write-only workgroups (zeros).  With row = blockIdx.x a left-padded sample
dispatches ~98 write-only workgroups in a run, then ~158 read+write ones.  Knobs (read once per process by ce.hip):
    DALM_CE_ORDER = 0 | i | <stride>      identity | sample-interleaved | t = (i * stride) % Tg inside a sample
    DALM_CE_FILL  = c | n                 cached | non-temporal stores for the zero rows

    python tools/ce_row_order_probe.py --sweep            # one subprocess per combination -> table on stdout
    python tools/ce_row_order_probe.py                    # this process's environment, one line
"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def one(workload: str, iters: int) -> dict:
    import torch

    import bench
    from dalm_amd.ops import default_ops

    dev = torch.device("cuda:0")
    V = bench.GENERATORS["falcon-7b" if workload == "cfg5" else "llama-2-7b"][1]
    ops = default_ops()
    g = torch.Generator(device="cpu").manual_seed(0)
    out = {}
    for tag in ("bench_masks", "all_ones"):
        times, bytes_ = [], []
        for i in range(4):
            b = bench.synthetic_batch(dev, 100 + i, V=V)
            ids, mask = b["generator_input_input_ids"], b["generator_input_attention_mask"]
            if tag == "all_ones":
                mask = torch.ones_like(mask)
            logits = torch.randn(ids.shape[0], ids.shape[1], V, generator=g).to(dev, torch.bfloat16)
            stats, _, _ = ops.ce_prep(mask, b["query_passage_input_len"])
            live = int((mask[:, 1:] != 0).sum())
            for _ in range(5):
                ops.ce_fwd(logits, ids, mask, stats, True, True)
            torch.cuda.synchronize()
            evs = []
            for _ in range(iters):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ops.ce_fwd(logits, ids, mask, stats, True, True, events=(a, e))
                evs.append((a, e))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(e) * 1e-3 for a, e in evs)
            times.append(ts[len(ts) // 2])
            bytes_.append((live + ids.shape[0] * ids.shape[1]) * V * 2)
        t = sum(times) / len(times)
        out[tag] = {"us": 1e6 * t, "frac": (sum(bytes_) / len(bytes_)) / t / 8e12}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--sweep", action="store_true")
    a = ap.parse_args()
    if not a.sweep:
        print(json.dumps(one(a.workload, a.iters)))
        return
    print(f"fused CE fwd+grad, bf16, in place, {a.workload}; median of {a.iters} launches per batch (HIP events), mean over the "
          "bench's 4 batches; frac = algorithmic bytes (live rows read + all rows written) / time / 8 TB/s")
    print(f"{'order':>8} {'fill':>6} | {'bench masks us':>15} {'frac':>6} | {'all-ones us':>12} {'frac':>6}")
    for rep in range(2):                       # two passes: the spread between them is the noise floor
        for order in ("0", "i", "97", "37", "129"):
            for fill in ("c", "n"):
                env = dict(os.environ, DALM_CE_ORDER=order, DALM_CE_FILL=fill)
                r = subprocess.run([sys.executable, __file__, "--workload", a.workload, "--iters", str(a.iters)], env=env,
                                   capture_output=True, text=True, timeout=600)
                line = [x for x in r.stdout.splitlines() if x.startswith("{")]
                if not line:
                    print(f"{order:>8} {fill:>6} | failed: {r.stderr[-200:]}")
                    continue
                d = json.loads(line[-1])
                print(f"{order:>8} {fill:>6} | {d['bench_masks']['us']:15.2f} {d['bench_masks']['frac']:6.3f} | "
                      f"{d['all_ones']['us']:12.2f} {d['all_ones']['frac']:6.3f}", flush=True)


if __name__ == "__main__":
    main()
