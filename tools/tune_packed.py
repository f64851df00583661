"""GEMM solution tuning (PyTorch TunableOp, hipBLASLt / rocBLAS) for the PACKED row counts the trainers can produce.

The packed tower path (dalm_amd/packed.py) runs every projection on n_pad live rows, n_pad a multiple of 256 (generator, query
tower) / 512 (passage tower) - a data-dependent GEMM M.  The replay-only table dalm_amd/tuning/tunableop_gfx950.csv is looked up
by exact shape, so this tool steps depth-1 towers of the real widths (same GEMM shapes as the full-depth models) through
`RagE2EStep` on synthetic batches whose live-token counts sweep those multiples, with tuning ON, and writes the solutions found to
gpurun_out/tunableop_packed.csv; `--merge` folds a results file into the package's table (union, new rows win).

    python tools/tune_packed.py [--generator llama-2-7b|falcon-7b] [--rows 1536,...,4608]
    python tools/tune_packed.py --merge gpurun_out/tunableop_packed.csv
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
TABLE = ROOT / "dalm_amd" / "tuning" / "tunableop_gfx950.csv"


def merge(src: str) -> None:
    def rows(path):
        val, ent = [], {}
        for line in Path(path).read_text().splitlines():
            if not line.strip():
                continue
            if line.startswith("Validator"):
                val.append(line)
            else:
                parts = line.split(",")
                ent[(parts[0], parts[1])] = line
        return val, ent

    v0, e0 = rows(TABLE)
    v1, e1 = rows(src)
    if v0 != v1:
        print("validator rows differ:\n ", v0, "\n ", v1)
        raise SystemExit("refusing to merge tables of different library versions")
    before = len(e0)
    e0.update(e1)
    TABLE.write_text("\n".join(v0 + [e0[k] for k in sorted(e0)]) + "\n")
    print(f"{TABLE}: {before} -> {len(e0)} solutions")


def batch_with_rows(B, Tq, Tp, Tg, V, n_gen, n_q, n_p, seed):
    """A synthetic batch whose packed row counts round up to (n_gen, n_q, n_p)."""
    import torch

    g = torch.Generator().manual_seed(seed)

    def lens_for(total, T, extra_per_row):
        total = min(total, B * T)
        per = max(1, min(T, (total - 8) // B - extra_per_row))
        lens = torch.full((B,), per, dtype=torch.long)
        if total >= B * T:
            lens[:] = T
        return lens

    gl = lens_for(n_gen, Tg, 1)
    ql, pl = lens_for(n_q, Tq, 0), lens_for(n_p, Tp, 0)
    ar = torch.arange
    return {
        "retriever_query_input_ids": torch.randint(1000, 30522, (B, Tq), generator=g),
        "retriever_query_attention_mask": (ar(Tq).unsqueeze(0) < ql.unsqueeze(1)).long(),
        "retriever_passage_input_ids": torch.randint(1000, 30522, (B, Tp), generator=g),
        "retriever_passage_attention_mask": (ar(Tp).unsqueeze(0) < pl.unsqueeze(1)).long(),
        "generator_input_input_ids": torch.randint(1000, V, (B, Tg), generator=g),
        "generator_input_attention_mask": (ar(Tg).unsqueeze(0) >= (Tg - gl).unsqueeze(1)).long(),
        "query_passage_input_len": (gl.float() * 0.8).long().clamp(min=1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--merge", default=None)
    ap.add_argument("--generator", default="llama-2-7b")
    ap.add_argument("--rows", default=",".join(str(r) for r in range(1536, 4608 + 1, 256)))
    ap.add_argument("--out", default="gpurun_out/tunableop_packed.csv")
    ap.add_argument("--duration-ms", type=int, default=20)
    args = ap.parse_args()
    if args.merge:
        return merge(args.merge)
    os.environ["DALM_TUNED_GEMMS"] = "0"
    import torch
    import torch.cuda.tunable as tunable

    import bench
    from dalm_amd import packed
    from dalm_amd.training.step import RagE2EStep

    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    tunable.enable(True)
    if TABLE.exists():
        tunable.read_file(str(TABLE))                     # shapes already in the table are not searched again
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(args.duration_ms)
    tunable.set_filename(args.out, insert_device_ordinal=False)
    dev = torch.device("cuda:0")
    V = bench.GENERATORS[args.generator][1]
    model = bench.build_models(dev, torch.bfloat16, 1, 1, generator=args.generator)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    step = RagE2EStep(model, torch.optim.Adam(params, lr=1e-5, fused=True), None, 100, autocast_dtype=torch.bfloat16, inplace_grad=True,
                      overlap_towers=False)
    B, Tq, Tp, Tg = bench.CFG["B"], bench.CFG["Tq"], bench.CFG["Tp"], bench.CFG["Tg"]
    mult = {k: packed.ROW_MULTIPLES[k] for k in ("generator", "retriever_query", "retriever_passage")}
    gens = [int(r) for r in args.rows.split(",")]
    qs = [256, 512, 768, 1024]
    ps = [512, 1024, 1536, 2048, 2304]
    n = max(len(gens), len(qs), len(ps))
    seen = set()
    for i in range(n):
        b = batch_with_rows(B, Tq, Tp, Tg, V, gens[i % len(gens)], qs[i % len(qs)], ps[i % len(ps)], i)
        b = packed.add_pack_plans(b, multiple=mult)
        shape = tuple(int(b[f"{k}_pack_rows"].numel()) for k in mult)
        seen.add(shape)
        step({k: v.to(dev) for k, v in b.items()})
        torch.cuda.synchronize()
        print("tuned packed rows", dict(zip(mult, shape)), flush=True)
    # the padded shapes of the same step (concatenated gate | up forward included)
    step({k: v.to(dev) for k, v in bench.synthetic_batch(torch.device("cpu"), 0, V=V).items()})
    torch.cuda.synchronize()
    # the results file is written when the process exits (TunableOp: write_file_on_exit)
    print("written", args.out, "shapes", sorted(seen))


if __name__ == "__main__":
    main()
