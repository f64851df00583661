"""SURVEY 8(f) rank 2, measured on the host: time to produce one training batch (host side, before the H2D copy)
  (a) the reference's way - a tokenised `datasets.Dataset` (python lists per row) through torch's DataLoader with
      transformers' default_data_collator (dalm/training/rag_e2e/train_rage2e.py:328-334), batch 18, shuffle;
  (b) `dalm_amd.training.common.ShardedBatches` - columns as contiguous int32 tensors, one index_select per column;
  (c) (b) + padding trim, (d) (b) + the live-row list of the fused lm_head path.
Synthetic cfg3-shaped rows (Tq 50 / Tp 128 / Tg 256).   python tools/data_path_bench.py [rows]
"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    B, Tq, Tp, Tg = 18, 50, 128, 256
    g = torch.Generator().manual_seed(0)

    def col(T, V):
        return torch.randint(1000, V, (n, T), generator=g)

    def right(T, lo):
        return (torch.arange(T).unsqueeze(0) < torch.randint(lo, T + 1, (n, 1), generator=g)).long()

    glen = torch.randint(60, Tg + 1, (n, 1), generator=g)
    data = {"retriever_query_input_ids": col(Tq, 30522), "retriever_query_attention_mask": right(Tq, 5),
            "retriever_passage_input_ids": col(Tp, 30522), "retriever_passage_attention_mask": right(Tp, 30),
            "generator_input_input_ids": col(Tg, 32000),
            "generator_input_attention_mask": (torch.arange(Tg).unsqueeze(0) >= (Tg - glen)).long(),
            "query_passage_input_len": (glen.float() * 0.8).long()}
    cols = list(data)
    dev = torch.device("cpu")
    print(f"{n} synthetic rows, batch {B}, host: {torch.get_num_threads()} torch threads")

    # (a) the reference's loader
    import datasets
    from torch.utils.data import DataLoader
    from transformers import default_data_collator

    ds = datasets.Dataset.from_dict({k: (v.squeeze(1) if v.shape[1] == 1 else v).tolist() for k, v in data.items()})
    dl = DataLoader(ds, shuffle=True, collate_fn=default_data_collator, batch_size=B)
    t0 = time.perf_counter()
    nb = 0
    for batch in dl:
        nb += 1
        if nb == 300:
            break
    ta = (time.perf_counter() - t0) / nb
    print(f"  (a) datasets.Dataset + DataLoader + default_data_collator   {ta * 1e3:8.3f} ms / batch")

    from dalm_amd.training.common import ShardedBatches

    def run(label, **kw):
        sb = ShardedBatches({k: (v.squeeze(1) if v.shape[1] == 1 else v) for k, v in data.items()}, B, 0, 1, 0, cols, **kw)
        t0 = time.perf_counter()
        k = 0
        for batch in sb.epoch(0, dev):
            k += 1
            if k == 1000:
                break
        t = (time.perf_counter() - t0) / k
        print(f"  {label:62s} {t * 1e3:8.3f} ms / batch   ({ta / t:5.1f} x)")

    run("(b) ShardedBatches (int32 columns, index_select)")
    run("(c) (b) + trim_padding", trim=dict(groups=[("retriever_query_input_ids", "retriever_query_attention_mask"),
                                                    ("retriever_passage_input_ids", "retriever_passage_attention_mask"),
                                                    ("generator_input_input_ids", "generator_input_attention_mask")],
                                            qlen_key="query_passage_input_len", qlen_follows="generator_input_attention_mask"))
    run("(d) (b) + live-row list for the fused lm_head path", live_rows=dict(mask="generator_input_attention_mask", multiple=512))


if __name__ == "__main__":
    main()
