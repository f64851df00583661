"""Diagnostic: for a cfg2-shaped packed retriever batch (queries + passages in ONE encoder call), is there a sequence whose row
count (cu[b + 1] - cu[b]) is <= 0 while its live-tile bytes say otherwise?  (The LDS-DMA attention kernels clamp a row index to
T - 1 and faulted on exactly that.)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from dalm_amd import packed  # noqa: E402

dev = torch.device("cuda:0")
B, Tq, Tp = 150, 50, 128
for seed in range(200, 204):
    g = torch.Generator().manual_seed(seed)
    ql = torch.randint(5, Tq + 1, (B, 1), generator=g)
    pl = torch.randint(Tp // 3, Tp + 1, (B, 1), generator=g)
    b = {"query_input_ids": torch.randint(1000, 30522, (B, Tq), generator=g), "query_attention_mask": (torch.arange(Tq).unsqueeze(0) < ql).long(),
         "passage_input_ids": torch.randint(1000, 30522, (B, Tp), generator=g), "passage_attention_mask": (torch.arange(Tp).unsqueeze(0) < pl).long()}
    b = packed.add_pack_plans(b, packed.RETRIEVER_GROUPS)
    b = {k: v.to(dev) for k, v in b.items()}
    parts, off = [], 0
    for side in ("query", "passage"):
        ids_p, pos, desc, valid = packed.packed_inputs(b[f"{side}_input_ids"], b[f"{side}_attention_mask"], b[f"{side}_pack_rows"], b[f"{side}_pack_cu"], False)
        parts.append((packed.packed_of(desc), off))
        off += int(b[f"{side}_pack_rows"].numel())
    T = max(p[0].T for p in parts)
    cu_all = torch.cat([parts[0][0].cu] + [(p[0].cu[1:] + p[1]) for p in parts[1:]])
    seqs = packed.PackedSeqs(cu=cu_all.to(torch.int32), nseq=int(cu_all.numel()) - 1, T=int(T), n=off,
                             key_live=torch.cat([p[0].key_live for p in parts]), causal=False)
    rows_bits, cols_bits, live = seqs.bits()
    torch.cuda.synchronize()
    W = (T + 31) // 32
    cu = cu_all.cpu()
    lens = (cu[1:] - cu[:-1])
    lv = live.cpu().view(seqs.nseq, W, W)
    anyl = lv.reshape(seqs.nseq, -1).any(1)
    bad = [(int(i), int(lens[i])) for i in range(seqs.nseq) if int(lens[i]) <= 0 and bool(anyl[i])]
    over = [(int(i), int(lens[i])) for i in range(seqs.nseq) if int(lens[i]) > T]
    print(f"seed {seed}: nseq {seqs.nseq}, T {T}, n {off}, cu[-1] {int(cu[-1])}, min len {int(lens.min())}, max len {int(lens.max())}, "
          f"empty-with-live {bad[:5]}, longer-than-T {over[:5]}, cu monotone {bool((lens >= 0).all())}")
