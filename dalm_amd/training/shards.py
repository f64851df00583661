"""Host data path, second form (SURVEY.md section 8 f, rank 2): pre-tokenised int32 shards, length bucketing and
loss-preserving padding trim.

The reference tokenises the whole dataset at every start (`dataset.map(preprocess_dataset)`,
dalm/training/rag_e2e/train_rage2e.py:298-322), keeps python lists, collates with `default_data_collator` and pads every
row to `max_length` (rag_e2e_dataloader_utils.py:7-68): at hundreds of pairs/s per GPU that is where a trainer starts to
wait for its host.

* `save_token_shards` / `load_token_shards`: the tokenised columns as int32 `.npy` shards (memory-mapped on load) plus an
  `index.json` carrying a fingerprint of everything that determines the tokens; a second run with the same fingerprint
  skips tokenisation.  int32 halves host memory and PCIe bytes; ids/masks become int64 on the device (what the kernels
  and `default_data_collator`-shaped code expect).
* `bucketed_order`: length bucketing - rows are shuffled, cut into mega-chunks, sorted by length inside a chunk, batched,
  and the batches shuffled again: batches hold rows of similar length without making the epoch order deterministic.
* `trim_batch`: drops the all-padding columns of a batch (leading ones of left-padded generator inputs, trailing ones of
  right-padded retriever inputs) in steps of `multiple` tokens and shifts `query_passage_input_len` by the number of
  leading columns removed, which keeps the reference's absolute-index marginalisation rule
  (`train_utils.py:100-103`: rows `t >= qlen-1` get the doc term) on exactly the same tokens: the loss and its gradients
  are unchanged (rotary / relative attention does not see the shift), only padded positions stop costing generator
  flops and CE rows.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

INDEX = "index.json"


def fingerprint(**what) -> str:
    """Stable hash of everything that determines the token ids (tokenizer identities, dataset content fingerprint, column
    names, max lengths, preprocessing version)."""
    blob = json.dumps(dict(what, version=PREPROCESS_VERSION), sort_keys=True, default=str).encode()
    return hashlib.sha256(blob).hexdigest()[:32]


PREPROCESS_VERSION = "dalm_amd-preprocess-1"   # bump when *_dataloader_utils.preprocess_dataset changes what it emits


def tokenizer_identity(tok) -> Dict[str, object]:
    """What makes two tokenizers produce different ids for the same text: name/path, a hash of the whole vocabulary
    (Llama-2 and Mistral are both 32000-token LlamaTokenizerFast; cased / retrained BERT vocabularies collide the same way
    on class name + size), the special tokens, and the padding / eos settings the preprocessing depends on."""
    try:
        vocab = tok.get_vocab()
        h = hashlib.sha256()
        for k, v in sorted(vocab.items(), key=lambda kv: kv[1]):
            h.update(f"{v}:{k}\n".encode("utf-8", "surrogatepass"))
        vocab_hash = h.hexdigest()[:32]
    except Exception:
        vocab_hash = None
    norm = None
    try:                                  # fast tokenizers: normaliser / pre-tokeniser / post-processor live in tokenizer.json
        norm = hashlib.sha256(tok.backend_tokenizer.to_str().encode()).hexdigest()[:32]
    except Exception:
        pass
    return {"class": type(tok).__name__, "name_or_path": getattr(tok, "name_or_path", None), "size": len(tok),
            "vocab": vocab_hash, "pipeline": norm, "special": {k: str(v) for k, v in sorted(tok.special_tokens_map.items())},
            "pad": (tok.pad_token, tok.pad_token_id, getattr(tok, "padding_side", None)),
            "truncation_side": getattr(tok, "truncation_side", None),
            "add_eos_token": getattr(tok, "add_eos_token", None), "add_bos_token": getattr(tok, "add_bos_token", None)}


def dataset_identity(dataset) -> Optional[str]:
    """`datasets` content fingerprint; None when the object has none - callers then do NOT cache (a row count alone would
    silently reuse stale ids for an edited file of the same length)."""
    fp = getattr(dataset, "_fingerprint", None)
    return str(fp) if fp else None


def columns_from_dataset(mapped, columns: Sequence[str]) -> Dict[str, np.ndarray]:
    """The tokenised columns of a `datasets.Dataset` as int32 arrays [N, T] ([N] for scalar columns), read straight from
    the Arrow buffers (`mapped[k]` materialises python lists - 200 000 rows x 256 tokens of python ints take minutes and
    gigabytes - and `with_format("numpy")` builds one small array per row before stacking them: 13 s per 20 000 rows;
    flattening the list column's child buffer takes milliseconds)."""
    import pyarrow as pa

    table = getattr(mapped, "data", None)
    out: Dict[str, np.ndarray] = {}
    if table is None or getattr(mapped, "_indices", None) is not None:     # an index mapping (shuffle/select): slow path
        view = mapped.with_format("numpy", columns=list(columns))
        for k in columns:
            arr = np.asarray(view[k])
            if arr.dtype == object:
                raise ValueError(f"column {k} is ragged: preprocess_dataset pads every row to max_length")
            out[k] = np.ascontiguousarray(arr.astype(np.int32, copy=False))
        return out
    for k in columns:
        parts = []
        for ch in table.column(k).chunks:
            if pa.types.is_list(ch.type) or pa.types.is_large_list(ch.type) or pa.types.is_fixed_size_list(ch.type):
                n = len(ch)
                flat = ch.flatten().to_numpy(zero_copy_only=False)     # flatten() honours the chunk's offset / length
                if n == 0:
                    continue
                if flat.size % n:
                    raise ValueError(f"column {k} is ragged: preprocess_dataset pads every row to max_length")
                width = flat.size // n
                if not pa.types.is_fixed_size_list(ch.type):
                    offs = ch.offsets.to_numpy()
                    if not np.all(np.diff(offs) == width):
                        raise ValueError(f"column {k} is ragged: preprocess_dataset pads every row to max_length")
                parts.append(flat.reshape(n, width))
            else:
                parts.append(ch.to_numpy(zero_copy_only=False))
        arr = np.concatenate(parts, axis=0) if len(parts) != 1 else parts[0]
        out[k] = np.ascontiguousarray(arr.astype(np.int32, copy=False))
    return out


def save_token_shards(columns: Dict[str, Sequence], path: str, fp: str, rows_per_shard: int = 1 << 16) -> None:
    os.makedirs(path, exist_ok=True)
    meta = {"fingerprint": fp, "columns": {}, "rows": None, "rows_per_shard": rows_per_shard, "dtype": "int32"}
    for name, col in columns.items():
        arr = np.asarray(col)
        if arr.ndim == 1:
            arr = arr[:, None]
        if arr.size and (arr.max() > np.iinfo(np.int32).max or arr.min() < np.iinfo(np.int32).min):
            raise ValueError(f"column {name} does not fit int32")
        arr = np.ascontiguousarray(arr.astype(np.int32))
        n = arr.shape[0]
        if meta["rows"] is None:
            meta["rows"] = n
        elif meta["rows"] != n:
            raise ValueError("columns differ in length")
        nsh = max(1, -(-n // rows_per_shard))
        for s in range(nsh):
            np.save(os.path.join(path, f"{name}.{s:05d}.npy"), arr[s * rows_per_shard:(s + 1) * rows_per_shard])
        meta["columns"][name] = {"width": int(arr.shape[1]), "shards": nsh}
    tmp = os.path.join(path, INDEX + ".tmp")
    with open(tmp, "w") as f:
        json.dump(meta, f, indent=1)
    os.replace(tmp, os.path.join(path, INDEX))      # the index appears last: a half-written cache is never used


def load_token_shards(path: str, fp: Optional[str] = None) -> Optional[Dict[str, torch.Tensor]]:
    """int32 tensors [N, T] per column (1-wide columns come back as [N]); None when absent or stale."""
    idx = os.path.join(path, INDEX)
    if not os.path.exists(idx):
        return None
    meta = json.load(open(idx))
    if fp is not None and meta.get("fingerprint") != fp:
        return None
    out = {}
    for name, info in meta["columns"].items():
        parts = [np.load(os.path.join(path, f"{name}.{s:05d}.npy"), mmap_mode="r") for s in range(info["shards"])]
        arr = np.concatenate(parts, axis=0) if len(parts) > 1 else np.ascontiguousarray(parts[0])
        t = torch.from_numpy(np.array(arr, copy=True))
        out[name] = t[:, 0].contiguous() if info["width"] == 1 else t
    return out


def bucketed_order(lengths: torch.Tensor, batch_rows: int, generator: torch.Generator, chunk_batches: int = 64) -> torch.Tensor:
    """A permutation of range(N) in which consecutive groups of `batch_rows` rows have similar `lengths`."""
    n = lengths.numel()
    perm = torch.randperm(n, generator=generator)
    chunk = max(batch_rows * chunk_batches, batch_rows)
    batches: List[torch.Tensor] = []
    tail: List[torch.Tensor] = []
    for c0 in range(0, n, chunk):
        rows = perm[c0:c0 + chunk]
        rows = rows[torch.argsort(lengths[rows], stable=True)]
        full = (rows.numel() // batch_rows) * batch_rows
        batches.extend(rows[:full].split(batch_rows))
        if full < rows.numel():
            tail.append(rows[full:])
    order = torch.randperm(len(batches), generator=generator).tolist() if batches else []
    out = [batches[i] for i in order] + tail          # leftovers (partial batches) go last, as without bucketing
    return torch.cat(out) if out else perm


def _pad_span(mask: torch.Tensor) -> Tuple[int, int]:
    """(first, last+1) column with any non-zero mask entry over the batch; (0, T) for an all-padding batch."""
    cols = (mask != 0).any(dim=0)
    nz = torch.nonzero(cols).flatten()
    if nz.numel() == 0:
        return 0, mask.shape[1]
    return int(nz[0]), int(nz[-1]) + 1


def trim_batch(batch: Dict[str, torch.Tensor], groups: Iterable[Tuple[str, str]], qlen_key: Optional[str] = None,
               qlen_follows: Optional[str] = None, multiple: int = 8, min_len: int = 8) -> Dict[str, torch.Tensor]:
    """Drop all-padding columns.  `groups`: (ids_key, mask_key) pairs sharing one time axis; `qlen_key` is shifted by the
    leading columns removed from the `qlen_follows` mask's axis.  Works on host or device tensors (one tiny reduction +
    two scalars per group; call it on the host copy, before the H2D, to keep the step free of syncs).

    Exactness: every loss row whose input position is a real token is unchanged.  With LEFT-padded generator inputs each
    sample also has one live row computed AT a padding position (the row that predicts its first token, m_bt = mask[b,t+1]):
    a fully masked query row attends uniformly over however many padding keys exist, so removing leading pads changes that
    row's hidden state - B rows per batch are approximate (measured 2e-5 relative on the step loss with rotary generators,
    tests/test_step_parity_gpu.py).  Generators with ABSOLUTE position embeddings (gpt2-style) see every token at a
    different position after a left trim: use trimming with rotary / relative-position generators only (Llama, Falcon,
    Mistral); right-padded inputs (the retriever side) are exact."""
    out = dict(batch)
    for ids_key, mask_key in groups:
        mask = batch[mask_key]
        T = mask.shape[1]
        lo, hi = _pad_span(mask)
        # keep at least ONE leading padding column when there is one: with left-padded generator inputs the reference's
        # shifted-label loss (train_utils.py:121-138) has a live row that predicts a sample's FIRST token from the pad
        # position before it (m_bt = mask[b,t+1]); removing every leading pad of the longest sample would drop that term
        lo = ((lo - 1) // multiple) * multiple if lo > 0 else 0   # and a multiple of `multiple`: GEMM-friendly shapes
        hi = min(T, -(-hi // multiple) * multiple)
        if hi - lo < min_len:
            lo = max(0, min(lo, T - min_len))
            hi = min(T, lo + min_len)
        if lo == 0 and hi == T:
            continue
        for k in (ids_key, mask_key):
            out[k] = batch[k][:, lo:hi].contiguous()
        if qlen_key is not None and qlen_follows == mask_key and lo > 0:
            # rows t >= qlen-1 (absolute, shifted coordinates) get the doc term: same tokens after removing `lo` columns.
            # A cut that fell inside the removed padding means "every remaining row": qlen 1.
            out[qlen_key] = torch.clamp(batch[qlen_key] - lo, min=1)
    return out
