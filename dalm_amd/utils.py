"""Two small host helpers that keep the reference's names and behaviour (dalm/utils.py:8-35)."""
from __future__ import annotations

import os

import torch


def load_dataset(dataset_or_path):
    """Resolve the trainer's `dataset_or_path` argument to a `datasets.Dataset`.

    Accepted, in the reference's order of precedence: an in-memory Dataset object (returned as is), a
    directory written by `Dataset.save_to_disk`, anything else is read as a csv file (its "train" split).
    """
    import datasets as hf_datasets

    source = dataset_or_path
    if isinstance(source, hf_datasets.Dataset):
        return source
    path = os.fspath(source)
    if os.path.isdir(path):
        return hf_datasets.load_from_disk(path)
    return hf_datasets.load_dataset("csv", data_files=path)["train"]


def eos_mask(mask: torch.Tensor, padding: str = "left") -> torch.Tensor:
    """Mask selecting only the LAST real token of every sequence (used to pool autoregressive retrievers).

    padding == "left"  (default, as upstream): sequences end in the last column -> that column is selected.
    padding == "right": the last real token sits at index sum(mask) - 1 of each row.
    Same dtype / shape / device as `mask`.
    """
    rows, cols = mask.shape
    if padding == "right":
        last = (mask.sum(dim=1) - 1) % cols                      # -1 wraps to the last column, like indexing does
    else:
        last = torch.full((rows,), cols - 1, device=mask.device, dtype=torch.long)
    picked = torch.zeros_like(mask)
    picked.scatter_(1, last.to(torch.long).unsqueeze(1), 1)
    return picked
