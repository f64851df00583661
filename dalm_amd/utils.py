"""Small helpers with the reference's names (dalm/utils.py:8-35)."""
from __future__ import annotations

import os

import torch


def load_dataset(dataset_or_path):
    """Dataset object, `save_to_disk` directory, or csv path -> datasets.Dataset (reference :8-19)."""
    import datasets

    if isinstance(dataset_or_path, datasets.Dataset):
        return dataset_or_path
    if os.path.isdir(dataset_or_path):
        return datasets.load_from_disk(dataset_or_path)
    return datasets.load_dataset("csv", data_files=dataset_or_path)["train"]


def eos_mask(mask: torch.Tensor, padding: str = "left") -> torch.Tensor:
    """One-hot mask on each sequence's last token (reference :22-35): last column under left
    padding, position sum(mask)-1 under right padding."""
    out = torch.zeros_like(mask)
    if padding == "right":
        last = mask.sum(dim=1) - 1
        out[torch.arange(mask.size(0), device=mask.device), last] = 1
    else:
        out[:, -1] = 1
    return out
