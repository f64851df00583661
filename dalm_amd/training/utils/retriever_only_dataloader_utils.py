"""Row -> fixed-length ids for the retriever-only trainer (reference:
dalm/training/utils/retriever_only_dataloader_utils.py:8-27; keys query_* / passage_*)."""
from __future__ import annotations

from typing import Any, Dict


def preprocess_dataset(examples, tokenizer, query_column_name: str, passage_column_name: str, query_max_len: int,
                       passage_max_len: int) -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for prefix, tag, col, n in (("query_", "#query# ", query_column_name, query_max_len),
                                ("passage_", "#passage# ", passage_column_name, passage_max_len)):
        enc = tokenizer([tag + str(t) for t in examples[col]], padding="max_length", max_length=n, truncation=True)
        for k, v in enc.items():
            out[prefix + k] = v
    return out
