"""Row -> fixed-length token ids for the RAG-e2e trainer; batch layout identical to the reference's
dalm/training/utils/rag_e2e_dataloader_utils.py:7-68 (including its quirks: the "#query#"/"#passage#"
tags end up doubled in the generator prompt, and `query_passage_input_len` is the UN-truncated token
count of the prompt including specials)."""
from __future__ import annotations

from typing import Any, Dict


def preprocess_dataset(examples, retriever_tokenizer, generator_tokenizer, query_column_name: str,
                       passage_column_name: str, answer_column_name: str, query_max_len: int,
                       passage_max_len: int, generator_max_len: int) -> Dict[str, Any]:
    tagged_q = ["#query# " + str(q) for q in examples[query_column_name]]
    tagged_p = ["#passage# " + str(p) for p in examples[passage_column_name]]
    answers = list(examples[answer_column_name])
    if not (len(tagged_q) == len(tagged_p) == len(answers)):
        raise ValueError("zip() argument lengths differ: query / passage / answer columns")

    def fixed(tok, texts, n):
        return tok(texts, padding="max_length", max_length=n, truncation=True)

    out: Dict[str, Any] = {}
    for prefix, enc in (("retriever_query_", fixed(retriever_tokenizer, tagged_q, query_max_len)),
                        ("retriever_passage_", fixed(retriever_tokenizer, tagged_p, passage_max_len))):
        for k, v in enc.items():
            out[prefix + k] = v

    # generator prompt is built from the already-tagged strings (hence "#query# #query# ...")
    prompts = [f"#query# {q} #passage# {p} #answer#" for q, p in zip(tagged_q, tagged_p)]
    full = [f"{pr} {a}" for pr, a in zip(prompts, answers)]
    for k, v in fixed(generator_tokenizer, full, generator_max_len).items():
        out["generator_input_" + k] = v
    out["query_passage_input_len"] = [len(x) for x in generator_tokenizer(prompts, padding=False)["input_ids"]]
    return out
