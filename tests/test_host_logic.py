"""Host-side parity with the reference (no GPU): CLI flags/defaults, train_* signatures, preprocess_dataset
outputs (golden dumped from the reference by oracle/make_golden.py), eos_mask, batching, LoRA injector."""
import inspect
import json
from pathlib import Path

import pytest
import torch

GOLD = json.loads((Path(__file__).parent / "golden" / "host_golden.json").read_text())


def _tok():
    from transformers import PreTrainedTokenizerFast

    return PreTrainedTokenizerFast.from_pretrained(str(Path(__file__).parent / "golden" / "wordlevel_tokenizer"))


def test_e2e_cli_defaults_match_reference():
    from dalm_amd.training.rag_e2e.train_rage2e import parse_args

    ns = vars(parse_args(["--retriever_name_or_path", "R", "--generator_name_or_path", "G"]))
    for k, v in GOLD["e2e_defaults"].items():
        got = ns[k].value if hasattr(ns[k], "value") else ns[k]
        assert got == v, (k, got, v)


def test_retriever_cli_defaults_match_reference():
    from dalm_amd.training.retriever_only.train_retriever_only import parse_args

    ns = vars(parse_args(["--retriever_name_or_path", "R"]))
    for k, v in GOLD["retriever_defaults"].items():
        assert ns[k] == v, (k, ns[k], v)


def test_cli_mode_flags():
    from dalm_amd.models import Mode
    from dalm_amd.training.rag_e2e.train_rage2e import parse_args

    a = parse_args(["--retriever_name_or_path", "R", "--generator_name_or_path", "G", "--use_peft", "both",
                    "--with_tracking", "--checkpointing_steps", "epoch"])
    assert a.use_peft is Mode.BOTH and a.with_tracking and a.checkpointing_steps == "epoch"
    with pytest.raises(SystemExit):
        parse_args(["--generator_name_or_path", "G"])  # retriever is required, as upstream


@pytest.mark.parametrize("which", ["train_e2e", "train_retriever"])
def test_train_function_signatures_match_reference(which):
    if which == "train_e2e":
        from dalm_amd.training.rag_e2e.train_rage2e import train_e2e as fn
    else:
        from dalm_amd.training.retriever_only.train_retriever_only import train_retriever as fn
    ref = GOLD[f"{which}_signature"]
    params = [(n, p) for n, p in inspect.signature(fn).parameters.items() if p.kind is not p.KEYWORD_ONLY]
    assert [n for n, _ in params] == [n for n, _ in ref]
    for (n, p), (_, d) in zip(params, ref):
        if d == "<required>":
            assert p.default is inspect._empty, n
        else:
            got = p.default.value if hasattr(p.default, "value") else p.default
            assert got == d, (n, got, d)


def test_preprocess_rag_e2e_matches_reference():
    from dalm_amd.training.utils.rag_e2e_dataloader_utils import preprocess_dataset

    tok = _tok()
    got = preprocess_dataset(GOLD["rows"], tok, tok, "Question", "Abstract", "Answer", 12, 24, 40)
    assert set(got) == set(GOLD["pre_e2e"])
    for k, v in GOLD["pre_e2e"].items():
        assert [list(r) if isinstance(r, (list, tuple)) else r for r in got[k]] == v, k


def test_preprocess_retriever_matches_reference():
    from dalm_amd.training.utils.retriever_only_dataloader_utils import preprocess_dataset

    tok = _tok()
    got = preprocess_dataset(GOLD["rows"], tok, "Question", "Abstract", 12, 24)
    assert set(got) == set(GOLD["pre_ret"])
    for k, v in GOLD["pre_ret"].items():
        assert [list(r) for r in got[k]] == v, k


def test_eos_mask_matches_reference_golden():
    import numpy as np

    from dalm_amd.utils import eos_mask

    z = np.load(Path(__file__).parent / "golden" / "pool_base.npz")
    mask = torch.from_numpy(z["mask"])
    assert torch.equal(eos_mask(mask), torch.from_numpy(z["eos_mask_left"]))
    assert torch.equal(eos_mask(mask, padding="right"), torch.from_numpy(z["eos_mask_right"]))


def test_sharded_batches_cover_each_row_once():
    import datasets

    from dalm_amd.training.common import ShardedBatches

    ds = datasets.Dataset.from_dict({"a": [[i, i] for i in range(23)]})
    one = ShardedBatches(ds, 4, 0, 1, 0, ["a"])
    assert len(one) == 6  # ceil(23/4): partial last batch like the reference's DataLoader
    seen = [int(r[0]) for b in one.epoch(0, torch.device("cpu")) for r in b["a"]]
    assert sorted(seen) == list(range(23))
    # two ranks: same permutation, disjoint slices, equal sizes per step
    parts = [ShardedBatches(ds, 4, r, 2, 0, ["a"]) for r in range(2)]
    per_rank = [[b["a"][:, 0].tolist() for b in p.epoch(0, torch.device("cpu"))] for p in parts]
    assert len(per_rank[0]) == len(per_rank[1]) == len(parts[0])
    flat = []
    for b0, b1 in zip(*per_rank):
        assert len(b0) == len(b1)
        flat += b0 + b1
    assert len(set(flat)) == len(flat) and len(flat) >= 22
    # resume: skipping k batches yields the tail of the same epoch
    tail = [b["a"][:, 0].tolist() for b in one.epoch(0, torch.device("cpu"), skip=4)]
    full = [b["a"][:, 0].tolist() for b in one.epoch(0, torch.device("cpu"))]
    assert tail == full[4:]


def test_parse_resume_naming():
    from dalm_amd.training.common import parse_resume, steps_and_epochs

    per_epoch, max_steps, epochs = steps_and_epochs(10, 1, 3, None)
    assert (per_epoch, max_steps, epochs) == (10, 30, 3)
    assert parse_resume("/x/epoch_1", 10, 10, 1) == (2, None, 20)
    assert parse_resume("/x/step_14", 10, 10, 1) == (1, 4, 14)


def test_lora_injector_matches_reference_config():
    from transformers import BertConfig, BertModel, LlamaConfig, LlamaForCausalLM

    from dalm_amd.models import lora

    g = LlamaForCausalLM(LlamaConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                                     intermediate_size=64, vocab_size=50))
    lora.inject_lora(g, ["q_proj", "v_proj"])
    names = [n for n, p in g.named_parameters() if p.requires_grad]
    assert len(names) == 2 * 2 * 2 and all(".lora_A." in n or ".lora_B." in n for n in names)
    assert g._dalm_lora_config["r"] == 8 and g._dalm_lora_config["lora_alpha"] == 16
    ids = torch.randint(0, 50, (2, 5))
    g.eval()
    base = g(input_ids=ids).logits
    # B is zero-initialised: the adapted model starts identical to the base model
    lora.merge_and_unload(g)
    torch.testing.assert_close(g(input_ids=ids).logits, base)
    r = BertModel(BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64, vocab_size=50))
    lora.inject_lora(r, ["key", "query", "value"])
    assert sum(p.requires_grad for p in r.parameters()) == 6
    with pytest.raises(ValueError):
        lora.inject_lora(BertModel(BertConfig(hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                                              intermediate_size=64, vocab_size=50)), ["nope"])


def test_use_bnb_is_rejected_loudly():
    from dalm_amd.models import AutoModelForRagE2E, Mode

    with pytest.raises(NotImplementedError, match="bitsandbytes"):
        AutoModelForRagE2E("r", "g", use_bnb=Mode.BOTH)
