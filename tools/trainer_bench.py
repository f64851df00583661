"""The TRAINER at real size: `train_e2e` / `train_retriever` themselves - not bench.py's loop over four
resident batches - on the full-depth random-init architectures of BASELINE.json's configs and a synthetic
Question,Abstract,Answer csv (tools/make_synthetic_csv.py):

    csv -> datasets -> preprocess_dataset (tokenise) -> --token_cache_dir shards -> ShardedBatches (pinned int32 columns,
    staged index_select, H2D on a copy stream) -> GraphedStep (whole step as a hipGraph; the partial last batch eagerly)
    -> step_N checkpoint -> the run is KILLED (exception out of on_step) -> a fresh process state (new model objects,
    re-seeded) resumes with --resume_from_checkpoint from the last step_N and finishes the epoch.

Prints ONE JSON line: the trainer's own pairs/s over a steady window (device-synchronised at both ends), tokenisation /
cache times, checkpoint cost, the losses of the steps both phases ran (continuity across the resume), peak HBM.

    python bench.py --through-trainer [--workload cfg3|cfg5|cfg2] [--trainer-rows 10000]
    python tools/trainer_bench.py --tokenise-only --rows 200000       # host only: tokenisation + cache write / hit

There is no network: tokenizers are WordLevel tokenizers over the csv's own 4000-word list (a word = a token), with the
BERT ([CLS] .. [SEP], right padding) and Llama (<s> .. </s>, pad = eos, left padding) conventions; ids stay below every
embedding table.  Sequence lengths are therefore those of the synthetic text (generator prompts of ~40-160 tokens padded
to 256), the step's SHAPES - what the GPU time depends on - are the named configuration's.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


class Killed(RuntimeError):
    """Raised out of on_step: the training process 'dies' between two steps."""


def make_tokenizers(out_dir: str):
    from make_synthetic_csv import word_list
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast

    words = word_list() + ["#query#", "#passage#", "#answer#"]

    def build(specials, template, **kw):
        vocab = {s: i for i, s in enumerate(specials)}
        for w in words:
            vocab.setdefault(w, len(vocab))
        tok = Tokenizer(models.WordLevel(vocab, unk_token=specials[1]))
        tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
        tok.post_processor = processors.TemplateProcessing(single=template, special_tokens=[(s, vocab[s]) for s in specials
                                                                                             if s in template])
        return PreTrainedTokenizerFast(tokenizer_object=tok, unk_token=specials[1], **kw)

    r_tok = build(["[PAD]", "[UNK]", "[CLS]", "[SEP]"], "[CLS] $A [SEP]", pad_token="[PAD]", cls_token="[CLS]",
                  sep_token="[SEP]")
    g_tok = build(["<pad>", "<unk>", "<s>", "</s>"], "<s> $A </s>", bos_token="<s>", eos_token="</s>", padding_side="left")
    r_tok.save_pretrained(os.path.join(out_dir, "retriever_tok"))
    g_tok.save_pretrained(os.path.join(out_dir, "generator_tok"))
    # reload: `name_or_path` and the serialised pipeline are then what a user's tokenizer directory would give
    return (PreTrainedTokenizerFast.from_pretrained(os.path.join(out_dir, "retriever_tok")),
            PreTrainedTokenizerFast.from_pretrained(os.path.join(out_dir, "generator_tok")))


def tokenise_only(rows: int, workdir: str) -> dict:
    """Host only: csv -> tokenised columns -> int32 shards -> reload."""
    import datasets as hf_datasets  # noqa: F401

    from dalm_amd.training import shards
    from dalm_amd.training.utils.rag_e2e_dataloader_utils import preprocess_dataset
    from dalm_amd.utils import load_dataset
    from make_synthetic_csv import write_csv

    r_tok, g_tok = make_tokenizers(workdir)
    g_tok.pad_token = g_tok.eos_token
    csv_path = os.path.join(workdir, f"rows_{rows}.csv")
    t0 = time.perf_counter()
    write_csv(csv_path, rows)
    t_csv = time.perf_counter() - t0
    t0 = time.perf_counter()
    ds = load_dataset(csv_path)
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    mapped = ds.map(lambda ex: preprocess_dataset(ex, retriever_tokenizer=r_tok, generator_tokenizer=g_tok,
                                                  query_column_name="Question", passage_column_name="Abstract",
                                                  answer_column_name="Answer", query_max_len=50, passage_max_len=128,
                                                  generator_max_len=256),
                    batched=True, remove_columns=ds.column_names, desc="tokenise")
    t_map = time.perf_counter() - t0
    cols = ["retriever_query_input_ids", "retriever_query_attention_mask", "retriever_passage_input_ids",
            "retriever_passage_attention_mask", "generator_input_input_ids", "generator_input_attention_mask",
            "query_passage_input_len"]
    t0 = time.perf_counter()
    processed = shards.columns_from_dataset(mapped, cols)
    t_cols = time.perf_counter() - t0
    cache = os.path.join(workdir, "cache")
    t0 = time.perf_counter()
    shards.save_token_shards(processed, cache, "fp")
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    again = shards.load_token_shards(cache, "fp")
    t_hit = time.perf_counter() - t0
    import numpy as np

    glen = np.asarray(again["generator_input_attention_mask"]).sum(axis=1)
    nbytes = sum(os.path.getsize(os.path.join(cache, f)) for f in os.listdir(cache))
    return {"rows": rows, "host_cpus": os.cpu_count(), "csv_write_s": t_csv, "csv_load_s": t_load, "tokenise_s": t_map,
            "rows_per_s_tokenise": rows / t_map, "columns_to_int_arrays_s": t_cols, "cache_write_s": t_write,
            "cache_hit_load_s": t_hit, "cache_bytes": nbytes,
            "generator_live_tokens": {"mean": float(glen.mean()), "min": int(glen.min()), "max": int(glen.max())},
            "steps_per_epoch_at_batch_18": -(-rows // 18)}


def run(args) -> dict:
    import torch

    import bench
    import dalm_amd
    from dalm_amd import hip
    from make_synthetic_csv import write_csv

    dalm_amd.configure_hw_queues(1)
    hip.load()
    dev = torch.device("cuda:0")
    retriever_only = args.workload == "cfg2"
    B = 150 if retriever_only else 18
    work = tempfile.mkdtemp(prefix="dalm_trainer_", dir=args.workdir)
    out_dir, cache = os.path.join(work, "out"), os.path.join(work, "token_cache")
    csv_path = os.path.join(work, "rows.csv")
    write_csv(csv_path, args.rows)
    r_tok, g_tok = make_tokenizers(work)
    nb = -(-args.rows // B)
    ckpt = args.checkpointing_steps or max(10, (nb * 5 // 9) // 10 * 10)
    kill_at = min(nb - 1, ckpt + max(5, ckpt // 10))
    window = (min(20, ckpt // 2), kill_at - 1)                 # steady window of phase A: after capture, before the kill
    gen_name = "falcon-7b" if args.workload == "cfg5" else "llama-2-7b"

    def build():
        if retriever_only:
            from transformers import BertConfig, BertModel

            from dalm_amd.models import AutoModelForSentenceEmbedding

            torch.manual_seed(0)
            with torch.device(dev):
                old = torch.get_default_dtype()
                torch.set_default_dtype(torch.bfloat16)
                try:
                    bert = BertModel(BertConfig(hidden_size=1024, num_hidden_layers=args.retriever_layers, num_attention_heads=16,
                                                intermediate_size=4096, vocab_size=30522, max_position_embeddings=512))
                finally:
                    torch.set_default_dtype(old)
            return AutoModelForSentenceEmbedding.from_modules(bert, r_tok, normalize=True, get_peft=True)
        m = bench.build_models(dev, torch.bfloat16, args.retriever_layers, args.generator_layers, generator=gen_name)
        m.retriever_tokenizer, m.generator_tokenizer = r_tok, g_tok
        return m

    rec = {"A": {}, "B": {}}
    marks = {}

    def on_step_factory(phase, first_step):
        losses = rec[phase]

        def on_step(step, loss):
            if phase == "A":
                if step == window[0]:
                    torch.cuda.synchronize()
                    marks["t0"] = time.perf_counter()
                if step == window[1]:
                    torch.cuda.synchronize()
                    marks["t1"] = time.perf_counter()
                if step > ckpt and step <= kill_at:
                    losses[step] = float(loss)
                if step == kill_at:
                    raise Killed(f"killed after step {step}")
            else:
                if step <= kill_at or step >= nb - 1:
                    losses[step] = float(loss)
                if step == kill_at + 1:         # past the eager + capture steps and the per-step float(loss) syncs
                    torch.cuda.synchronize()
                    marks["b0"], marks["b0_step"] = time.perf_counter(), step
                if step == nb - 1:           # the last FULL batch; the partial one follows
                    torch.cuda.synchronize()
                    marks["b1"], marks["b1_step"] = time.perf_counter(), step
        return on_step

    def train(model, phase, resume=None):
        kw = dict(per_device_train_batch_size=B, query_max_len=50, passage_max_len=128, learning_rate=1e-4, logit_scale=100,
                  num_train_epochs=1, output_dir=out_dir, seed=42, checkpointing_steps=ckpt, resume_from_checkpoint=resume,
                  with_tracking=True, mixed_precision="bf16", token_cache_dir=cache, pack_tokens=bool(args.pack_tokens),
                  on_step=on_step_factory(phase, ckpt if resume else 0))
        t0 = time.perf_counter()
        try:
            if retriever_only:
                from dalm_amd.training.retriever_only.train_retriever_only import train_retriever

                train_retriever("", csv_path, num_warmup_steps=0, use_peft=True, use_bnb=False, model=model, **kw)
            else:
                from dalm_amd.training.rag_e2e.train_rage2e import train_e2e

                train_e2e(csv_path, "", "", generator_max_len=256, num_warmup_steps=100, rag_model=model, **kw)
        except Killed:
            pass
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    import logging

    logging.basicConfig(level=logging.INFO, stream=sys.stderr, format="%(asctime)s %(name)s %(message)s")
    torch.cuda.reset_peak_memory_stats()
    model = build()
    wall_a = train(model, "A")
    peak_a = torch.cuda.max_memory_allocated() / 1e9
    steps_dirs = sorted(d for d in os.listdir(out_dir) if d.startswith("step_"))
    ck_dir = os.path.join(out_dir, f"step_{ckpt}")
    ck_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(ck_dir) for f in fs)
    del model
    import gc

    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    model = build()                     # a new process would do exactly this: same seed -> same frozen base weights
    wall_b = train(model, "B", resume=ck_dir)
    peak_b = torch.cuda.max_memory_allocated() / 1e9
    both = sorted(set(rec["A"]) & set(rec["B"]))
    rel = [abs(rec["A"][s] - rec["B"][s]) / abs(rec["A"][s]) for s in both]
    rate_a = (window[1] - window[0]) * B / (marks["t1"] - marks["t0"])
    rate_b = (marks["b1_step"] - marks["b0_step"]) * B / (marks["b1"] - marks["b0"]) if "b1" in marks and "b0" in marks else None
    logs = [json.loads(x) for x in open(os.path.join(out_dir, "logs", "peft_contrastive_learning.jsonl" if retriever_only
                                                     else "peft_rag_e2e_learning.jsonl")) if x.strip()]
    logged = [x for x in logs if "train/pairs_per_sec_recent" in x]
    final = sorted(os.listdir(out_dir))
    res = {
        "metric": "training pairs/sec through the trainer entry point (" + ("train_retriever" if retriever_only else "train_e2e") + ")",
        "value": rate_a, "unit": "pairs/s", "n_gpus": 1, "dtype": "bf16", "data": "synthetic csv",
        "config": {"workload": args.workload, "rows": args.rows, "pack_tokens": bool(args.pack_tokens), "per_device_train_batch_size": B, "batches_per_epoch": nb,
                   "partial_last_batch_rows": args.rows - (nb - 1) * B, "retriever_layers": args.retriever_layers,
                   "generator_layers": None if retriever_only else args.generator_layers,
                   "generator": None if retriever_only else gen_name,
                   "tokenizers": "WordLevel over the csv's 4000-word list, BERT / Llama special-token conventions (no network)"},
        "steady_window_steps": list(window), "ms_per_step": 1e3 * B / rate_a,
        "phase_B_pairs_per_s_after_resume": rate_b,
        "trainer_log_pairs_per_sec_recent": [round(x["train/pairs_per_sec_recent"], 2) for x in logged],
        "checkpointing_steps": ckpt, "checkpoints_written_phase_A": steps_dirs, "checkpoint_bytes": ck_bytes,
        "killed_after_step": kill_at, "resumed_from": f"step_{ckpt}",
        "losses_phase_A": {str(k): v for k, v in sorted(rec["A"].items())},
        "losses_phase_B": {str(k): v for k, v in sorted(rec["B"].items())},
        "resume_loss_rel_diff_max": max(rel) if rel else None,
        "resume_loss_note": "steps both phases ran (after the checkpoint, before the kill); dropout is ON (bench-equal model: BERT "
                            "hidden 0.1, LoRA 0.05) and its RNG stream is not part of a checkpoint (nor of the reference's), so the two "
                            "phases agree to dropout noise, not bitwise; the exact check (dropout 0, fp32) is "
                            "tests/test_trainer_realwidth_gpu.py",
        "phase_B_last_steps": {str(k): v for k, v in sorted(rec["B"].items()) if k >= nb - 1},
        "wall_s": {"phase_A_total": wall_a, "phase_B_total": wall_b},
        "peak_hbm_gb": {"phase_A": peak_a, "phase_B": peak_b},
        "output_dir_entries": final,
    }
    if args.bench_line and os.path.exists(args.bench_line):
        for line in open(args.bench_line):
            if line.startswith("{"):
                b = json.loads(line)
                res["bench_line"] = {"value": b["value"], "ms_per_step": b["ms_per_step"]}
                res["trainer_over_bench"] = rate_a / b["value"]
    shutil.rmtree(work, ignore_errors=True)
    return res


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg5", "cfg2"])
    ap.add_argument("--rows", type=int, default=10000)
    ap.add_argument("--checkpointing-steps", type=int, default=None)
    ap.add_argument("--retriever-layers", type=int, default=24)
    ap.add_argument("--generator-layers", type=int, default=32)
    ap.add_argument("--workdir", default="/tmp")
    ap.add_argument("--bench-line", default=None, help="file holding bench.py's JSON line of the same workload (ratio)")
    ap.add_argument("--tokenise-only", action="store_true")
    ap.add_argument("--pack-tokens", action="store_true", help="the trainers' --pack_tokens: towers on the live tokens only")
    a = ap.parse_args(argv)
    if a.tokenise_only:
        d = tempfile.mkdtemp(prefix="dalm_tok_", dir=a.workdir)
        try:
            print(json.dumps(tokenise_only(a.rows, d)), flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)
        return
    print(json.dumps(run(a)), flush=True)


if __name__ == "__main__":
    main()
