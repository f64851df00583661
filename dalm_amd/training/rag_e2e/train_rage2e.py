"""RAG-end2end trainer: the reference's CLI and `train_e2e` signature
(dalm/training/rag_e2e/train_rage2e.py:54-226, 229-260) driving the MI355X step in dalm_amd.training.step.

    python -m dalm_amd.training.rag_e2e.train_rage2e --dataset_path rows.csv \
        --retriever_name_or_path <bert-like> --generator_name_or_path <causal-lm> [--use_peft both] ...
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m dalm_amd.training.rag_e2e.train_rage2e ...

Differences from the reference that a user can observe are listed in INTEGRATION.md section 5.
"""
from __future__ import annotations

import logging
import os
from argparse import Namespace
from typing import Optional, Union

import torch

from ...models.rag_e2e_base_model import AutoModelForRagE2E, Mode
from ...sharded import barrier, init_distributed
from ...utils import load_dataset
from .. import common
from .._hooks import load_submodel, save_submodel
from ..step import RagE2EStep
from ..utils.rag_e2e_dataloader_utils import preprocess_dataset

logger = logging.getLogger("dalm_amd.train_rage2e")

_S, _I, _F = str, int, float
FLAGS = [
    ("dataset_path", dict(type=_S, default=None, help="Dataset path: a huggingface dataset directory or a csv file.")),
    ("passage_column_name", dict(type=_S, default="Abstract", help="Column holding the passage")),
    ("query_column_name", dict(type=_S, default="Question", help="Column holding the query")),
    ("answer_column_name", dict(type=_S, default="Answer", help="Column holding the answer")),
    ("query_max_len", dict(type=_I, default=50, help="Max query length after tokenization (truncates)")),
    ("passage_max_len", dict(type=_I, default=160, help="Max passage length after tokenization (truncates)")),
    ("generator_max_len", dict(type=_I, default=256, help="Max generator input length after tokenization (truncates)")),
    ("retriever_name_or_path", dict(type=_S, required=True, help="Retriever model path or hub id")),
    ("generator_name_or_path", dict(type=_S, required=True, help="Generator model path or hub id")),
    ("per_device_train_batch_size", dict(type=_I, default=32, help="Batch size per device")),
    ("learning_rate", dict(type=_F, default=1e-4, help="Initial learning rate (after warmup)")),
    ("logit_scale", dict(type=_I, default=100, help="Logit scale of the contrastive loss")),
    ("weight_decay", dict(type=_F, default=0.0, help="Weight decay (accepted; Adam ignores it, as upstream)")),
    ("num_train_epochs", dict(type=_I, default=1, help="Number of training epochs")),
    ("max_train_steps", dict(type=_I, default=None, help="Total training steps; overrides num_train_epochs")),
    ("gradient_accumulation_steps", dict(type=_I, default=1, help="Accepted for CLI compatibility")),
    ("lr_scheduler_type", dict(type=_S, default="linear", choices=common.SCHEDULERS, help="LR scheduler")),
    ("num_warmup_steps", dict(type=_I, default=100, help="Warmup steps of the LR scheduler")),
    ("output_dir", dict(type=_S, default=None, help="Where to store the final model")),
    ("seed", dict(type=_I, default=None, help="Seed for reproducible training")),
    ("hub_model_id", dict(type=_S, help="Unused (kept for CLI compatibility)")),
    ("hub_token", dict(type=_S, help="Unused (kept for CLI compatibility)")),
    ("checkpointing_steps", dict(type=_S, default=None, help="Save state every n steps, or 'epoch'")),
    ("resume_from_checkpoint", dict(type=_S, default=None, help="Checkpoint folder to continue from")),
    ("with_tracking", dict(action="store_true", help="Enable experiment tracking")),
    ("report_to", dict(type=_S, default="all", help="Tracker name(s); only with --with_tracking")),
    ("sanity_test", dict(action="store_true", help="Unused (kept for CLI compatibility)")),
    ("use_peft", dict(type=Mode, choices=list(Mode), required=False, help="LoRA on generator / retriever / both")),
    ("use_bnb", dict(type=Mode, choices=list(Mode), help="nf4 storage of the frozen base weights of this tower (HIP kernels; needs the GPU)")),
    ("retriever_is_autoregressive", dict(action="store_true", help="Retriever is an autoregressive LM")),
    # extensions (not in the reference)
    ("mixed_precision", dict(type=_S, default=None, choices=["no", "bf16"], help="[ext] autocast dtype of the towers; default: $ACCELERATE_MIXED_PRECISION, else 'no' (the reference's Accelerator() default); 'bf16' is the fast setting")),
    ("no_hip_graph", dict(action="store_true", help="[ext] launch every step eagerly instead of replaying a hipGraph")),
    ("token_cache_dir", dict(type=_S, default=None, help="[ext] keep the tokenised dataset as int32 shards here; reused when unchanged")),
    ("length_bucketing", dict(action="store_true", help="[ext] batch rows of similar generator length together")),
    ("trim_padding", dict(action="store_true", help="[ext] drop all-padding columns per batch (exact for real-token rows; the B first-token rows of left-padded generator inputs are approximate, and generators with absolute position embeddings are refused)")),
    ("async_checkpoint", dict(action="store_true", help="[ext] write optimizer/scheduler state from a background thread")),
    ("fuse_lm_head", dict(action="store_true", help="[ext] lm_head + loss in chunks over the rows that carry loss; the [B,T,V] logits never exist")),
    ("pack_tokens", dict(action="store_true", help="[ext] run both towers on the live tokens only (un-padded rows, per-sequence attention, original positions): same loss and gradients, ~1/3 fewer generator rows at the default lengths")),
]


def parse_args(argv=None) -> Namespace:
    return common.build_parser("training a PEFT model for Semantic Search task", FLAGS).parse_args(argv)


def train_e2e(
    dataset_or_path,
    retriever_name_or_path: str,
    generator_name_or_path: str,
    passage_column_name: str = "Abstract",
    query_column_name: str = "Question",
    answer_column_name: str = "Answer",
    query_max_len: int = 50,
    passage_max_len: int = 128,
    generator_max_len: int = 256,
    per_device_train_batch_size: int = 32,
    learning_rate: float = 1e-4,
    logit_scale: int = 100,
    weight_decay: float = 0.0,
    num_train_epochs: int = 1,
    max_train_steps: Optional[int] = None,
    gradient_accumulation_steps: int = 1,
    lr_scheduler_type="linear",
    num_warmup_steps: int = 100,
    output_dir: Optional[str] = None,
    seed: int = 42,
    hub_model_id: Optional[str] = None,
    hub_token: Optional[str] = None,
    checkpointing_steps: Optional[Union[int, str]] = None,
    resume_from_checkpoint: Optional[str] = None,
    with_tracking: bool = True,
    report_to: str = "all",
    sanity_test: bool = True,
    use_peft: Optional[Mode] = None,
    use_bnb: Optional[Mode] = None,
    retriever_is_autoregressive: bool = False,
    *,
    mixed_precision: Optional[str] = None,
    no_hip_graph: bool = False,
    token_cache_dir: Optional[str] = None,
    length_bucketing: bool = False,
    trim_padding: bool = False,
    async_checkpoint: bool = False,
    fuse_lm_head: bool = False,
    pack_tokens: bool = False,
    rag_model: Optional[AutoModelForRagE2E] = None,
    on_step=None,
) -> None:
    """Train retriever and generator jointly.  `rag_model` (pre-built wrapper) and `on_step`
    (callback(step, loss)) are extensions used by tests and benchmarks."""
    config = {k: v for k, v in dict(locals()).items() if v is None or isinstance(v, (float, int, str))}
    comm, device = init_distributed()
    mixed_precision = common.resolve_mixed_precision(mixed_precision)
    gradient_accumulation_steps = common.effective_grad_accum(gradient_accumulation_steps)
    if device.type != "cuda":
        raise RuntimeError("train_e2e needs an MI355X: the loss path has no CPU implementation in this package")
    is_main = comm.rank == 0
    from ...tuning import enable_tuned_gemms

    enable_tuned_gemms()  # replay-only GEMM solution table for the towers; unknown shapes use library defaults
    common.seed_everything(seed)
    if rag_model is None:
        dtype = torch.bfloat16 if (mixed_precision == "bf16" and use_peft is not None) else None
        rag_model = AutoModelForRagE2E(retriever_name_or_path, generator_name_or_path, get_peft=use_peft,
                                       use_bnb=use_bnb, retriever_is_autoregressive=retriever_is_autoregressive,
                                       torch_dtype=dtype)
    rag_model.to(device)
    if is_main and output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
    barrier(comm)

    # ---- data: tokenise once on the host (reference :298-322) -------------------------------
    dataset = load_dataset(dataset_or_path)
    r_tok, g_tok = rag_model.retriever_tokenizer, rag_model.generator_tokenizer
    g_tok.pad_token = g_tok.eos_token
    g_tok.add_eos_token = True
    columns = ["retriever_query_input_ids", "retriever_query_attention_mask", "retriever_passage_input_ids",
               "retriever_passage_attention_mask", "generator_input_input_ids", "generator_input_attention_mask",
               "query_passage_input_len"]
    from .. import shards

    data_id = shards.dataset_identity(dataset)
    if token_cache_dir and data_id is None:
        import warnings

        warnings.warn("--token_cache_dir ignored: the dataset carries no content fingerprint, a cache could not be invalidated")
        token_cache_dir = None
    fp = shards.fingerprint(data=data_id, rows=len(dataset), r_tok=shards.tokenizer_identity(r_tok),
                            g_tok=shards.tokenizer_identity(g_tok), cols=(query_column_name, passage_column_name, answer_column_name),
                            lens=(query_max_len, passage_max_len, generator_max_len))
    processed = shards.load_token_shards(token_cache_dir, fp) if token_cache_dir else None
    if processed is None:
        mapped = dataset.map(
            lambda ex: preprocess_dataset(ex, retriever_tokenizer=r_tok, generator_tokenizer=g_tok,
                                          query_column_name=query_column_name, passage_column_name=passage_column_name,
                                          answer_column_name=answer_column_name, query_max_len=query_max_len,
                                          passage_max_len=passage_max_len, generator_max_len=generator_max_len),
            # in-process on purpose: the reference passes num_proc=1 (train_rage2e.py:316), which makes `datasets` FORK a
            # worker - forking a process that already holds a HIP context and its threads crashed the worker now and
            # then ("One of the subprocesses has abruptly died during map operation")
            batched=True, remove_columns=dataset.column_names, desc="Running tokenizer on dataset")
        processed = shards.columns_from_dataset(mapped, columns)
        if token_cache_dir and is_main:
            shards.save_token_shards(processed, token_cache_dir, fp)
    trim = None
    if trim_padding:
        why = common.has_absolute_positions(rag_model.generator_model)
        if why is not None:
            raise ValueError("--trim_padding shifts every token of a left-padded row: it needs a rotary / ALiBi / "
                             f"relative-position generator ({why})")
        trim = dict(groups=[("retriever_query_input_ids", "retriever_query_attention_mask"),
                            ("retriever_passage_input_ids", "retriever_passage_attention_mask"),
                            ("generator_input_input_ids", "generator_input_attention_mask")],
                    qlen_key="query_passage_input_len", qlen_follows="generator_input_attention_mask")
    live_rows = None
    if fuse_lm_head:   # padding rows skip the lm_head GEMMs and the CE
        from ...fused import gemm_wave_rows

        vocab = rag_model.generator_model.get_output_embeddings().weight.shape[0]
        live_rows = dict(mask="generator_input_attention_mask", multiple=gemm_wave_rows(vocab))
    pack = None
    if pack_tokens:
        # coarse row multiples: the packed row counts are part of a hipGraph's shape (one graph per combination seen)
        from ... import packed as packed_mod

        pack = dict(groups=packed_mod.RAG_GROUPS, multiple=packed_mod.ROW_MULTIPLES)
        live_rows = None          # the packed generator path lists its own rows
    batches = common.ShardedBatches(processed, per_device_train_batch_size, comm.rank, comm.world_size,
                                    seed if seed is not None else 0, columns,
                                    bucket_by="generator_input_attention_mask" if length_bucketing else None, trim=trim,
                                    live_rows=live_rows, pack=pack)

    # ---- optimiser / schedule (reference :336-362) ----------------------------------------------
    params = [p for p in rag_model.parameters() if p.requires_grad]
    # one GPU: the whole step is captured once and replayed as a hipGraph (capturable Adam + tensor lr);
    # W > 1 launches eagerly (RCCL collectives stay outside graphs for now)
    from ...fused import LocalComm
    from ..graphed import GraphedStep, TensorLRScheduler, make_capturable_adam

    # DALM_STEP_GRAPHS=towers: one rank, but the W > 1 launch structure (single-stream tower graphs, eager loss / optimizer): a
    # graph captured across two streams replays with a dependency bubble per node, 5-6 ms per cfg3 step (bench.py's default since
    # round 6); the trainers keep the whole-step graph by default: its graphs share one memory pool across any number of batch
    # shapes (packed rows), tower-graph sets do not (DALM_TOWER_SETS)
    towers_mode = os.environ.get("DALM_STEP_GRAPHS", "whole") == "towers"
    use_graph = isinstance(comm, LocalComm) and not no_hip_graph and gradient_accumulation_steps == 1 and not towers_mode
    optimizer = (make_capturable_adam(params, learning_rate, device) if use_graph
                 else torch.optim.Adam(params, lr=learning_rate, fused=True))
    per_epoch, max_train_steps, num_train_epochs = common.steps_and_epochs(
        len(batches), gradient_accumulation_steps, num_train_epochs, max_train_steps)
    from transformers import get_scheduler

    name = getattr(lr_scheduler_type, "value", lr_scheduler_type)

    def make_schedule(o):
        return get_scheduler(name=name, optimizer=o, num_warmup_steps=num_warmup_steps, num_training_steps=max_train_steps)

    scheduler = TensorLRScheduler(optimizer, learning_rate, make_schedule) if use_graph else make_schedule(optimizer)
    if checkpointing_steps is not None and str(checkpointing_steps).isdigit():
        checkpointing_steps = int(checkpointing_steps)
    tracker = common.Tracker(with_tracking, output_dir, "peft_rag_e2e_learning", config, is_main)

    def save_models(path: str) -> None:
        save_submodel(rag_model.generator_model, os.path.join(path, "generator"))
        save_submodel(rag_model.retriever_model, os.path.join(path, "retriever"))

    starting_epoch, resume_step, completed = 0, None, 0
    if resume_from_checkpoint:
        logger.info("Resumed from checkpoint: %s", resume_from_checkpoint)
        load_submodel(rag_model.generator_model, os.path.join(resume_from_checkpoint, "generator"))
        load_submodel(rag_model.retriever_model, os.path.join(resume_from_checkpoint, "retriever"))
        saved = common.load_training_state(resume_from_checkpoint, optimizer, scheduler)
        starting_epoch, resume_step, completed = common.parse_resume(resume_from_checkpoint, per_epoch, len(batches),
                                                                     gradient_accumulation_steps, saved)

    if is_main:
        logger.info("***** Running E2E training *****  examples=%d epochs=%d per-device batch=%d global batch=%d steps=%d",
                    batches.n, num_train_epochs, per_device_train_batch_size,
                    per_device_train_batch_size * comm.world_size, max_train_steps)
    step_fn = RagE2EStep(rag_model, optimizer, scheduler, logit_scale, comm=comm,
                         autocast_dtype=torch.bfloat16 if mixed_precision == "bf16" else None,
                         # W > 1: graph the tower fwd/bwd, keep collectives + loss + optimizer eager
                         graph_towers=(not use_graph) and (not no_hip_graph), graph_after=2, fuse_lm_head=fuse_lm_head,
                         grad_accum=gradient_accumulation_steps)
    if use_graph:
        # partial last batches (other shapes) run eagerly; with live rows the padded row count is part of the shape
        # (multiples of 256-512 rows: at most B*Tg/256 values), all graphs share one memory pool
        step_fn = GraphedStep(step_fn, warmup=0, eager_steps=2, max_graphs=48 if pack_tokens else (24 if fuse_lm_head else 8))
    meter = common.Throughput()
    saver = common.AsyncSaver() if async_checkpoint else None

    def save_state(path: str, position) -> None:
        common.save_training_state(path, rag_model, optimizer, scheduler, position, save_models, rank=comm.rank,
                                   world=comm.world_size, saver=saver, barrier=lambda: barrier(comm))

    progress = common.Progress(comm=comm, is_main=is_main, tracker=tracker, meter=meter, on_step=on_step,
                               checkpointing_steps=checkpointing_steps, output_dir=output_dir,
                               max_train_steps=max_train_steps, save_state=save_state, num_batches=len(batches),
                               grad_accum=gradient_accumulation_steps, completed=completed, log=logger)
    for epoch in range(starting_epoch, num_train_epochs):
        rag_model.train()
        total_loss = torch.zeros((), device=device)
        skip = resume_step if (resume_from_checkpoint and epoch == starting_epoch and resume_step) else 0
        step, loss, stop = -1, None, False
        for step, batch in enumerate(batches.epoch(epoch, device, skip)):
            loss = step_fn(batch)  # rank share of the global-batch loss
            total_loss += loss
            meter.add(batch["query_passage_input_len"].shape[0] * comm.world_size)
            if not getattr(step_fn, "synced", True):
                continue                  # gradient accumulation: a micro-batch that did not take the optimizer step
            stop = progress.after_optimizer_step(epoch, step, skip, loss, total_loss)
            if stop:
                break
        if not stop and gradient_accumulation_steps > 1 and step_fn.flush():
            # micro-batches still pending at the end of the epoch: that optimizer step is counted, reported, checkpointed
            # and checked against max_train_steps like any other
            progress.after_optimizer_step(epoch, step, skip, loss, total_loss)
        tl = comm.all_reduce_sum_(total_loss.clone())
        tracker.log({"train/epoch_loss": float(tl) / max(len(batches), 1)}, progress.completed)
        if output_dir is not None:
            barrier(comm)
            if isinstance(checkpointing_steps, str):
                save_state(os.path.join(output_dir, f"epoch_{epoch}"), progress.position(epoch + 1, 0))
            if is_main:
                save_models(output_dir)  # <output_dir>/retriever, <output_dir>/generator (reference :508-524)
                r_tok.save_pretrained(os.path.join(output_dir, "retriever"))
                g_tok.save_pretrained(os.path.join(output_dir, "generator"))
            barrier(comm)
    if saver is not None:
        saver.wait()
    tracker.close()


def main() -> None:
    from ... import configure_hw_queues

    configure_hw_queues()  # before the HIP runtime starts (see dalm_amd/__init__.py)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(name)s - %(message)s")
    a = parse_args()
    kw = {k: v for k, v in vars(a).items() if k not in ("dataset_path", "retriever_name_or_path", "generator_name_or_path")}
    train_e2e(a.dataset_path, a.retriever_name_or_path, a.generator_name_or_path, **kw)


if __name__ == "__main__":
    main()
