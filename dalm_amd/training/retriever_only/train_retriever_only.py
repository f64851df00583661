"""Retriever-only contrastive trainer: the reference's CLI and `train_retriever` signature
(dalm/training/retriever_only/train_retriever_only.py:43-203) on the MI355X step."""
from __future__ import annotations

import logging
import os
from argparse import Namespace
from typing import Optional, Union

import torch

from ...models.retriever_only_base_model import AutoModelForSentenceEmbedding
from ...sharded import barrier, init_distributed
from ...utils import load_dataset
from .. import common
from .._hooks import load_submodel, save_submodel
from ..step import RetrieverStep
from ..utils.retriever_only_dataloader_utils import preprocess_dataset

logger = logging.getLogger("dalm_amd.train_retriever_only")

_S, _I, _F = str, int, float
FLAGS = [
    ("dataset_path", dict(type=_S, default=None, help="dataset path in the local dir (dataset directory or csv)")),
    ("query_column_name", dict(type=_S, default="Question", help="name of the query col")),
    ("passage_column_name", dict(type=_S, default="Abstract", help="name of the passage col")),
    ("query_max_len", dict(type=_I, default=50, help="Max query length after tokenization (truncates)")),
    ("passage_max_len", dict(type=_I, default=160, help="Max passage length after tokenization (truncates)")),
    ("retriever_name_or_path", dict(type=_S, required=True, help="Retriever model path or hub id")),
    ("per_device_train_batch_size", dict(type=_I, default=8, help="Batch size per device")),
    ("learning_rate", dict(type=_F, default=1e-4, help="Initial learning rate (after warmup)")),
    ("logit_scale", dict(type=_I, default=100, help="Logit scale of the contrastive loss")),
    ("weight_decay", dict(type=_F, default=0.0, help="Weight decay (accepted; Adam ignores it, as upstream)")),
    ("num_train_epochs", dict(type=_I, default=3, help="Number of training epochs")),
    ("max_train_steps", dict(type=_I, default=None, help="Total training steps; overrides num_train_epochs")),
    ("gradient_accumulation_steps", dict(type=_I, default=1, help="Accepted for CLI compatibility")),
    ("lr_scheduler_type", dict(type=_S, default="linear", choices=common.SCHEDULERS, help="LR scheduler")),
    ("num_warmup_steps", dict(type=_I, default=0, help="Warmup steps of the LR scheduler")),
    ("output_dir", dict(type=_S, default=None, help="Where to store the final model")),
    ("seed", dict(type=_I, default=None, help="Seed for reproducible training")),
    ("hub_model_id", dict(type=_S, help="Unused (kept for CLI compatibility)")),
    ("hub_token", dict(type=_S, help="Unused (kept for CLI compatibility)")),
    ("checkpointing_steps", dict(type=_S, default=None, help="Save state every n steps, or 'epoch'")),
    ("resume_from_checkpoint", dict(type=_S, default=None, help="Checkpoint folder to continue from")),
    ("with_tracking", dict(action="store_true", help="Enable experiment tracking")),
    ("report_to", dict(type=_S, default="all", help="Tracker name(s); only with --with_tracking")),
    ("sanity_test", dict(action="store_true", help="Unused (kept for CLI compatibility)")),
    ("use_peft", dict(action="store_true", help="LoRA on the retriever")),
    ("use_bnb", dict(action="store_true", help="nf4 storage of the frozen base weights (HIP kernels; needs the GPU)")),
    ("is_autoregressive", dict(action="store_true", help="Retriever is an autoregressive LM")),
    ("mixed_precision", dict(type=_S, default=None, choices=["no", "bf16"], help="[ext] autocast dtype; default: $ACCELERATE_MIXED_PRECISION, else 'no' (the reference's Accelerator() default); 'bf16' is the fast setting")),
    ("no_hip_graph", dict(action="store_true", help="[ext] launch every step eagerly instead of replaying a hipGraph")),
    ("token_cache_dir", dict(type=_S, default=None, help="[ext] keep the tokenised dataset as int32 shards here; reused when unchanged")),
    ("length_bucketing", dict(action="store_true", help="[ext] batch rows of similar passage length together")),
    ("trim_padding", dict(action="store_true", help="[ext] drop all-padding columns per batch (loss-preserving)")),
    ("async_checkpoint", dict(action="store_true", help="[ext] write optimizer/scheduler state from a background thread")),
    ("pack_tokens", dict(action="store_true", help="[ext] run the encoder on the live tokens only (un-padded rows, per-sequence attention, original positions): same loss and gradients")),
]


def parse_args(argv=None) -> Namespace:
    return common.build_parser("training a PEFT model for Sematic Search task", FLAGS).parse_args(argv)


def train_retriever(
    retriever_name_or_path: str,
    dataset_or_path,
    passage_column_name: str = "Abstract",
    query_column_name: str = "Question",
    query_max_len: int = 50,
    passage_max_len: int = 128,
    per_device_train_batch_size: int = 32,
    learning_rate: float = 1e-4,
    logit_scale: int = 100,
    weight_decay: float = 0.0,
    num_train_epochs: int = 1,
    max_train_steps: Optional[int] = None,
    gradient_accumulation_steps: int = 1,
    lr_scheduler_type="linear",
    num_warmup_steps: int = 0,
    output_dir: Optional[str] = None,
    seed: int = 42,
    hub_model_id: Optional[str] = None,
    hub_token: Optional[str] = None,
    checkpointing_steps: Optional[Union[int, str]] = None,
    resume_from_checkpoint: Optional[str] = None,
    with_tracking: bool = True,
    report_to: str = "all",
    sanity_test: bool = True,
    use_peft: bool = True,
    use_bnb: bool = True,
    is_autoregressive: bool = False,
    *,
    mixed_precision: Optional[str] = None,
    no_hip_graph: bool = False,
    token_cache_dir: Optional[str] = None,
    length_bucketing: bool = False,
    trim_padding: bool = False,
    async_checkpoint: bool = False,
    pack_tokens: bool = False,
    model: Optional[AutoModelForSentenceEmbedding] = None,
    on_step=None,
) -> None:
    config = {k: v for k, v in dict(locals()).items() if v is None or isinstance(v, (float, int, str))}
    comm, device = init_distributed()
    mixed_precision = common.resolve_mixed_precision(mixed_precision)
    gradient_accumulation_steps = common.effective_grad_accum(gradient_accumulation_steps)
    if device.type != "cuda":
        raise RuntimeError("train_retriever needs an MI355X: the loss path has no CPU implementation in this package")
    is_main = comm.rank == 0
    from ...tuning import enable_tuned_gemms

    enable_tuned_gemms()  # replay-only GEMM solution table for the towers; unknown shapes use library defaults
    common.seed_everything(seed)
    if is_main and output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
    barrier(comm)
    if model is None:
        model = AutoModelForSentenceEmbedding(retriever_name_or_path, use_bnb=use_bnb, get_peft=use_peft,
                                              is_autoregressive=is_autoregressive, device=str(device))
    model.to(device)
    tokenizer = model.tokenizer
    dataset = load_dataset(dataset_or_path)
    columns = ["query_input_ids", "query_attention_mask", "passage_input_ids", "passage_attention_mask"]
    from .. import shards

    data_id = shards.dataset_identity(dataset)
    if token_cache_dir and data_id is None:
        import warnings

        warnings.warn("--token_cache_dir ignored: the dataset carries no content fingerprint, a cache could not be invalidated")
        token_cache_dir = None
    fp = shards.fingerprint(data=data_id, rows=len(dataset), tok=shards.tokenizer_identity(tokenizer),
                            cols=(query_column_name, passage_column_name), lens=(query_max_len, passage_max_len))
    processed = shards.load_token_shards(token_cache_dir, fp) if token_cache_dir else None
    if processed is None:
        mapped = dataset.map(
            lambda ex: preprocess_dataset(ex, tokenizer, query_column_name=query_column_name,
                                          passage_column_name=passage_column_name, query_max_len=query_max_len,
                                          passage_max_len=passage_max_len),
            batched=True, remove_columns=dataset.column_names, desc="Running tokenizer on dataset")
        processed = shards.columns_from_dataset(mapped, columns)
        if token_cache_dir and is_main:
            shards.save_token_shards(processed, token_cache_dir, fp)
    if use_peft and is_main:
        model.print_trainable_parameters()
    trim = dict(groups=[("query_input_ids", "query_attention_mask"), ("passage_input_ids", "passage_attention_mask")]) \
        if trim_padding else None
    pack = None
    if pack_tokens and not is_autoregressive:
        from ... import packed as packed_mod

        pack = dict(groups=packed_mod.RETRIEVER_GROUPS, multiple=packed_mod.ROW_MULTIPLES)
    batches = common.ShardedBatches(processed, per_device_train_batch_size, comm.rank, comm.world_size,
                                    seed if seed is not None else 0, columns,
                                    bucket_by="passage_attention_mask" if length_bucketing else None, trim=trim,
                                    pack=pack)
    params = [p for p in model.parameters() if p.requires_grad]
    # one GPU: the whole step is captured once and replayed as a hipGraph (capturable Adam + tensor lr);
    # W > 1 launches eagerly (RCCL collectives stay outside graphs for now)
    from ...fused import LocalComm
    from ..graphed import GraphedStep, TensorLRScheduler, make_capturable_adam

    use_graph = isinstance(comm, LocalComm) and not no_hip_graph and gradient_accumulation_steps == 1
    optimizer = (make_capturable_adam(params, learning_rate, device) if use_graph
                 else torch.optim.Adam(params, lr=learning_rate, fused=True))
    per_epoch, max_train_steps, num_train_epochs = common.steps_and_epochs(
        len(batches), gradient_accumulation_steps, num_train_epochs, max_train_steps)
    from transformers import get_scheduler

    def make_schedule(o):
        return get_scheduler(name=getattr(lr_scheduler_type, "value", lr_scheduler_type), optimizer=o,
                             num_warmup_steps=num_warmup_steps, num_training_steps=max_train_steps)

    scheduler = TensorLRScheduler(optimizer, learning_rate, make_schedule) if use_graph else make_schedule(optimizer)
    if checkpointing_steps is not None and str(checkpointing_steps).isdigit():
        checkpointing_steps = int(checkpointing_steps)
    tracker = common.Tracker(with_tracking, output_dir, "peft_contrastive_learning", config, is_main)

    def save_models(path: str) -> None:
        save_submodel(model.model, path)

    starting_epoch, resume_step, completed = 0, None, 0
    if resume_from_checkpoint:
        load_submodel(model.model, resume_from_checkpoint)
        saved = common.load_training_state(resume_from_checkpoint, optimizer, scheduler)
        starting_epoch, resume_step, completed = common.parse_resume(resume_from_checkpoint, per_epoch, len(batches),
                                                                     gradient_accumulation_steps, saved)
    step_fn = RetrieverStep(model, optimizer, scheduler, logit_scale, comm=comm,
                            autocast_dtype=torch.bfloat16 if mixed_precision == "bf16" else None,
                            grad_accum=gradient_accumulation_steps)
    if use_graph:
        step_fn = GraphedStep(step_fn, warmup=0, eager_steps=2, max_graphs=32 if pack else 8)
    meter = common.Throughput()
    saver = common.AsyncSaver() if async_checkpoint else None

    def save_state(path: str, position) -> None:
        common.save_training_state(path, model, optimizer, scheduler, position, save_models, rank=comm.rank,
                                   world=comm.world_size, saver=saver, barrier=lambda: barrier(comm))

    progress = common.Progress(comm=comm, is_main=is_main, tracker=tracker, meter=meter, on_step=on_step,
                               checkpointing_steps=checkpointing_steps, output_dir=output_dir,
                               max_train_steps=max_train_steps, save_state=save_state, num_batches=len(batches),
                               grad_accum=gradient_accumulation_steps, completed=completed, log=logger)
    for epoch in range(starting_epoch, num_train_epochs):
        model.train()
        total_loss = torch.zeros((), device=device)
        skip = resume_step if (resume_from_checkpoint and epoch == starting_epoch and resume_step) else 0
        step, loss, stop = -1, None, False
        for step, batch in enumerate(batches.epoch(epoch, device, skip)):
            loss = step_fn(batch)  # rank share of the global-batch loss
            total_loss += loss
            meter.add(batch["query_input_ids"].shape[0] * comm.world_size)
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
                save_models(os.path.join(output_dir, "retriever"))
                tokenizer.save_pretrained(os.path.join(output_dir, "retriever"))
            barrier(comm)
    if saver is not None:
        saver.wait()
    tracker.close()


def main() -> None:
    from ... import configure_hw_queues

    configure_hw_queues()  # before the HIP runtime starts (see dalm_amd/__init__.py)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(name)s - %(message)s")
    a = parse_args()
    kw = {k: v for k, v in vars(a).items() if k not in ("dataset_path", "retriever_name_or_path")}
    train_retriever(a.retriever_name_or_path, a.dataset_path, **kw)


if __name__ == "__main__":
    main()
