"""Shared trainer machinery: flag tables -> argparse, data loading, schedules, checkpoint naming, logging.

Mirrors the behaviour of the reference drivers (train_rage2e.py:229-527, train_retriever_only.py:175-422)
without `accelerate`: one process per GPU (torchrun env), torch.distributed(nccl = RCCL) for the few
collectives, explicit HIP streams for overlap.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import logging
import math
import os
import random
import time
from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple

import torch

logger = logging.getLogger("dalm_amd.train")

SCHEDULERS = ["linear", "cosine", "cosine_with_restarts", "polynomial", "constant", "constant_with_warmup"]


def build_parser(description: str, flags: List[Tuple[str, Dict[str, Any]]]) -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=description)
    for name, spec in flags:
        ap.add_argument("--" + name, **spec)
    return ap


def seed_everything(seed: Optional[int]) -> None:
    if seed is None:
        return
    import numpy as np

    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class ShardedBatches:
    """Shuffled fixed-size batches of a tokenised `datasets.Dataset`, sharded over ranks.

    Rank r takes rows [r*B, (r+1)*B) of every global batch of W*B rows (the layout the sharded loss
    assumes).  Like the reference's DataLoader there is no drop_last on one GPU (the last batch is
    partial); with W > 1 a tail that cannot give every rank the same number of rows is dropped so that
    the all-gathers stay rectangular (the reference would pad by re-using samples instead).

    Host data path (SURVEY section 8 f, rank 2): every column is materialised ONCE as a contiguous int64
    tensor [N, T] (pinned when a GPU is present), a batch is one index_select per column into a pinned
    staging buffer, and the host->device copy of batch i+1 is issued on a copy stream while step i runs
    (the reference tokenises to python lists and collates + copies synchronously every step)."""

    def __init__(self, dataset, batch_size: int, rank: int, world: int, seed: int, columns: List[str], *,
                 bucket_by: Optional[str] = None, trim: Optional[Dict[str, Any]] = None,
                 live_rows: Optional[Dict[str, Any]] = None, pack: Optional[Dict[str, Any]] = None):
        """dataset: a tokenised `datasets.Dataset` or a dict of int tensors/arrays (e.g. `shards.load_token_shards`).
        bucket_by: name of an attention-mask column - batches are formed from rows of similar length (opt-in: it changes
        which rows share a batch, i.e. the in-batch negatives; the reference batches at random).
        trim: kwargs of `shards.trim_batch` - all-padding columns are dropped on the host before the copy (opt-in:
        shapes then vary from batch to batch; loss-preserving, see shards.py).
        live_rows: dict(mask=<generator attention-mask column>, multiple=<row granularity>) - every batch also carries
        `generator_live_rows` (fused.live_row_index of the host copy of that mask) for the fused lm_head path.
        pack: dict(groups=<packed.RAG_GROUPS / RETRIEVER_GROUPS>, multiple={prefix: rows}) - every batch also carries the
        packed row lists of its towers (`<prefix>_pack_rows` / `<prefix>_pack_cu`, dalm_amd/packed.py: the towers then run
        on the live tokens only; same loss and gradients)."""
        self.B, self.rank, self.world, self.seed = batch_size, rank, world, seed
        self.columns = columns
        self.n = len(dataset[columns[0]]) if isinstance(dataset, dict) else len(dataset)
        pin = torch.cuda.is_available()
        self.data: Dict[str, torch.Tensor] = {}
        for k in columns:
            # int32 on the host (half the memory and PCIe bytes); int64 again on the device
            t = torch.as_tensor(dataset[k]).to(torch.int32).contiguous()
            self.data[k] = t.pin_memory() if pin else t
        self.bucket_by, self.trim, self.live_rows = bucket_by, trim, live_rows
        self.pack = pack
        self._lengths = (self.data[bucket_by] != 0).sum(dim=1) if bucket_by else None
        n = self.n
        if world == 1:
            self.num_batches = math.ceil(n / batch_size)
        else:
            full = n // (batch_size * world)
            rest = (n - full * batch_size * world) // world
            self.num_batches = full + (1 if rest > 0 else 0)
        self._copy_stream = torch.cuda.Stream() if pin else None

    def __len__(self) -> int:
        return self.num_batches

    def _rows(self, perm: torch.Tensor, i: int) -> torch.Tensor:
        W, B = self.world, self.B
        pos = i * W * B
        remaining = self.n - pos
        b = B if remaining >= W * B else (remaining // W if W > 1 else remaining)
        return perm[pos + self.rank * b: pos + (self.rank + 1) * b]

    def _stage(self, rows: torch.Tensor, device: torch.device):
        if device.type != "cuda" or self.trim:
            host = {k: v.index_select(0, rows) for k, v in self.data.items()}
            if self.trim:
                from .shards import trim_batch

                host = trim_batch(host, **self.trim)
            if device.type != "cuda":
                dev = {k: v.long() for k, v in host.items()}
                self._add_live_rows(dev, host, device)
                return dev, None
            ev = torch.cuda.Event()
            with torch.cuda.stream(self._copy_stream):   # shapes vary: per-batch pinned copies instead of fixed staging
                dev = {k: v.pin_memory().to(device, non_blocking=True).long() for k, v in host.items()}
                self._add_live_rows(dev, host, device)
                ev.record(self._copy_stream)
            return dev, ev
        # two persistent pinned staging sets (batch i+1 is staged while batch i's copy may still be in flight);
        # index_select writes straight into the pinned buffer: no per-step pinned allocation
        slot = self._slot = (getattr(self, "_slot", -1) + 1) % 2
        if not hasattr(self, "_staging"):
            self._staging = [{k: torch.empty((self.B,) + tuple(v.shape[1:]), dtype=v.dtype).pin_memory()
                              for k, v in self.data.items()} for _ in range(2)]
            self._staging_free = [None, None]
        if self._staging_free[slot] is not None:
            self._staging_free[slot].synchronize()  # the copy that last read this staging set has finished
        n = rows.numel()
        ev = torch.cuda.Event()
        with torch.cuda.stream(self._copy_stream):
            dev = {}
            for k, v in self.data.items():
                host = self._staging[slot][k][:n]
                torch.index_select(v, 0, rows, out=host)
                dev[k] = host.to(device, non_blocking=True).long()
            self._add_live_rows(dev, {k: self._staging[slot][k][:n] for k in self.data}, device, slot)
            ev.record(self._copy_stream)
        self._staging_free[slot] = ev
        return dev, ev

    def _add_live_rows(self, dev: Dict[str, torch.Tensor], host: Dict[str, torch.Tensor], device: torch.device,
                       slot: Optional[int] = None) -> None:
        """The rows of the generator batch that carry loss, listed where the mask is still host memory (no device sync).
        slot: index of the persistent pinned staging set this batch uses (its previous copy has completed) - the list
        then goes through a persistent pinned buffer of that slot as well; None: a per-batch pinned copy."""
        self._add_pack_plans(dev, host, device)
        if not self.live_rows:
            return
        from ..fused import live_row_index

        mask = host[self.live_rows["mask"]]
        idx = live_row_index(mask, int(self.live_rows.get("multiple", 256)))
        if idx is None:
            return
        if device.type != "cuda":
            dev["generator_live_rows"] = idx
            return
        if slot is None:
            dev["generator_live_rows"] = idx.pin_memory().to(device, non_blocking=True)
            return
        if not hasattr(self, "_live_staging"):
            cap = self.B * int(self.data[self.live_rows["mask"]].shape[1])
            self._live_staging = [torch.empty((cap,), dtype=torch.int64).pin_memory() for _ in range(2)]
        buf = self._live_staging[slot][:idx.numel()]
        buf.copy_(idx)
        dev["generator_live_rows"] = buf.to(device, non_blocking=True)

    def _add_pack_plans(self, dev: Dict[str, torch.Tensor], host: Dict[str, torch.Tensor], device: torch.device) -> None:
        """Packed row lists of every tower (dalm_amd/packed.py), from the HOST copy of the masks; tens of KB per batch, copied
        on the copy stream with the batch."""
        if not self.pack:
            return
        from ..packed import PACK_MULTIPLE, pack_plan

        mult = self.pack.get("multiple", {})
        for prefix, _ids, mask_key, shifted in self.pack["groups"]:
            if mask_key not in host:
                continue
            rows, cu = pack_plan(host[mask_key], shifted, int(mult.get(prefix, PACK_MULTIPLE)))
            if device.type == "cuda":
                rows, cu = rows.pin_memory().to(device, non_blocking=True), cu.pin_memory().to(device, non_blocking=True)
            dev[f"{prefix}_pack_rows"], dev[f"{prefix}_pack_cu"] = rows, cu

    def epoch(self, epoch: int, device: torch.device, skip: int = 0) -> Iterable[Dict[str, torch.Tensor]]:
        g = torch.Generator().manual_seed(self.seed + epoch)
        if self.bucket_by:
            from .shards import bucketed_order

            perm = bucketed_order(self._lengths, self.B * self.world, g)
        else:
            perm = torch.randperm(self.n, generator=g)
        order = list(range(skip, self.num_batches))
        nxt = self._stage(self._rows(perm, order[0]), device) if order else None
        for j, i in enumerate(order):
            cur, ev = nxt
            nxt = self._stage(self._rows(perm, order[j + 1]), device) if j + 1 < len(order) else None
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
                for t in cur.values():
                    t.record_stream(torch.cuda.current_stream())
            yield cur


_ABS_POS_MODULES = ("wpe", "embed_positions", "position_embeddings", "positions_embed", "position_embedding", "pos_emb")


def has_absolute_positions(model) -> Optional[str]:
    """Why a generator cannot have its left padding trimmed, or None when it can.  Trimming leading all-padding columns
    shifts every token of a left-padded row; that is only loss-preserving when positions enter through rotary / ALiBi /
    relative terms.  Detected structurally (a denylist of model_type strings missed gpt_bigcode, ctrl, xglm,
    biogpt, ...): a learned or sinusoidal position TABLE among the modules, or a config that says "absolute", refuses;
    so does a config that names no relative scheme at all."""
    cfg = getattr(model, "config", None)
    if getattr(cfg, "position_embedding_type", None) == "absolute":
        return "config.position_embedding_type == 'absolute'"
    for name, _ in model.named_modules():
        leaf = name.rsplit(".", 1)[-1]
        if leaf in _ABS_POS_MODULES:
            return f"module {name!r} is an absolute position table"
    relative = any(getattr(cfg, k, None) not in (None, False) for k in
                   ("rope_theta", "rope_parameters", "rope_scaling", "rotary_pct", "rotary_dim", "rotary_emb_base", "alibi",
                    "use_alibi", "relative_attention_num_buckets")) or \
        getattr(cfg, "position_embedding_type", None) in ("relative_key", "relative_key_query", "rotary", "alibi") or \
        any(type(m).__name__.lower().endswith("rotaryembedding") for m in model.modules())
    if not relative:
        return f"model_type={getattr(cfg, 'model_type', None)!r} names no rotary / ALiBi / relative position scheme"
    return None


def resolve_mixed_precision(mixed_precision: Optional[str]) -> str:
    """The reference builds `Accelerator()` without arguments (train_rage2e.py:276, train_retriever_only.py:218): its precision
    is accelerate's - the ACCELERATE_MIXED_PRECISION environment variable (what `accelerate launch --mixed_precision` /
    `accelerate config` export), else "no" (fp32).  An explicit argument wins; None follows the same rule, so an unchanged
    caller gets the reference's precision.  "bf16" is the fast setting (bench.py, tools/trainer_bench.py pass it)."""
    if mixed_precision is None:
        mixed_precision = os.environ.get("ACCELERATE_MIXED_PRECISION", "no").lower() or "no"
    if mixed_precision not in ("no", "bf16"):
        raise ValueError(f"mixed_precision {mixed_precision!r}: this package trains in 'no' (fp32) or 'bf16'")
    return mixed_precision


def effective_grad_accum(gradient_accumulation_steps: int) -> int:
    """--gradient_accumulation_steps N: the trainers accumulate N micro-batches (each scaled 1/N) per optimizer / scheduler
    step, so the number of optimizer steps, the LR schedule and the step_N / resume arithmetic are the reference's
    (ceil(batches / N) steps per epoch).  One deliberate difference: the reference zeroes the model's gradients after
    every micro-batch (train_rage2e.py:474), which leaves only the LAST micro-batch in each update; here all N count.
    With N > 1 the whole-step hipGraph is not used (the graph would bake the optimizer step into every replay)."""
    n = 1 if gradient_accumulation_steps is None else int(gradient_accumulation_steps)
    if n < 1:
        raise ValueError("gradient_accumulation_steps must be >= 1")
    return n


def steps_and_epochs(num_batches: int, grad_accum: int, num_train_epochs: int, max_train_steps: Optional[int]):
    per_epoch = math.ceil(num_batches / grad_accum)
    if max_train_steps is None:
        max_train_steps = num_train_epochs * per_epoch
    num_train_epochs = math.ceil(max_train_steps / max(per_epoch, 1))
    return per_epoch, max_train_steps, num_train_epochs


def parse_resume(path: str, per_epoch: int, num_batches: int, grad_accum: int, extra: Optional[Dict[str, Any]] = None):
    """`step_N` / `epoch_N` folder name -> (starting_epoch, batches_to_skip, completed_steps).

    An epoch takes per_epoch = ceil(num_batches / grad_accum) optimizer steps (the last one may be the end-of-epoch flush
    over fewer micro-batches), so step_K lies in epoch K // per_epoch, (K % per_epoch) * grad_accum micro-batches into it.
    With grad_accum == 1 this is the reference's arithmetic (train_rage2e.py:400-411); with grad_accum > 1 the reference
    divides K * grad_accum by the number of batches, which drifts by one batch per epoch whenever num_batches % grad_accum
    != 0 and then loses the completed epochs from `completed_steps` - not reproduced.  When the checkpoint's
    trainer_state carries the position it was written at (`extra`: epoch, batch_in_epoch, completed_steps - written by
    this trainer for the same batch geometry) that position is used as is."""
    tag = os.path.splitext(os.path.basename(path.rstrip("/")))[0]
    if "epoch" in tag:
        e = int(tag.replace("epoch_", "")) + 1
        return e, None, e * per_epoch
    k = int(tag.replace("step_", ""))
    if extra and extra.get("completed_steps") == k and extra.get("num_batches") == num_batches \
            and extra.get("grad_accum") == grad_accum and "epoch" in extra and "batch_in_epoch" in extra:
        return int(extra["epoch"]), int(extra["batch_in_epoch"]), k
    e = k // max(per_epoch, 1)
    return e, (k % max(per_epoch, 1)) * grad_accum, k


class Progress:
    """What both trainers do after an optimizer step - the normal one and the end-of-epoch flush of pending micro-batches
    go through the same block: count it, `on_step`, the every-100-batches log line, `checkpointing_steps`, and the
    `max_train_steps` stop (reference train_rage2e.py:476-494, train_retriever_only.py:381-397)."""

    def __init__(self, *, comm, is_main: bool, tracker: "Tracker", meter: "Throughput", on_step, checkpointing_steps,
                 output_dir: Optional[str], max_train_steps: int, save_state: Callable[[str, Dict[str, Any]], None],
                 num_batches: int, grad_accum: int, completed: int = 0, log=None):
        self.comm, self.is_main, self.tracker, self.meter, self.on_step = comm, is_main, tracker, meter, on_step
        self.checkpointing_steps, self.output_dir, self.max_train_steps = checkpointing_steps, output_dir, max_train_steps
        self.save_state, self.num_batches, self.grad_accum = save_state, num_batches, grad_accum
        self.completed = completed
        self.log = log or logger

    def position(self, epoch: int, batches_done: int) -> Dict[str, Any]:
        return {"completed_steps": self.completed, "epoch": epoch, "batch_in_epoch": batches_done,
                "num_batches": self.num_batches, "grad_accum": self.grad_accum}

    def after_optimizer_step(self, epoch: int, step: int, skipped: int, loss, total_loss: torch.Tensor) -> bool:
        """step: index of the micro-batch that completed the optimizer step among the batches this run has consumed in this
        epoch (the reference's `step`, which restarts at 0 after a resume); skipped: batches skipped by the resume.
        Returns True when training has to stop (max_train_steps reached)."""
        batch_index = step
        self.completed += 1
        if self.on_step is not None:
            self.on_step(self.completed, loss)
        if (batch_index + 1) % 100 == 0:
            tl = self.comm.all_reduce_sum_(total_loss.clone())
            rate, recent = self.meter.rate(), self.meter.window_rate()
            if self.is_main:
                self.log.info("Step: %d, Loss: %.6f, pairs/s: %.1f (last 100 batches: %.1f)", batch_index + 1,
                              float(tl) / (batch_index + 1), rate, recent)
            self.tracker.log({"train/loss": float(tl) / (batch_index + 1), "train/pairs_per_sec": rate,
                              "train/pairs_per_sec_recent": recent}, self.completed)
        if isinstance(self.checkpointing_steps, int) and self.completed % self.checkpointing_steps == 0 and self.output_dir:
            self.save_state(os.path.join(self.output_dir, f"step_{self.completed}"),
                            self.position(epoch, skipped + batch_index + 1))
        return self.completed >= self.max_train_steps


class Tracker:
    """Tiny stand-in for accelerate's trackers: JSON-lines under <output_dir>/logs (tensorboard when present)."""

    def __init__(self, enabled: bool, output_dir: Optional[str], run_name: str, config: Dict[str, Any], is_main: bool):
        self.f = None
        self.tb = None
        if not (enabled and is_main and output_dir):
            return
        d = os.path.join(output_dir, "logs")
        os.makedirs(d, exist_ok=True)
        self.f = open(os.path.join(d, f"{run_name}.jsonl"), "a")
        self.f.write(json.dumps({"config": config}) + "\n")
        try:
            from torch.utils.tensorboard import SummaryWriter  # optional

            self.tb = SummaryWriter(os.path.join(d, run_name))
        except Exception:
            self.tb = None

    def log(self, values: Dict[str, float], step: int) -> None:
        if self.f:
            self.f.write(json.dumps({"step": step, **values}) + "\n")
            self.f.flush()
        if self.tb:
            for k, v in values.items():
                self.tb.add_scalar(k, v, step)

    def close(self) -> None:
        if self.f:
            self.f.close()
        if self.tb:
            self.tb.close()


class Throughput:
    """pairs/sec over rows actually consumed (the reference has no throughput metric).  `rate()` is since construction (it
    includes model warm-up and hipGraph capture); `window_rate()` is since its previous call - both are host clocks around
    asynchronously launched steps: the loader lets the host run at most two batches ahead of the GPU, so over 100 batches
    they follow the device rate."""

    def __init__(self):
        self.t0 = self._wt = time.perf_counter()
        self.rows = self._wrows = 0

    def add(self, rows: int) -> None:
        self.rows += rows

    def rate(self) -> float:
        return self.rows / max(time.perf_counter() - self.t0, 1e-9)

    def window_rate(self) -> float:
        now = time.perf_counter()
        r = (self.rows - self._wrows) / max(now - self._wt, 1e-9)
        self._wt, self._wrows = now, self.rows
        return r


# ---------------------------------------------------------------------------
# checkpoints (SURVEY section 8 f, rank 3): the reference's directory layout (train_rage2e.py:486-524:
# <output_dir>/{step_N,epoch_N}/{retriever,generator} + accelerate's optimizer/scheduler state), written without
# stalling the step (async) and, with W > 1, without every rank writing the same bytes (sharded)
# ---------------------------------------------------------------------------
def _to_host(obj, stream=None):
    """Deep copy of a (nested) state dict with every tensor on the host; device tensors are copied on `stream` into
    pinned memory (non-blocking) so that the training stream never waits for a checkpoint."""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            host = torch.empty(obj.shape, dtype=obj.dtype, device="cpu", pin_memory=True)
            with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
                host.copy_(obj, non_blocking=True)
            return host
        return obj.detach().clone()
    if isinstance(obj, dict):
        return {k: _to_host(v, stream) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_host(v, stream) for v in obj)
    return obj


def shard_optimizer_state(sd: Dict[str, Any], rank: int, world: int) -> Dict[str, Any]:
    """Data-parallel ranks hold identical optimizer state: rank r keeps every world-th entry of `state` (by sorted key),
    rank 0 additionally the param_groups.  `merge_optimizer_shards` is the inverse."""
    keys = sorted(sd["state"].keys())
    mine = {k: sd["state"][k] for i, k in enumerate(keys) if i % world == rank}
    return {"state": mine, "param_groups": sd["param_groups"] if rank == 0 else None, "num_state": len(keys)}


def merge_optimizer_shards(shards: List[Dict[str, Any]]) -> Dict[str, Any]:
    state: Dict[Any, Any] = {}
    groups = None
    for sh in shards:
        state.update(sh["state"])
        if sh.get("param_groups") is not None:
            groups = sh["param_groups"]
    want = shards[0].get("num_state")
    if groups is None or (want is not None and len(state) != want):
        raise RuntimeError(f"incomplete optimizer shards: {len(state)} of {want} entries, param_groups {'present' if groups else 'missing'}")
    return {"state": state, "param_groups": groups}


class AsyncSaver:
    """One background writer thread.  `submit(snapshot, write)`: `snapshot()` runs NOW on the caller's thread (device ->
    pinned host copies enqueued on a side stream, then one event), `write(host_state)` runs on the worker after the
    event has completed.  At most one checkpoint is in flight: a new submit first waits for the previous one.
    The training stream is made to wait for the copies (GPU-side), never for the file system."""

    def __init__(self):
        import threading

        self._threading = threading
        self._thread = None
        self._error: Optional[BaseException] = None
        self._stream = torch.cuda.Stream() if torch.cuda.is_available() else None

    def submit(self, snapshot: Callable[[Any], Any], write: Callable[[Any], None]) -> None:
        self.wait()
        ev = None
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())     # the values as of this point of the training stream
        host_state = snapshot(self._stream)
        if self._stream is not None:
            ev = torch.cuda.Event()
            ev.record(self._stream)
            # the next optimizer update must not overwrite the state while the copies are still reading it: the
            # training stream waits for them ON THE GPU (a few ms of PCIe time for LoRA state); the host returns at
            # once and the file writing stays off the critical path
            torch.cuda.current_stream().wait_event(ev)

        def work():
            try:
                if ev is not None:
                    ev.synchronize()
                write(host_state)
            except BaseException as e:  # surfaced by the next wait()
                self._error = e

        self._thread = self._threading.Thread(target=work, name="dalm-checkpoint-writer", daemon=False)
        self._thread.start()

    def wait(self) -> None:
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._error is not None:
            e, self._error = self._error, None
            raise RuntimeError("asynchronous checkpoint write failed") from e


def _shard_name(rank: int, world: int) -> str:
    return f"optimizer-{rank:05d}-of-{world:05d}.pt"


def save_training_state(path: str, model, optimizer, scheduler, extra: Dict[str, Any], save_models: Callable[[str], None],
                        *, rank: int = 0, world: int = 1, saver: Optional[AsyncSaver] = None,
                        shard_wait_s: float = 600.0, barrier: Optional[Callable[[], None]] = None) -> None:
    """<path>/{retriever,generator,...} via `save_models` (rank 0), <path>/trainer_state.pt (scheduler, extra, and the
    optimizer state when world == 1) and, with world > 1, <path>/optimizer-RRRRR-of-WWWWW.pt per rank.  With `saver` the
    device->host copies are enqueued and the files are written by a background thread (the adapters / models are
    written synchronously: they go through HF / safetensors writers that read the live tensors).

    Re-used directories (the same output_dir and world size as an earlier run): the commit file and this rank's shard of
    the EARLIER checkpoint are unlinked first and, with world > 1, `barrier()` (all ranks) runs before anything new is
    written - from then on a file that exists was written by this save.  Every shard also carries `stamp` (the
    completed-step count), which `load_training_state` checks against the commit file."""
    os.makedirs(path, exist_ok=True)
    if saver is not None:
        saver.wait()              # an earlier asynchronous write into this very directory must not race the unlinks
    stale = [_shard_name(rank, world)] + (["trainer_state.pt"] if rank == 0 else [])
    for name in stale:
        with contextlib.suppress(FileNotFoundError):
            os.unlink(os.path.join(path, name))
    if world > 1:
        if barrier is None:
            raise ValueError("save_training_state with world > 1 needs `barrier` (stale files of a re-used directory "
                             "must be gone on every rank before rank 0 starts waiting for the new shards)")
        barrier()
    if rank == 0:
        save_models(path)
    sched_sd = scheduler.state_dict() if scheduler else None
    stamp = extra.get("completed_steps")

    def snapshot(stream):
        osd = optimizer.state_dict()
        if world > 1:
            osd = shard_optimizer_state(osd, rank, world)
            osd["stamp"] = stamp
        return _to_host({"optimizer": osd, "scheduler": sched_sd, "extra": dict(extra)}, stream)

    def atomic_save(obj, name):
        tmp = os.path.join(path, name + f".tmp{os.getpid()}")
        torch.save(obj, tmp)
        os.replace(tmp, os.path.join(path, name))

    def write(host):
        # COMMIT POINT = trainer_state.pt, written last and atomically: every rank renames its optimizer shard into place
        # first; rank 0 then waits until all W shard files exist before it writes trainer_state.pt - a directory holding
        # that file is complete, one without it is ignored by parse_resume / load_training_state (crash or early resume)
        if world > 1:
            atomic_save(host["optimizer"], _shard_name(rank, world))
            if rank == 0:
                files = [os.path.join(path, _shard_name(r, world)) for r in range(world)]
                deadline = time.time() + shard_wait_s
                while not all(os.path.exists(x) for x in files):
                    if time.time() > deadline:
                        raise RuntimeError(f"checkpoint {path}: optimizer shards of other ranks did not appear within "
                                           f"{shard_wait_s:.0f} s; trainer_state.pt NOT written (checkpoint stays uncommitted)")
                    time.sleep(0.05)
                atomic_save({"optimizer": None, "sharded": world, "stamp": stamp, "scheduler": host["scheduler"],
                             "extra": host["extra"]}, "trainer_state.pt")
        else:
            atomic_save({"optimizer": host["optimizer"], "scheduler": host["scheduler"], "extra": host["extra"]},
                        "trainer_state.pt")

    if saver is not None:
        saver.submit(snapshot, write)
    else:
        host = snapshot(None)
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()   # the non-blocking copies into pinned memory have landed
        write(host)


def load_training_state(path: str, optimizer, scheduler) -> Dict[str, Any]:
    f = os.path.join(path, "trainer_state.pt")
    if not os.path.exists(f):
        import glob

        if glob.glob(os.path.join(path, "optimizer-*-of-*.pt")) or glob.glob(os.path.join(path, "*.tmp*")):
            # shards without the commit file: the run died (or is still writing) between the shard renames and the commit
            raise RuntimeError(f"checkpoint {path} was never committed (optimizer shards present, trainer_state.pt missing); "
                               "resume from the previous step_N / epoch_N directory")
        return {}      # a checkpoint written by the reference trainer: models / adapters only
    st = torch.load(f, map_location="cpu")
    if st.get("sharded"):
        W = int(st["sharded"])
        files = [os.path.join(path, f"optimizer-{r:05d}-of-{W:05d}.pt") for r in range(W)]
        missing = [x for x in files if not os.path.exists(x)]
        if missing:
            raise FileNotFoundError(f"optimizer shards missing: {missing[:2]}")
        shards_ = [torch.load(x, map_location="cpu") for x in files]
        bad = [r for r, sh in enumerate(shards_) if "stamp" in sh and "stamp" in st and sh["stamp"] != st["stamp"]]
        if bad:
            raise RuntimeError(f"checkpoint {path}: optimizer shards of ranks {bad} were written at another step than "
                               f"trainer_state.pt (stamp {st['stamp']}): a mix of two saves, not loadable")
        st["optimizer"] = merge_optimizer_shards(shards_)
    lr_devices = [g["lr"].device if torch.is_tensor(g["lr"]) else None for g in optimizer.param_groups]
    optimizer.load_state_dict(st["optimizer"])
    # per-parameter state follows the parameter's memory layout (the fused Adam kernel requires it): lora_B lives in
    # transposed-dense memory (models/lora.py), checkpoints hold plain contiguous tensors
    for g in optimizer.param_groups:
        for p in g["params"]:
            ps = optimizer.state.get(p)
            if not ps:
                continue
            for k, v in list(ps.items()):
                if torch.is_tensor(v) and v.shape == p.shape and v.dim() > 1 and v.stride() != p.stride():
                    ps[k] = torch.empty_like(p, dtype=v.dtype).copy_(v)
    for g, dev in zip(optimizer.param_groups, lr_devices):  # a tensor lr (capturable Adam) must stay on its device
        if dev is not None:
            g["lr"] = torch.as_tensor(g["lr"], dtype=torch.float32).to(dev)
    if scheduler and st.get("scheduler"):
        scheduler.load_state_dict(st["scheduler"])
    return st.get("extra", {})
