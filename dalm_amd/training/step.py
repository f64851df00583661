"""One optimisation step of the RAG-end2end / retriever-only trainers.

Same work as the reference's step bodies (train_rage2e.py:429-474, train_retriever_only.py:365-379)
with the loss path on the HIP kernels:
    passage tower -> [async all-gather of P on a side stream] -> query tower -> generator ->
    fused contrastive + marginalised-CE loss (gradient of the logits written in the same pass) ->
    backward -> gradient all-reduce (W>1) -> Adam -> scheduler -> zero_grad
The reference runs the query tower first; the order is swapped so the passage all-gather overlaps
the query tower (numerically irrelevant with dropout off; changes only the dropout RNG stream).
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional

import torch

from .. import streams as _streams

from .. import packed as _packed
from ..fused import (GatherHandle, LocalComm, contrastive_loss, pool_l2norm, rag_e2e_loss, rag_e2e_loss_from_hidden,
                     rag_e2e_loss_packed)
from ..sharded import GradBucket, allreduce_grads



def _advance_dropout(batch) -> None:
    """New LoRA dropout masks for this step: one device op on the current stream (captured with the step when it runs as a
    hipGraph, so every replay advances too).  See dalm_amd/models/lora_ops.py."""
    t = next((v for v in batch.values() if torch.is_tensor(v)), None)
    if t is not None and t.is_cuda:
        from ..models import lora_ops

        lora_ops.advance_dropout_seed(t.device)


class _StepBase:
    def __init__(self, model, optimizer, lr_scheduler, logit_scale, comm=None, autocast_dtype=None, ops=None,
                 grad_overlap: bool = True, track_grad_norm: bool = False, grad_accum: int = 1):
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler
        # global L2 norm of the (all-reduced) trainable gradients, left on the device in self.grad_norm right before
        # the optimizer consumes them - the quantity north_star's tolerance is stated on next to the loss
        self.track_grad_norm = track_grad_norm
        self.grad_norm: Optional[torch.Tensor] = None
        # --gradient_accumulation_steps N: N micro-batches (each scaled 1/N, as accelerate scales them) per optimizer /
        # scheduler step; `synced` tells the trainer whether the call it just made took that step.  The reference zeroes the
        # model's gradients after EVERY micro-batch (train_rage2e.py:474), so its update only ever sees the last one of the N -
        # the step COUNT and schedule are the reference's, the dropped micro-batches are not reproduced.
        self.grad_accum = max(1, int(grad_accum))
        self._micro = 0
        self.synced = True
        if self.grad_accum > 1:
            grad_overlap = False          # the all-reduce runs once per optimizer step, after the last micro-batch
        self.logit_scale = logit_scale
        self.comm = comm or LocalComm()
        self.autocast_dtype = autocast_dtype
        self.ops = ops
        self.side_stream = None
        if not isinstance(self.comm, LocalComm) and torch.cuda.is_available():
            self.side_stream = torch.cuda.Stream()
        self.trainable = [p for p in model.parameters() if p.requires_grad]
        # W > 1: gradients live in one flat bucket (no per-step flatten / copy-back); fp32 trainables only
        self.bucket = None
        import os as _os

        if _os.environ.get("DALM_GRAD_BUCKET", "1") != "0" and not isinstance(self.comm, LocalComm) and self.trainable and \
                all(p.dtype == torch.float32 for p in self.trainable) and \
                len({p.device for p in self.trainable}) == 1:
            # bucketed all-reduce issued from post-accumulate hooks while the backward is still running
            self.bucket = GradBucket(self.trainable, self.comm, overlap=None if grad_overlap else False)

    # autocast keeps a cache of the low-precision copies of fp32 parameters (here: the LoRA matrices).  Two towers of
    # the SAME model running on two streams would share those copies without any stream dependency - the second
    # stream can read a copy the first is still writing (seen as NaN losses under hipGraph replay at the bge-small
    # shape).  Steps that run one model on two streams therefore turn the cache off.
    autocast_cache = True

    def _autocast(self):
        if self.autocast_dtype is None:
            return contextlib.nullcontext()
        return torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=self.autocast_cache)

    def _retrieve_pair(self, batch, q_prefix: str, p_prefix: str):
        """Queries AND passages through the encoder in one packed call (one rank: there is no passage all-gather to overlap with
        the query tower) - or None when the batch / model does not allow it."""
        import os as _os

        m = self.model
        enc = getattr(m, "retriever_model", None) or getattr(m, "model", None)
        if (_os.environ.get("DALM_PACK_PAIR", "1") == "0" or not isinstance(self.comm, LocalComm)
                or f"{q_prefix}_pack_rows" not in batch or f"{p_prefix}_pack_rows" not in batch
                or getattr(m, "retriever_is_autoregressive", getattr(m, "is_autoregressive", False))
                or not _packed.attention_is_packable(enc)):
            return None
        q = (batch[f"{q_prefix}_input_ids"], batch[f"{q_prefix}_attention_mask"], batch[f"{q_prefix}_pack_rows"], batch[f"{q_prefix}_pack_cu"])
        p = (batch[f"{p_prefix}_input_ids"], batch[f"{p_prefix}_attention_mask"], batch[f"{p_prefix}_pack_rows"], batch[f"{p_prefix}_pack_cu"])
        hp, hq = _packed.retrieval_hidden_pair(enc, p, q)
        return pool_l2norm(hp, p[1], m.normalize), pool_l2norm(hq, q[1], m.normalize)

    def _finish(self, loss: torch.Tensor) -> torch.Tensor:
        if self.grad_accum > 1:
            (loss / self.grad_accum).backward()
            self._micro += 1
            self.synced = self._micro % self.grad_accum == 0
            if not self.synced:
                return loss.detach()
        else:
            loss.backward()
        return self._apply(loss)

    def flush(self) -> bool:
        """End of an epoch with micro-batches still pending: take the optimizer step on what has accumulated (accelerate
        syncs on the last batch of the dataloader in the same way).  True when a step was taken."""
        if self.grad_accum > 1 and self._micro % self.grad_accum != 0:
            self._micro = 0
            self.synced = True
            self._apply(None)
            return True
        return False

    def _apply(self, loss: Optional[torch.Tensor]):
        if self.bucket is not None:
            self.bucket.all_reduce()
        else:
            allreduce_grads(self.trainable, self.comm)
        if self.track_grad_norm:
            if self.bucket is not None:
                self.grad_norm = torch.linalg.vector_norm(self.bucket.flat)
            else:
                grads = [p.grad for p in self.trainable if p.grad is not None]
                self.grad_norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads)))
        self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        if self.bucket is not None:
            self.bucket.zero()
        else:
            self.model.zero_grad(set_to_none=True)
        return loss.detach() if loss is not None else None


class RagE2EStep(_StepBase):
    """batch keys as produced by preprocess_dataset (rag_e2e_dataloader_utils.py:56-68)."""

    def __init__(self, *a, inplace_grad: bool = True, overlap_towers: bool = True, fuse_lm_head: bool = False,
                 lm_head_chunk: Optional[int] = None, graph_towers: bool = False, graph_after: int = 2, **kw):
        super().__init__(*a, **kw)
        self.inplace_grad = inplace_grad
        # run the decoder without its lm_head and let the loss consume the hidden states chunk by chunk -
        # the [B,Tg,V] logits and their gradient never exist (frozen bf16 head: the library's own MFMA kernels end to end,
        # fused._lm_head_train_kernel).  True / False, or "auto": fused whenever the materialised logits of a batch would exceed
        # DALM_LOGITS_BUDGET_MB (default 1024 MB; resolved on the first batch - the BASELINE batches stay below it, where the
        # materialised path measures ~1 ms per step faster: profiles/r05_lm_head_train_kernels_vs_library.txt)
        self.fuse_lm_head = fuse_lm_head
        self.lm_head_chunk = lm_head_chunk
        # W > 1: graph the towers (fwd + bwd), keep collectives / loss / optimizer eager (GraphedTowers)
        self.graph_towers = graph_towers and torch.cuda.is_available()
        self.towers = None
        self.towers_failed: Optional[str] = None
        self.calls = 0
        self.graph_after = graph_after
        import os as _os2

        self.early_gather = _os2.environ.get("DALM_EARLY_GATHER", "1") != "0"
        self.aux: Dict[str, torch.Tensor] = {}
        # the two retriever towers are many small kernels (3204 tokens through BERT) and are independent of
        # the generator until the loss: run them on their own HIP stream so they fill the gaps between the
        # generator's large GEMMs; autograd replays each backward on its forward stream, so the backward
        # overlaps the same way
        self.tower_stream = _streams.tower_stream() if (overlap_towers and torch.cuda.is_available()) else None

    def _maybe_build_towers(self, batch) -> None:
        """One set of tower graphs per batch shape (packed batches: per row-count combination), at most DALM_TOWER_SETS (4) alive:
        every set keeps its own activations between its forward and backward graphs.  Other shapes launch eagerly."""
        if not self.graph_towers or self.towers_failed is not None:
            return
        if self.calls <= self.graph_after:  # first real steps run eagerly (library warm-up, as GraphedStep)
            return
        from .graphed import GraphedTowers

        key = GraphedTowers.key_of(batch)
        sets = self.__dict__.setdefault("_tower_sets", {})
        if key in sets:
            self.towers = sets[key]
            return
        import os as _os3

        if len(sets) >= int(_os3.environ.get("DALM_TOWER_SETS", "4")):
            self.towers = None
            return
        if any(k in batch for k in GraphedTowers.PACK_KEYS) and not (
                GraphedTowers.is_packed(batch) and _packed.attention_is_packable(self.model.generator_model)
                and _packed.attention_is_packable(self.model.retriever_model)):
            self.towers = None              # row lists for some towers only, or a model the packed call cannot serve: eager
            return
        try:
            torch.cuda.synchronize()
            sets[key] = self.towers = GraphedTowers(self.model, self.autocast_dtype, batch, hidden_only=self.fuse_lm_head)
        except Exception as e:  # same kernels, eager launches
            self.towers_failed = repr(e)
            self.towers = None
            torch.cuda.synchronize()

    def _use_graphs(self, batch) -> bool:
        return self.towers is not None and self.towers.matches(batch)

    def _gather(self, emb):
        """Start the all-gather of an embedding matrix early on the side stream (overlaps the other tower) -
        or, with early_gather off, leave it to the loss (gathered on the main stream right before use)."""
        if not self.early_gather:
            return None
        return GatherHandle(emb.float(), self.comm, self.side_stream)

    def _retrieve(self, batch, side: str):
        """One retriever tower call; on the PACKED rows when the batch carries their list (`retriever_{side}_pack_rows` /
        `_pack_cu`, dalm_amd/packed.py) - same embeddings, the padding never enters the encoder."""
        m = self.model
        ids, mask = batch[f"retriever_{side}_input_ids"], batch[f"retriever_{side}_attention_mask"]
        rows = batch.get(f"retriever_{side}_pack_rows")
        if rows is not None and not m.retriever_is_autoregressive and _packed.attention_is_packable(m.retriever_model):
            h = _packed.retrieval_hidden(m.retriever_model, ids, mask, rows, batch[f"retriever_{side}_pack_cu"])
            return pool_l2norm(h, mask, m.normalize)
        return m("retrieval", ids, mask)

    def _towers(self, batch):
        if not self._use_graphs(batch):
            pair = self._retrieve_pair(batch, "retriever_query", "retriever_passage")
            if pair is not None:
                p_emb, q_emb = pair
                return p_emb, q_emb, self._gather(p_emb), self._gather(q_emb)
        if self._use_graphs(batch):
            p_emb = self.towers.passage(*self.towers.call_args(batch, "retriever_passage", "retriever_passage_input_ids",
                                                               "retriever_passage_attention_mask"))
        else:
            p_emb = self._retrieve(batch, "passage")
        p_gather = self._gather(p_emb)
        if self._use_graphs(batch):
            q_emb = self.towers.query(*self.towers.call_args(batch, "retriever_query", "retriever_query_input_ids",
                                                             "retriever_query_attention_mask"))
        else:
            q_emb = self._retrieve(batch, "query")
        q_gather = self._gather(q_emb)
        return p_emb, q_emb, p_gather, q_gather

    def _packed_generator(self, batch) -> bool:
        if self._use_graphs(batch):
            return self.towers.packed
        return "generator_pack_rows" in batch and _packed.attention_is_packable(self.model.generator_model)

    def _generator(self, batch):
        m = self.model
        if self._use_graphs(batch):            # padded graphs: logits / hidden states; packed graphs: hidden rows [n, H]
            return self.towers.generator(*self.towers.call_args(batch, "generator", "generator_input_input_ids",
                                                                "generator_input_attention_mask"))
        if self._packed_generator(batch):      # final hidden states of the live rows only, [n, H]
            return _packed.generator_hidden(m.generator_model, batch["generator_input_input_ids"],
                                            batch["generator_input_attention_mask"], batch["generator_pack_rows"],
                                            batch["generator_pack_cu"])
        if not self.fuse_lm_head:
            return m("generation", batch["generator_input_input_ids"], batch["generator_input_attention_mask"])
        gm = m.generator_model
        return gm.base_model(input_ids=batch["generator_input_input_ids"],
                             attention_mask=batch["generator_input_attention_mask"], use_cache=False)[0]

    def _resolve_fuse(self, batch) -> None:
        """fuse_lm_head="auto": decided PER BATCH SHAPE (cached) - a later, longer batch that exceeds DALM_LOGITS_BUDGET_MB takes
        the logits-free path even when the first batch was small, and a large first batch does not lock the slower path in
        (ADVICE r5).  Inside a hipGraph the decision is part of the captured shape."""
        if getattr(self, "_fuse_auto", None) is None:
            self._fuse_auto = {} if self.fuse_lm_head == "auto" else False
        if self._fuse_auto is False:
            return
        import os

        head = self.model.generator_model.get_output_embeddings()
        ids = batch["generator_input_input_ids"]
        key = tuple(ids.shape)
        if key not in self._fuse_auto:
            el = 2 if self.autocast_dtype in (torch.bfloat16, torch.float16) or head.weight.dtype != torch.float32 else 4
            mb = ids.shape[0] * ids.shape[1] * head.weight.shape[0] * el / 2 ** 20
            self._fuse_auto[key] = bool(mb > float(os.environ.get("DALM_LOGITS_BUDGET_MB", "1024")) and getattr(head, "bias", None) is None)
        self.fuse_lm_head = self._fuse_auto[key]

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        m = self.model
        self.calls += 1
        self.graph_after = getattr(self, "graph_after", 2)
        self._resolve_fuse(batch)
        self._maybe_build_towers(batch)
        _advance_dropout(batch)
        with self._autocast():
            if self.tower_stream is not None:
                cur = torch.cuda.current_stream()
                self.tower_stream.wait_stream(cur)
                if self._use_graphs(batch):
                    # graph launches cost host milliseconds each: get the generator's big graph onto the GPU
                    # first, then feed the small tower graphs into the gaps (the tower stream only waits for
                    # what was on the main stream BEFORE this point)
                    logits = self._generator(batch)
                    with torch.cuda.stream(self.tower_stream):
                        p_emb, q_emb, p_gather, q_gather = self._towers(batch)
                else:
                    with torch.cuda.stream(self.tower_stream):
                        p_emb, q_emb, p_gather, q_gather = self._towers(batch)
                    logits = self._generator(batch)
                cur.wait_stream(self.tower_stream)
                for t in (p_emb, q_emb, getattr(p_gather, "result", None), getattr(q_gather, "result", None)):
                    if t is not None and t.is_cuda:
                        t.record_stream(cur)
            else:
                p_emb, q_emb, p_gather, q_gather = self._towers(batch)
                logits = self._generator(batch)
        if (self._packed_generator(batch) or self.fuse_lm_head) and self.autocast_dtype is not None \
                and logits.dtype != self.autocast_dtype:
            # the reference's lm_head runs INSIDE the autocast'ed model forward: nn.Linear casts its input to the autocast dtype.
            # A decoder whose final norm is an nn.LayerNorm (Falcon: autocast runs layer_norm in f32) hands back f32 hidden
            # states - without this cast the head GEMMs of the from-hidden paths ran in f32 (2.2 ms each at cfg5)
            logits = logits.to(self.autocast_dtype)
        if self._packed_generator(batch):
            head = m.generator_model.get_output_embeddings()
            if getattr(head, "bias", None) is not None:
                raise NotImplementedError("the packed generator path needs a bias-free lm_head")
            labels, weights = _packed.packed_labels(batch["generator_input_input_ids"], batch["generator_input_attention_mask"],
                                                    batch["generator_pack_rows"])
            loss = rag_e2e_loss_packed(q_emb, p_emb, logits, head.weight, labels, weights,
                                       batch["generator_input_attention_mask"], batch["query_passage_input_len"],
                                       self.logit_scale, comm=self.comm, ops=self.ops, q_gather=q_gather, p_gather=p_gather,
                                       aux=self.aux)
            return self._finish(loss)
        if self.fuse_lm_head:
            head = m.generator_model.get_output_embeddings()
            if getattr(head, "bias", None) is not None:
                raise NotImplementedError("fuse_lm_head needs a bias-free lm_head")
            loss = rag_e2e_loss_from_hidden(q_emb, p_emb, logits, head.weight, batch["generator_input_input_ids"],
                                            batch["generator_input_attention_mask"], batch["query_passage_input_len"],
                                            self.logit_scale, comm=self.comm, ops=self.ops,
                                            chunk_samples=self.lm_head_chunk, q_gather=q_gather, p_gather=p_gather,
                                            aux=self.aux, live_rows=batch.get("generator_live_rows"))
            return self._finish(loss)
        loss = rag_e2e_loss(q_emb, p_emb, logits, batch["generator_input_input_ids"],
                            batch["generator_input_attention_mask"], batch["query_passage_input_len"],
                            self.logit_scale, comm=self.comm, ops=self.ops, inplace_grad=self.inplace_grad,
                            q_gather=q_gather, p_gather=p_gather, aux=self.aux)
        return self._finish(loss)


class RetrieverStep(_StepBase):
    """batch keys as produced by retriever_only_dataloader_utils.preprocess_dataset."""

    def __init__(self, *a, overlap_towers: bool = True, graph_towers: bool = False, graph_after: int = 2, **kw):
        super().__init__(*a, **kw)
        # the query pass (Tq = 50) is small next to the passage pass (Tp = 128): run it on its own stream
        self.tower_stream = _streams.tower_stream() if (overlap_towers and torch.cuda.is_available()) else None
        self.autocast_cache = self.tower_stream is None   # one model on two streams: no shared cast cache
        # the two encoder calls as single-stream hipGraphs (GraphedEncoders), loss / optimizer / collectives eager: padded batches
        # only (the packed batch goes through the encoder in ONE call - a whole-step graph of that is a single stream already)
        # ONE RANK ONLY (LocalComm): next to the W > 1 gradient bucket the second replay of a set gave an infinite gradient norm
        self.graph_towers = (graph_towers and torch.cuda.is_available() and self.tower_stream is not None
                             and isinstance(self.comm, LocalComm))
        self.graph_after = graph_after
        self.calls = 0
        self.towers = None
        self.towers_failed: Optional[str] = None
        self._encoder_sets: Dict[tuple, object] = {}

    def _maybe_build_encoders(self, batch) -> None:
        self.towers = None
        if not self.graph_towers or self.towers_failed is not None or "query_pack_rows" in batch or self.calls <= self.graph_after:
            return
        from .graphed import GraphedEncoders

        key = GraphedEncoders.key_of(batch)
        if key in self._encoder_sets:
            self.towers = self._encoder_sets[key]
            return
        import os as _os4

        if len(self._encoder_sets) >= int(_os4.environ.get("DALM_TOWER_SETS", "4")):
            return
        try:
            torch.cuda.synchronize()
            self._encoder_sets[key] = self.towers = GraphedEncoders(self.model, self.autocast_dtype, batch)
        except Exception as e:  # same kernels, eager launches
            self.towers_failed = repr(e)
            self.towers = None
            torch.cuda.synchronize()

    def _embed(self, batch, side: str):
        """One encoder call; on the PACKED rows when the batch carries their list (`{side}_pack_rows` / `_pack_cu`)."""
        m = self.model
        ids, mask = batch[f"{side}_input_ids"], batch[f"{side}_attention_mask"]
        rows = batch.get(f"{side}_pack_rows")
        if rows is not None and not getattr(m, "is_autoregressive", False) and _packed.attention_is_packable(m.model):
            h = _packed.retrieval_hidden(m.model, ids, mask, rows, batch[f"{side}_pack_cu"])
            return pool_l2norm(h, mask, m.normalize)
        return m(ids, mask)

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        self.calls += 1
        self._maybe_build_encoders(batch)
        _advance_dropout(batch)
        if self.towers is not None:       # graphed encoder calls: the structure of the two-stream branch below
            cur = torch.cuda.current_stream()
            self.tower_stream.wait_stream(cur)
            with torch.cuda.stream(self.tower_stream):
                q_emb = self.towers.query(batch["query_input_ids"], batch["query_attention_mask"])
                q_gather = GatherHandle(q_emb.float(), self.comm, self.side_stream)
            p_emb = self.towers.passage(batch["passage_input_ids"], batch["passage_attention_mask"])
            p_gather = GatherHandle(p_emb.float(), self.comm, self.side_stream)
            cur.wait_stream(self.tower_stream)
            for t in (q_emb, q_gather.result):
                if t is not None and t.is_cuda:
                    t.record_stream(cur)
            loss = contrastive_loss(q_emb, p_emb, self.logit_scale, comm=self.comm, ops=self.ops, q_gather=q_gather,
                                    p_gather=p_gather)
            return self._finish(loss)
        with self._autocast():
            pair = self._retrieve_pair(batch, "query", "passage")
        if pair is not None:             # packed, one rank: both inputs through the encoder in ONE call, one stream
            p_emb, q_emb = pair
            loss = contrastive_loss(q_emb, p_emb, self.logit_scale, comm=self.comm, ops=self.ops,
                                    q_gather=GatherHandle(q_emb.float(), self.comm, self.side_stream),
                                    p_gather=GatherHandle(p_emb.float(), self.comm, self.side_stream))
            return self._finish(loss)
        with self._autocast():
            if self.tower_stream is not None:
                cur = torch.cuda.current_stream()
                self.tower_stream.wait_stream(cur)
                with torch.cuda.stream(self.tower_stream):
                    q_emb = self._embed(batch, "query")
                    q_gather = GatherHandle(q_emb.float(), self.comm, self.side_stream)
                p_emb = self._embed(batch, "passage")
                p_gather = GatherHandle(p_emb.float(), self.comm, self.side_stream)
                cur.wait_stream(self.tower_stream)
                for t in (q_emb, q_gather.result):
                    if t is not None and t.is_cuda:
                        t.record_stream(cur)
            else:
                p_emb = self._embed(batch, "passage")
                p_gather = GatherHandle(p_emb.float(), self.comm, self.side_stream)
                q_emb = self._embed(batch, "query")
                q_gather = GatherHandle(q_emb.float(), self.comm, self.side_stream)
        loss = contrastive_loss(q_emb, p_emb, self.logit_scale, comm=self.comm, ops=self.ops, q_gather=q_gather,
                                p_gather=p_gather)
        return self._finish(loss)
