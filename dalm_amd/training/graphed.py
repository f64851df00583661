"""hipGraph capture of a whole optimisation step (towers + HIP loss path + backward + Adam).

The eager step issues ~10k kernel launches (HF eager elementwise ops dominate the count); replaying
them from one captured graph removes the host launch path from the critical loop.  Static-shape
batches are copied into captured input buffers; a batch with another shape (the partial last batch)
runs eagerly.  Requirements: optimizer built with `capturable=True` and a tensor `lr`
(`make_capturable_adam`), the LR scheduler stepped outside the graph (it `fill_`s the lr tensor).
Single-GPU only for now: with W > 1 the step stays eager (RCCL inside a captured step is left for a
later round).
"""
from __future__ import annotations

import contextlib

from typing import Callable, Dict, Optional, Tuple

import torch


def make_capturable_adam(params, lr: float, device) -> torch.optim.Adam:
    return torch.optim.Adam(params, lr=torch.tensor(float(lr), device=device), fused=True, capturable=True)


def init_adam_state(optimizer: torch.optim.Adam) -> None:
    """Materialise Adam's lazily created state (step, exp_avg, exp_avg_sq) BEFORE a capture.

    If the first optimizer.step() happens inside the capture, the zero-initialisation of the state is
    recorded in the graph and every replay resets the moments - the first replay is right, all later
    ones are wrong (caught by tests/test_step_parity_gpu.py)."""
    for group in optimizer.param_groups:
        for p in group["params"]:
            if not p.requires_grad or len(optimizer.state[p]) != 0:
                continue
            st = optimizer.state[p]
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if group.get("amsgrad"):
                st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)


class TensorLRScheduler:
    """Drives the tensor-valued lr of a capturable optimizer from an ordinary scheduler.

    The scheduler is built on a shadow optimizer with a FLOAT lr (schedulers keep `initial_lr` by
    reference, so attaching one directly to a tensor lr compounds the decay: lr_t = lr_{t-1} * f(t));
    after every shadow step the float is written into the real optimizer's lr tensor with fill_(),
    which a captured graph picks up on its next replay."""

    def __init__(self, optimizer, lr: float, factory: Callable[[torch.optim.Optimizer], object]):
        self.optimizer = optimizer
        self._shadow = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=float(lr))
        self.scheduler = factory(self._shadow)
        self._push()

    def _push(self) -> None:
        lr = float(self._shadow.param_groups[0]["lr"])
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(lr)
            else:
                g["lr"] = lr

    def step(self) -> None:
        self._shadow.step()
        self.scheduler.step()
        self._push()

    def get_last_lr(self):
        return self.scheduler.get_last_lr()

    def state_dict(self):
        return self.scheduler.state_dict()

    def load_state_dict(self, sd) -> None:
        self.scheduler.load_state_dict(sd)
        # LambdaLR.load_state_dict restores the schedule position but not the shadow optimizer's lr:
        # take it from the restored schedule, otherwise _push() would overwrite the (correct) lr that
        # optimizer.load_state_dict has just restored with the shadow's construction-time value
        last = sd.get("_last_lr") or self.scheduler.get_last_lr()
        for g, lr in zip(self._shadow.param_groups, last):
            g["lr"] = float(lr)
        self._push()


class _WarmBlas(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        torch.cuda.current_blas_handle()          # runs on autograd's device thread: that thread's handle
        return g


def warm_blas_handles(device=None) -> None:
    """Create the hipBLAS / rocBLAS handles of THIS thread and of autograd's device thread now.  They are made lazily on first
    use; the tuned-solution table (dalm_amd/tuning) maps some GEMM shapes to rocBLAS solutions, and a shape that meets its first
    rocBLAS solution INSIDE a hipGraph capture made the capture fail in `hipblasCreate` (seen in the retriever-only trainer on a
    new packed row count) - after which the eager fall-back died in torch's dropout (its generator was left in capture mode)."""
    if not torch.cuda.is_available():
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    torch.cuda.current_blas_handle()
    x = torch.zeros(1, device=dev, requires_grad=True)
    _WarmBlas.apply(x).sum().backward()
    a = torch.zeros(8, 8, device=dev)
    torch.mm(a, a)
    torch.cuda.synchronize()


class GraphedStep:
    def __init__(self, step, warmup: int = 3, eager_steps: int = 0, max_graphs: int = 8):
        """warmup: hidden extra steps run on a side stream right before the capture (benchmarks);
        eager_steps: the first N REAL steps are launched eagerly and the capture happens after them
        (trainers: no hidden steps, and every library - hipBLASLt workspaces, GEMM solution lookup,
        allocator pools - has seen the shapes before anything is captured)."""
        self.step = step
        self.warmup = warmup
        self.eager_steps = eager_steps
        self.max_graphs = max_graphs
        self.calls = 0
        # one graph per batch shape (trimmed / bucketed batches come in a handful of lengths); all graphs share ONE
        # memory pool - they never run concurrently, so their activations can overlay each other
        self.graphs: Dict[Tuple, Tuple[torch.cuda.CUDAGraph, Dict[str, torch.Tensor], torch.Tensor]] = {}
        self._outputs: Dict[Tuple, Tuple[Optional[torch.Tensor], dict]] = {}
        self.pool = None
        self.failed: Optional[str] = None
        self.replays = 0
        self.eager_calls = 0
        # the LR scheduler must not be captured: it runs on the host and writes the lr tensor
        self.scheduler = step.lr_scheduler
        step.lr_scheduler = None

    # first captured graph, for callers that only ever see one shape (bench.py, tests)
    @property
    def graph(self) -> Optional[torch.cuda.CUDAGraph]:
        return next(iter(self.graphs.values()))[0] if self.graphs else None

    @property
    def grad_norm(self) -> Optional[torch.Tensor]:
        return getattr(self.step, "grad_norm", None)

    @staticmethod
    def _key(batch) -> Tuple:
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()))

    def _capture(self, batch) -> None:
        static = {k: v.clone() for k, v in batch.items()}
        if getattr(self, "_warm_stream", None) is None:
            self._warm_stream = torch.cuda.Stream()
        s = self._warm_stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up off the default stream, as the capture API asks
            for _ in range(self.warmup):
                self.step(static)
                if self.scheduler is not None:
                    self.scheduler.step()
        torch.cuda.current_stream().wait_stream(s)
        init_adam_state(self.step.optimizer)  # no-op after a warm-up step; essential with warmup == 0
        if not getattr(self, "_blas_warm", False):
            warm_blas_handles()
            self._blas_warm = True
        torch.cuda.synchronize()
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # manual begin/end instead of `with torch.cuda.graph(g)`: when the body raises (an op the capture refuses),
        # that context manager's exit raises a second error from capture_end() and never restores the current stream -
        # every later eager step then runs on a stream stuck in an invalidated capture (seen with HF Falcon's
        # list-indexed head split: the fall-back step died in dropout's RNG-state lookup)
        g = torch.cuda.CUDAGraph()
        # ONE capture stream for every graph of this step: the caching allocator hands a freed block only to requests on the
        # stream it was allocated on, so a fresh stream per capture meant that no graph could reuse the pool memory of the graphs
        # before it (reserved memory grew by a full set of activations per batch shape: 12 trimmed shapes at cfg3 ran out of
        # 288 GB; tools/graph_pool_probe.py)
        if getattr(self, "_capture_stream", None) is None:
            self._capture_stream = torch.cuda.Stream()
        cs = self._capture_stream
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            g.capture_begin(pool=self.pool, capture_error_mode="thread_local")   # other threads (a process group's watchdog) may query events
            try:
                static_loss = self.step(static)
            except BaseException:
                try:
                    g.capture_end()
                except Exception:
                    pass
                raise
            g.capture_end()
        torch.cuda.current_stream().wait_stream(cs)
        self.graphs[self._key(batch)] = (g, static, static_loss)
        # per-graph views of what the step leaves behind for its caller (each capture allocates its own tensors):
        # the replay path points the step back at THIS graph's outputs
        self._outputs[self._key(batch)] = (getattr(self.step, "grad_norm", None), dict(getattr(self.step, "aux", None) or {}))

    def __call__(self, batch) -> torch.Tensor:
        self.calls += 1
        key = self._key(batch)
        if (self.failed is None and key not in self.graphs and len(self.graphs) < self.max_graphs
                and self.calls > self.eager_steps):
            try:
                self._capture(batch)
                # the capture itself does not execute the step; fall through to the replay below
            except Exception as e:  # keep training: same kernels, eager launches
                self.failed = repr(e)
                import warnings

                warnings.warn(f"dalm_amd: hipGraph capture of the step failed ({self.failed[:500]}); the steps launch eagerly from here on")
                torch.cuda.synchronize()
        hit = self.graphs.get(key)
        if hit is not None:
            g, static, static_loss = hit
            for k, v in batch.items():
                static[k].copy_(v, non_blocking=True)
            g.replay()
            loss = static_loss
            self.replays += 1
            gn, aux = self._outputs.get(key, (None, {}))
            if gn is not None:
                self.step.grad_norm = gn
            if aux and isinstance(getattr(self.step, "aux", None), dict):
                self.step.aux.update(aux)
        else:
            loss = self.step(batch)
            self.eager_calls += 1
        if self.scheduler is not None:
            self.scheduler.step()
        return loss


# ---------------------------------------------------------------------------
# Multi-GPU form: graph the towers, keep the collectives eager
# ---------------------------------------------------------------------------
class _RetrievalCall(torch.nn.Module):
    """retriever forward + HIP pool/normalise as one graph-capturable callable (owns only the retriever)."""

    def __init__(self, rag_model, autocast_dtype):
        super().__init__()
        self.retriever = rag_model.retriever_model
        self.normalize = rag_model.normalize
        self.autocast_dtype = autocast_dtype

    def forward(self, input_ids, attention_mask):
        from ..fused import pool_l2norm

        if self.autocast_dtype is None:
            h = self.retriever(input_ids, attention_mask)[0]
        else:
            with torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=False):
                h = self.retriever(input_ids, attention_mask)[0]
        return pool_l2norm(h, attention_mask, self.normalize)


@contextlib.contextmanager
def thread_local_capture():
    """torch.cuda.make_graphed_callables captures in hipStreamCaptureModeGlobal: while the capture is open ANY thread's event
    query is an error - and torch.distributed's ProcessGroupNCCL watchdog thread polls the events of outstanding collectives.
    With a live process group the tower captures aborted the process in 5 of 8 runs of the one-rank tests ("operation not permitted
    when stream is capturing", raised in the watchdog thread, which terminates the process).  Thread-local mode keeps the check for the
    capturing thread and lets other threads be."""
    orig = torch.cuda.graph.__init__

    def init(self, cuda_graph, pool=None, stream=None, capture_error_mode="thread_local"):
        orig(self, cuda_graph, pool=pool, stream=stream, capture_error_mode="thread_local")

    torch.cuda.graph.__init__ = init
    try:
        yield
    finally:
        torch.cuda.graph.__init__ = orig


class _EncoderCall(torch.nn.Module):
    """AutoModelForSentenceEmbedding.forward (encoder + pooling + normalisation) as one graph-capturable callable."""

    def __init__(self, model, autocast_dtype):
        super().__init__()
        self.model = model
        self.autocast_dtype = autocast_dtype

    def forward(self, input_ids, attention_mask):
        if self.autocast_dtype is None:
            return self.model(input_ids, attention_mask)
        with torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=False):
            return self.model(input_ids, attention_mask)


class GraphedEncoders:
    """The retriever-only step's two encoder calls (query, passage), forward AND backward, as single-stream hipGraphs
    (torch.cuda.make_graphed_callables, one call each: separate pools, so the query graphs replay on the tower stream while the
    passage graphs run on the main stream); loss, optimizer and collectives stay eager.  Same reason as GraphedTowers at one rank:
    a whole-step graph captured across the two streams replays with a dependency bubble per node."""

    def __init__(self, model, autocast_dtype, sample_batch: Dict[str, torch.Tensor]):
        if getattr(model, "is_autoregressive", False):
            raise NotImplementedError("graphed encoders: autoregressive retrievers run eagerly")
        warm_blas_handles()
        self.key = self.key_of(sample_batch)
        calls = [_EncoderCall(model, autocast_dtype) for _ in range(2)]
        for c in calls:
            c.train(model.training)
        b = sample_batch
        with thread_local_capture():
            self.query = torch.cuda.make_graphed_callables(
                calls[0], (b["query_input_ids"].clone(), b["query_attention_mask"].clone()), num_warmup_iters=3, allow_unused_input=True)
            self.passage = torch.cuda.make_graphed_callables(
                calls[1], (b["passage_input_ids"].clone(), b["passage_attention_mask"].clone()), num_warmup_iters=3, allow_unused_input=True)

    KEYS = ("query_input_ids", "passage_input_ids")

    @classmethod
    def key_of(cls, batch):
        return tuple(tuple(batch[k].shape) for k in cls.KEYS)

    def matches(self, batch) -> bool:
        return self.key_of(batch) == self.key


class _GeneratorCall(torch.nn.Module):
    """hidden_only: stop at the decoder's final (normed) hidden states - the fused lm_head path consumes those."""

    def __init__(self, rag_model, autocast_dtype, hidden_only: bool = False):
        super().__init__()
        self.generator = rag_model.generator_model
        self.autocast_dtype = autocast_dtype
        self.hidden_only = hidden_only

    def _run(self, input_ids, attention_mask):
        if self.hidden_only:
            return self.generator.base_model(input_ids=input_ids, attention_mask=attention_mask, use_cache=False)[0]
        return self.generator(input_ids=input_ids, attention_mask=attention_mask, use_cache=False).logits   # no KV cache copies

    def forward(self, input_ids, attention_mask):
        if self.autocast_dtype is None:
            return self._run(input_ids, attention_mask)
        with torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=False):
            return self._run(input_ids, attention_mask)


class _PackedRetrievalCall(torch.nn.Module):
    """`_RetrievalCall` on the packed rows of a batch (dalm_amd/packed.py): ids, mask, rows, cu -> embeddings."""

    def __init__(self, rag_model, autocast_dtype):
        super().__init__()
        self.retriever = rag_model.retriever_model
        self.normalize = rag_model.normalize
        self.autocast_dtype = autocast_dtype

    def forward(self, input_ids, attention_mask, rows, cu):
        from .. import packed
        from ..fused import pool_l2norm

        if self.autocast_dtype is None:
            h = packed.retrieval_hidden(self.retriever, input_ids, attention_mask, rows, cu)
        else:
            with torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=False):
                h = packed.retrieval_hidden(self.retriever, input_ids, attention_mask, rows, cu)
        return pool_l2norm(h, attention_mask, self.normalize)


class _PackedGeneratorCall(torch.nn.Module):
    """The decoder on the packed rows: ids, mask, rows, cu -> final hidden states [n, H] (in the autocast dtype: what the
    reference's lm_head, an nn.Linear inside the autocast'ed forward, would read)."""

    def __init__(self, rag_model, autocast_dtype):
        super().__init__()
        self.generator = rag_model.generator_model
        self.autocast_dtype = autocast_dtype

    def forward(self, input_ids, attention_mask, rows, cu):
        from .. import packed

        if self.autocast_dtype is None:
            return packed.generator_hidden(self.generator, input_ids, attention_mask, rows, cu)
        with torch.autocast("cuda", dtype=self.autocast_dtype, cache_enabled=False):
            h = packed.generator_hidden(self.generator, input_ids, attention_mask, rows, cu)
        return h if h.dtype == self.autocast_dtype else h.to(self.autocast_dtype)


class GraphedTowers:
    """Forward AND backward of the three tower calls (passage, query, generator) as hipGraphs
    (torch.cuda.make_graphed_callables), everything that talks to other GPUs - the embedding all-gathers, the
    loss with its stats exchange, the gradient all-reduce - and the optimizer stay eager.  That removes >99 %
    of the per-step launches from the host path while no collective is ever captured; it is what W > 1 uses."""

    def __init__(self, rag_model, autocast_dtype, sample_batch: Dict[str, torch.Tensor], hidden_only: bool = False):
        if getattr(rag_model, "retriever_is_autoregressive", False):
            raise NotImplementedError("graphed towers: autoregressive retrievers run eagerly")
        b = sample_batch
        warm_blas_handles()
        self.key = self.key_of(b)
        # a batch that carries the packed row lists of all three tower inputs gets PACKED graphs (round 6): the same captures
        # around dalm_amd/packed.py's calls; the row counts are part of the key (one set of graphs per combination)
        self.packed = self.is_packed(b)
        if self.packed:
            calls = (_PackedRetrievalCall(rag_model, autocast_dtype), _PackedRetrievalCall(rag_model, autocast_dtype),
                     _PackedGeneratorCall(rag_model, autocast_dtype))
        else:
            calls = (_RetrievalCall(rag_model, autocast_dtype), _RetrievalCall(rag_model, autocast_dtype),
                     _GeneratorCall(rag_model, autocast_dtype, hidden_only))
        for c in calls:
            c.train(rag_model.training)

        def arg(prefix, ids, mask):
            a = (b[ids].clone(), b[mask].clone())
            return a + (b[f"{prefix}_pack_rows"].clone(), b[f"{prefix}_pack_cu"].clone()) if self.packed else a

        args = (arg("retriever_passage", "retriever_passage_input_ids", "retriever_passage_attention_mask"),
                arg("retriever_query", "retriever_query_input_ids", "retriever_query_attention_mask"),
                arg("generator", "generator_input_input_ids", "generator_input_attention_mask"))
        # Callables captured in ONE make_graphed_callables call share a memory pool and must replay in capture
        # order on one stream.  The retrieval pair runs on the tower stream concurrently with the generator on
        # the main stream, so the generator gets its own capture (own pool); within the pair the order is
        # passage -> query forward and (autograd: later nodes first) query -> passage backward, as required.
        with thread_local_capture():
            self.passage, self.query = torch.cuda.make_graphed_callables(
                calls[:2], args[:2], num_warmup_iters=3, allow_unused_input=True)
            self.generator = torch.cuda.make_graphed_callables(
                calls[2], args[2], num_warmup_iters=3, allow_unused_input=True)

    KEYS = ("retriever_passage_input_ids", "retriever_query_input_ids", "generator_input_input_ids")
    PACK_KEYS = ("retriever_passage_pack_rows", "retriever_query_pack_rows", "generator_pack_rows",
                 "retriever_passage_pack_cu", "retriever_query_pack_cu", "generator_pack_cu")

    @classmethod
    def is_packed(cls, batch) -> bool:
        return all(k in batch for k in cls.PACK_KEYS)

    @classmethod
    def key_of(cls, batch):
        keys = cls.KEYS + (cls.PACK_KEYS if cls.is_packed(batch) else ())
        return tuple(tuple(batch[k].shape) for k in keys)

    def matches(self, batch) -> bool:
        return self.key_of(batch) == self.key

    def call_args(self, batch, prefix: str, ids: str, mask: str):
        a = (batch[ids], batch[mask])
        return a + (batch[f"{prefix}_pack_rows"], batch[f"{prefix}_pack_cu"]) if self.packed else a
