#!/usr/bin/env python
"""bench.py - training pairs/sec of the RAG-end2end step on N MI355X (one process per GPU).

Workload (BASELINE.json configs[2], the config the headline metric is quoted on; fits one GPU):
    retriever  bge-large-en architecture (BERT 24L/1024/16H/4096, vocab 30522), LoRA r=8 on q/k/v
    generator  Llama-2-7b-hf architecture (32L/4096/32H/11008, vocab 32000), LoRA r=8 on q_proj/v_proj
    per-GPU batch 18, Tq=50, Tp=128, Tg=256, logit_scale 100, Adam lr 1e-4, bf16 autocast
    random-init weights of those architectures + synthetic (Passage, Query, Answer) token rows
    (no network for checkpoints/datasets).
A "step" = passage tower + query tower + generator forward, the fused HIP loss path
(pool/normalise, f32-MFMA similarity + contrastive, marginalised CE with the logits gradient written
in the same pass), backward, gradient all-reduce (N>1), Adam, scheduler, zero_grad - nothing skipped.
With N>1 every rank keeps batch 18 (weak scaling) and the in-batch negatives span the global batch
(RCCL all-gather of the embeddings over xGMI, overlapped with the query tower on a side stream).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant hand-written kernel
(marg_ce_row_kernel: 2*R*V*el algorithmic bytes per launch, R = B*(Tg-1) dense rows), timed live with
HIP events on the launch stream.  `cpu_baseline` times the reference-equivalent CPU path (oracle
restatement + the same HF architectures) on the host cores, bounded by depth-scaling (see
cpu_reference_baseline).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import dalm_amd  # noqa: E402,F401
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
A100_README_PAIRS_PER_S = 200000.0 / (7 * 3600.0)  # reference README.md:34-40 (1x A100 80GB), BASELINE.md section 1

CFG = dict(B=18, Tq=50, Tp=128, Tg=256, D=1024, V=32000, logit_scale=100)


GENERATORS = {
    # name -> (config factory, vocabulary).  Defaults of the HF configs == the published 7B architectures.
    "llama-2-7b": (lambda layers: __import__("transformers").LlamaConfig(num_hidden_layers=layers), 32000),
    "falcon-7b": (lambda layers: __import__("transformers").FalconConfig(num_hidden_layers=layers, hidden_dropout=0.0,
                                                                          attention_dropout=0.0), 65024),
}


def build_models(device, dtype, bert_layers=24, llama_layers=32, lora=True, generator="llama-2-7b"):
    from transformers import AutoModelForCausalLM, BertConfig, BertModel

    from dalm_amd.models import AutoModelForRagE2E, Mode

    bc = BertConfig(hidden_size=1024, num_hidden_layers=bert_layers, num_attention_heads=16, intermediate_size=4096,
                    vocab_size=30522, max_position_embeddings=512)
    gc = GENERATORS[generator][0](llama_layers)
    torch.manual_seed(0)  # identical weights on every rank
    with torch.device(device):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            retriever = BertModel(bc)
            gen = AutoModelForCausalLM.from_config(gc)
        finally:
            torch.set_default_dtype(old)
    return AutoModelForRagE2E.from_modules(retriever, gen, None, None, normalize=True,
                                           get_peft=Mode.BOTH if lora else None)


def synthetic_batch(device, seed, B=CFG["B"], Tq=CFG["Tq"], Tp=CFG["Tp"], Tg=CFG["Tg"], V=CFG["V"]):
    """Token-id level synthetic (Passage, Query, Answer) rows, SURVEY.md section 8(d)."""
    g = torch.Generator().manual_seed(seed)

    def ids(n, T, V):
        return torch.randint(1000, V, (n, T), generator=g)

    def right_mask(T, lo, hi):
        lens = torch.randint(lo, hi + 1, (B, 1), generator=g)
        return (torch.arange(T).unsqueeze(0) < lens).long()

    glen = torch.randint(60, Tg + 1, (B, 1), generator=g)
    gmask = (torch.arange(Tg).unsqueeze(0) >= (Tg - glen)).long()  # Llama tokenizers pad left
    batch = {
        "retriever_query_input_ids": ids(B, Tq, 30522), "retriever_query_attention_mask": right_mask(Tq, 5, 15),
        "retriever_passage_input_ids": ids(B, Tp, 30522), "retriever_passage_attention_mask": right_mask(Tp, 30, Tp),
        "generator_input_input_ids": ids(B, Tg, V), "generator_input_attention_mask": gmask,
        "query_passage_input_len": (glen.squeeze(1).float() * 0.8).long().clamp(min=1),
    }
    return {k: v.to(device) for k, v in batch.items()}


def bucketed_batches(device, seed, V, n_batches=12):
    """The trainer's opt-in host data path on synthetic rows: a pool of n_batches * B rows is ordered by generator length
    (`shards.bucketed_order`), cut into batches of B and every batch loses its all-padding columns (`shards.trim_batch`)."""
    from dalm_amd.training.shards import bucketed_order, trim_batch

    B = CFG["B"]
    pool = [synthetic_batch(torch.device("cpu"), seed + i, V=V) for i in range(n_batches)]
    rows = {k: torch.cat([b[k] for b in pool]) for k in pool[0]}
    lengths = (rows["generator_input_attention_mask"] != 0).sum(dim=1)
    order = bucketed_order(lengths, B, torch.Generator().manual_seed(seed))
    groups = [("retriever_query_input_ids", "retriever_query_attention_mask"),
              ("retriever_passage_input_ids", "retriever_passage_attention_mask"),
              ("generator_input_input_ids", "generator_input_attention_mask")]
    out = []
    for i in range(n_batches):
        idx = order[i * B:(i + 1) * B]
        host = trim_batch({k: v.index_select(0, idx) for k, v in rows.items()}, groups,
                          qlen_key="query_passage_input_len", qlen_follows="generator_input_attention_mask")
        out.append({k: v.to(device) for k, v in host.items()})
    return out


class TimedOps:
    """HipOps with HIP events around the dominant kernel launch (marginalised CE, fused fwd+grad)."""

    def __init__(self):
        from dalm_amd.ops import HipOps

        self._ops = HipOps()
        self.events = []
        self.enabled = False

    def __getattr__(self, name):
        return getattr(self._ops, name)

    def ce_fwd(self, logits, ids, mask, stats, want_grad, inplace=False):
        if not self.enabled:
            return self._ops.ce_fwd(logits, ids, mask, stats, want_grad, inplace)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # recorded inside HipOps.ce_fwd on torch's current stream (== the stream the kernel is launched on), right
        # around the launch: output allocation and argument marshalling are not inside the interval
        out = self._ops.ce_fwd(logits, ids, mask, stats, want_grad, inplace, events=(a, b))
        # bytes this launch really has to move: live rows are read once; with want_grad every row of the
        # [B,Tg,V] gradient is written (zeros for masked rows and the last position)
        B, Tg, V = logits.shape
        live = (mask[:, 1:] != 0).sum()  # stays on the device: no host sync inside the timed region
        self.events.append((a, b, live, (B * Tg if want_grad else 0), V * logits.element_size()))
        return out


def cpu_reference_baseline(max_seconds=90.0):
    """Reference CPU path on the host cores, bounded (~25 s): the oracle restatement of the reference's
    loss code at the full cfg3 shapes around the same HF architectures (LoRA, fp32) at depth 1 and
    depth 2 instead of 24 / 32 layers, one step each; the per-layer increment is extrapolated to full
    depth.  Threads: min(host cores, 16) - measured on the 256-core GPU-box host, 16 threads is the
    fastest setting for this eager-torch workload (16: 4.9 s, 32: 5.5 s, 64: 6.8 s, 256: 80 s per
    depth-1 step), so this is the CPU path at its best, and `cores` reports the threads used."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import dalm_oracle as O

    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    dev = torch.device("cpu")
    times = {}
    t_start = time.time()
    for depth in (1, 2):
        model = build_models(dev, torch.float32, bert_layers=depth, llama_layers=depth)
        model.train()
        params = [p for p in model.parameters() if p.requires_grad]
        opt = torch.optim.Adam(params, lr=1e-4)
        batch = synthetic_batch(dev, 0)

        def step():
            qh = model.retriever_model(batch["retriever_query_input_ids"], batch["retriever_query_attention_mask"])[0]
            ph = model.retriever_model(batch["retriever_passage_input_ids"], batch["retriever_passage_attention_mask"])[0]
            q = O.ref_retrieval_embed(qh, batch["retriever_query_attention_mask"])
            p = O.ref_retrieval_embed(ph, batch["retriever_passage_attention_mask"])
            logits = model.generator_model(input_ids=batch["generator_input_input_ids"],
                                           attention_mask=batch["generator_input_attention_mask"]).logits
            out = O.ref_step_loss(q, p, logits, batch["generator_input_input_ids"],
                                  batch["generator_input_attention_mask"], batch["query_passage_input_len"],
                                  CFG["logit_scale"])
            out["loss"].backward()
            opt.step()
            model.zero_grad()

        if depth == 1:
            step()  # warm-up (thread pools, allocator)
        t0 = time.time()
        step()
        times[depth] = time.time() - t0
        del model, opt
        if time.time() - t_start > max_seconds:
            break
    if len(times) == 2:
        delta = max(times[2] - times[1], 0.0)   # one more BERT layer (both towers) + one more Llama layer
        fixed = max(times[1] - delta, 0.0)      # embeddings, lm_head, loss path, Adam
        # split the increment by layer FLOPs (Llama layer 202 M params x 4608 tokens vs BERT layer
        # 12.6 M x 3204 tokens -> 95.8 % / 4.2 %) and scale each share to its true depth (32 / 24)
        full = fixed + delta * (0.958 * 32 + 0.042 * 24)
        note = (f"reference-equivalent CPU step (oracle loss code at full cfg3 shapes + HF towers, LoRA, fp32, "
                f"{threads} threads): depth 1 = {times[1]:.2f} s, depth 2 = {times[2]:.2f} s per step; per-layer "
                f"increment extrapolated to 24 BERT / 32 Llama layers -> {full:.1f} s per 18-pair step")
    else:
        full = times[1] * 30.0
        note = f"depth-1 towers only ({times[1]:.2f} s/step, {threads} threads) x30 (time bound hit before depth 2)"
    return {"value": CFG["B"] / full, "unit": "training pairs/s", "cores": threads, "kind": "port", "sample": note}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-overlap", action="store_true", help="run the retriever towers on the main stream")
    ap.add_argument("--graph-collectives", action="store_true",
                    help="EXPERIMENTAL: capture the whole step including the RCCL collectives (W > 1)")
    ap.add_argument("--graph-towers", action="store_true",
                    help="graph the tower fwd/bwd and keep collectives eager (the default when --gpus > 1)")
    ap.add_argument("--fuse-lm-head", action="store_true",
                    help="SURVEY 8(f) rank 1: chunked lm_head + CE, the [B,Tg,V] logits are never materialised")
    ap.add_argument("--data-path", default="fixed", choices=["fixed", "bucketed"],
                    help="fixed (default, the named configuration): every batch padded to Tq50/Tp128/Tg256; bucketed: the "
                         "trainer's opt-in --length_bucketing + --trim_padding applied to a pool of synthetic rows "
                         "(an extra line next to the headline, never the headline)")
    ap.add_argument("--all-rows", action="store_true",
                    help="with --fuse-lm-head: run the padding rows through the lm_head GEMMs too (sample chunks)")
    ap.add_argument("--workload", default="cfg3", choices=["cfg3", "cfg5", "cfg2", "cfg1"],
                    help="cfg3 = RAG-e2e bge-large + Llama-2-7b batch 18 (headline, default); "
                         "cfg5 = RAG-e2e bge-large + Falcon-7B architecture (V = 65024: the 1024-thread CE rows); "
                         "cfg2 = retriever-only bge-large batch 150 (BASELINE.json configs[1]); "
                         "cfg1 = retriever-only bge-small batch 19 (the toy-CSV shape of configs[0])")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"],
                    help="bf16 = bf16 weights + autocast (default); fp32 = fp32 weights, no autocast "
                         "(the reference's default precision; quoted beside the bf16 line in DESIGN.md)")
    ap.add_argument("--retriever-layers", type=int, default=24, help=argparse.SUPPRESS)
    ap.add_argument("--generator-layers", type=int, default=32, help=argparse.SUPPRESS)
    args = ap.parse_args()

    from dalm_amd.launch import in_distributed_env, spawn_ranks

    if args.gpus > 1 and not in_distributed_env():
        # `python bench.py --gpus N`: no torchrun needed - spawn one rank per GPU ourselves (RANK / LOCAL_RANK /
        # WORLD_SIZE / MASTER_ADDR=127.0.0.1), rank 0 prints the JSON line; fewer than N visible GPUs -> exit 2
        raise SystemExit(spawn_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus))
    args.hw_queues = dalm_amd.configure_hw_queues(args.gpus)   # before the HIP runtime starts

    from dalm_amd import hip
    from dalm_amd.sharded import barrier, init_distributed
    from dalm_amd.training.step import RagE2EStep

    hip.load()  # no HIP extension -> fail here, loudly
    from dalm_amd.tuning import enable_tuned_gemms

    args.tuned_gemms = enable_tuned_gemms()  # replay-only hipBLASLt/rocBLAS solution table for the tower GEMMs
    if args.workload in ("cfg2", "cfg1"):
        return main_retriever_only(args)
    comm, dev = init_distributed()
    if dev.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if comm.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={comm.world_size}")
    rank = comm.rank
    gen_name = "falcon-7b" if args.workload == "cfg5" else "llama-2-7b"
    V = GENERATORS[gen_name][1]
    wdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    autocast = torch.bfloat16 if args.dtype == "bf16" else None

    model = build_models(dev, wdtype, args.retriever_layers, args.generator_layers, generator=gen_name)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    from transformers import get_scheduler

    from dalm_amd.fused import LocalComm
    from dalm_amd.training.graphed import GraphedStep, TensorLRScheduler, make_capturable_adam

    use_graph = (isinstance(comm, LocalComm) or args.graph_collectives) and not args.no_graph and not args.graph_towers
    opt = make_capturable_adam(params, 1e-4, dev) if use_graph else torch.optim.Adam(params, lr=1e-4, fused=True)

    def mk_sched(o):
        return get_scheduler("linear", optimizer=o, num_warmup_steps=100, num_training_steps=100000)

    sched = TensorLRScheduler(opt, 1e-4, mk_sched) if use_graph else mk_sched(opt)
    ops = TimedOps()
    step = RagE2EStep(model, opt, sched, CFG["logit_scale"], comm=comm, autocast_dtype=autocast, ops=ops,
                      inplace_grad=True, overlap_towers=not args.no_overlap, fuse_lm_head=args.fuse_lm_head,
                      graph_towers=(args.graph_towers or not isinstance(comm, LocalComm)) and not args.no_graph
                      and not args.graph_collectives,
                      graph_after=0, grad_overlap=not args.graph_collectives)
    if use_graph:
        step = GraphedStep(step, max_graphs=8 if args.data_path == "fixed" else 32)
    # a few distinct pre-staged batches (inputs resident in HBM before the timed region)
    if args.data_path == "fixed":
        batches = [synthetic_batch(dev, 100 + 17 * rank + i, V=V) for i in range(4)]
    else:
        batches = bucketed_batches(dev, 100 + 17 * rank, V)
        args.warmup = max(args.warmup, len(batches))   # one hipGraph per trimmed shape, all captured before the timed region
    if args.fuse_lm_head and not args.all_rows:
        # the data loader's job (ShardedBatches(live_rows=...)): list the rows that carry loss while the mask is host memory
        from dalm_amd.fused import gemm_wave_rows, live_row_index

        for b in batches:
            idx = live_row_index(b["generator_input_attention_mask"], gemm_wave_rows(V))
            if idx is not None:
                b["generator_live_rows"] = idx.to(dev)

    # graphs are captured during the first untimed step; with --warmup 0 one untimed step still runs so that
    # the capture never lands inside the timed region
    for i in range(max(args.warmup, 1)):
        step(batches[i % len(batches)])
    torch.cuda.synchronize()
    barrier(comm)
    graphed = use_graph and getattr(step, "graph", None) is not None
    ops.enabled = not graphed  # HIP events cannot bracket a kernel inside a replayed graph
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(batches[i % len(batches)])
    torch.cuda.synchronize()
    barrier(comm)
    elapsed = time.perf_counter() - t0
    ops.enabled = False
    loss_val = float(loss)
    probe_note = "HIP events around the launch in every timed step"
    if graphed:
        # the timed region replays a hipGraph; time the dominant kernel live with HIP events in a few
        # eager launches of the very same step right after it (not part of `value`)
        ops.enabled = True
        for i in range(min(args.steps, 5)):
            step.step(batches[i % len(batches)])
        torch.cuda.synchronize()
        ops.enabled = False
        probe_note = "HIP events around the launch in 5 eager steps run right after the timed hipGraph-replay region"
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        import torch.distributed as dist

        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    ranks_seen, backend = 1, "none (one process)"
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        ranks_seen, backend = torch.distributed.get_world_size(), torch.distributed.get_backend()
    if rank == 0:
        B, Tg = CFG["B"], CFG["Tg"]
        el = 2 if args.dtype == "bf16" else 4   # logits element size (bf16 lm_head output / fp32)
        ce_ms = [e[0].elapsed_time(e[1]) for e in ops.events]
        ce_avg_s = (sum(ce_ms) / max(len(ce_ms), 1)) * 1e-3
        lives = [int(e[2]) for e in ops.events]
        alg_bytes = sum((lv + e[3]) * e[4] for lv, e in zip(lives, ops.events)) / max(len(ops.events), 1)  # per launch
        live_rows = sum(lives) / max(len(lives), 1)
        dense_bytes = 2 * B * (Tg - 1) * V * el                               # SURVEY 8(d) dense definition
        achieved = alg_bytes / ce_avg_s / 1e9 if ce_avg_s > 0 else 0.0
        # HBM traffic of that kernel comes from rocprofv3 PMC passes of THIS command (FETCH_SIZE and WRITE_SIZE
        # cannot share a pass, and counters perturb timing, so they are never collected inside the timed run):
        # tools/pmc_bench.sh writes profiles/roofline_traffic.json; the value is per launch, FETCH doubled as the
        # gfx950 guide prescribes.  Only quoted for the workload/dtype it was collected on.
        traffic, traffic_source = None, None
        tfile = ROOT / "profiles" / "roofline_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                if (tj.get("workload", "cfg3") == args.workload and tj.get("dtype", "bf16") == args.dtype
                        and not args.fuse_lm_head):   # collected on the materialised-logits launch
                    traffic = tj.get("bench_marg_ce_bytes_per_launch")
                    traffic_source = tj.get("source", "profiles/roofline_traffic.json (separate rocprofv3 --pmc passes)")
            except Exception:
                traffic = None
        value = args.gpus * B * args.steps / elapsed
        out = {
            "metric": "training pairs/sec (global batch) RAG-e2e bge-large+" + ("Llama-2-7b" if gen_name == "llama-2-7b" else "Falcon-7B"),
            "value": value, "unit": "pairs/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": (value / A100_README_PAIRS_PER_S) if (args.gpus == 1 and args.workload == "cfg3"
                                                                   and args.data_path == "fixed") else None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} RAG-e2e: bge-large-en + {gen_name} architectures (random init, V={V}), "
                                   "LoRA r=8 both towers, per-GPU batch 18, Tq50/Tp128/Tg256, logit_scale 100, Adam, "
                                   + ("bf16 weights + autocast" if args.dtype == "bf16" else "fp32 weights, no autocast")
                                   + ("" if args.data_path == "fixed" else
                                      "; DATA PATH: length-bucketed batches with all-padding columns trimmed (the trainer's opt-in "
                                      "--length_bucketing --trim_padding), fewer tokens per pair than the named configuration"),
                       "global_batch": args.gpus * B, "parallelism": f"dp{args.gpus} + sharded in-batch negatives",
                       "ranks_seen_by_process_group": ranks_seen, "collective_backend": backend + (" (= RCCL)" if backend == "nccl" else ""),
                       "spawned_by": os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or
                                     ("dalm_amd.launch (python bench.py --gpus N)" if ranks_seen > 1 else "single process"),
                       "gpu_max_hw_queues": args.hw_queues,
                       "retriever_layers": args.retriever_layers, "generator_layers": args.generator_layers,
                       "lm_head": (("fused with the CE in row chunks over the rows that carry loss (no logits tensor)" if not args.all_rows
                                    else "fused with the CE in sample chunks (no logits tensor)") if args.fuse_lm_head
                                   else f"logits materialised ({args.dtype})"),
                       "tower_gemms": "pre-tuned solution table (dalm_amd/tuning)" if args.tuned_gemms else "library defaults",
                       "launch": ("hipGraph replay of the whole step" if (use_graph and getattr(step, "graph", None) is not None)
                                  else ("hipGraph replay of tower fwd/bwd, eager collectives+loss+optimizer"
                                        if getattr(step, "towers", None) is not None else
                                        "eager" + (f" (capture failed: {getattr(step, 'failed', None) or getattr(step, 'towers_failed', None)})"
                                                   if (getattr(step, "failed", None) or getattr(step, "towers_failed", None)) else ""))),
                       "baseline_ref": "reference README.md:34-40: 200k rows in 7 h on 1x A100-80GB = 7.94 pairs/s",
                       # informational step-level roofline: ~124 TFLOP of tower work per 18-pair step with
                       # LoRA (generator 2*6.74e9*4608 fwd, x2 for activation grads; retriever 2.1 TFLOP fwd x3)
                       "peak_hbm_gb": torch.cuda.max_memory_allocated() / 1e9,
                       "step_model_tflops": 124.0 + 6.3,
                       "step_frac_of_bf16_mfma_peak": (124.0 + 6.3) / (elapsed / args.steps) / 2500.0,
                       "final_loss": loss_val},
            "roofline": {"bound": "hbm", "kernel": f"marg_ce_row kernel (fused fwd+grad, {args.dtype} logits, V={V})",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "avg_launch_us": ce_avg_s * 1e6, "algorithmic_bytes": alg_bytes,
                         "live_rows_per_launch": live_rows, "dense_rows_per_launch": B * (Tg - 1),
                         "dense_definition_GBps": dense_bytes / ce_avg_s / 1e9 if ce_avg_s > 0 else 0.0,
                         "note": "achieved counts only bytes the launch must move (padded rows are skipped on the read "
                                 "side); the all-ones-mask roofline point is in profiles/ (tools/kernel_bench.py)",
                         "timing": probe_note},
        }
        if args.gpus == 1 and not args.no_cpu_baseline and args.workload == "cfg3":
            try:
                out["cpu_baseline"] = cpu_reference_baseline()
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "training pairs/s", "cores": os.cpu_count(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    barrier(comm)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main_retriever_only(args):
    """BASELINE.json configs[1]: retriever-only contrastive step, bge-large-en architecture, batch 150 per GPU,
    Tq=50 / Tp=128, LoRA on q/k/v.  Secondary workload (not the headline line)."""
    from transformers import BertConfig, BertModel, get_scheduler

    from dalm_amd.fused import LocalComm
    from dalm_amd.models import AutoModelForSentenceEmbedding
    from dalm_amd.sharded import barrier, init_distributed
    from dalm_amd.training.graphed import GraphedStep, TensorLRScheduler, make_capturable_adam
    from dalm_amd.training.step import RetrieverStep

    comm, dev = init_distributed()
    small = args.workload == "cfg1"   # bge-small-en: 384 wide, 12 layers, 12 heads; the toy csv has 19 rows (< batch 32)
    B, Tq, Tp = (19 if small else 150), 50, 128
    wdtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    autocast = torch.bfloat16 if args.dtype == "bf16" else None
    layers = min(args.retriever_layers, 12) if small else args.retriever_layers
    torch.manual_seed(0)
    with torch.device(dev):
        old = torch.get_default_dtype()
        torch.set_default_dtype(wdtype)
        try:
            bert = BertModel(BertConfig(hidden_size=384 if small else 1024, num_hidden_layers=layers,
                                        num_attention_heads=12 if small else 16, intermediate_size=1536 if small else 4096,
                                        vocab_size=30522, max_position_embeddings=512))
        finally:
            torch.set_default_dtype(old)
    model = AutoModelForSentenceEmbedding.from_modules(bert, None, normalize=True, get_peft=True)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    use_graph = isinstance(comm, LocalComm) and not args.no_graph
    opt = make_capturable_adam(params, 1e-4, dev) if use_graph else torch.optim.Adam(params, lr=1e-4, fused=True)
    def mk_sched(o):
        return get_scheduler("linear", optimizer=o, num_warmup_steps=0, num_training_steps=100000)

    sched = TensorLRScheduler(opt, 1e-4, mk_sched) if use_graph else mk_sched(opt)
    step = RetrieverStep(model, opt, sched, CFG["logit_scale"], comm=comm, autocast_dtype=autocast,
                         overlap_towers=not args.no_overlap)
    if use_graph:
        step = GraphedStep(step)

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        ql = torch.randint(5, 16, (B, 1), generator=g)
        pl = torch.randint(30, Tp + 1, (B, 1), generator=g)
        return {k: v.to(dev) for k, v in {
            "query_input_ids": torch.randint(1000, 30522, (B, Tq), generator=g),
            "query_attention_mask": (torch.arange(Tq).unsqueeze(0) < ql).long(),
            "passage_input_ids": torch.randint(1000, 30522, (B, Tp), generator=g),
            "passage_attention_mask": (torch.arange(Tp).unsqueeze(0) < pl).long()}.items()}

    batches = [batch(200 + 17 * comm.rank + i) for i in range(4)]
    for i in range(max(args.warmup, 1)):
        step(batches[i % 4])
    torch.cuda.synchronize()
    barrier(comm)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(batches[i % 4])
    torch.cuda.synchronize()
    barrier(comm)
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    if comm.rank == 0:
        value = args.gpus * B * args.steps / float(t.item())
        print(json.dumps({
            "metric": "training pairs/sec (global batch) retriever-only " + ("bge-small" if small else "bge-large"), "value": value, "unit": "pairs/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * float(t.item()) / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload} retriever-only: {'bge-small-en' if small else 'bge-large-en'} architecture (random init), "
                                   f"LoRA r=8 q/k/v, per-GPU batch {B}, Tq50/Tp128, logit_scale 100, Adam, {args.dtype}",
                       "global_batch": args.gpus * B,
                       "parallelism": f"dp{args.gpus} + sharded in-batch negatives", "final_loss": float(loss),
                       "launch": "hipGraph replay" if (use_graph and getattr(step, "graph", None) is not None) else "eager"}}),
              flush=True)
    barrier(comm)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
