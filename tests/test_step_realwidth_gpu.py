"""a12 at the REAL widths of BASELINE.json's configs: towers -> pool -> similarity/contrastive ->
marginalised CE -> backward at D = 1024 / H = 4096 (4544) / V = 32000 (65024) / B = 18 (150), compared with the
reference's step (train_rage2e.py:429-474, train_retriever_only.py:365-379) on LOSSES AND GRADIENT NORMS.

Two independent checks per configuration, both on depth-1 towers built from a CPU seed (oracle/realwidth.py):
  1. full fine-tune vs tests/golden/realwidth_golden.json - numbers produced by the reference's OWN classes and loss
     code in the build container (oracle/make_golden.py::main_realwidth_golden), fp32 and bf16-autocast;
  2. LoRA (the bench's configuration; the image has no peft, so the in-tree injector is used on both sides) vs the
     oracle's `ref_*` restatement run on this host's CPU in fp32.
Tolerances asserted here (north_star: 1e-3 relative in fp32, stated tolerance in bf16); measured on MI355X in round 3
(profiles/history/r03_realwidth_parity.json): fp32 <= 1e-6 everywhere, bf16 loss <= 4e-6 / gradient norm <= 6e-5 at real width:
  fp32            loss / contrastive / generator / grad-norm  <= 1e-4
  bf16 autocast   loss / contrastive / generator              <= 1e-4   (vs the reference under CPU bf16 autocast)
                  grad-norm (global and per tower)            <= 5e-4   (bf16 has 8 mantissa bits; CPU and GPU autocast
                                                                         round at different operators)
Round 4: the bf16 bounds were 1e-3 / 5e-3, 15-250x what was measured - a regression of two orders of
magnitude would have passed; they are now ~25x / ~8x the measured deviations.  A host whose CPU RNG does not reproduce the
seeded weights FAILS these tests (they carry row a12) unless DALM_ALLOW_RNG_SKIP=1 is set.
"""
import copy
import json
import os
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"
OUT = Path(__file__).resolve().parent.parent / "gpurun_out"

TOL = {"fp32": {"loss": 1e-4, "grad": 1e-4}, "bf16_autocast": {"loss": 1e-4, "grad": 5e-4}}


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-30)


def _record(name, payload):
    """Measured deviations are kept (gpurun_out/ is merged back) so DESIGN.md can quote them."""
    try:
        OUT.mkdir(exist_ok=True)
        path = OUT / "realwidth_parity.json"
        cur = json.loads(path.read_text()) if path.exists() else {}
        cur[name] = payload
        path.write_text(json.dumps(cur, indent=1))
    except OSError:
        pass


class _TowerNorms:
    """optimizer pre-step hook: per-tower gradient norms while the gradients still exist."""

    def __init__(self, groups):
        self.groups, self.norms = groups, {}

    def __call__(self, *_):
        import realwidth as RW

        self.norms = {k: RW.grad_norm(ps) for k, ps in self.groups.items()}


_BUILT = {}


def _build(case):
    """Seeded towers, built once per session (465 M - 720 M parameters of CPU randn each) and handed out as copies."""
    import realwidth as RW

    if case not in _BUILT:      # ~5 GB of host memory for the three configurations together
        _BUILT[case] = RW.build_case(case)
    r, g = _BUILT[case]
    return copy.deepcopy(r), copy.deepcopy(g)


def _seeded_case(case, gold):
    import realwidth as RW

    retriever, generator = _build(case)
    cs = RW.checksum(retriever)
    if _rel(cs, gold["checksum_retriever"]) > 1e-9:
        msg = f"this host's torch CPU RNG does not reproduce the golden's seeded weights ({cs} vs {gold['checksum_retriever']})"
        if os.environ.get("DALM_ALLOW_RNG_SKIP") == "1":
            pytest.skip(msg)
        pytest.fail(msg + "; the real-width parity tests cannot run here (set DALM_ALLOW_RNG_SKIP=1 to skip them knowingly)")
    if generator is not None:
        assert _rel(RW.checksum(generator), gold["checksum_generator"]) <= 1e-9
    return retriever, generator


@pytest.mark.parametrize("precision", ["fp32", "bf16_autocast"])
@pytest.mark.parametrize("case", ["cfg3", "cfg5", "cfg3_d2"])
def test_full_finetune_step_matches_the_reference_at_real_width(case, precision):
    """cfg3_d2 (round 4): two layers per tower - the bf16-autocast comparison then crosses two attention + MLP blocks per
    tower, where CPU and GPU autocast round at different operators (the depth-1 cases cannot show tower-side drift)."""
    import realwidth as RW

    from dalm_amd.models import AutoModelForRagE2E
    from dalm_amd.training.step import RagE2EStep

    gold = json.loads((G / "realwidth_golden.json").read_text())[case]
    retriever, generator = _seeded_case(case, gold)
    dev = torch.device("cuda:0")
    model = AutoModelForRagE2E.from_modules(retriever, generator, None, None, normalize=True, get_peft=None).to(dev)
    model.train()
    params = list(model.parameters())
    opt = torch.optim.SGD(params, lr=0.0)
    towers = _TowerNorms({"retriever": list(model.retriever_model.parameters()),
                          "generator": list(model.generator_model.parameters())})
    opt.register_step_pre_hook(towers)
    step = RagE2EStep(model, opt, None, 100, autocast_dtype=torch.bfloat16 if precision == "bf16_autocast" else None,
                      inplace_grad=True, overlap_towers=True, track_grad_norm=True)
    batch = {k: v.to(dev) for k, v in RW.synthetic_batch(case).items()}
    loss = float(step(batch))
    got = {"loss": loss, "contrastive": float(step.aux["contrastive"]), "generator": float(step.aux["generator"]),
           "grad_norm": float(step.grad_norm), "grad_norm_retriever": towers.norms["retriever"],
           "grad_norm_generator": towers.norms["generator"]}
    ref = gold[precision]
    rel = {k: _rel(got[k], ref[k]) for k in ref}
    _record(f"{case}/full_ft/{precision}", {"got": got, "reference": ref, "rel": rel})
    tol = TOL[precision]
    if case.endswith("_d2") and precision == "bf16_autocast":
        # two blocks per tower: the CPU and the GPU autocast round at different operators in EVERY block, and the reference's
        # own bf16 result already sits 2.8e-3 (generator gradient norm) from its fp32 one at this depth - stated bound 2e-4 on the
        # losses (measured 4.5e-5 ... 6.4e-5), 5e-4 on the gradient norms (measured <= 3.7e-5); profiles/history/r04_realwidth_parity.json
        tol = {"loss": 2e-4, "grad": 5e-4}
    for k, r in rel.items():
        assert r <= (tol["grad"] if k.startswith("grad_norm") else tol["loss"]), (k, rel, got, ref)


@pytest.mark.parametrize("precision", ["fp32", "bf16_autocast"])
def test_full_finetune_retriever_step_matches_the_reference_at_real_width(precision):
    """BASELINE configs[1]: retriever-only, bge-large width, batch 150."""
    import realwidth as RW

    from dalm_amd.models import AutoModelForSentenceEmbedding
    from dalm_amd.training.step import RetrieverStep

    gold = json.loads((G / "realwidth_golden.json").read_text())["cfg2"]
    bert, _ = _seeded_case("cfg2", gold)
    dev = torch.device("cuda:0")
    model = AutoModelForSentenceEmbedding.from_modules(bert, None, normalize=True, get_peft=False).to(dev)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    step = RetrieverStep(model, opt, None, 100, autocast_dtype=torch.bfloat16 if precision == "bf16_autocast" else None,
                         overlap_towers=True, track_grad_norm=True)
    batch = {k: v.to(dev) for k, v in RW.synthetic_batch("cfg2").items()}
    got = {"loss": float(step(batch)), "grad_norm": float(step.grad_norm)}
    ref = gold[precision]
    rel = {k: _rel(got[k], ref[k]) for k in ref}
    _record(f"cfg2/full_ft/{precision}", {"got": got, "reference": ref, "rel": rel})
    tol = TOL[precision]
    assert rel["loss"] <= tol["loss"] and rel["grad_norm"] <= tol["grad"], (rel, got, ref)


def _randomise_lora_b(module, seed):
    """peft initialises lora_B to zero, which makes every lora_A gradient exactly zero: give B small seeded values so
    the check covers both factors."""
    from dalm_amd.models.lora import LoRALinear

    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, LoRALinear):
            w = m.lora_B["default"].weight
            with torch.no_grad():
                w.copy_(0.02 * torch.randn(w.shape, generator=g))


@pytest.mark.parametrize("case", ["cfg3", "cfg5", "cfg2"])
def test_lora_step_matches_the_oracle_on_this_host_at_real_width(case):
    """The configuration bench.py times (LoRA r=8 on both towers, frozen base) against the reference's op sequence
    (`oracle.ref_*` around the plain HF towers) executed on this host's CPU in fp32: loss, its two parts and the
    global gradient norm of the trainable parameters within 1e-3."""
    import dalm_oracle as O
    import realwidth as RW

    from dalm_amd.models import AutoModelForRagE2E, AutoModelForSentenceEmbedding, lora
    from dalm_amd.training.step import RagE2EStep, RetrieverStep

    retriever, generator = _build(case)
    lora.inject_lora(retriever, ["key", "query", "value"], lora_dropout=0.0)
    _randomise_lora_b(retriever, 11)
    if generator is not None:
        lora.inject_lora(generator, ["q_proj", "v_proj"], lora_dropout=0.0)   # Falcon: resolved to query_key_value
        _randomise_lora_b(generator, 12)
    batch = RW.synthetic_batch(case)
    dev = torch.device("cuda:0")
    # product side first (deep copies: the wrapper swaps in HIP rms_norm / rope modules that refuse CPU tensors)
    r_gpu, g_gpu = copy.deepcopy(retriever), copy.deepcopy(generator)
    if generator is not None:
        model = AutoModelForRagE2E.from_modules(r_gpu, g_gpu, None, None, normalize=True, get_peft=None).to(dev)
    else:
        model = AutoModelForSentenceEmbedding.from_modules(r_gpu, None, normalize=True, get_peft=False).to(dev)
    model.train()
    trainable = [p for p in model.parameters() if p.requires_grad]
    assert trainable and all("lora_" in n for n, p in model.named_parameters() if p.requires_grad)
    opt = torch.optim.SGD(trainable, lr=0.0)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    if generator is not None:
        step = RagE2EStep(model, opt, None, 100, autocast_dtype=None, inplace_grad=True, overlap_towers=True,
                          track_grad_norm=True)
        got = {"loss": float(step(dbatch)), "contrastive": float(step.aux["contrastive"]),
               "generator": float(step.aux["generator"]), "grad_norm": float(step.grad_norm)}
    else:
        step = RetrieverStep(model, opt, None, 100, autocast_dtype=None, overlap_towers=True, track_grad_norm=True)
        got = {"loss": float(step(dbatch)), "grad_norm": float(step.grad_norm)}
    del model, step, opt, r_gpu, g_gpu
    torch.cuda.empty_cache()

    # reference op sequence on the host (bounded threads: eager torch on a 256-core host is fastest at ~16)
    old_threads = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    try:
        qk, pk = (("retriever_query", "retriever_passage") if generator is not None else ("query", "passage"))
        q = O.ref_retrieval_embed(retriever(batch[f"{qk}_input_ids"], batch[f"{qk}_attention_mask"])[0], batch[f"{qk}_attention_mask"])
        p = O.ref_retrieval_embed(retriever(batch[f"{pk}_input_ids"], batch[f"{pk}_attention_mask"])[0], batch[f"{pk}_attention_mask"])
        logits = None
        if generator is not None:
            logits = generator(input_ids=batch["generator_input_input_ids"],
                               attention_mask=batch["generator_input_attention_mask"]).logits
        out = O.ref_step_loss(q, p, logits, batch.get("generator_input_input_ids"), batch.get("generator_input_attention_mask"),
                              batch.get("query_passage_input_len"), 100)
        out["loss"].backward()
    finally:
        torch.set_num_threads(old_threads)
    cpu_params = [p for m in (retriever, generator) if m is not None for p in m.parameters() if p.requires_grad]
    ref = {"loss": float(out["loss"]), "grad_norm": RW.grad_norm(cpu_params)}
    if generator is not None:
        ref.update(contrastive=float(out["contrastive"]), generator=float(out["generator"]))
    rel = {k: _rel(got[k], ref[k]) for k in ref}
    _record(f"{case}/lora/fp32_vs_oracle_on_host", {"got": got, "reference": ref, "rel": rel})
    assert max(rel.values()) <= 1e-4, (rel, got, ref)


# ---------------------------------------------------------------------------------------------------------------
# bf16: the 5-step trajectories of the tiny golden models against the REFERENCE RUN UNDER bf16 AUTOCAST
# (fp32 master weights, forward under autocast, fp32 loss code on the up-cast outputs - accelerate's bf16 mode)
# ---------------------------------------------------------------------------------------------------------------
def test_bf16_autocast_trajectory_matches_the_reference_under_bf16_autocast():
    """Stated bf16 tolerance: per-step loss <= 2e-4 relative and per-step gradient norm <= 5e-4 relative against the
    reference's own bf16-autocast trajectory (step_golden.json["bf16_autocast"]) over 5 Adam steps (measured: 2e-5 /
    8e-5; round 3 asserted 2e-3 / 5e-3); the bound against the reference's fp32 trajectory lives in test_step_parity_gpu.py."""
    from transformers import get_scheduler

    from dalm_amd.models import AutoModelForRagE2E
    from dalm_amd.training.step import RagE2EStep
    from test_step_parity_gpu import _batches

    gold = json.loads((G / "step_golden.json").read_text())
    dev = torch.device("cuda:0")
    rag = AutoModelForRagE2E(str(G / "tiny_retriever"), str(G / "tiny_generator")).to(dev)
    g_tok = rag.generator_tokenizer
    g_tok.pad_token = g_tok.eos_token
    rag.train()
    opt = torch.optim.Adam(rag.parameters(), lr=gold["lr"])
    sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=gold["warmup"], num_training_steps=gold["total_steps"])
    step = RagE2EStep(rag, opt, sched, 100, autocast_dtype=torch.bfloat16, inplace_grad=True, overlap_towers=True,
                      track_grad_norm=True)
    losses, gnorms = [], []
    for b in _batches(rag.retriever_tokenizer, g_tok, gold, dev):
        losses.append(float(step(b)))
        gnorms.append(float(step.grad_norm))
    ref = gold["bf16_autocast"]
    rel_l = [_rel(a, b) for a, b in zip(losses, ref["losses"])]
    rel_g = [_rel(a, b) for a, b in zip(gnorms, ref["grad_norms"])]
    rel_fp32 = [_rel(a, b) for a, b in zip(losses, gold["losses"])]
    _record("tiny/e2e/bf16_autocast_trajectory", {"losses": losses, "grad_norms": gnorms, "rel_loss": rel_l,
                                                  "rel_grad_norm": rel_g, "rel_loss_vs_fp32_reference": rel_fp32})
    assert max(rel_l) <= 2e-4, (rel_l, losses, ref["losses"])
    assert max(rel_g) <= 5e-4, (rel_g, gnorms, ref["grad_norms"])


@pytest.mark.parametrize("graph", [False, True])
def test_fp32_trajectories_match_the_reference_gradient_norms(graph):
    """Per-step global gradient norm of the 5 golden steps (e2e, all parameters trainable) within 1e-3 of the reference's,
    eager and as a replayed hipGraph (the norm is computed inside the captured step)."""
    from transformers import get_scheduler

    from dalm_amd.models import AutoModelForRagE2E
    from dalm_amd.training.graphed import GraphedStep, TensorLRScheduler, make_capturable_adam
    from dalm_amd.training.step import RagE2EStep
    from test_step_parity_gpu import _batches

    gold = json.loads((G / "step_golden.json").read_text())
    dev = torch.device("cuda:0")
    rag = AutoModelForRagE2E(str(G / "tiny_retriever"), str(G / "tiny_generator")).to(dev)
    g_tok = rag.generator_tokenizer
    g_tok.pad_token = g_tok.eos_token
    rag.train()
    opt = make_capturable_adam(rag.parameters(), gold["lr"], dev) if graph else torch.optim.Adam(rag.parameters(), lr=gold["lr"])

    def mk(o):
        return get_scheduler("linear", optimizer=o, num_warmup_steps=gold["warmup"], num_training_steps=gold["total_steps"])

    sched = TensorLRScheduler(opt, gold["lr"], mk) if graph else mk(opt)
    inner = RagE2EStep(rag, opt, sched, 100, autocast_dtype=None, inplace_grad=True, overlap_towers=graph,
                       track_grad_norm=True)
    step = GraphedStep(inner, warmup=0) if graph else inner
    losses, gnorms = [], []
    for b in _batches(rag.retriever_tokenizer, g_tok, gold, dev):
        losses.append(float(step(b)))
        gnorms.append(float(inner.grad_norm))
    for got, ref in zip(losses, gold["losses"]):
        assert _rel(got, ref) <= 1e-3, (losses, gold["losses"])
    for got, ref in zip(gnorms, gold["grad_norms"]):
        assert _rel(got, ref) <= 1e-3, (gnorms, gold["grad_norms"])


def test_retriever_only_trajectory_matches_the_reference_gradient_norms_fp32_and_bf16():
    from transformers import PreTrainedTokenizerFast, get_scheduler

    from dalm_amd.models import AutoModelForSentenceEmbedding
    from dalm_amd.training.step import RetrieverStep
    from dalm_amd.training.utils.retriever_only_dataloader_utils import preprocess_dataset
    from test_step_parity_gpu import _tiny_bge_small

    gold = json.loads((G / "retriever_step_golden.json").read_text())
    tok = PreTrainedTokenizerFast.from_pretrained(str(G / "wordlevel_tokenizer"))
    dev = torch.device("cuda:0")
    enc = preprocess_dataset(gold["rows"], tok, query_column_name="Question", passage_column_name="Abstract",
                             query_max_len=gold["query_max_len"], passage_max_len=gold["passage_max_len"])
    full = {k: torch.tensor(v, device=dev) for k, v in enc.items()}
    for precision, ref, tol_l, tol_g in (("fp32", gold, 1e-3, 1e-3), ("bf16", gold["bf16_autocast"], 2e-3, 5e-3)):
        bert = _tiny_bge_small(len(tok), gold["seed"])
        model = AutoModelForSentenceEmbedding.from_modules(bert, tok, normalize=True, get_peft=False).to(dev)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=gold["lr"])
        sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=gold["warmup"], num_training_steps=gold["total_steps"])
        step = RetrieverStep(model, opt, sched, 100, autocast_dtype=torch.bfloat16 if precision == "bf16" else None,
                             overlap_towers=False, track_grad_norm=True)
        losses, gnorms = [], []
        for a, b in gold["batch_rows"]:
            losses.append(float(step({k: v[a:b] for k, v in full.items()})))
            gnorms.append(float(step.grad_norm))
        rel_l = [_rel(a, b) for a, b in zip(losses, ref["losses"])]
        rel_g = [_rel(a, b) for a, b in zip(gnorms, ref["grad_norms"])]
        _record(f"tiny/retriever_only/{precision}_trajectory", {"rel_loss": rel_l, "rel_grad_norm": rel_g})
        assert max(rel_l) <= tol_l, (precision, rel_l)
        assert max(rel_g) <= tol_g, (precision, rel_g)
