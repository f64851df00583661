"""a12 on the GPU: our step (towers on PyTorch-ROCm, loss path on the HIP kernels, Adam + schedule) vs the
REFERENCE's step body run on CPU by oracle/make_golden.py with the same tiny models (tests/golden/tiny_*),
the same tokenised batches (reference preprocess_dataset), dropout 0, fp32.
Tolerance: per-step loss within 1e-3 relative (north-star), observed ~1e-6."""
import json
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
G = Path(__file__).parent / "golden"


def _batches(tok_r, tok_g, gold, dev):
    from dalm_amd.training.utils.rag_e2e_dataloader_utils import preprocess_dataset

    rows = json.loads((G / "host_golden.json").read_text())["rows"]
    enc = preprocess_dataset(rows, tok_r, tok_g, "Question", "Abstract", "Answer", gold["query_max_len"],
                             gold["passage_max_len"], gold["generator_max_len"])
    full = {k: torch.tensor(v, device=dev) for k, v in enc.items()}
    return [{k: v[a:b] for k, v in full.items()} for a, b in gold["batch_rows"]]


@pytest.mark.parametrize("inplace,graph", [(False, False), (True, False), (True, True), (True, "towers")])
def test_rag_e2e_step_trajectory_matches_reference(inplace, graph):
    """graph=True: the step is captured into a hipGraph (5-row batches replay it, the 3- and 4-row batches
    run eagerly), towers overlapped on a side stream - same trajectory as the reference either way."""
    from transformers import get_scheduler

    from dalm_amd.models import AutoModelForRagE2E
    from dalm_amd.training.graphed import GraphedStep, TensorLRScheduler, make_capturable_adam
    from dalm_amd.training.step import RagE2EStep

    gold = json.loads((G / "step_golden.json").read_text())
    dev = torch.device("cuda:0")
    rag = AutoModelForRagE2E(str(G / "tiny_retriever"), str(G / "tiny_generator")).to(dev)
    g_tok = rag.generator_tokenizer
    g_tok.pad_token = g_tok.eos_token
    rag.train()
    towers = graph == "towers"  # tower fwd/bwd as graphs, collectives + loss + optimizer eager (the W > 1 mode)
    graph = bool(graph) and not towers
    opt = make_capturable_adam(rag.parameters(), gold["lr"], dev) if graph else torch.optim.Adam(rag.parameters(), lr=gold["lr"])
    def mk(o):
        return get_scheduler("linear", optimizer=o, num_warmup_steps=gold["warmup"], num_training_steps=gold["total_steps"])

    sched = TensorLRScheduler(opt, gold["lr"], mk) if graph else mk(opt)
    step = RagE2EStep(rag, opt, sched, 100, autocast_dtype=None, inplace_grad=inplace, overlap_towers=graph or towers,
                      graph_towers=towers, graph_after=0)
    if graph:
        step = GraphedStep(step, warmup=0)  # no hidden warm-up steps: the trajectory must start at step 0
    losses = []
    for b in _batches(rag.retriever_tokenizer, g_tok, gold, dev):
        losses.append(float(step(b)))
    if graph:
        assert step.failed is None and step.graph is not None, step.failed
    if towers:
        assert step.towers_failed is None and step.towers is not None, step.towers_failed
    for got, ref in zip(losses, gold["losses"]):
        assert abs(got - ref) <= 1e-3 * abs(ref), (losses, gold["losses"])
    final = float(sum(p.detach().abs().sum() for p in rag.parameters()))
    assert abs(final - gold["final_param_abs_sum"]) <= 1e-4 * gold["final_param_abs_sum"]


def test_retriever_only_step_runs_and_decreases_loss():
    from transformers import AutoModel, AutoTokenizer

    from dalm_amd.models import AutoModelForSentenceEmbedding
    from dalm_amd.training.step import RetrieverStep
    from dalm_amd.training.utils.retriever_only_dataloader_utils import preprocess_dataset

    dev = torch.device("cuda:0")
    tok = AutoTokenizer.from_pretrained(str(G / "tiny_retriever"))
    model = AutoModelForSentenceEmbedding.from_modules(AutoModel.from_pretrained(str(G / "tiny_retriever")), tok,
                                                      get_peft=True).to(dev)
    rows = json.loads((G / "host_golden.json").read_text())["rows"]
    enc = preprocess_dataset(rows, tok, "Question", "Abstract", 12, 24)
    batch = {k: torch.tensor(v, device=dev) for k, v in enc.items()}
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-3)
    step = RetrieverStep(model, opt, None, 100, autocast_dtype=None)
    model.eval()  # LoRA dropout off: deterministic descent check
    losses = [float(step(batch)) for _ in range(8)]
    assert losses[-1] < losses[0]


def test_train_e2e_end_to_end_on_csv(tmp_path):
    """The trainer entry point itself: csv -> tokenise -> hipGraph-replayed steps -> checkpoints in the
    reference's directory layout -> resume."""
    import csv

    from dalm_amd.training.rag_e2e.train_rage2e import train_e2e

    rows = json.loads((G / "host_golden.json").read_text())["rows"]
    path = tmp_path / "rows.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Question", "Abstract", "Answer"])
        for i in range(12):
            k = i % 5
            w.writerow([rows["Question"][k], rows["Abstract"][k], rows["Answer"][k]])
    out = tmp_path / "out"
    losses = []
    train_e2e(str(path), str(G / "tiny_retriever"), str(G / "tiny_generator"), query_max_len=12, passage_max_len=24,
              generator_max_len=40, per_device_train_batch_size=4, learning_rate=1e-3, num_train_epochs=2,
              num_warmup_steps=0, output_dir=str(out), checkpointing_steps="epoch", with_tracking=True,
              mixed_precision="no", on_step=lambda s, l: losses.append(float(l)))
    assert len(losses) == 6 and losses[-1] < losses[0]
    for sub in ("retriever", "generator", "epoch_0/retriever", "epoch_1/generator", "logs"):
        assert (out / sub).exists(), sub
    assert (out / "epoch_1" / "trainer_state.pt").exists()
    more = []
    train_e2e(str(path), str(G / "tiny_retriever"), str(G / "tiny_generator"), query_max_len=12, passage_max_len=24,
              generator_max_len=40, per_device_train_batch_size=4, learning_rate=1e-3, num_train_epochs=3,
              num_warmup_steps=0, output_dir=str(out), resume_from_checkpoint=str(out / "epoch_1"), with_tracking=False,
              mixed_precision="no", on_step=lambda s, l: more.append((s, float(l))))
    assert [s for s, _ in more] == [7, 8, 9]  # epochs 0-1 are skipped, one more epoch of 3 steps runs


def test_multi_gpu_code_path_on_one_rank_matches_reference(monkeypatch):
    """The W > 1 step (RCCL all-gathers on a side stream, stats exchange, flat gradient all-reduce, graphed
    towers) run through a real one-rank RCCL process group: collectives are identities, so the trajectory
    must still be the reference's."""
    import torch.distributed as dist
    from transformers import get_scheduler

    from dalm_amd.fused import TorchDistComm
    from dalm_amd.models import AutoModelForRagE2E
    from dalm_amd.sharded import init_distributed
    from dalm_amd.training.step import RagE2EStep

    monkeypatch.setenv("DALM_FORCE_DIST", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29641")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("LOCAL_RANK", "0")
    comm, dev = init_distributed()
    try:
        assert isinstance(comm, TorchDistComm) and comm.world_size == 1 and dist.get_backend() == "nccl"
        gold = json.loads((G / "step_golden.json").read_text())
        rag = AutoModelForRagE2E(str(G / "tiny_retriever"), str(G / "tiny_generator")).to(dev)
        g_tok = rag.generator_tokenizer
        g_tok.pad_token = g_tok.eos_token
        rag.train()
        opt = torch.optim.Adam(rag.parameters(), lr=gold["lr"])
        sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=gold["warmup"], num_training_steps=gold["total_steps"])
        step = RagE2EStep(rag, opt, sched, 100, comm=comm, autocast_dtype=None, inplace_grad=True, overlap_towers=True,
                          graph_towers=True, graph_after=0)
        assert step.side_stream is not None
        losses = [float(step(b)) for b in _batches(rag.retriever_tokenizer, g_tok, gold, dev)]
        assert step.towers is not None, step.towers_failed
        for got, ref in zip(losses, gold["losses"]):
            assert abs(got - ref) <= 1e-3 * abs(ref), (losses, gold["losses"])
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_falcon_architecture_generator_runs_through_the_step():
    """cfg5 uses a Falcon generator (vocab 65024, bf16): a tiny random-init Falcon goes through the same step
    (1024-thread packed bf16 CE rows) and learns."""
    from transformers import AutoModel, AutoTokenizer, FalconConfig, FalconForCausalLM

    from dalm_amd.models import AutoModelForRagE2E
    from dalm_amd.training.step import RagE2EStep

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tok = AutoTokenizer.from_pretrained(str(G / "tiny_retriever"))
    falcon = FalconForCausalLM(FalconConfig(vocab_size=65024, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                                            new_decoder_architecture=False, multi_query=True, parallel_attn=True,
                                            bias=False, hidden_dropout=0.0, attention_dropout=0.0))
    rag = AutoModelForRagE2E.from_modules(AutoModel.from_pretrained(str(G / "tiny_retriever")), falcon, tok, tok).to(dev)
    rag.train()
    g = torch.Generator().manual_seed(1)
    B, Tg = 4, 32
    lens = torch.randint(10, Tg + 1, (B, 1), generator=g)
    batch = {
        "retriever_query_input_ids": torch.randint(4, 50, (B, 12), generator=g),
        "retriever_query_attention_mask": torch.ones(B, 12, dtype=torch.int64),
        "retriever_passage_input_ids": torch.randint(4, 50, (B, 24), generator=g),
        "retriever_passage_attention_mask": torch.ones(B, 24, dtype=torch.int64),
        "generator_input_input_ids": torch.randint(0, 65024, (B, Tg), generator=g),
        "generator_input_attention_mask": (torch.arange(Tg).unsqueeze(0) < lens).long(),
        "query_passage_input_len": (lens.squeeze(1).float() * 0.8).long().clamp(min=1),
    }
    batch = {k: v.to(dev) for k, v in batch.items()}
    opt = torch.optim.Adam(rag.parameters(), lr=2e-3)
    step = RagE2EStep(rag, opt, None, 100, autocast_dtype=torch.bfloat16, inplace_grad=True)
    losses = [float(step(batch)) for _ in range(8)]
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
