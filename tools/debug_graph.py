import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from transformers import get_scheduler
from dalm_amd.models import AutoModelForRagE2E
from dalm_amd.training.graphed import GraphedStep, TensorLRScheduler, make_capturable_adam
from dalm_amd.training.step import RagE2EStep
from test_step_parity_gpu import _batches, G

gold = json.loads((G / "step_golden.json").read_text())
dev = torch.device("cuda:0")
def run(graph, overlap, capturable, same_shape=False, inplace=True):
    rag = AutoModelForRagE2E(str(G / "tiny_retriever"), str(G / "tiny_generator")).to(dev)
    g_tok = rag.generator_tokenizer; g_tok.pad_token = g_tok.eos_token; rag.train()
    opt = make_capturable_adam(rag.parameters(), gold["lr"], dev) if capturable else torch.optim.Adam(rag.parameters(), lr=gold["lr"])
    mk = lambda o: get_scheduler("linear", optimizer=o, num_warmup_steps=0, num_training_steps=20)
    sched = TensorLRScheduler(opt, gold["lr"], mk) if capturable else mk(opt)
    step = RagE2EStep(rag, opt, sched, 100, autocast_dtype=None, inplace_grad=inplace, overlap_towers=overlap)
    if graph: step = GraphedStep(step, warmup=0)
    bs = _batches(rag.retriever_tokenizer, g_tok, gold, dev)
    if same_shape: bs = [bs[0]] * 5
    out = [round(float(step(b)), 5) for b in bs]
    return out
print("gold          ", [round(x, 5) for x in gold["losses"]])
print("eager         ", run(False, False, False))
print("eager capt    ", run(False, False, True))
print("eager ovl     ", run(False, True, False))
print("graph no-ovl  ", run(True, False, True))
print("graph ovl     ", run(True, True, True))
print("eager same    ", run(False, False, False, same_shape=True))
print("graph same    ", run(True, False, True, same_shape=True))
print("graph same ovl", run(True, True, True, same_shape=True))
print("graph noinpl  ", run(True, False, True, inplace=False))
