import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from transformers import get_scheduler
from dalm_amd.models import AutoModelForRagE2E
from dalm_amd.training.step import RagE2EStep
from test_step_parity_gpu import _batches, G

gold = json.loads((G / "step_golden.json").read_text())
dev = torch.device("cuda:0")
def run(graph_towers, inplace, overlap, same_shape=True, n=6):
    rag = AutoModelForRagE2E(str(G / "tiny_retriever"), str(G / "tiny_generator")).to(dev)
    g_tok = rag.generator_tokenizer; g_tok.pad_token = g_tok.eos_token; rag.train()
    opt = torch.optim.Adam(rag.parameters(), lr=gold["lr"])
    sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=0, num_training_steps=20)
    step = RagE2EStep(rag, opt, sched, 100, autocast_dtype=None, inplace_grad=inplace, overlap_towers=overlap,
                      graph_towers=graph_towers, graph_after=1)
    bs = _batches(rag.retriever_tokenizer, g_tok, gold, dev)
    bs = [bs[0]] * n if same_shape else bs
    out = [round(float(step(b)), 5) for b in bs]
    return out, step.towers is not None, step.towers_failed
print("eager            ", run(False, True, False))
print("towers inpl      ", run(True, True, False))
print("towers noinpl    ", run(True, False, False))
print("towers inpl ovl  ", run(True, True, True))
print("mixed shapes     ", run(True, True, True, same_shape=False))
print("gold             ", [round(x, 5) for x in gold["losses"]])
