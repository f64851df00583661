"""Why the encoder-call graphs (GraphedEncoders) are refused next to a communicator: with the W > 1 gradient bucket the gradient norm
goes infinite on the second replay of a set.  Variants: both encoder graphs on ONE stream / on two streams; bucket overlap on / off."""
import json
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ.update(DALM_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
G = ROOT / "tests" / "golden"
from transformers import PreTrainedTokenizerFast, get_scheduler  # noqa: E402

import test_step_parity_gpu as T  # noqa: E402
from dalm_amd.models import AutoModelForSentenceEmbedding  # noqa: E402
from dalm_amd.sharded import init_distributed  # noqa: E402
from dalm_amd.training.step import RetrieverStep  # noqa: E402
from dalm_amd.training.utils.retriever_only_dataloader_utils import preprocess_dataset  # noqa: E402

comm, dev = init_distributed()
gold = json.loads((G / "retriever_step_golden.json").read_text())
tok = PreTrainedTokenizerFast.from_pretrained(str(G / "wordlevel_tokenizer"))
for one_stream, overlap in ((False, True), (True, True), (True, False)):
    model = AutoModelForSentenceEmbedding.from_modules(T._tiny_bge_small(len(tok), gold["seed"]), tok, normalize=True, get_peft=False).to(dev)
    model.train()
    enc = preprocess_dataset(gold["rows"], tok, query_column_name="Question", passage_column_name="Abstract",
                             query_max_len=gold["query_max_len"], passage_max_len=gold["passage_max_len"])
    full = {k: torch.tensor(v, device=dev) for k, v in enc.items()}
    opt = torch.optim.Adam(model.parameters(), lr=gold["lr"])
    sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=gold["warmup"], num_training_steps=gold["total_steps"])
    step = RetrieverStep(model, opt, sched, 100, comm=comm, autocast_dtype=None, overlap_towers=True, graph_towers=True, graph_after=0,
                         grad_overlap=overlap, track_grad_norm=True)
    step.graph_towers = True                       # past the one-rank guard, on purpose
    if one_stream:
        step.tower_stream = torch.cuda.current_stream()
    out = []
    for a, b in gold["batch_rows"]:
        loss = float(step({k: v[a:b] for k, v in full.items()}))
        out.append((round(loss, 5), round(float(step.grad_norm), 4)))
    print(f"one stream {one_stream}  bucket overlap {overlap}: {out}")
print("reference losses", [round(x, 5) for x in gold["losses"]])
