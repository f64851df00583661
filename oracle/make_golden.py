"""Generate tests/golden/*.npz by running the REFERENCE's own code on CPU.

Runs only in the build container (needs /root/reference).  Recipe (SURVEY.md
section 8c): import transformers first, plant a 4-name dummy `peft` module so
`dalm.models.*` imports, import the reference modules, remove the stub again.
The functions exercised are exactly
    dalm.training.utils.train_utils.{get_cosine_sim,get_nt_xent_loss,get_nll,
        marginalize_log_probs,compute_marginalized_loss_from_logits}
    dalm.models.rag_e2e_base_model.AutoModelForRagE2E.mean_pooling (+F.normalize)
    dalm.utils.eos_mask
executed in float64 AND float32 under autograd; inputs, outputs and gradients
are stored.  The committed .npz files travel to the GPU box; this script and
/root/reference do not need to.

    python oracle/make_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import sys
import os
import types
import zlib
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
# DALM_GOLDEN_OUT=<dir>: regenerate somewhere else and diff against tests/golden (reproducibility check)
OUT = Path(os.environ.get("DALM_GOLDEN_OUT") or Path(__file__).resolve().parent.parent / "tests" / "golden")


def import_reference():
    import transformers  # noqa: F401  (must be imported before the stub goes in)

    stub = types.ModuleType("peft")
    for name in ("LoraConfig", "PeftModel", "TaskType", "get_peft_model"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["peft"] = stub
    sys.path.insert(0, str(REF))
    try:
        import dalm.models.rag_e2e_base_model as m_rag
        import dalm.training.utils.train_utils as tu
        import dalm.utils as du
    finally:
        sys.modules.pop("peft", None)
        sys.path.remove(str(REF))
    return tu, m_rag, du


def make_batch(gen: torch.Generator, B, D, Tg, V, *, pad_side="right", qlen_mode="normal", dead_row=False,
               logit_gain=2.0, normalize=True):
    q = torch.randn(B, D, generator=gen, dtype=torch.float64)
    p = torch.randn(B, D, generator=gen, dtype=torch.float64)
    if normalize:
        q = q / q.norm(dim=1, keepdim=True)
        p = p / p.norm(dim=1, keepdim=True)
    logits = logit_gain * torch.randn(B, Tg, V, generator=gen, dtype=torch.float64)
    ids = torch.randint(0, V, (B, Tg), generator=gen)
    lens = torch.randint(max(2, Tg // 3), Tg + 1, (B,), generator=gen)
    mask = torch.zeros(B, Tg, dtype=torch.int64)
    for b in range(B):
        n = int(lens[b])
        if pad_side == "right":
            mask[b, :n] = 1
        else:
            mask[b, Tg - n:] = 1
    if dead_row and B > 1:
        mask[B - 1] = 0
    if qlen_mode == "normal":
        qlen = torch.clamp((lens.float() * 0.8).floor().long(), min=1)
    elif qlen_mode == "beyond":      # qlen >= Tg: no doc term anywhere (un-truncated prompt longer than Tg)
        qlen = torch.full((B,), Tg + 7, dtype=torch.int64)
    elif qlen_mode == "one":         # qlen == 1: every shifted row gets the doc term
        qlen = torch.ones(B, dtype=torch.int64)
    elif qlen_mode == "mixed":
        qlen = torch.tensor([(1, Tg - 1, Tg, Tg + 3, 2)[i % 5] for i in range(B)], dtype=torch.int64)
    else:
        raise ValueError(qlen_mode)
    return q, p, logits, ids, mask, qlen


def run_reference_loss(tu, q, p, logits, ids, mask, qlen, scale, dtype):
    q = q.to(dtype).clone().requires_grad_(True)
    p = p.to(dtype).clone().requires_grad_(True)
    lg = logits.to(dtype).clone().requires_grad_(True)
    S = tu.get_cosine_sim(q, p, scale)
    S.retain_grad()
    loss_q = tu.get_nt_xent_loss(S)
    loss_p = tu.get_nt_xent_loss(S.t())
    con = (loss_q + loss_p) / 2.0
    gen = tu.compute_marginalized_loss_from_logits(lg, ids, mask, S, qlen)
    total = con + gen
    total.backward()
    return {
        "S": S.detach(), "loss_query": loss_q.detach(), "loss_passage": loss_p.detach(),
        "contrastive": con.detach(), "generator": gen.detach(), "loss": total.detach(),
        "dS": S.grad, "dq": q.grad, "dp": p.grad, "dlogits": lg.grad,
    }


def run_reference_con_only(tu, q, p, scale, dtype):
    q = q.to(dtype).clone().requires_grad_(True)
    p = p.to(dtype).clone().requires_grad_(True)
    S = tu.get_cosine_sim(q, p, scale)
    con = (tu.get_nt_xent_loss(S) + tu.get_nt_xent_loss(S.t())) / 2.0
    con.backward()
    return {"con_only": con.detach(), "con_only_dq": q.grad, "con_only_dp": p.grad}


def to_np(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


LOSS_CASES = {
    # name: (B, D, Tg, V, kwargs)
    "base_right_pad":      (4, 32, 12, 50, {}),
    "left_pad":            (3, 32, 12, 50, {"pad_side": "left"}),
    "batch_one":           (1, 16, 10, 30, {}),
    "partial_batch_3":     (3, 48, 9, 37, {"qlen_mode": "mixed"}),
    "qlen_beyond_Tg":      (4, 32, 12, 50, {"qlen_mode": "beyond"}),
    "qlen_one":            (4, 32, 12, 50, {"qlen_mode": "one"}),
    "dead_row":            (5, 32, 12, 50, {"dead_row": True, "qlen_mode": "mixed"}),
    "bge_small_dim":       (5, 384, 16, 201, {"qlen_mode": "mixed"}),
    "unnormalised_embs":   (4, 24, 8, 33, {"normalize": False}),
    "odd_vocab_1000":      (6, 64, 24, 1003, {"pad_side": "left", "qlen_mode": "mixed", "logit_gain": 4.0}),
}


def main() -> None:
    tu, m_rag, du = import_reference()
    OUT.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)

    # ---- loss path ------------------------------------------------------------
    for i, (name, (B, D, Tg, V, kw)) in enumerate(LOSS_CASES.items()):
        gen = torch.Generator().manual_seed(1000 + i)
        scale = 100 if kw.get("normalize", True) else 1
        q, p, logits, ids, mask, qlen = make_batch(gen, B, D, Tg, V, **kw)
        rec = {"q": q, "p": p, "logits": logits, "ids": ids, "mask": mask, "qlen": qlen,
               "scale": np.int64(scale)}
        r64 = run_reference_loss(tu, q, p, logits, ids, mask, qlen, scale, torch.float64)
        r32 = run_reference_loss(tu, q, p, logits, ids, mask, qlen, scale, torch.float32)
        rec.update({f"ref64_{k}": v for k, v in r64.items()})
        rec.update({f"ref32_{k}": v for k, v in r32.items()})
        rec.update({f"ref64_{k}": v for k, v in run_reference_con_only(tu, q, p, scale, torch.float64).items()})
        np.savez_compressed(OUT / f"loss_{name}.npz", **to_np(rec))
        print(f"loss_{name}: loss={float(r64['loss']):.12f} (fp32 {float(r32['loss']):.8f})")

    # ---- get_nll / marginalize_log_probs on their own ----------------------------
    gen = torch.Generator().manual_seed(77)
    lp = torch.log_softmax(torch.randn(3, 7, 19, generator=gen, dtype=torch.float64), dim=2)
    labels = torch.randint(0, 19, (3, 7), generator=gen)
    rec = {"lp": lp, "labels": labels, "nll": tu.get_nll(lp, labels)}
    for ql in (1, 2, 4, 7, 8, 12):
        rec[f"marg_q{ql}"] = tu.marginalize_log_probs(lp[0], torch.tensor([[-1.25]], dtype=torch.float64)[0],
                                                      torch.tensor(ql))
    np.savez_compressed(OUT / "pieces.npz", **to_np(rec))

    # ---- pooling + normalise + eos_mask -----------------------------------------------
    pool = m_rag.AutoModelForRagE2E.mean_pooling
    for name, (B, T, D, normalize, dead) in {
        "pool_base": (4, 9, 32, True, False),
        "pool_bge_small": (3, 13, 384, True, False),
        "pool_no_norm": (3, 7, 20, False, False),
        "pool_dead_row": (4, 6, 16, True, True),
    }.items():
        gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 1000)
        h = torch.randn(B, T, D, generator=gen, dtype=torch.float64)
        lens = torch.randint(1, T + 1, (B,), generator=gen)
        mask = (torch.arange(T).unsqueeze(0) < lens.unsqueeze(1)).long()
        if dead:
            mask[-1] = 0
        up = torch.randn(B, D, generator=gen, dtype=torch.float64)
        rec = {"h": h, "mask": mask, "upstream": up, "normalize": np.bool_(normalize)}
        for dt, tag in ((torch.float64, "ref64"), (torch.float32, "ref32")):
            hh = h.to(dt).clone().requires_grad_(True)
            e = pool(None, hh, mask)
            if normalize:
                e = torch.nn.functional.normalize(e, p=2, dim=1)
            (e * up.to(dt)).sum().backward()
            rec[f"{tag}_emb"] = e.detach()
            rec[f"{tag}_dh"] = hh.grad
        rec["eos_mask_left"] = du.eos_mask(mask)
        rec["eos_mask_right"] = du.eos_mask(mask.clamp(min=0) if not dead else torch.ones_like(mask), padding="right")
        np.savez_compressed(OUT / f"{name}.npz", **to_np(rec))
        print(name, "ok")


def make_tokenizer(words, path):
    """Small local WordLevel tokenizer (no network): [PAD]=0 [UNK]=1 [CLS]=2 [SEP]=3 + words."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast

    vocab = {"[PAD]": 0, "[UNK]": 1, "[CLS]": 2, "[SEP]": 3}
    for w in words:
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="[UNK]"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(single="[CLS] $A [SEP]", special_tokens=[("[CLS]", 2), ("[SEP]", 3)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="[PAD]", unk_token="[UNK]", cls_token="[CLS]",
                                   sep_token="[SEP]", eos_token="[SEP]")
    fast.save_pretrained(path)
    return fast


ROWS = {
    "Question": ["what is a heat pump", "who invented the transistor", "why is the sky blue at noon",
                 "how do vaccines train the immune system", "what does a compiler do"],
    "Abstract": ["a heat pump moves thermal energy from a cold space to a warm space using a refrigeration cycle",
                 "the transistor was invented at bell labs in 1947 by bardeen brattain and shockley",
                 "rayleigh scattering of sunlight by air molecules is stronger for short blue wavelengths",
                 "vaccines present harmless antigens so that the adaptive immune system forms memory cells",
                 "a compiler translates source code written in one language into another language usually machine code"],
    "Answer": ["it moves heat", "bardeen brattain shockley", "rayleigh scattering", "memory cells", "translates code"],
}


def main_host_goldens() -> None:
    """CLI defaults and preprocess_dataset outputs of the reference (host-side parity)."""
    import json

    tu, m_rag, du = import_reference()
    import transformers  # noqa: F401
    # resolve every lazy transformers / accelerate import the trainers need while `peft` is still absent
    import accelerate  # noqa: F401
    import datasets  # noqa: F401
    from accelerate import Accelerator  # noqa: F401
    from transformers import (AutoModel, AutoModelForCausalLM, AutoTokenizer, BitsAndBytesConfig,  # noqa: F401
                              SchedulerType, default_data_collator, get_scheduler)

    stub = types.ModuleType("peft")
    for name in ("LoraConfig", "PeftModel", "TaskType", "get_peft_model"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["peft"] = stub
    sys.path.insert(0, str(REF))
    try:
        import dalm.training.rag_e2e.train_rage2e as ref_e2e
        import dalm.training.retriever_only.train_retriever_only as ref_ret
        from dalm.training.utils.rag_e2e_dataloader_utils import preprocess_dataset as ref_pre_e2e
        from dalm.training.utils.retriever_only_dataloader_utils import preprocess_dataset as ref_pre_ret
    finally:
        sys.modules.pop("peft", None)
        sys.path.remove(str(REF))

    out = {}
    old = sys.argv
    try:
        sys.argv = ["x", "--retriever_name_or_path", "R", "--generator_name_or_path", "G"]
        ns = vars(ref_e2e.parse_args())
        out["e2e_defaults"] = {k: (v.value if hasattr(v, "value") else v) for k, v in ns.items()}
        sys.argv = ["x", "--retriever_name_or_path", "R"]
        ns = vars(ref_ret.parse_args())
        out["retriever_defaults"] = {k: (v.value if hasattr(v, "value") else v) for k, v in ns.items()}
    finally:
        sys.argv = old
    import inspect

    out["train_e2e_signature"] = [[n, (p.default.value if hasattr(p.default, "value") else p.default)
                                   if p.default is not inspect._empty else "<required>"]
                                  for n, p in inspect.signature(ref_e2e.train_e2e).parameters.items()]
    out["train_retriever_signature"] = [[n, (p.default.value if hasattr(p.default, "value") else p.default)
                                         if p.default is not inspect._empty else "<required>"]
                                        for n, p in inspect.signature(ref_ret.train_retriever).parameters.items()]
    words = sorted({w for col in ROWS.values() for t in col for w in t.split()} | {"#query#", "#passage#", "#answer#"})
    tok = make_tokenizer(words, str(OUT / "wordlevel_tokenizer"))
    out["rows"] = ROWS
    out["pre_e2e"] = ref_pre_e2e(ROWS, tok, tok, "Question", "Abstract", "Answer", 12, 24, 40)
    out["pre_e2e"] = {k: v for k, v in out["pre_e2e"].items()}
    out["pre_ret"] = dict(ref_pre_ret(ROWS, tok, "Question", "Abstract", 12, 24))
    (OUT / "host_golden.json").write_text(json.dumps(out, indent=1, default=str))
    print("host_golden.json ok")


def main_step_golden() -> None:
    """a12: the reference's step body (train_rage2e.py:431-474) on tiny random-init models, fixed batches,
    dropout 0, fp32, Adam + linear schedule -> per-step losses.  The tiny models are saved under
    tests/golden/ so the GPU test loads identical weights."""
    import json

    from transformers import BertConfig, BertModel, LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast, get_scheduler

    tu, m_rag, du = import_reference()
    sys.path.insert(0, str(REF))
    try:
        from dalm.training.utils.rag_e2e_dataloader_utils import preprocess_dataset as ref_pre_e2e
    finally:
        sys.path.remove(str(REF))
    tok_dir = OUT / "wordlevel_tokenizer"
    tok = PreTrainedTokenizerFast.from_pretrained(str(tok_dir))
    V = len(tok)
    torch.manual_seed(1234)
    bert = BertModel(BertConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                                vocab_size=V, max_position_embeddings=64, hidden_dropout_prob=0.0,
                                attention_probs_dropout_prob=0.0))
    llama = LlamaForCausalLM(LlamaConfig(hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                         num_key_value_heads=2, intermediate_size=64, vocab_size=V,
                                         max_position_embeddings=64, attention_dropout=0.0, pad_token_id=0))
    for name, model in (("tiny_retriever", bert), ("tiny_generator", llama)):
        d = OUT / name
        model.save_pretrained(str(d))
        tok.save_pretrained(str(d))
    import contextlib

    def run(autocast: bool):
        """autocast=True: what `accelerate --mixed_precision bf16` gives the reference on this box - the model forward
        under torch.autocast(bfloat16), its outputs up-cast to fp32 (accelerator.py convert_outputs_to_fp32), the
        reference's loss code outside autocast on those fp32 values, fp32 master weights."""
        rag = m_rag.AutoModelForRagE2E(str(OUT / "tiny_retriever"), str(OUT / "tiny_generator"))
        g_tok = rag.generator_tokenizer
        g_tok.pad_token = g_tok.eos_token
        enc = ref_pre_e2e(ROWS, rag.retriever_tokenizer, g_tok, "Question", "Abstract", "Answer", 12, 24, 40)
        full = {k: torch.tensor(v) for k, v in enc.items()}
        batches = [full, {k: v[:3] for k, v in full.items()}, full, {k: v[1:5] for k, v in full.items()}, full]
        opt = torch.optim.Adam(rag.parameters(), lr=1e-3)
        sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=0, num_training_steps=20)
        rag.train()
        ctx = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
        losses, cons, gens, gnorms = [], [], [], []
        for b in batches:
            with ctx():
                q = rag("retrieval", b["retriever_query_input_ids"], b["retriever_query_attention_mask"]).float()
                p = rag("retrieval", b["retriever_passage_input_ids"], b["retriever_passage_attention_mask"]).float()
            S = tu.get_cosine_sim(q, p, 100)
            con = (tu.get_nt_xent_loss(S) + tu.get_nt_xent_loss(S.t())) / 2.0
            with ctx():
                lg = rag("generation", b["generator_input_input_ids"], b["generator_input_attention_mask"]).float()
            gen = tu.compute_marginalized_loss_from_logits(lg, b["generator_input_input_ids"],
                                                           b["generator_input_attention_mask"], S,
                                                           b["query_passage_input_len"])
            loss = con + gen
            loss.backward()
            gnorms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in rag.parameters() if p.grad is not None))))
            opt.step(); sched.step(); rag.zero_grad()
            losses.append(float(loss)); cons.append(float(con)); gens.append(float(gen))
        return rag, losses, cons, gens, gnorms

    rag, losses, cons, gens, gnorms = run(False)
    rec = {"losses": losses, "contrastive": cons, "generator": gens, "lr": 1e-3, "warmup": 0, "total_steps": 20,
           "batch_rows": [[0, 5], [0, 3], [0, 5], [1, 5], [0, 5]], "query_max_len": 12, "passage_max_len": 24,
           "generator_max_len": 40,
           "final_param_abs_sum": float(sum(p.detach().abs().sum() for p in rag.parameters())),
           # round 3: global L2 norm of all parameter gradients at every step (before the optimizer consumes them)
           "grad_norms": gnorms}
    _, bl, bc, bg, bn = run(True)
    rec["bf16_autocast"] = {"losses": bl, "contrastive": bc, "generator": bg, "grad_norms": bn,
                            "how": "reference step body, model forwards under torch.autocast('cpu', bfloat16), outputs "
                                   "up-cast to fp32, loss code in fp32 outside autocast, fp32 master weights (what "
                                   "accelerate mixed_precision=bf16 hands the reference)"}
    (OUT / "step_golden.json").write_text(json.dumps(rec, indent=1))
    print("step_golden.json", losses, "bf16", bl)


def tiny_bge_small(vocab: int):
    """bge-small-en's width (384, 12 heads) at 2 layers: built from a fixed seed on CPU, so the weights are
    reproduced on the GPU box without committing 4 MB of safetensors; the golden records a checksum."""
    from transformers import BertConfig, BertModel

    torch.manual_seed(4321)
    return BertModel(BertConfig(hidden_size=384, num_hidden_layers=2, num_attention_heads=12, intermediate_size=256,
                                vocab_size=vocab, max_position_embeddings=64, hidden_dropout_prob=0.0,
                                attention_probs_dropout_prob=0.0))


def retriever_rows(n=19):
    """n synthetic (Question, Abstract) rows over the committed word-level vocabulary (the toy csv of the reference
    has 19 rows and the default batch is 32, so one partial batch of 19 is what configs[0] really trains on)."""
    import json
    import random

    base = json.loads((OUT / "host_golden.json").read_text())["rows"]
    words = sorted({w for col in base.values() for t in col for w in t.split()})
    rnd = random.Random(1234)
    return {"Question": [" ".join(rnd.choice(words) for _ in range(rnd.randint(3, 9))) for _ in range(n)],
            "Abstract": [" ".join(rnd.choice(words) for _ in range(rnd.randint(8, 30))) for _ in range(n)]}


def main_retriever_step_golden() -> None:
    """a12, retriever-only: the reference's step body (train_retriever_only.py:365-379) with the reference's own
    AutoModelForSentenceEmbedding (retriever_only_base_model.py) and preprocess_dataset, on a bge-small-width
    model, batches of 19 / 7 / 19 / 12 / 19 rows, dropout 0, fp32, Adam + linear schedule."""
    import json
    import tempfile

    from transformers import AutoModel, PreTrainedTokenizerFast, get_scheduler

    tu, m_rag, du = import_reference()
    stub = types.ModuleType("peft")
    for name in ("LoraConfig", "PeftModel", "TaskType", "get_peft_model"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["peft"] = stub
    sys.path.insert(0, str(REF))
    try:
        import dalm.models.retriever_only_base_model as m_ret
        from dalm.training.utils.retriever_only_dataloader_utils import preprocess_dataset as ref_pre_ret
    finally:
        sys.modules.pop("peft", None)
        sys.path.remove(str(REF))
    tok = PreTrainedTokenizerFast.from_pretrained(str(OUT / "wordlevel_tokenizer"))
    rows = retriever_rows(19)
    spans = [[0, 19], [0, 7], [0, 19], [5, 17], [0, 19]]

    def run(autocast: bool):
        import contextlib

        bert = tiny_bge_small(len(tok))
        init_sum = float(sum(p.detach().double().abs().sum() for p in bert.parameters()))
        with tempfile.TemporaryDirectory() as td:
            bert.save_pretrained(td)
            tok.save_pretrained(td)
            # the reference pins device 0 (device_map={"": 0}, retriever_only_base_model.py:25): strip it on CPU
            orig = AutoModel.from_pretrained

            def cpu_from_pretrained(*a, **kw):
                kw.pop("device_map", None)
                kw.pop("quantization_config", None)
                return orig(*a, **kw)

            AutoModel.from_pretrained = cpu_from_pretrained
            try:
                model = m_ret.AutoModelForSentenceEmbedding(td, use_bnb=False, get_peft=False)
            finally:
                AutoModel.from_pretrained = orig
        enc = ref_pre_ret(rows, model.tokenizer, "Question", "Abstract", 12, 32)
        full = {k: torch.tensor(v) for k, v in enc.items()}
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        sched = get_scheduler("linear", optimizer=opt, num_warmup_steps=0, num_training_steps=20)
        model.train()
        ctx = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
        losses, gnorms = [], []
        for a, b in spans:
            batch = {k: v[a:b] for k, v in full.items()}
            with ctx():
                q = model(batch["query_input_ids"], batch["query_attention_mask"]).float()
                p = model(batch["passage_input_ids"], batch["passage_attention_mask"]).float()
            logits = tu.get_cosine_sim(q, p, 100)
            loss = (tu.get_nt_xent_loss(logits) + tu.get_nt_xent_loss(logits.t())) / 2.0
            loss.backward()
            gnorms.append(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))))
            opt.step(); sched.step(); model.zero_grad()
            losses.append(float(loss))
        return model, enc, init_sum, losses, gnorms

    model, enc, init_sum, losses, gnorms = run(False)
    # only parameters that received gradients move (the BERT pooler is unused: rag_e2e_base_model.py:93 takes [0])
    rec = {"losses": losses, "lr": 1e-3, "warmup": 0, "total_steps": 20, "batch_rows": spans, "query_max_len": 12,
           "passage_max_len": 32, "rows": rows, "init_param_abs_sum": init_sum, "seed": 4321,
           "final_param_abs_sum": float(sum(p.detach().double().abs().sum() for p in model.parameters())),
           "pre_ret": {k: v for k, v in enc.items()},
           "grad_norms": gnorms}
    _, _, _, bl, bn = run(True)
    rec["bf16_autocast"] = {"losses": bl, "grad_norms": bn}
    (OUT / "retriever_step_golden.json").write_text(json.dumps(rec, indent=1))
    print("retriever_step_golden.json", losses, "bf16", bl)


def main_realwidth_golden(only=None) -> None:
    """Round 3: ONE step of the reference's own step body at the REAL widths of BASELINE.json's
    configs - bge-large (D = 1024), Llama-2-7b (4096, V = 32000), Falcon-7B (4544, V = 65024), batch 18 / 150 - on
    depth-1 random-init towers built from a CPU seed (oracle/realwidth.py), through the reference's OWN classes
    (`AutoModelForRagE2E` / `AutoModelForSentenceEmbedding`, get_peft=None: every parameter trains) and loss functions,
    in fp32 and under bf16 autocast.  Only scalars are committed: losses, the global gradient norm (and its split per
    tower) and a checksum of the seeded weights."""
    import contextlib
    import json
    import tempfile

    from transformers import AutoModel, PreTrainedTokenizerFast

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import realwidth as RW

    tu, m_rag, du = import_reference()
    stub = types.ModuleType("peft")
    for name in ("LoraConfig", "PeftModel", "TaskType", "get_peft_model"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["peft"] = stub
    sys.path.insert(0, str(REF))
    try:
        import dalm.models.retriever_only_base_model as m_ret
    finally:
        sys.modules.pop("peft", None)
        sys.path.remove(str(REF))
    tok = PreTrainedTokenizerFast.from_pretrained(str(OUT / "wordlevel_tokenizer"))
    path = OUT / "realwidth_golden.json"
    out = json.loads(path.read_text()) if (only and path.exists()) else {}
    for case in RW.CASES:
        if only and case not in only:
            continue
        retriever, generator = RW.build_case(case)
        rec = {"seed": RW.SEED, "depth": RW.CASES[case].get("depth", 1), "checksum_retriever": RW.checksum(retriever),
               "checksum_generator": RW.checksum(generator) if generator is not None else None}
        batch = RW.synthetic_batch(case)
        with tempfile.TemporaryDirectory() as td:
            rdir, gdir = f"{td}/retriever", f"{td}/generator"
            retriever.save_pretrained(rdir); tok.save_pretrained(rdir)
            if generator is not None:
                generator.save_pretrained(gdir); tok.save_pretrained(gdir)
            del retriever, generator
            if RW.CASES[case]["generator"] is not None:
                model = m_rag.AutoModelForRagE2E(rdir, gdir)
            else:
                orig = AutoModel.from_pretrained

                def cpu_from_pretrained(*a, **kw):
                    kw.pop("device_map", None)
                    kw.pop("quantization_config", None)
                    return orig(*a, **kw)

                AutoModel.from_pretrained = cpu_from_pretrained
                try:
                    model = m_ret.AutoModelForSentenceEmbedding(rdir, use_bnb=False, get_peft=False)
                finally:
                    AutoModel.from_pretrained = orig
        model.train()   # every dropout probability of these configs is 0
        for tag, autocast in (("fp32", False), ("bf16_autocast", True)):
            ctx = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
            model.zero_grad()
            if RW.CASES[case]["generator"] is not None:
                with ctx():
                    q = model("retrieval", batch["retriever_query_input_ids"], batch["retriever_query_attention_mask"]).float()
                    p = model("retrieval", batch["retriever_passage_input_ids"], batch["retriever_passage_attention_mask"]).float()
                S = tu.get_cosine_sim(q, p, 100)
                con = (tu.get_nt_xent_loss(S) + tu.get_nt_xent_loss(S.t())) / 2.0
                with ctx():
                    lg = model("generation", batch["generator_input_input_ids"], batch["generator_input_attention_mask"]).float()
                gen = tu.compute_marginalized_loss_from_logits(lg, batch["generator_input_input_ids"],
                                                               batch["generator_input_attention_mask"], S,
                                                               batch["query_passage_input_len"])
                loss = con + gen
                loss.backward()
                rec[tag] = {"loss": float(loss), "contrastive": float(con), "generator": float(gen),
                            "grad_norm": RW.grad_norm(model.parameters()),
                            "grad_norm_retriever": RW.grad_norm(model.retriever_model.parameters()),
                            "grad_norm_generator": RW.grad_norm(model.generator_model.parameters())}
                del lg, S, q, p, loss, gen, con
            else:
                with ctx():
                    q = model(batch["query_input_ids"], batch["query_attention_mask"]).float()
                    p = model(batch["passage_input_ids"], batch["passage_attention_mask"]).float()
                S = tu.get_cosine_sim(q, p, 100)
                loss = (tu.get_nt_xent_loss(S) + tu.get_nt_xent_loss(S.t())) / 2.0
                loss.backward()
                rec[tag] = {"loss": float(loss), "grad_norm": RW.grad_norm(model.parameters())}
            print(case, tag, rec[tag], flush=True)
        out[case] = rec
        del model
    path.write_text(json.dumps(out, indent=1))
    print("realwidth_golden.json ok")


def trainer_rows(n: int = 18, seed: int = 77):
    """n synthetic (Question, Abstract, Answer) rows over the words of tests/golden/wordlevel_tokenizer."""
    import random

    words = sorted({w for col in ROWS.values() for t in col for w in t.split()})
    r = random.Random(seed)

    def text(lo, hi):
        return " ".join(r.choice(words) for _ in range(r.randint(lo, hi)))

    return {"Question": [text(4, 12) for _ in range(n)], "Abstract": [text(20, 110) for _ in range(n)],
            "Answer": [text(1, 8) for _ in range(n)]}


def main_trainer_golden() -> None:
    """Round 4: the reference's own ENTRY POINT - `dalm.training.rag_e2e.train_rage2e.train_e2e`,
    unmodified, csv in, accelerate + DataLoader + Adam + linear schedule inside - at the REAL width of configs[2]
    (bge-large 1024 / Llama-2-7b 4096, V = 32000, Tq 50 / Tp 128 / Tg 256, batch 18), depth 1 (oracle/realwidth.py), fp32,
    every parameter training (no peft in the image).  The csv holds exactly one batch (18 rows), so the reference's
    shuffling DataLoader and ours present the same SET of rows to every step and the loss is order-independent: one
    optimizer step per epoch, 3 epochs.  The per-step loss is read where the reference hands it to
    `accelerator.backward`.  Committed: the rows and 3 scalars (+ the weight checksums)."""
    import csv
    import json
    import tempfile

    import accelerate
    from transformers import PreTrainedTokenizerFast

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import realwidth as RW

    import_reference()
    # resolve every lazy transformers / accelerate import the trainer needs while `peft` is still absent
    import datasets  # noqa: F401
    from accelerate import Accelerator  # noqa: F401
    from transformers import (AutoModel, AutoModelForCausalLM, AutoTokenizer, BitsAndBytesConfig,  # noqa: F401
                              SchedulerType, default_data_collator, get_scheduler)

    stub = types.ModuleType("peft")
    for name in ("LoraConfig", "PeftModel", "TaskType", "get_peft_model"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["peft"] = stub
    sys.path.insert(0, str(REF))
    try:
        import dalm.training.rag_e2e.train_rage2e as ref_e2e
    finally:
        sys.modules.pop("peft", None)
        sys.path.remove(str(REF))
    tok = PreTrainedTokenizerFast.from_pretrained(str(OUT / "wordlevel_tokenizer"))
    rows = trainer_rows()
    retriever, generator = RW.build_case("cfg3")
    rec = {"case": "cfg3", "depth": 1, "seed": RW.SEED, "checksum_retriever": RW.checksum(retriever),
           "checksum_generator": RW.checksum(generator), "rows": rows,
           "args": {"per_device_train_batch_size": 18, "query_max_len": 50, "passage_max_len": 128, "generator_max_len": 256,
                    "learning_rate": 1e-4, "num_warmup_steps": 0, "num_train_epochs": 3, "logit_scale": 100, "seed": 42}}
    losses = []
    orig_backward = accelerate.Accelerator.backward

    def recording_backward(self, loss, **kw):
        losses.append(float(loss.detach()))
        return orig_backward(self, loss, **kw)

    accelerate.Accelerator.backward = recording_backward
    try:
        with tempfile.TemporaryDirectory() as td:
            rdir, gdir, path = f"{td}/retriever", f"{td}/generator", f"{td}/rows.csv"
            retriever.save_pretrained(rdir); tok.save_pretrained(rdir)
            generator.save_pretrained(gdir); tok.save_pretrained(gdir)
            del retriever, generator
            with open(path, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Question", "Abstract", "Answer"])
                for i in range(len(rows["Question"])):
                    w.writerow([rows["Question"][i], rows["Abstract"][i], rows["Answer"][i]])
            ref_e2e.train_e2e(path, rdir, gdir, with_tracking=False, output_dir=None, use_peft=None, use_bnb=None,
                              sanity_test=False, **rec["args"])
    finally:
        accelerate.Accelerator.backward = orig_backward
    rec["losses"] = losses
    # the same entry point in accelerate's bf16 mode (ACCELERATE_MIXED_PRECISION=bf16: fp32 master weights, forward under
    # torch.autocast(bfloat16), outputs up-cast to fp32 before the reference's loss code): the precision dalm_amd's trainers
    # default to (--mixed_precision bf16)
    retriever, generator = RW.build_case("cfg3")
    bf16_losses = []

    def recording_backward16(self, loss, **kw):
        bf16_losses.append(float(loss.detach()))
        return orig_backward(self, loss, **kw)

    accelerate.Accelerator.backward = recording_backward16
    os.environ["ACCELERATE_MIXED_PRECISION"] = "bf16"
    try:
        from accelerate.state import AcceleratorState, PartialState

        AcceleratorState._reset_state(reset_partial_state=True)
        with tempfile.TemporaryDirectory() as td:
            rdir, gdir, path = f"{td}/retriever", f"{td}/generator", f"{td}/rows.csv"
            retriever.save_pretrained(rdir); tok.save_pretrained(rdir)
            generator.save_pretrained(gdir); tok.save_pretrained(gdir)
            del retriever, generator
            with open(path, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Question", "Abstract", "Answer"])
                for i in range(len(rows["Question"])):
                    w.writerow([rows["Question"][i], rows["Abstract"][i], rows["Answer"][i]])
            ref_e2e.train_e2e(path, rdir, gdir, with_tracking=False, output_dir=None, use_peft=None, use_bnb=None,
                              sanity_test=False, **rec["args"])
    finally:
        accelerate.Accelerator.backward = orig_backward
        os.environ.pop("ACCELERATE_MIXED_PRECISION", None)
        AcceleratorState._reset_state(reset_partial_state=True)
    rec["bf16_autocast_losses"] = bf16_losses
    (OUT / "trainer_golden.json").write_text(json.dumps(rec, indent=1))
    print("trainer_golden.json", losses, "bf16", bf16_losses)


def main_retriever_trainer_golden() -> None:
    """Round 4: the reference's own `train_retriever` (dalm/training/retriever_only/train_retriever_only.py:175-422,
    unmodified) at the REAL width and batch of configs[1] - bge-large (D = 1024), batch 150, Tq 50 / Tp 128 - on the depth-1
    seeded BERT of oracle/realwidth.py, fp32, every parameter training (use_peft=False, use_bnb=False).  The csv holds exactly
    one batch (150 rows): 3 epochs of one step.  `AutoModel.from_pretrained` is patched only to drop the reference's
    `device_map={"": 0}` (there is no GPU here).  Committed: the rows' seed and 3 per-step losses."""
    import csv
    import json
    import tempfile

    import accelerate
    import datasets  # noqa: F401
    from accelerate import Accelerator  # noqa: F401
    from transformers import (AutoModel, AutoModelForCausalLM, AutoTokenizer, BitsAndBytesConfig,  # noqa: F401
                              PreTrainedTokenizerFast, SchedulerType, default_data_collator, get_scheduler)

    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import realwidth as RW

    import_reference()
    stub = types.ModuleType("peft")
    for name in ("LoraConfig", "PeftModel", "TaskType", "get_peft_model"):
        setattr(stub, name, type(name, (), {}))
    sys.modules["peft"] = stub
    sys.path.insert(0, str(REF))
    try:
        import dalm.training.retriever_only.train_retriever_only as ref_ret
    finally:
        sys.modules.pop("peft", None)
        sys.path.remove(str(REF))
    tok = PreTrainedTokenizerFast.from_pretrained(str(OUT / "wordlevel_tokenizer"))
    rows = trainer_rows(n=150, seed=78)
    retriever, _ = RW.build_case("cfg2")
    rec = {"case": "cfg2", "depth": 1, "seed": RW.SEED, "checksum_retriever": RW.checksum(retriever),
           "rows_seed": 78, "rows_n": 150, "first_row": {k: v[0] for k, v in rows.items()},
           "args": {"per_device_train_batch_size": 150, "query_max_len": 50, "passage_max_len": 128, "learning_rate": 1e-4,
                    "num_warmup_steps": 0, "num_train_epochs": 3, "logit_scale": 100, "seed": 42}}
    losses = []
    orig_backward = accelerate.Accelerator.backward
    orig_fp = AutoModel.from_pretrained

    def recording_backward(self, loss, **kw):
        losses.append(float(loss.detach()))
        return orig_backward(self, loss, **kw)

    def cpu_from_pretrained(*a, **kw):
        kw.pop("device_map", None)
        kw.pop("quantization_config", None)
        return orig_fp(*a, **kw)

    accelerate.Accelerator.backward = recording_backward
    AutoModel.from_pretrained = cpu_from_pretrained
    try:
        with tempfile.TemporaryDirectory() as td:
            rdir, path = f"{td}/retriever", f"{td}/rows.csv"
            retriever.save_pretrained(rdir); tok.save_pretrained(rdir)
            del retriever
            with open(path, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Question", "Abstract", "Answer"])
                for i in range(len(rows["Question"])):
                    w.writerow([rows["Question"][i], rows["Abstract"][i], rows["Answer"][i]])
            ref_ret.train_retriever(rdir, path, with_tracking=False, output_dir=None, use_peft=False, use_bnb=False,
                                    sanity_test=False, **rec["args"])
    finally:
        accelerate.Accelerator.backward = orig_backward
        AutoModel.from_pretrained = orig_fp
    rec["losses"] = losses
    (OUT / "retriever_trainer_golden.json").write_text(json.dumps(rec, indent=1))
    print("retriever_trainer_golden.json", losses)


if __name__ == "__main__":
    if "--trainer-only" in sys.argv:
        main_trainer_golden()
        sys.exit(0)
    if "--retriever-trainer-only" in sys.argv:
        main_retriever_trainer_golden()
        sys.exit(0)
    if "--retriever-step-only" in sys.argv:
        main_retriever_step_golden()
        sys.exit(0)
    if "--realwidth-only" in sys.argv:
        main_realwidth_golden([a for a in sys.argv[1:] if not a.startswith("--")] or None)
        sys.exit(0)
    if not REF.exists():
        sys.exit("/root/reference not present: golden vectors can only be regenerated in the build container")
    main()
    main_host_goldens()
    main_step_golden()
    main_retriever_step_golden()
    main_realwidth_golden()
    main_trainer_golden()
    main_retriever_trainer_golden()
