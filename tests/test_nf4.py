"""nf4 storage of the frozen base weights (`use_bnb`: rag_e2e_base_model.py:137-142 -> bitsandbytes, absent here).

CPU half: the restated algorithm (oracle/nf4.py) against what is published about the format.  GPU half: the HIP kernels
bit-for-bit against that restatement through the C ABI, `NF4Linear` against a plain Linear holding the dequantised
weight, and the model wrappers' `use_bnb` switch (footprint, LoRA on top, a training step, hipGraph capture).
"""
import json
import warnings
from pathlib import Path

import numpy as np
import pytest
import torch

import nf4 as O   # oracle/nf4.py (conftest puts oracle/ on sys.path)

G = Path(__file__).parent / "golden"


# ---------------------------------------------------------------- CPU: the oracle against the published format
def test_levels_are_the_qlora_quantile_construction():
    """QLoRA appendix E / bitsandbytes create_normal_map(offset=0.9677083): 8 positive quantile steps, 7 negative, an
    exact zero, normalised to [-1, 1]."""
    from scipy.stats import norm

    offset = 0.9677083
    pos = norm.ppf(np.linspace(offset, 0.5, 9)[:-1]).tolist()
    neg = (-norm.ppf(np.linspace(offset, 0.5, 8)[:-1])).tolist()
    v = np.array(sorted(pos + [0.0] + neg))
    v /= v.max()
    assert np.abs(v - O.LEVELS.astype(np.float64)).max() < 2e-7
    assert O.LEVELS[0] == -1.0 and O.LEVELS[7] == 0.0 and O.LEVELS[15] == 1.0
    assert np.all(np.diff(O.LEVELS) > 0)
    mid = (O.LEVELS[:-1].astype(np.float64) + O.LEVELS[1:]) / 2
    assert np.abs(mid - O.MIDPOINTS).max() < 1e-7


def test_oracle_block_layout_and_nibble_order():
    w = np.zeros(128, dtype=np.float32)
    w[0], w[1], w[2] = 2.0, -2.0, 0.0          # block 0: absmax 2 -> levels 15, 0, 7
    w[64], w[65] = 0.5, 0.25                    # block 1: absmax 0.5 -> 1.0 (15) and 0.5 (below the .5017 threshold -> .4407 = 12)
    p, a = O.quantize(w)
    assert p.dtype == np.uint8 and p.size == 64 and a.size == 2
    assert a.tolist() == [2.0, 0.5]
    assert p[0] == (15 << 4) | 0 and p[1] == (7 << 4) | 7      # element 2j in the HIGH nibble
    assert p[32] == (15 << 4) | 12
    back = O.dequantize(p, a, 128)
    assert back[0] == 2.0 and back[1] == -2.0 and back[2] == 0.0 and back[64] == 0.5
    assert back[65] == np.float32(0.5) * O.LEVELS[12]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 1000, 4096])
def test_oracle_roundtrip_error_is_bounded_by_half_the_widest_gap(n):
    rng = np.random.default_rng(n)
    w = (rng.standard_normal(n) * 0.02).astype(np.float32)
    p, a = O.quantize(w)
    assert p.size == (n + 1) // 2 and a.size == (n + 63) // 64
    back = O.dequantize(p, a, n)
    scale = np.repeat(a, 64)[:n]
    half_gap = np.diff(O.LEVELS).max() / 2
    assert np.all(np.abs(back - w) <= scale * half_gap * (1 + 1e-6))
    # idempotent: a dequantised tensor quantises to itself
    p2, a2 = O.quantize(back)
    assert np.array_equal(p2, p) and np.array_equal(a2, a)


def test_oracle_ties_and_zero_blocks():
    w = np.zeros(64, dtype=np.float32)
    p, a = O.quantize(w)
    assert a[0] == 0 and np.all(p == 0x77) and np.all(O.dequantize(p, a, 64) == 0)
    # a value exactly on a threshold goes to the LOWER level (the tree compares with `>`)
    w = np.zeros(64, dtype=np.float32)
    w[0] = 1.0
    w[1] = O.MIDPOINTS[14]
    p, _ = O.quantize(w)
    assert p[0] == (15 << 4) | 14


def test_use_bnb_without_a_gpu_degrades_with_a_warning(monkeypatch):
    """`train_retriever(use_bnb=True)` is the reference's default: on a box without a GPU (this suite) it must still run."""
    from dalm_amd.models import AutoModelForSentenceEmbedding
    from dalm_amd.models.rag_e2e_base_model import nf4_enabled

    assert nf4_enabled(None) is False and nf4_enabled(False) is False
    monkeypatch.setenv("DALM_NF4", "0")
    with pytest.warns(UserWarning, match="DALM_NF4=0"):
        assert nf4_enabled(True) is False
    monkeypatch.delenv("DALM_NF4")
    if not torch.cuda.is_available():
        with pytest.warns(UserWarning, match="no GPU is visible"):
            m = AutoModelForSentenceEmbedding(str(G / "tiny_retriever"), use_bnb=True, get_peft=False, device="cpu")
        assert m.model is not None and not getattr(m.model, "_dalm_nf4", False)


def test_which_linears_are_converted_is_transformers_own_rule():
    """The reference hands the model to transformers' 4-bit loader without `llm_int8_skip_modules`
    (rag_e2e_base_model.py:50-58): which Linears become 4-bit is transformers' decision.  The installed transformers'
    own `get_keys_to_not_convert` / `should_convert_module` are what the product calls; the in-tree fallback rule
    (output head only) converts exactly the same modules on the golden BERT, Llama and a Falcon-architecture model."""
    import re

    from transformers import AutoModel, AutoModelForCausalLM, FalconConfig
    from transformers.quantizers.base import get_keys_to_not_convert
    from transformers.quantizers.quantizers_utils import should_convert_module

    from dalm_amd.models import nf4

    falcon = AutoModelForCausalLM.from_config(FalconConfig(vocab_size=64, hidden_size=64, num_hidden_layers=2,
                                                           num_attention_heads=4, new_decoder_architecture=False,
                                                           multi_query=True, parallel_attn=True, bias=False))
    models = {"bert": AutoModel.from_pretrained(str(G / "tiny_retriever")),
              "llama": AutoModelForCausalLM.from_pretrained(str(G / "tiny_generator")), "falcon": falcon}
    for name, m in models.items():
        theirs = list(get_keys_to_not_convert(m))
        assert sorted(nf4.modules_kept_in_full_precision(m)) == sorted(theirs), name
        # transformers 5.x converts `type(module) is nn.Linear` only; 4.x (the reference's `transformers>4.35`) converted
        # every isinstance - Falcon's FalconLinear included, which the product keeps converting
        want = [n for n, mod in m.named_modules() if isinstance(mod, torch.nn.Linear) and should_convert_module(n, theirs)]
        got = [x[0] for x in nf4.linears_to_convert(m, nf4.modules_kept_in_full_precision(m))]
        own = [x[0] for x in nf4.linears_to_convert(m, nf4._own_keep_rule(m))]
        assert got == want and own == want and len(want) > 0, (name, set(got) ^ set(want), set(own) ^ set(want))
        head = m.get_output_embeddings()
        if head is not None:
            assert all(not re.fullmatch(r"lm_head", n) for n in want)
    for n, pats in (("lm_head", ["lm_head"]), ("a.b.c", ["b"]), ("a.b.c", ["a.b"]), ("a.b.c", ["c"]),
                    ("pooler.dense", ["pooler.dense.bias"]), ("encoder.layer.1.x", ["encoder.layer.*"])):
        assert nf4._skipped(n, pats) == (not should_convert_module(n, pats)), (n, pats)


# ---------------------------------------------------------------- GPU: kernels, module, wrappers
def _q(w_t):
    from dalm_amd.models import nf4

    return nf4.quantize(w_t)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 1000, 4096 * 33 + 5])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_kernels_match_the_oracle_bit_for_bit(n, dtype):
    from dalm_amd.models import nf4

    g = torch.Generator().manual_seed(n)
    w = (torch.randn(n, generator=g) * 0.03).to(dtype)
    if n >= 64:
        w[:64] = 0                               # an all-zero block
    want_p, want_a = O.quantize(w.float().numpy())
    p, a = nf4.quantize(w.cuda())
    assert np.array_equal(a.cpu().numpy(), want_a)
    assert np.array_equal(p.cpu().numpy(), want_p)
    back32 = nf4.dequantize(p, a, (n,), torch.float32).cpu().numpy()
    assert np.array_equal(back32, O.dequantize(want_p, want_a, n))
    back16 = nf4.dequantize(p, a, (n,), torch.bfloat16).cpu()
    assert torch.equal(back16, torch.from_numpy(back32).to(torch.bfloat16))      # RNE, as torch rounds


@pytest.mark.gpu
def test_kernel_argument_errors_are_reported_not_launched():
    import ctypes as C

    from dalm_amd import hip

    lib = hip.load()
    buf = torch.zeros(256, device="cuda")
    assert lib.dalm_nf4_quantize(None, 0, 64, hip.ptr(buf), hip.ptr(buf), None) == -1          # DALM_E_NULL
    assert lib.dalm_nf4_quantize(hip.ptr(buf), 7, 64, hip.ptr(buf), hip.ptr(buf), None) == -3  # DALM_E_DTYPE
    assert lib.dalm_nf4_dequantize(hip.ptr(buf), hip.ptr(buf), -1, 0, hip.ptr(buf), None) == -2
    assert lib.dalm_nf4_dequantize(hip.ptr(buf), hip.ptr(buf), 64, 0, hip.ptr(buf) + 4, None) == -4
    assert lib.dalm_nf4_quantize(None, 0, 0, None, None, None) == 0
    assert lib.dalm_nf4_packed_bytes(65) == 33 and lib.dalm_nf4_absmax_count(65) == 2
    assert b"dalm_nf4" in C.c_char_p(lib.dalm_last_error_string()).value


@pytest.mark.gpu
@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bias", [False, True])
def test_nf4_linear_is_a_bf16_linear_over_the_dequantised_weight(xdtype, bias):
    """bitsandbytes' Linear4bit recipe: x -> bf16, W' = dequant(W) in bf16, y = x W'^T (+ b) -> x.dtype; the gradient
    reaches x through the same W'."""
    from dalm_amd.models import nf4

    torch.manual_seed(0)
    lin = torch.nn.Linear(192, 80, bias=bias).cuda()
    q = nf4.NF4Linear(lin)
    w_ref = torch.from_numpy(O.roundtrip(lin.weight.detach().cpu().numpy())).cuda().to(torch.bfloat16)
    assert torch.equal(q.weight, w_ref)
    x = torch.randn(5, 7, 192, device="cuda", dtype=xdtype, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    y = q(x)
    b = None if not bias else lin.bias.detach().to(torch.bfloat16)
    y_ref = torch.nn.functional.linear(x2.to(torch.bfloat16), w_ref, b).to(xdtype)
    assert y.dtype == xdtype and torch.equal(y, y_ref)
    gy = torch.randn_like(y)
    y.backward(gy)
    y_ref.backward(gy)
    assert torch.equal(x.grad, x2.grad)
    assert not any(p.requires_grad for p in q.parameters())
    assert nf4.weight_bytes(q) == 192 * 80 // 2 + 192 * 80 // 64 * 4 + (80 * 4 if bias else 0)


def _tiny_rag(dev, **kw):
    from transformers import AutoModel, AutoModelForCausalLM, AutoTokenizer

    from dalm_amd.models import AutoModelForRagE2E

    tok = AutoTokenizer.from_pretrained(str(G / "tiny_retriever"))
    gtok = AutoTokenizer.from_pretrained(str(G / "tiny_generator"))
    gtok.pad_token = gtok.eos_token
    return AutoModelForRagE2E.from_modules(AutoModel.from_pretrained(str(G / "tiny_retriever")).to(dev),
                                           AutoModelForCausalLM.from_pretrained(str(G / "tiny_generator")).to(dev),
                                           tok, gtok, **kw)


@pytest.mark.gpu
def test_use_bnb_quantises_every_linear_but_the_head_and_lora_sits_on_top():
    from dalm_amd.models import lora, nf4
    from dalm_amd.models.rag_e2e_base_model import Mode

    dev = torch.device("cuda:0")
    plain = _tiny_rag(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # a served request does not warn
        rag = _tiny_rag(dev, use_bnb=Mode.BOTH, get_peft=Mode.BOTH)
    gen = rag.generator_model
    head = gen.get_output_embeddings()
    assert isinstance(head, torch.nn.Linear) and not isinstance(head, nf4.NF4Linear)   # lm_head keeps its precision
    kinds = {type(m).__name__ for m in gen.modules()}
    assert "NF4Linear" in kinds and "LoRALinear" in kinds
    lin_left = [n for n, m in gen.named_modules() if type(m) is torch.nn.Linear and m is not head and "lora_" not in n]
    assert lin_left == []
    assert all(isinstance(m.base_layer, nf4.NF4Linear) for m in gen.modules() if isinstance(m, lora.LoRALinear))
    assert not any(type(m) is torch.nn.Linear and "lora_" not in n for n, m in rag.retriever_model.named_modules())
    # only the adapters train
    names = [n for n, p in rag.named_parameters() if p.requires_grad]
    assert names and all("lora_" in n for n in names)

    def linear_bytes(model, skip):
        return sum(nf4.weight_bytes(m) for m in model.modules()
                   if (type(m) is torch.nn.Linear or isinstance(m, nf4.NF4Linear)) and m is not skip
                   and not isinstance(m, lora.LoRALinear) and m.in_features != 8 and m.out_features != 8)

    before = linear_bytes(plain.generator_model, plain.generator_model.get_output_embeddings())
    after = linear_bytes(gen, head)
    w = sum(m.in_features * m.out_features for m in gen.modules() if isinstance(m, nf4.NF4Linear))
    assert after == w // 2 + (w // 64) * 4                  # 0.5625 bytes per weight (no biases in the tiny Llama)
    assert before == 4 * w                                  # the fp32 load it replaces
    # the embeddings the quantised tower produces stay close to the unquantised tower's
    ids = torch.randint(5, 40, (4, 12), device=dev)
    mask = torch.ones_like(ids)
    rag.eval(), plain.eval()
    with torch.no_grad():
        e_q, e_p = rag.retrieval_forward(ids, mask), plain.retrieval_forward(ids, mask)
    cos = (e_q * e_p).sum(-1)
    assert cos.min() > 0.9, cos


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_training_step_on_nf4_bases_matches_the_same_step_on_dequantised_bf16_bases(graph):
    """The step over NF4Linear bases == the step over plain bf16 Linears holding the dequantised weights (what the
    kernel stands for), LoRA adapters on both: same losses over 4 steps, eagerly and as a captured hipGraph."""
    from dalm_amd.models import lora, nf4
    from dalm_amd.models.rag_e2e_base_model import Mode
    from dalm_amd.training.graphed import GraphedStep, make_capturable_adam
    from dalm_amd.training.step import RagE2EStep

    dev = torch.device("cuda:0")

    def build(quantised):
        torch.manual_seed(3)
        rag = _tiny_rag(dev, use_bnb=Mode.BOTH)
        if not quantised:
            for tower in (rag.retriever_model, rag.generator_model):
                for parent in list(tower.modules()):
                    for name, child in list(parent.named_children()):
                        if isinstance(child, nf4.NF4Linear):
                            setattr(parent, name, _Bf16Linear(child))
        torch.manual_seed(4)                      # same adapter initialisation on both sides
        lora.inject_lora(rag.retriever_model, ["key", "query", "value"], lora_dropout=0.0)
        lora.inject_lora(rag.generator_model, ["q_proj", "v_proj"], lora_dropout=0.0)
        return rag.train()

    class _Bf16Linear(torch.nn.Linear):
        """nn.Linear subclass computing the Linear4bit recipe on a resident bf16 weight."""

        def __init__(self, q):
            super().__init__(q.in_features, q.out_features, bias=q.bias is not None, device=dev, dtype=torch.bfloat16)
            with torch.no_grad():
                self.weight.copy_(q.weight)
                if q.bias is not None:
                    self.bias.copy_(q.bias)
            self.requires_grad_(False)

        def forward(self, x):
            return torch.nn.functional.linear(x.to(torch.bfloat16), self.weight, self.bias).to(x.dtype)

    gold = json.loads((G / "step_golden.json").read_text())
    import test_step_parity_gpu as T

    out = {}
    for quantised in (True, False):
        rag = build(quantised)
        params = [p for p in rag.parameters() if p.requires_grad]
        opt = make_capturable_adam(params, 1e-3, dev) if graph else torch.optim.Adam(params, lr=1e-3)
        step = RagE2EStep(rag, opt, None, 100, autocast_dtype=None, inplace_grad=True, graph_after=0)
        if graph:
            step = GraphedStep(step, warmup=0)
        batches = [b for b in T._batches(rag.retriever_tokenizer, rag.generator_tokenizer, gold, dev) if b["retriever_query_input_ids"].shape[0] == 5][:4]
        out[quantised] = [float(step(b)) for b in batches]
        if graph:
            assert step.failed is None and step.graph is not None, step.failed
    assert len(out[True]) >= 3
    for a, b in zip(out[True], out[False]):
        assert abs(a - b) <= 1e-6 * abs(b), out
    assert out[True][-1] != out[True][0]
