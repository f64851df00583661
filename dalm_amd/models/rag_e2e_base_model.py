"""AutoModelForRagE2E with the reference's surface (dalm/models/rag_e2e_base_model.py:16-160).

Encoder / causal-LM forward and backward run through PyTorch-ROCm (hipBLASLt, SDPA); the tail
of the retrieval path - masked mean-pool + L2 normalise - is the hand-written HIP kernel
`dalm_pool_l2norm_{fwd,bwd}` instead of the reference's three [B,T,D] temporaries.
"""
from __future__ import annotations

from enum import Enum
from typing import Optional

import torch

from ..fused import pool_l2norm
from ..utils import eos_mask
from . import lora
from .attention import use_hip_attention_backward
from .fastpath import (use_bert_layer_kernels, use_capturable_falcon_heads, use_falcon_attention_kernels, use_falcon_layer_kernels,
                       use_fused_residual_norm, use_llama_attention_node, use_native_rms_norm, use_roll_rope, use_swiglu_kernel)


class Mode(Enum):
    GENERATOR = "generator"
    RETRIEVER = "retriever"
    BOTH = "both"


_BNB_MSG = ("use_bnb=%r: %s; the base weights stay in their loaded dtype (a 7B model in bf16 is 13 GB of the 288 GB "
            "HBM3E)")


def nf4_enabled(use_bnb) -> bool:
    """Whether a `use_bnb` request is served with real nf4 storage (dalm_amd/models/nf4.py).  It needs the GPU library
    (the quantise / dequantise kernels have no CPU form); `DALM_NF4=0` opts out.  When it cannot be served the request
    degrades to unquantised weights with a warning, so that default invocations of the reference's signatures
    (`train_retriever` has use_bnb=True) still run on a box without a GPU."""
    import os
    import warnings

    if not use_bnb:
        return False
    if os.environ.get("DALM_NF4", "1") == "0":
        warnings.warn(_BNB_MSG % (use_bnb, "nf4 switched off by DALM_NF4=0"), stacklevel=3)
        return False
    if not torch.cuda.is_available():
        warnings.warn(_BNB_MSG % (use_bnb, "nf4 storage needs the GPU kernels and no GPU is visible"), stacklevel=3)
        return False
    return True


def to_nf4(model: torch.nn.Module) -> torch.nn.Module:
    """The reference's 4-bit load (rag_e2e_base_model.py:50-58,137-142): every Linear outside the output head becomes
    nf4 storage on the current GPU (transformers places a quantised model there too)."""
    from . import nf4

    # the keep-list is decided where the model was loaded (transformers 4.x deep-copies the model to find tied weights:
    # on the GPU that would double a 7B model's footprint for a moment)
    keep = nf4.modules_kept_in_full_precision(model)
    where = next(model.parameters()).device
    if where.type != "cuda":
        model = model.to(torch.device("cuda", torch.cuda.current_device()))
    nf4.quantize_linears(model, keep)
    return model


class AutoModelForRagE2E(torch.nn.Module):
    def __init__(
        self,
        retriever_name: str,
        generator_name: str,
        normalize: bool = True,
        get_peft: Optional[Mode] = None,
        use_bnb: Optional[Mode] = None,
        retriever_is_autoregressive: bool = False,
        *,
        torch_dtype: Optional[torch.dtype] = None,
    ) -> None:
        super().__init__()
        from transformers import AutoModel, AutoModelForCausalLM, AutoTokenizer

        kw = {} if torch_dtype is None else {"dtype": torch_dtype}
        retriever = AutoModel.from_pretrained(retriever_name, **kw)
        generator = AutoModelForCausalLM.from_pretrained(generator_name, trust_remote_code=True, **kw)
        self._assemble(retriever, generator, AutoTokenizer.from_pretrained(retriever_name),
                       AutoTokenizer.from_pretrained(generator_name), normalize, get_peft,
                       retriever_is_autoregressive, use_bnb)

    @classmethod
    def from_modules(cls, retriever_model, generator_model, retriever_tokenizer=None, generator_tokenizer=None,
                     normalize: bool = True, get_peft: Optional[Mode] = None,
                     retriever_is_autoregressive: bool = False, use_bnb: Optional[Mode] = None) -> "AutoModelForRagE2E":
        """Build from already-constructed modules (random-init benchmarks, tests)."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self._assemble(retriever_model, generator_model, retriever_tokenizer, generator_tokenizer, normalize,
                       get_peft, retriever_is_autoregressive, use_bnb)
        return self

    def _assemble(self, retriever, generator, r_tok, g_tok, normalize, get_peft, autoregressive, use_bnb=None) -> None:
        if use_bnb is not None and nf4_enabled(use_bnb):      # quantise first, then adapters on top (reference :50-81)
            use_bnb = Mode(use_bnb)
            if use_bnb in (Mode.RETRIEVER, Mode.BOTH):
                retriever = to_nf4(retriever)
            if use_bnb in (Mode.GENERATOR, Mode.BOTH):
                generator = to_nf4(generator)
        self.retriever_model = retriever
        self.generator_model = generator
        self.retriever_tokenizer = r_tok
        self.generator_tokenizer = g_tok
        if autoregressive and r_tok is not None:
            r_tok.add_eos_token = True
            r_tok.pad_token = r_tok.eos_token
        self.normalize = normalize
        self.retriever_is_autoregressive = autoregressive
        use_native_rms_norm(self.generator_model)
        use_roll_rope(self.generator_model)
        use_swiglu_kernel(self.generator_model)
        use_fused_residual_norm(self.generator_model)
        use_capturable_falcon_heads(self.generator_model)
        use_falcon_layer_kernels(self.generator_model)
        use_falcon_attention_kernels(self.generator_model)
        use_hip_attention_backward(self.generator_model)
        use_llama_attention_node(self.generator_model)
        use_hip_attention_backward(self.retriever_model)          # BERT: head width 64, attention dropout inside the kernels
        if autoregressive:
            use_native_rms_norm(self.retriever_model)
        if get_peft is not None:
            get_peft = Mode(get_peft)
            if get_peft in (Mode.RETRIEVER, Mode.BOTH):
                targets = ["key", "query", "value"] if not autoregressive else ["q_proj", "v_proj"]
                lora.inject_lora(self.retriever_model, targets)
            if get_peft in (Mode.GENERATOR, Mode.BOTH):
                lora.inject_lora(self.generator_model, ["q_proj", "v_proj"])
        # frozen projections: the backward GEMM through a transposed weight copy (models/frozen_linear.py; after the adapters
        # are in, so that LoRA-wrapped projections and the plain Linears grouped with them keep their own modules)
        from .frozen_linear import use_transposed_dgrad

        use_transposed_dgrad(self.generator_model)
        use_transposed_dgrad(self.retriever_model)
        use_bert_layer_kernels(self.retriever_model)              # dropout + add + LayerNorm of a BERT layer: one launch each way

    # ---- retrieval tower ---------------------------------------------------------------
    def retrieval_hidden(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        """Token states + the mask that pools them (reference :84-93)."""
        if self.retriever_is_autoregressive:
            h = self.retriever_model(input_ids, attention_mask=attention_mask, output_hidden_states=True,
                                     return_dict=True).hidden_states[-1]
            return h, eos_mask(attention_mask)
        return self.retriever_model(input_ids, attention_mask)[0], attention_mask

    def retrieval_forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        h, pool_mask = self.retrieval_hidden(input_ids, attention_mask)
        return pool_l2norm(h, pool_mask, self.normalize)

    def forward(self, task: str, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        if task == "retrieval":
            return self.retrieval_forward(input_ids, attention_mask)
        # use_cache=False: the reference leaves transformers' default (config.use_cache = True), which makes every decoder layer
        # copy its keys and values into a DynamicCache nobody reads - two 38 MB torch.cat per layer at cfg3, 3.4 ms of the
        # step (profiles/history/r04_step_by_stream.txt).  The logits are the same tensor either way.
        return self.generator_model(input_ids=input_ids, attention_mask=attention_mask, use_cache=False).logits

    def mean_pooling(self, model_output: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        return pool_l2norm(model_output, attention_mask, False)

    def attach_pre_trained_peft_layers(self, peft_retriever_path: Optional[str], peft_generator_path: Optional[str],
                                       device: str) -> None:
        if peft_retriever_path is not None:
            lora.load_adapter(self.retriever_model, peft_retriever_path)
            self.retriever_model = lora.merge_and_unload(self.retriever_model).to(device).eval()
        if peft_generator_path is not None:
            lora.load_adapter(self.generator_model, peft_generator_path)
            self.generator_model = lora.merge_and_unload(self.generator_model).to(device).eval()
