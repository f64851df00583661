"""AutoModelForSentenceEmbedding with the reference's surface
(dalm/models/retriever_only_base_model.py:10-110); pooling/normalise on the HIP kernel."""
from __future__ import annotations

from typing import Optional

import torch

from ..fused import pool_l2norm
from ..utils import eos_mask
from . import lora
from .fastpath import use_native_rms_norm
from .rag_e2e_base_model import nf4_enabled, to_nf4


class AutoModelForSentenceEmbedding(torch.nn.Module):
    def __init__(self, model_name: str, normalize: bool = True, use_bnb: bool = True, get_peft: bool = True,
                 is_autoregressive: bool = False, *, torch_dtype: Optional[torch.dtype] = None,
                 device: Optional[str] = None) -> None:
        super().__init__()
        from transformers import AutoModel, AutoModelForCausalLM, AutoTokenizer

        model_type = AutoModel if not is_autoregressive else AutoModelForCausalLM
        kw = {} if torch_dtype is None else {"dtype": torch_dtype}
        model = model_type.from_pretrained(model_name, **kw)
        # the reference pins device 0 (device_map={"": 0}, :25); keep that default when a GPU exists
        if device is None and torch.cuda.is_available():
            device = "cuda:0"
        if device is not None:
            model = model.to(device)
        self._assemble(model, AutoTokenizer.from_pretrained(model_name), normalize, get_peft, is_autoregressive,
                       use_bnb)

    @classmethod
    def from_modules(cls, model, tokenizer=None, normalize: bool = True, get_peft: bool = False,
                     is_autoregressive: bool = False, use_bnb: bool = False) -> "AutoModelForSentenceEmbedding":
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        self._assemble(model, tokenizer, normalize, get_peft, is_autoregressive, use_bnb)
        return self

    def _assemble(self, model, tokenizer, normalize, get_peft, is_autoregressive, use_bnb=False) -> None:
        if use_bnb and nf4_enabled(use_bnb):          # the reference's 4-bit load (:23-27), before the adapters
            model = to_nf4(model)
        self.model = model
        if is_autoregressive:
            use_native_rms_norm(self.model)
        from .attention import use_hip_attention_backward

        use_hip_attention_backward(self.model)        # BERT / Llama-family: dalm_attn_* (attention dropout inside the kernels)
        if get_peft:
            lora.inject_lora(self.model, ["key", "query", "value"] if not is_autoregressive else ["q_proj", "v_proj"])
        # frozen projections: backward GEMM through a transposed weight copy; BERT layers: dropout + add + LayerNorm in one launch
        # each way (both no-ops for a model that is fine-tuned in full: they look at requires_grad)
        from .fastpath import use_bert_layer_kernels
        from .frozen_linear import use_transposed_dgrad

        use_transposed_dgrad(self.model)
        use_bert_layer_kernels(self.model)
        self.normalize = normalize
        self.is_autoregressive = is_autoregressive
        self.tokenizer = tokenizer
        if is_autoregressive and tokenizer is not None:
            tokenizer.add_eos_token = True
            tokenizer.pad_token = tokenizer.eos_token

    def hidden(self, input_ids: torch.Tensor, attention_mask: torch.Tensor):
        if self.is_autoregressive:
            h = self.model(input_ids, attention_mask=attention_mask, output_hidden_states=True,
                           return_dict=True).hidden_states[-1]
            return h, eos_mask(attention_mask)
        return self.model(input_ids, attention_mask)[0], attention_mask

    def forward(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        h, pool_mask = self.hidden(input_ids, attention_mask)
        return pool_l2norm(h, pool_mask, self.normalize)

    def mean_pooling(self, token_embeddings: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        return pool_l2norm(token_embeddings, attention_mask, False)

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.model, name)

    def print_trainable_parameters(self) -> None:
        print(lora.trainable_parameter_summary(self.model))

    def attach_pre_trained_peft_layers(self, peft_retriever_path: str, device: str) -> None:
        lora.load_adapter(self.model, peft_retriever_path)
        self.model = lora.merge_and_unload(self.model).to(device).eval()
