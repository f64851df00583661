"""Checkpoint helpers keeping the reference's on-disk layout
(dalm/training/utils/train_utils.py:12-73; train_rage2e.py:486-524):
    <dir>/retriever, <dir>/generator   (HF save_pretrained, or a LoRA adapter when one is attached)
"""
from __future__ import annotations

import os
from typing import List

import torch

from ..models import lora
from ..models.rag_e2e_base_model import AutoModelForRagE2E
from ..models.retriever_only_base_model import AutoModelForSentenceEmbedding


def save_submodel(model: torch.nn.Module, path: str) -> None:
    if lora.has_lora(model):
        lora.save_adapter(model, path)
    else:
        from ..models.frozen_linear import uncat_weights

        uncat_weights(model)          # projection pairs that run as one GEMM share one weight tensor: separate them for the writer
        model.save_pretrained(path)


def load_submodel(model: torch.nn.Module, path: str) -> torch.nn.Module:
    if os.path.exists(os.path.join(path, lora.ADAPTER_CONFIG)):   # a peft-format adapter directory
        return lora.load_adapter(model, path)
    from transformers import AutoModel, AutoModelForCausalLM

    cls = AutoModelForCausalLM if hasattr(model, "lm_head") else AutoModel
    loaded = cls.from_pretrained(path)
    model.load_state_dict(loaded.state_dict())
    return model


def save_model_hook(models: List[torch.nn.Module], weights: List, output_dir: str) -> None:
    for model in models:
        if isinstance(model, AutoModelForSentenceEmbedding):
            save_submodel(model.model, output_dir)
        elif isinstance(model, AutoModelForRagE2E):
            save_submodel(model.generator_model, os.path.join(output_dir, "generator"))
            save_submodel(model.retriever_model, os.path.join(output_dir, "retriever"))
        else:
            raise NotImplementedError
        if weights:
            weights.pop()


def load_model_hook(models: List[torch.nn.Module], input_dir: str) -> None:
    while models:
        model = models.pop()
        if isinstance(model, AutoModelForRagE2E):
            load_submodel(model.generator_model, os.path.join(input_dir, "generator"))
            load_submodel(model.retriever_model, os.path.join(input_dir, "retriever"))
        elif isinstance(model, AutoModelForSentenceEmbedding):
            load_submodel(model.model, input_dir)
        else:
            raise NotImplementedError
