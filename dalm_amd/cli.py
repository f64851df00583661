"""`dalm`-style command line for the two trainers on the MI355X path (typer), mirroring the argument surface of
the reference's `dalm train-rag-e2e` / `dalm train-retriever-only` (dalm/cli.py:41-167, 170-277): the same
positional arguments, the same option names and defaults as `train_e2e` / `train_retriever`.

    python -m dalm_amd.cli train-rag-e2e rows.csv BAAI/bge-large-en meta-llama/Llama-2-7b-hf --use-peft both
    python -m dalm_amd.cli train-retriever-only BAAI/bge-large-en rows.csv --per-device-train-batch-size 150

The eval / qa-gen commands of the reference are outside this package's scope.
The commands are generated from the trainer functions' own signatures, so the CLI cannot drift from them.
"""
from __future__ import annotations

import inspect
from enum import Enum
from typing import Optional

import typer

from . import __version__ as _v  # noqa: F401
from .models.rag_e2e_base_model import Mode


class DALMSchedulerType(str, Enum):
    LINEAR = "linear"
    COSINE = "cosine"
    COSINE_WITH_RESTARTS = "cosine_with_restarts"
    POLYNOMIAL = "polynomial"
    CONSTANT = "constant"
    CONSTANT_WITH_WARMUP = "constant_with_warmup"


cli = typer.Typer(add_completion=False, help="MI355X-native RAG-end2end / retriever-only training (DALM surface)")

_HELP = {
    "dataset_or_path": "Path to the dataset to train with: an hf dataset dir or a csv file.",
    "retriever_name_or_path": "Path to pretrained retriever or identifier from huggingface.co/models.",
    "generator_name_or_path": "Path to pretrained (causal) generator or identifier from huggingface.co/models.",
    "per_device_train_batch_size": "Batch size (per device).",
    "logit_scale": "Logit scale of the contrastive loss.",
    "use_peft": "LoRA fine-tuning (which tower(s)).",
    "use_bnb": "nf4 storage of the frozen base weights (HIP kernels; needs the GPU).",
    "checkpointing_steps": "Save state every n steps, or 'epoch'.",
    "no_hip_graph": "Launch every step eagerly instead of replaying a hipGraph.",
}
_SKIP = {"rag_model", "model", "on_step"}


def _cli_type(name: str, default):
    if name == "lr_scheduler_type":
        return "DALMSchedulerType", "DALMSchedulerType.LINEAR"
    if name in ("use_peft", "use_bnb") and not isinstance(default, bool):
        return "Optional[Mode]", "None"
    if name == "checkpointing_steps":
        return "Optional[str]", "None"
    if isinstance(default, bool):
        return "bool", repr(default)
    if isinstance(default, int):
        return "int", repr(default)
    if isinstance(default, float):
        return "float", repr(default)
    if isinstance(default, str):
        return "str", repr(default)
    if name in ("max_train_steps",):
        return "Optional[int]", "None"
    return "Optional[str]", "None"


def _make_command(fn, positional, cmd_name):
    sig = inspect.signature(fn)
    params, call = [], []
    for name in positional:  # positional CLI arguments, in the reference's order
        cli_name = "dataset_path" if name == "dataset_or_path" else name
        params.append(f'{cli_name}: str = typer.Argument(..., help={_HELP.get(name, name)!r}, show_default=False)')
        call.append(f"{name}={cli_name}")
    for name, p in sig.parameters.items():
        if name in positional or name in _SKIP:
            continue
        t, d = _cli_type(name, p.default)
        params.append(f"{name}: {t} = typer.Option({d}, help={_HELP.get(name, name.replace('_', ' '))!r})")
        if name == "lr_scheduler_type":
            call.append(f"{name}={name}.value")
        else:
            call.append(f"{name}={name}")
    src = f"def {cmd_name}(\n    " + ",\n    ".join(params) + f"\n) -> None:\n    _fn({', '.join(call)})\n"
    ns = {"typer": typer, "Optional": Optional, "Mode": Mode, "DALMSchedulerType": DALMSchedulerType, "_fn": fn}
    exec(src, ns)  # the signature typer introspects is built from the trainer's own signature
    ns[cmd_name].__doc__ = (fn.__doc__ or "").strip() or f"{cmd_name.replace('_', '-')} on MI355X"
    return ns[cmd_name]


@cli.command()
def version() -> None:
    """Print the version of this package."""
    from . import __version__

    print(f"dalm_amd version: {__version__}")


def _register() -> None:
    from .training.rag_e2e.train_rage2e import train_e2e
    from .training.retriever_only.train_retriever_only import train_retriever

    cli.command(name="train-rag-e2e")(_make_command(
        train_e2e, ["dataset_or_path", "retriever_name_or_path", "generator_name_or_path"], "train_rag_e2e"))
    cli.command(name="train-retriever-only")(_make_command(
        train_retriever, ["retriever_name_or_path", "dataset_or_path"], "train_retriever_only"))


_register()

if __name__ == "__main__":
    cli()
