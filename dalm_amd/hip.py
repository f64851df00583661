"""ctypes binding of libdalm_hip.so (the C ABI in include/dalm_hip.h).

This module is the only place that touches the shared object.  There is NO CPU
fallback: if the library is missing, or a tensor is not on a HIP device, the
call raises.  torch is used for device memory and the current stream only.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import torch

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libdalm_hip.so"
_lib: Optional[C.CDLL] = None

F32, BF16 = 0, 1
_i64, _f32, _int, _vp, _sz = C.c_int64, C.c_float, C.c_int, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/dalm_hip.h one to one
SIGNATURES = {
    "dalm_version": (_int, []),
    "dalm_last_error_string": (C.c_char_p, []),
    "dalm_pool_l2norm_fwd": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "dalm_pool_l2norm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _int, _vp]),
    "dalm_sim_matmul": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _i64, _vp]),
    "dalm_gemm_f32": (_int, [_int, _int, _i64, _i64, _i64, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
    "dalm_sim_rowstats_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dalm_sim_rowstats": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dalm_sim_rowstats_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dalm_sim_rowstats_bf16x3_supported": (_int, [_i64, _i64, _i64]),
    "dalm_sim_rowstats_bf16x3_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dalm_sim_rowstats_bf16x3": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dalm_sim_grad_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dalm_sim_grad": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dalm_nt_xent_fwd": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp]),
    "dalm_nt_xent_bwd": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _vp]),
    "dalm_marg_ce_prep": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "dalm_marg_ce_fwd": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_marg_ce_bwd": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_scale_inplace": (_int, [_vp, _int, _i64, _vp, _vp]),
    "dalm_marg_ce_finalize": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dalm_marg_ce_finalize_topk": (_int, [_vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_marg_ce_bwd_weighted": (_int, [_vp, _int, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_doc_scores_topk_fwd": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _vp, _vp, _vp]),
    "dalm_doc_scores_topk_bwd": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_doc_logprob_fwd": (_int, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "dalm_doc_logprob_bwd": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _int, _vp]),
    "dalm_gather_nll": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "dalm_marginalize_rows": (_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "dalm_marginalize_rows_dev": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "dalm_contrastive_finalize": (_int, [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "dalm_pool_l2norm_fwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "dalm_pool_l2norm_fwd_ws": (_int, [_vp, _int, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dalm_sim_small_supported": (_int, [_i64, _i64, _i64]),
    "dalm_sim_small_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "dalm_sim_small_fwd": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dalm_sim_small_fwd1_preferred": (_int, [_i64, _i64, _i64]),
    "dalm_sim_small_fwd1_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "dalm_sim_small_fwd1_ticket_words": (_sz, [_i64, _i64]),
    "dalm_sim_small_fwd1": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "dalm_sim_small_bwd": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_sim_small_bwd_workspace_bytes": (_sz, [_i64, _i64, _i64, _int, _int]),
    "dalm_sim_small_bwd1_workspace_bytes": (_sz, [_i64, _i64, _i64, _int, _int]),
    "dalm_sim_small_bwd1_ticket_words": (_sz, [_i64, _i64, _i64]),
    "dalm_sim_small_bwd1": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "dalm_sim_small_bwd_ws": (_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dalm_rag_loss_finalize": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "dalm_sim_topk_supported": (_int, [_i64, _i64]),
    "dalm_sim_topk_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64]),
    "dalm_sim_topk": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dalm_lm_head_lse_workspace_bytes": (_sz, [_i64, _i64]),
    "dalm_lm_head_lse_fwd": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _sz, _vp]),
    "dalm_comm_unique_id": (_int, [_vp]),
    "dalm_comm_init": (_int, [C.POINTER(_vp), _vp, _int, _int, _int]),
    "dalm_comm_destroy": (_int, [_vp]),
    "dalm_comm_rank": (_int, [_vp]),
    "dalm_comm_world": (_int, [_vp]),
    "dalm_comm_wait_stream": (_int, [_vp, _vp]),
    "dalm_comm_stream_wait": (_int, [_vp, _vp]),
    "dalm_comm_allgather": (_int, [_vp, _vp, _vp, _sz]),
    "dalm_comm_allreduce_sum_f32": (_int, [_vp, _vp, _sz]),
    "dalm_comm_allgather_on": (_int, [_vp, _vp, _vp, _sz, _vp]),
    "dalm_comm_allreduce_sum_f32_on": (_int, [_vp, _vp, _sz, _vp]),
    "dalm_nf4_packed_bytes": (_sz, [_i64]),
    "dalm_nf4_absmax_count": (_sz, [_i64]),
    "dalm_nf4_quantize": (_int, [_vp, _int, _i64, _vp, _vp, _vp]),
    "dalm_nf4_dequantize": (_int, [_vp, _vp, _i64, _int, _vp, _vp]),
    "dalm_rope_qk": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "dalm_swiglu_fwd": (_int, [_vp, _vp, _vp, _int, _i64, _vp]),
    "dalm_swiglu_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp]),
    "dalm_swiglu_fwd_2d": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _vp]),
    "dalm_swiglu_bwd_2d": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    "dalm_rms_norm_fwd": (_int, [_vp, _vp, _vp, _int, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "dalm_rms_norm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _i64, _vp, _vp]),
    "dalm_layer_norm_fwd": (_int, [_vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp, _vp]),
    "dalm_layer_norm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "dalm_gelu_fwd": (_int, [_vp, _vp, _i64, _vp]),
    "dalm_gelu_bwd": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "dalm_add3": (_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "dalm_bert_add_norm_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _i64, _i64, _f32, _f32, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dalm_bert_add_norm_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _i64, _i64, _f32, _vp, _vp, _vp]),
    "dalm_attn_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _f32, _vp, C.c_uint32, _vp, _vp, _vp]),
    "dalm_attn_mask_bits": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "dalm_attn_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _i64, _i64,
                             _f32, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "dalm_attn_mask_bits_packed": (_int, [_vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp]),
    "dalm_attn_fwd_packed": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _f32, _vp, C.c_uint32, _vp, _vp, _vp]),
    "dalm_attn_bwd_packed": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _f32, _vp, _vp, _vp, _i64,
                                    _f32, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "dalm_lora_rowdot": (_int, [_vp, _int, _vp, _int, _i64, _i64, _int, _f32, _f32, _vp, C.c_uint32, _vp, _vp]),
    "dalm_lora_rankupd": (_int, [_vp, _int, _vp, _vp, _int, _i64, _i64, _int, _f32, _f32, _vp, C.c_uint32, _vp]),
    "dalm_lora_colacc_workspace_bytes": (_sz, [_i64, _i64, _int]),
    "dalm_lora_colacc": (_int, [_vp, _int, _vp, _i64, _i64, _int, _f32, _f32, _vp, C.c_uint32, _vp, _int, _vp, _sz, _vp]),
    "dalm_lora2_rowdot": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f32, _f32, _vp, C.c_uint32, C.c_uint32,
                                 _int, _vp]),
    "dalm_lora2_rankupd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f32, _int, _vp]),
    "dalm_lora2_colacc_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "dalm_lora2_colacc_ticket_words": (_sz, [_i64, _int]),
    "dalm_sim_grad_bf16x3_supported": (_int, [_i64, _i64, _i64]),
    "dalm_sim_grad_bf16x3_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "dalm_sim_grad_bf16x3": (_int, [_vp, _vp, _i64, _i64, _i64, _f32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dalm_lm_head_dlogits": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dalm_lm_head_logits": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dalm_lm_head_dhidden": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _int, _vp]),
    "dalm_transpose_bf16": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp]),
    "dalm_f32_to_bf16": (_int, [_vp, _vp, _i64, _vp]),
    "dalm_lora2_colacc": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _f32, _int, _vp, _sz, _vp, _vp]),
}


def library_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load (once) and return the shared library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m dalm_amd._build` (needs hipcc, targets gfx950)."
        )
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class DalmHipError(RuntimeError):
    pass


def _check(rc: int, name: str) -> None:
    if rc == 0:
        return
    msg = load().dalm_last_error_string().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError(f"{name} rejected its arguments (code {rc}): {msg}")
    raise DalmHipError(f"{name} failed with hipError {rc}: {msg}")


def call(name: str, *args) -> None:
    _check(getattr(load(), name)(*args), name)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on the same HIP device; returns it."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "dalm_amd: tensor is on %s; the loss path runs only as HIP kernels on an MI355X "
                "(there is no CPU implementation in this package)" % t.device
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"dalm_amd: tensors on different devices ({dev} vs {t.device})")
    if dev is None:
        raise RuntimeError("dalm_amd: no tensor arguments")
    return dev


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"dalm_amd: unsupported dtype {t.dtype} (float32 or bfloat16 expected)")


def as_i64(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.int64 and t.is_contiguous()) else t.to(torch.int64).contiguous()


def as_f32c(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
