"""ctypes binding of libomlm_hip.so (include/omlm.h).

The product path has NO CPU fallback: every op in ``ops.py`` goes through :func:`call`, which raises if the
shared library is missing or if a kernel reports an error.  PyTorch is used only for device memory, streams
and (in ``parallel.py``) torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OMLM_LIB_PATH") or os.path.join(_HERE, "libomlm_hip.so")      # override: A/B builds while tuning

_lib: Optional[C.CDLL] = None

vp, i32, i64, f32, u64 = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_ulonglong

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/omlm.h one to one
SIGNATURES = {
    "omlm_version": [],
    "omlm_last_error": [],
    "omlm_set_error": [C.c_char_p],
    "omlm_gemm": [vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp],
    "omlm_gemm_tail_workspace_bytes": [i32, i32],
    "omlm_gemm_wgrad_group": [vp, i32, i32, i32, vp],
    "omlm_gemm_planes": [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, i64, vp],
    "omlm_split_planes": [vp, vp, i64, i64, vp],
    "omlm_gemm_planes16": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp],
    "omlm_gemm_mx16": [vp, vp, i64, vp, vp, vp, i64, vp, vp, vp, i32, vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp],
    "omlm_gemm_mx16_workspace_bytes": [i32, i32, i32],
    "omlm_layernorm_fwd_mx": [vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, f32, vp],
    "omlm_ffmid_fwd_mx": [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, i32, f32, f32, u64, vp, vp, vp, vp],
    "omlm_quant_rows_mx": [vp, i32, vp],
    "omlm_layernorm_fwd_planes": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "omlm_ffmid_fwd_planes": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u64, vp, vp, vp, i32, vp],
    "omlm_layernorm_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "omlm_layernorm_bwd_workspace_bytes": [i32],
    "omlm_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp],
    "omlm_layernorm_bwd2": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp],
    "omlm_qk_norm_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "omlm_qk_norm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "omlm_gemm_qknorm": [vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp],
    "omlm_qk_norm_bwd2": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "omlm_attn_bias_table_floats": [i32, i32],
    "omlm_attn_bias_prepare": [vp, vp, i32, i32, i32, vp, vp, f32, f32, i32, vp],
    "omlm_attn_bias_prepare_group": [vp, vp, i32, i32, i32, i32, vp, vp, f32, f32, i32, vp],
    "omlm_mqa_attn_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp],
    "omlm_mqa_attn_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, i32, vp],
    "omlm_mqa_attn_bwd_workspace_bytes": [i32, i32, i32],
    "omlm_ffmid_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, u64, vp, vp, vp, i32, vp],
    "omlm_ffmid_bwd_workspace_bytes": [i32, i32],
    "omlm_ffmid_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp, vp, vp, i32, vp],
    "omlm_colsum_accumulate": [vp, vp, i32, i32, i32, vp],
    "omlm_colsum_group": [vp, i32, vp],
    "omlm_ffmid_set_impl": [i32],
    "omlm_embed_gather_fwd": [vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, vp, vp, vp, vp],
    "omlm_embed_gather_bwd": [vp, vp, vp, vp, vp, vp, i32, vp, i32, i32, i32, f32, vp, vp, vp, vp],
    "omlm_prepare_train_batch": [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, i32, vp],
    "omlm_cross_entropy_fwd": [vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "omlm_cross_entropy_bwd": [vp, vp, vp, vp, f32, vp, i32, i32, i32, i32, i32, vp],
    "omlm_sumsq_accumulate": [vp, i64, vp, vp, vp],
    "omlm_adamw_clip_step": [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp, f32, i32, i32, i32, vp, vp],
    "omlm_loss_scale_update": [vp, vp, f32, f32, i32, f32, f32, vp],
    "omlm_cast_pad": [vp, vp, i64, i32, i32, i32, i32, vp],
    "omlm_transpose_cast": [vp, vp, i32, i32, i32, i32, i32, vp],
    "omlm_cast_pad_group": [vp, i32, i32, vp],
    "omlm_sample_topk_gumbel_at": [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "omlm_sample_embed_at": [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp, i64, i64, vp, i32, vp],
    "omlm_decode_step": [vp, vp, vp],
    "omlm_decode_advance": [vp, vp, vp],
    "omlm_relpos_first_fwd": [vp, vp, vp, vp, i32, i32, vp],
    "omlm_relpos_first_bwd": [vp, vp, i32, i32, vp],
    "omlm_bias_silu_fwd": [vp, vp, vp, vp, i64, i32, vp],
    "omlm_silu_bwd": [vp, vp, vp, i64, vp],
    "omlm_bias_add": [vp, vp, vp, i32, i32, i32, vp],
    "omlm_relpos_mlp_fwd": [vp] * 15 + [i32, i32, i32, i32, vp],
    "omlm_relpos_mlp_bwd": [vp] * 19 + [i32, i32, i32, i32, vp],
    "omlm_rvq_encode": [vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "omlm_nearest_centroid": [vp, vp, vp, i32, i32, i32, vp],
    "omlm_rvq_encode_strided": [vp, vp, vp, i32, vp, i32, i32, i32, vp],
    "omlm_vq_accumulate": [vp, vp, i32, vp, vp, i32, i32, i32, vp],
    "omlm_vq_kmeans_update": [vp, vp, vp, vp, i32, i32, vp],
    "omlm_vq_ema_update": [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp],
    "omlm_sample_topk_gumbel": [vp, vp, vp, i32, i32, i32, i32, f32, i32, vp],
    "omlm_probe_tr16": [vp, vp],
}
_RESTYPES = {"omlm_last_error": C.c_char_p, "omlm_gemm_tail_workspace_bytes": C.c_longlong, "omlm_gemm_mx16_workspace_bytes": C.c_longlong, "omlm_ffmid_bwd_workspace_bytes": C.c_longlong,
             "omlm_attn_bias_table_floats": C.c_longlong, "omlm_mqa_attn_bwd_workspace_bytes": C.c_longlong,
             "omlm_layernorm_bwd_workspace_bytes": C.c_longlong, "omlm_set_error": None}


class HipLibraryMissing(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C open_musiclm_amd/csrc`). There is no CPU fallback for the MI355X hot path.")
        l = C.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)            # AttributeError here = header / library drift
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        _lib = l
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return t.data_ptr()


def call(name: str, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        msg = lib().omlm_last_error()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg.decode() if msg else '?'}")


def require_gpu(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(
            f"open_musiclm_amd: {what} is on {t.device}; the TokenConditionedTransformer path only runs on an "
            "MI355X through libomlm_hip.so (no CPU fallback). Move the model and inputs to 'cuda'.")
    cur = torch.cuda.current_device()
    if t.device.index != cur:
        # kernels are launched on the current device's stream with raw pointers: a tensor of another GPU would be read through
        # the wrong context (one process drives ONE GPU; under torchrun that is cuda:LOCAL_RANK)
        raise RuntimeError(f"open_musiclm_amd: {what} lives on cuda:{t.device.index} but the current device is cuda:{cur}; "
                           "call torch.cuda.set_device(LOCAL_RANK) and keep the model and its inputs on that device")
