"""ctypes binding of the C-ABI library ``csrc/libimmunostruct_hip.so``.

The library is the product: if it is missing, or a tensor is not a contiguous
fp32/int32 ROCm tensor, the call raises -- there is NO CPU or eager-PyTorch
fallback anywhere in ``immunostruct_amd`` (the CPU oracle lives in ``oracle/``
and is test infrastructure only).

Entry points are declared in ``include/immunostruct_hip.h``; every function
returns 0 on success or a negative errno-style code, never throws, never
allocates and never owns memory: the caller (PyTorch) owns every buffer and
passes the HIP stream the kernels are enqueued on.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IMMUNOSTRUCT_LIB") or os.path.join(_HERE, "csrc", "libimmunostruct_hip.so")   # override: A/B builds

_P, _I, _F, _LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

# symbol -> argtypes ; keep in sync with include/immunostruct_hip.h
_RETURNS_LONGLONG = {"is_attn_colmean_probs_floats", "is_contrastive_scratch_floats", "is_contrastive_work_floats",
                     "is_linear_dgrad_scratch_floats"}
SIGNATURES = {
    "is_version": [],
    "is_last_error_string": [],
    "is_mfma_selftest": [_P, _P, _P, _P],
    "is_mfma_outer_selftest": [_P, _P, _P, _P],
    "is_egnn_layer_fwd": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I,
                          _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "is_egnn_layer_bwd": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I] + [_P] * 10 + [_I, _P, _P, _I, _I, _I] + [_P] * 18 + [_P],
    "is_egnn_layer_bwd_paired": [_P, _P, _I, _P, _P, _P, _P, _P, _I, _I] + [_P] * 10 + [_I, _P, _P, _I, _I, _I] + [_P] * 18 + [_P],
    "is_egnn_layer_bwd_paired_supported": [_I],
    "is_layer_saves_m1": [],
    "is_layer_saves_geo": [],
    "is_node_proj_fwd": [_P, _I, _I, _P, _I, _P, _P, _P, _I, _P],
    "is_node_proj_bwd": [_P, _P, _P, _I, _I, _P, _I, _P, _P, _I, _I, _P],
    "is_reduce_partials_scratch_floats": [_I],
    "is_reduce_partials": [_P, _I, _I, _I, _P, _P, _P, _P],
    "is_node_pack_floats": [],
    "is_stack_prologue": [_P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P],
    "is_egnn_node_wgrad_stride": [],
    "is_egnn_node_wgrad_proj_floats": [],
    "is_egnn_node_wgrad_batched": [_P, _I, _I, _I, _I, _P],
    "is_reduce_partials_batched": [_P, _I, _P],
    "is_multi_copy": [_P, _I, _P],
    "is_step_random": [_P, _I, _P, _P],
    "is_batch_gather": [_P, _I, _I, _I, _I] + [_P] * 15 + [_P, _I, _P],
    "is_chunk_partition": [_P, _I, _I, _I, _P, _P],
    "is_adam_step": [_P, _I, _P, _P, _P],
    "is_adam_prepare": [_P, _P, _P],
    "is_adam_apply": [_P, _I, _P, _P, _P],
    "is_linear_wgrad": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P],
    "is_linear_dgrad_scratch_floats": [_I, _I, _I],
    "is_linear_fwd_long": [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "is_linear_dgrad": [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P],
    "is_contrastive_scratch_floats": [_I],
    "is_contrastive_work_floats": [_I],
    "is_contrastive_fwd": [_P, _P, _I, _I, _P, _P, _P, _P, _P, ctypes.c_float, _P, _P, _P, ctypes.c_float, _I, _P],
    "is_contrastive_bwd": [_P, _P, _P, _P, ctypes.c_float, _P, _P, _P, _P, ctypes.c_float, _P, _P, _I, _I, _I, _P],
    "is_contrastive_targets": [_P, _P, _P, _I, _P],
    "is_mlp2_fwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "is_mlp2_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "is_mlp2_bwd_records": [_I],
    "is_mlp2_bwd_record_floats": [_I, _I, _I],
    "is_debug_timestamp": [_P, _P],
    "is_debug_emulated_collective": [_P, _LL, _I, _I, _LL, _P, _P],
    "is_debug_stream_copy": [_P, _P, _LL, _I, _P],
    "is_gather_segment_sum": [_P, _P, _P, _P, _P, _I, _P, _I, _P, _P],
    "is_segment_pool_fwd": [_P, _I, _P, _P, _P, _I, _I, _P],
    "is_segment_pool_bwd": [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_attn_colmean_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_attn_colmean_fwd_tail": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "is_attn_colmean_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_attn_colmean_bwd_tail": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "is_attn_colmean_probs_floats": [_I, _I, _I],
    "is_comb_attn_stats_floats": [_I, _I],
    "is_comb_attn_partials_floats": [_I],
    "is_comb_attn_grad_floats": [_I],
    "is_comb_attn_fwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_comb_attn_bwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_comb_attn_cls_fwd": [_P, _I] + [_P] * 16 + [_I] * 6 + [_P],
    "is_comb_attn_cls_grad_floats": [_I, _I, _I],
    "is_comb_attn_cls_bwd": [_P, _I] + [_P] * 18 + [_I] * 6 + [_P],
    "is_loss_partials_floats": [],
    "is_vae_latent_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_vae_latent_grad_floats": [_I, _I],
    "is_vae_latent_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_vae_latent_bwd_data": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "is_vae_latent_bwd_wgrad": [_P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P],
    "is_recon_mse": [_P, _P, _P, _LL, _F, _P, _P],
    "is_vae_loss": [_P, _P, _P, _LL, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P],
}



class WgradLayer(ctypes.Structure):
    """one layer of is_egnn_node_wgrad_batched (mirrors `WgradLayer` in csrc/egnn_node16.hip)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("g_psd", "h_out", "dh", "zn1", "dzn1", "h", "h_neigh", "partials")] + \
               [(n, ctypes.c_int) for n in ("ld_h", "din", "ld_hn", "ld_ho", "dho", "pad")]


class NodePackJob(ctypes.Structure):
    """one layer of is_node_pack_weights (mirrors `NodePackJob` in csrc/egnn_node16.hip)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("Wn1", "Wn2", "W1n", "W1nb", "fpack", "bpack")] + \
               [(n, ctypes.c_int) for n in ("din", "ldw_n")]


class RowGather(ctypes.Structure):
    """one per-sample row gather of is_batch_gather (mirrors `RowGather` in csrc/segment_ops.hip)"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("floats", ctypes.c_int), ("pad", ctypes.c_int)]


class ReduceJob(ctypes.Structure):
    """one job of is_reduce_partials_batched (mirrors `ReduceJob` in csrc/egnn_node.hip)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ("partials", "map", "dst", "scratch")] + \
               [(n, ctypes.c_int) for n in ("nparts", "stride", "count", "pad")]


class CaPart(ctypes.Structure):
    """one piece of the combined attention's token row (mirrors `CaPart` in csrc/combined_attention.hip)"""
    _fields_ = [("x", ctypes.c_void_p), ("dx", ctypes.c_void_p), ("width", ctypes.c_int), ("ld", ctypes.c_int)]


class RandJob(ctypes.Structure):
    """one job of is_step_random"""
    _fields_ = [("out", ctypes.c_void_p), ("n", ctypes.c_longlong), ("kind", ctypes.c_int), ("p", ctypes.c_float)]


class CopyJob(ctypes.Structure):
    """one job of is_multi_copy"""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_longlong)]


_lib = None


class HipExtensionError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: build it with `make -C immunostruct_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "immunostruct_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name in _RETURNS_LONGLONG else (ctypes.c_char_p if name == "is_last_error_string" else _I)
    _lib = lib
    return lib


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (``None`` -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HipExtensionError(
                "immunostruct_amd kernels run on a ROCm device only (got a CPU tensor); "
                "there is no CPU fallback -- move the model and the batch to 'cuda'.")


def f32c(t):
    """fp32, contiguous (copies only when needed)."""
    if t.dtype != torch.float32:
        raise ValueError(f"expected float32 tensor, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def rows_ld(t):
    """Return (tensor, leading dimension) for a 2-d fp32 tensor with unit column stride."""
    if t.dtype != torch.float32:
        raise ValueError(f"expected float32 tensor, got {t.dtype}")
    if t.dim() != 2:
        raise ValueError("expected a 2-d tensor")
    if t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t, int(t.stride(0))


def check(code, what):
    if code != 0:
        detail = load().is_last_error_string().decode("utf-8", "replace")      # (same thread: the failing call was just made)
        raise HipExtensionError(f"{what} failed with code {code}" + (f": {detail}" if detail else ""))
