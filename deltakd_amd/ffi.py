"""ctypes binding of libdkd.so (C ABI: include/dkd.h).

The library is the product: there is NO fallback.  ``lib()`` raises if the shared object is missing, and every wrapper
raises if handed a non-GPU tensor -- a silent eager/CPU path would void every parity claim.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DKD_LIB") or os.path.join(_HERE, "lib", "libdkd.so")     # DKD_LIB: A/B another build of the same ABI

EPI_BIAS, EPI_GELU, EPI_DGELU, EPI_RESID = 1, 2, 4, 8
EPI_OUT_F32, EPI_TAP_F32, EPI_RELU, EPI_ACCUM, EPI_RELU_GATE = 16, 32, 64, 128, 256


class RowMap(C.Structure):
    _fields_ = [("rpg", C.c_int32), ("gstride", C.c_int32), ("off", C.c_int32)]


IDENT = RowMap(0, 0, 0)


def strip_map(tokens_per_sample: int, n_prefix: int) -> RowMap:
    """rows of x[:, n_prefix:] viewed as [B*(tokens-n_prefix), D] inside a contiguous [B*tokens, D] buffer."""
    return RowMap(tokens_per_sample - n_prefix, tokens_per_sample, n_prefix)


class TnProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("a_colsum", C.c_void_p),
                ("M", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("amap", RowMap), ("bmap", RowMap)]


class Gemm(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
        ("amap", RowMap), ("cmap", RowMap),
        ("epi", C.c_uint32),
        ("bias", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int32), ("rmap", RowMap),
        ("rowscale", C.c_void_p), ("rows_per_sample", C.c_int32),
        ("preact", C.c_void_p), ("ldp", C.c_int32),
        ("tap", C.c_void_p), ("ldt", C.c_int32), ("conv_hw", C.c_int32),
        ("xb", C.c_void_p), ("ldxb", C.c_int32), ("rowstats", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_c", C.c_void_p), ("ln_eps", C.c_float),
    ]


class Block(C.Structure):
    """DkdBlock (include/dkd.h): one transformer block's parameters, shadows and activation buffers."""
    _fields_ = ([(n, C.c_int32) for n in ("B", "N", "D", "H", "hidden")] + [("eps", C.c_float)] +
                [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "ln2_w", "ln2_b", "qkv_b", "proj_b", "fc1_b", "fc2_b",
                                           "qkv_w", "proj_w", "fc1_w", "fc2_w", "qkv_wt", "proj_wt", "fc1_wt", "fc2_wt",
                                           "s1", "s2", "x", "x1", "x2", "y1", "qkv", "o", "y2", "pre", "h", "tap",
                                           "mean1", "rstd1", "mean2", "rstd2", "lse")] + [("fuse_mlp", C.c_int32), ("ln_fold", C.c_int32)] +
                [(n, C.c_void_p) for n in ("qkv_c", "fc1_c", "stats1", "stats2", "stats_next", "xb")] + [("ln1_ready", C.c_int32)] +
                [(n, C.c_void_p) for n in ("next_ln1_w", "next_ln1_b", "next_y1", "next_mean1", "next_rstd1")] + [("fuse_attn", C.c_int32)])


class BlockGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("g", "gtap", "d_ln1_w", "d_ln1_b", "d_ln2_w", "d_ln2_b", "d_qkv_w", "d_qkv_b",
                                          "d_proj_w", "d_proj_b", "d_fc1_w", "d_fc1_b", "d_fc2_w", "d_fc2_b", "dF", "dH", "dqkv", "dT",
                                          "ln_ws", "dF2", "ln_ws2", "ln_defer")] + [("defer_wgrad", C.c_int32)]


class LnReduce(C.Structure):
    _fields_ = [("part", C.c_void_p), ("nblk", C.c_int32), ("D", C.c_int32), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


_lib = None
_lock = threading.Lock()

_SIGS = {
    "dkd_version": (C.c_int, []),
    "dkd_last_error": (C.c_char_p, []),
    "dkd_device_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "dkd_gemm_nt": (C.c_int, [C.POINTER(Gemm), C.c_void_p]),
    "dkd_gemm_tn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                              C.c_int32, RowMap, RowMap, C.c_void_p, C.c_void_p]),
    "dkd_gemm_tn_group": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dkd_attn192_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_attn192_bwd": (C.c_int, [C.c_void_p] * 15 + [C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_attn192_fwd_proj": (C.c_int, [C.c_void_p] * 11 + [C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_block_wgrad_group": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dkd_ln_bwd_reduce_group": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "dkd_conv3x3_wgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_gram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, RowMap, C.c_void_p]),
    "dkd_attn_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_attn_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p]),
    "dkd_layernorm_fwd": (C.c_int, [C.c_void_p, C.c_int32, RowMap, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "dkd_layernorm_bwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, RowMap, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, RowMap, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p]),
    "dkd_gemm_nt_lnbwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_void_p]),
    "dkd_mlp192_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_mlp192_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_im2col_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_prefix_tokens_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_embed_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_scale_cast_bf16": (C.c_int, [C.c_void_p, C.c_int32, RowMap, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_cast_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_cast_weight_group": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_colsum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, RowMap, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_add_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, RowMap, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p]),
    "dkd_mixup": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                            C.c_int32, C.c_void_p]),
    "dkd_mixup_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_mixup_to_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_mixup_targets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    "dkd_ema_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "dkd_topk_correct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "dkd_logit_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_float,
                                 C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_mse_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, RowMap, C.c_void_p, C.c_float,
                               C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_mask_select": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_mask_select_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_sort_l1_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, RowMap, C.c_float, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_im2col3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_col2im3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_normalize_mse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p]),
    "dkd_diffkd_prepare": (C.c_int, [C.c_void_p, C.c_int32, RowMap, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_dropout_mse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int64,
                                  C.c_void_p]),
    "dkd_saliency_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_probe_begin": (C.c_int, []),
    "dkd_probe_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "dkd_probe_end_ex": (C.c_int, [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "dkd_blocks_fwd": (C.c_int, [C.POINTER(Block), C.c_int32, C.c_void_p]),
    "dkd_block_bwd": (C.c_int, [C.POINTER(Block), C.POINTER(BlockGrads), C.c_void_p]),
    "dkd_layernorm_bwd_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dkd_block_fwd_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dkd_block_bwd_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dkd_block_bwd_workspace_carve": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(BlockGrads)]),
    "dkd_jacobi_eigh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_lowrank_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dkd_lowrank_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "dkd_gram_batched": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, RowMap,
                                   C.c_void_p]),
    "dkd_scale_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dkd_cast_pad_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dkd_droppath_scales": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]),
    "dkd_lowrank_chain_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dkd_lowrank_chain_zero_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "dkd_lowrank_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "dkd_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_int32, C.c_float, C.c_void_p]),
    "dkd_adamw_step_gated": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)


def lib():
    """Load libdkd.so (once).  Raises RuntimeError when it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  deltakd_amd has no CPU/eager fallback.")
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in _SIGS.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().dkd_last_error().decode()
        if rc == -1:
            raise ValueError(f"libdkd {what}: {msg}")
        raise RuntimeError(f"libdkd {what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Refuses host tensors: the kernels dereference device memory."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("deltakd_amd ops need tensors on an MI355X (cuda) device; there is no CPU path")
    return t.data_ptr()


def stream():
    """Raw hipStream_t of torch's current stream on the current device (the fast accessor: this runs once per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
