"""Tensor-level wrappers over the C ABI (one Python function per ``dkd_*`` entry point).

No arithmetic happens here: the functions only check shapes/dtypes, pass raw device pointers + sizes to libdkd.so on
torch's current HIP stream, and return the output tensors they allocated.
"""
import ctypes as C

import torch

from . import ffi
from .ffi import (EPI_ACCUM, EPI_BIAS, EPI_DGELU, EPI_GELU, EPI_OUT_F32, EPI_RELU, EPI_RELU_GATE, EPI_RESID, EPI_TAP_F32, IDENT, RowMap,
                  check, lib, ptr, stream)

BF16, F32 = torch.bfloat16, torch.float32


def _is_f32(t):
    if t.dtype == F32:
        return 1
    if t.dtype == BF16:
        return 0
    raise TypeError(f"expected bf16 or f32 tensor, got {t.dtype}")


def gemm_nt(a, b, out=None, *, M=None, bias=None, gelu=False, dgelu=False, relu=False, preact=None, resid=None, rowscale=None,
            rows_per_sample=0, tap=None, out_f32=False, accumulate=False, amap=IDENT, cmap=IDENT, rmap=IDENT, out_rows=None,
            K=None, N=None, conv_hw=0, relu_gate=None, xb=None, rowstats=None, ln_stats=None, ln_c=None, ln_eps=1e-6):
    """out[M, N] = epilogue(a[M, K] @ b[N, K]^T); a, b bf16 (2-D, row stride = stride(0)).

    ``M`` = logical rows (defaults to a.shape[0]; with ``amap`` the rows are gathered through the map).
    ``out_rows`` = rows of the allocated output when ``cmap`` scatters into a larger buffer.
    ``conv_hw`` > 0: a is the activation [B * hw * hw, Cin] of a 3 x 3 / pad 1 convolution on the hw x hw token grid, b its weight as
    [N, (ky, kx, cin)] (K = 9 Cin): implicit GEMM, the neighbourhood is gathered by the kernel (include/dkd.h, DkdGemm.conv_hw).
    ``relu_gate`` (bf16 [M, N]): out = gate > 0 ? out : 0 -- the backward of a ReLU given its output.
    LayerNorm fold (include/dkd.h, DkdGemm.xb): ``xb`` (bf16 [M, N]) + ``rowstats`` (f32 [M, 2], zeroed by the caller) on the f32-residual
    GEMM that produces x; ``ln_stats`` (those sums) + ``ln_c`` (f32 [N] row sums of b = bf16(gamma * W)) on the Linear behind the norm.
    """
    if conv_hw:
        K = 9 * a.shape[1]
    assert a.dtype == BF16 and b.dtype == BF16 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1
    M = a.shape[0] if M is None else M
    K = a.shape[1] if K is None else K
    N = b.shape[0] if N is None else N
    if out is None:
        out = torch.empty(out_rows or M, N, device=a.device, dtype=F32 if out_f32 else BF16)
    else:
        out_f32 = out.dtype == F32
    g = ffi.Gemm()
    g.A, g.B, g.C = ptr(a), ptr(b), ptr(out)
    g.M, g.N, g.K = M, N, K
    g.lda, g.ldb, g.ldc = a.stride(0), b.stride(0), out.stride(0)
    g.amap, g.cmap, g.rmap = amap, cmap, rmap
    epi = 0
    if bias is not None:
        assert bias.dtype == F32 and bias.numel() >= N
        epi |= EPI_BIAS
        g.bias = ptr(bias)
    if gelu:
        epi |= EPI_GELU
    if dgelu:
        epi |= EPI_DGELU
    if relu:
        epi |= EPI_RELU
    if relu_gate is not None:
        assert preact is None and relu_gate.dtype == BF16
        epi |= EPI_RELU_GATE
        preact = relu_gate
    g.conv_hw = conv_hw
    if preact is not None:
        assert preact.dtype == BF16
        g.preact, g.ldp = ptr(preact), preact.stride(0)
    if resid is not None:
        assert resid.dtype == F32
        epi |= EPI_RESID
        g.resid, g.ldr = ptr(resid), resid.stride(0)
    if rowscale is not None:
        assert rowscale.dtype == F32
        g.rowscale, g.rows_per_sample = ptr(rowscale), rows_per_sample
    if tap is not None:
        g.tap, g.ldt = ptr(tap), tap.stride(0)
        if tap.dtype == F32:
            epi |= EPI_TAP_F32
    if out_f32:
        epi |= EPI_OUT_F32
    if accumulate:
        epi |= EPI_ACCUM
    if xb is not None:
        assert xb.dtype == BF16 and rowstats.dtype == F32 and rowstats.is_contiguous()
        g.xb, g.ldxb, g.rowstats = ptr(xb), xb.stride(0), ptr(rowstats)
    if ln_stats is not None:
        assert ln_stats.dtype == F32 and ln_c.dtype == F32 and ln_stats.is_contiguous()
        g.ln_stats, g.ln_c, g.ln_eps = ptr(ln_stats), ptr(ln_c), ln_eps
    g.epi = epi
    check(lib().dkd_gemm_nt(C.byref(g), stream()), "gemm_nt")
    return out


def gemm_tn(a, b, out, *, M=None, N1=None, N2=None, amap=IDENT, bmap=IDENT, colsum=None):
    """out[N1, N2] (f32) += a[M, N1]^T @ b[M, N2]   (a, b bf16; rows through amap / bmap).
    ``colsum`` (f32 [N1], optional) += column sums of a: the bias gradient, fused into the same pass."""
    assert a.dtype == BF16 and b.dtype == BF16 and out.dtype == F32
    M = a.shape[0] if M is None else M
    N1 = a.shape[1] if N1 is None else N1
    N2 = b.shape[1] if N2 is None else N2
    check(lib().dkd_gemm_tn(ptr(a), ptr(b), ptr(out), M, N1, N2, a.stride(0), b.stride(0), out.stride(0), amap, bmap, ptr(colsum),
                            stream()), "gemm_tn")
    return out


def gram(a, out, *, M=None, amap=IDENT, mirror=True):
    """out[N, N] (f32) += a[M, N]^T a[M, N], computed on the 128 x 128 tile pairs of the upper triangle; ``mirror``: fill the tiles below
    the diagonal from their transposes (they must be zero on entry) -- ``lowrank_step`` reads the upper tiles only and does not need it."""
    assert a.dtype == BF16 and out.dtype == F32 and out.shape[0] == out.shape[1] == a.shape[1]
    N = a.shape[1]
    check(lib().dkd_gram(ptr(a), ptr(out), a.shape[0] if M is None else M, N, a.stride(0), out.stride(0), amap, stream()), "gram")
    if N > 128 and mirror:
        blk = torch.arange(N, device=out.device) // 128
        lower = blk[:, None] > blk[None, :]                  # tiles strictly below the diagonal
        out.copy_(torch.where(lower, out.t(), out))
    return out


def gram_batched(a0, out, L, stride_a, *, M, amap=IDENT):
    """out[l] (f32 [L, N, N], zero on entry) += A_l^T A_l for A_l = the bf16 [rows, N] matrix ``stride_a`` elements behind A_{l-1} (a0 = A_0):
    the upper 128 x 128 tile pairs of all L matrices in one launch (include/dkd.h: dkd_gram_batched)."""
    assert a0.dtype == BF16 and out.dtype == F32 and out.is_contiguous() and out.shape == (L, a0.shape[1], a0.shape[1])
    N = a0.shape[1]
    check(lib().dkd_gram_batched(ptr(a0), stride_a, ptr(out), N * N, L, M, N, a0.stride(0), N, amap, stream()), "gram_batched")
    return out


def gemm_tn_group(problems):
    """Up to four ``gemm_tn`` problems in one launch.  problems: sequence of dicts with the arguments of ``gemm_tn``
    (a, b, out and optionally M, N1, N2, amap, bmap, colsum)."""
    from .ffi import TnProblem
    arr = (TnProblem * len(problems))()
    for q, kw in zip(arr, problems):
        a, b, out = kw["a"], kw["b"], kw["out"]
        assert a.dtype == BF16 and b.dtype == BF16 and out.dtype == F32
        q.A, q.B, q.C, q.a_colsum = ptr(a), ptr(b), ptr(out), ptr(kw.get("colsum"))
        q.M = kw.get("M", a.shape[0])
        q.N1, q.N2 = kw.get("N1", a.shape[1]), kw.get("N2", b.shape[1])
        q.lda, q.ldb, q.ldc = a.stride(0), b.stride(0), out.stride(0)
        q.amap, q.bmap = kw.get("amap", IDENT), kw.get("bmap", IDENT)
    import ctypes
    check(lib().dkd_gemm_tn_group(ctypes.cast(arr, ctypes.c_void_p), len(problems), stream()), "gemm_tn_group")


def attn_fwd(qkv, B, N, H, need_lse=True):
    """qkv bf16 [B*N, 3*H*64] -> (out bf16 [B*N, H*64], lse f32 [B, H, N] | None)."""
    assert qkv.dtype == BF16 and qkv.is_contiguous() and qkv.numel() == B * N * 3 * H * 64
    out = torch.empty(B * N, H * 64, device=qkv.device, dtype=BF16)
    lse = torch.empty(B, H, N, device=qkv.device, dtype=F32) if need_lse else None
    check(lib().dkd_attn_fwd(ptr(qkv), ptr(out), ptr(lse), B, N, H, stream()), "attn_fwd")
    return out, lse


def attn192_fwd(y1, wqkv, bqkv, B, N, need_lse=True):
    """y1 bf16 [B*N, 192], wqkv bf16 [576, 192], bqkv f32 [576] -> (qkv bf16 [B*N, 576], out bf16 [B*N, 192], lse f32 [B, 3, N] | None)."""
    assert y1.dtype == BF16 and y1.is_contiguous() and wqkv.dtype == BF16 and wqkv.is_contiguous() and bqkv.dtype == F32
    qkv = torch.empty(B * N, 576, device=y1.device, dtype=BF16)
    out = torch.empty(B * N, 192, device=y1.device, dtype=BF16)
    lse = torch.empty(B, 3, N, device=y1.device, dtype=F32) if need_lse else None
    check(lib().dkd_attn192_fwd(ptr(y1), ptr(wqkv), ptr(bqkv), ptr(qkv), ptr(out), ptr(lse), B, N, stream()), "attn192_fwd")
    return qkv, out, lse


def attn192_fwd_proj(y1, wqkv, bqkv, proj_w, proj_b, x, B, N, rowscale=None, need_lse=True):
    """attn192_fwd carried through proj and the residual: -> (qkv, out, lse, x1 f32 [B*N, 192] = x + rowscale[b] (out proj_w^T + proj_b))."""
    assert y1.dtype == BF16 and y1.is_contiguous() and wqkv.dtype == BF16 and wqkv.is_contiguous() and bqkv.dtype == F32
    assert proj_w.dtype == BF16 and proj_w.is_contiguous() and proj_w.shape == (192, 192) and proj_b.dtype == F32 and x.dtype == F32 and x.is_contiguous()
    qkv = torch.empty(B * N, 576, device=y1.device, dtype=BF16)
    out = torch.empty(B * N, 192, device=y1.device, dtype=BF16)
    lse = torch.empty(B, 3, N, device=y1.device, dtype=F32) if need_lse else None
    x1 = torch.empty_like(x)
    check(lib().dkd_attn192_fwd_proj(ptr(y1), ptr(wqkv), ptr(bqkv), ptr(qkv), ptr(out), ptr(lse), ptr(proj_w), ptr(proj_b), ptr(x), ptr(rowscale),
                                     ptr(x1), B, N, stream()), "attn192_fwd_proj")
    return qkv, out, lse, x1


def attn192_bwd(dy, proj_wt, qkv, out, lse, B, N, *, qkv_wt=None, x=None, ln_w=None, mean=None, rstd=None, g=None, d_ln_w=None, d_ln_b=None):
    """dy bf16 [B*N, 192] (gradient w.r.t. proj's output), proj_wt bf16 [192, 192] = proj.weight^T, qkv / out / lse as attn192_fwd returned
    them -> dqkv bf16 [B*N, 576]: proj dgrad + attention backward in one launch (dO never leaves the chip).  With ``qkv_wt`` (bf16 [192, 576] =
    qkv.weight^T) the same launch also runs the qkv dgrad and norm1's backward: g f32 [B*N, 192] += LN'(dqkv Wqkv), d_ln_w / d_ln_b +=."""
    assert dy.dtype == BF16 and dy.is_contiguous() and proj_wt.dtype == BF16 and proj_wt.is_contiguous() and proj_wt.shape == (192, 192)
    assert qkv.dtype == BF16 and qkv.is_contiguous() and out.dtype == BF16 and out.is_contiguous() and lse.dtype == F32
    dqkv = torch.empty(B * N, 576, device=dy.device, dtype=BF16)
    ws = None
    if qkv_wt is not None:
        assert qkv_wt.dtype == BF16 and qkv_wt.is_contiguous() and qkv_wt.shape == (192, 576)
        for t in (x, ln_w, mean, rstd, g, d_ln_w, d_ln_b):
            assert t is not None and t.dtype == F32 and t.is_contiguous()
        ws = torch.empty(lib().dkd_layernorm_bwd_workspace_bytes(B * N, 192) // 4, device=dy.device, dtype=F32)
    check(lib().dkd_attn192_bwd(ptr(dy), ptr(proj_wt), ptr(qkv), ptr(out), ptr(lse), ptr(dqkv), ptr(qkv_wt), ptr(x), ptr(ln_w), ptr(mean),
                                ptr(rstd), ptr(g), ptr(d_ln_w), ptr(d_ln_b), ptr(ws), B, N, stream()), "attn192_bwd")
    return dqkv


def attn_bwd(qkv, out, dout, lse, B, N, H):
    assert dout.dtype == BF16 and dout.is_contiguous() and out.is_contiguous() and qkv.is_contiguous()
    dqkv = torch.empty_like(qkv)
    check(lib().dkd_attn_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), B, N, H, stream()), "attn_bwd")
    return dqkv


def layernorm_fwd(x, gamma, beta, *, M=None, xmap=IDENT, eps=1e-6, save_stats=True, out_f32=False):
    """x f32 [rows, D] -> y bf16 [M, D] (+ mean, rstd f32 [M])."""
    assert x.dtype == F32 and x.dim() == 2 and x.stride(1) == 1
    M = x.shape[0] if M is None else M
    D = x.shape[1]
    y = torch.empty(M, D, device=x.device, dtype=F32 if out_f32 else BF16)
    mean = torch.empty(M, device=x.device, dtype=F32) if save_stats else None
    rstd = torch.empty(M, device=x.device, dtype=F32) if save_stats else None
    check(lib().dkd_layernorm_fwd(ptr(x), x.stride(0), xmap, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, D, eps,
                                  int(out_f32), stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, *, M=None, xmap=IDENT, dxmap=IDENT, accumulate=False, ws=None):
    """dx (f32, in place) (+)= LN'(dy); dgamma/dbeta (f32 [D]) += .  ws: optional f32 scratch of >= 2 * D * ceil(M / 64) elements
    (partial sums per block instead of same-address atomics)."""
    M = dy.shape[0] if M is None else M
    D = x.shape[1]
    assert dy.is_contiguous() and dy.shape[1] == D
    assert ws is None or (ws.dtype == F32 and ws.numel() >= 2 * D * ((M + 63) // 64))
    check(lib().dkd_layernorm_bwd(ptr(dy), _is_f32(dy), ptr(x), x.stride(0), xmap, ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                  dx.stride(0), dxmap, int(accumulate), ptr(dgamma), ptr(dbeta), M, D, ptr(ws), stream()), "layernorm_bwd")
    return dx


def gemm_nt_lnbwd(a, w, x, gamma, mean, rstd, dx, dgamma, dbeta, ws, *, cast_out=None, rowscale=None, rows_per_sample=0):
    """dx += LN'(a @ w^T) with LayerNorm width 192: the dgrad GEMM whose epilogue is the LayerNorm backward (include/dkd.h).
    a bf16 [M, K], w bf16 [192, K]; x, dx f32 [M, 192]; ws f32 scratch of layernorm_bwd's size; cast_out bf16 [M, 192] optional."""
    assert a.dtype == BF16 and w.dtype == BF16 and w.shape[0] == 192 and x.dtype == F32 and dx.dtype == F32 and ws.dtype == F32
    M, K = a.shape
    check(lib().dkd_gemm_nt_lnbwd(ptr(a), ptr(w), M, K, a.stride(0), w.stride(0), ptr(x), x.stride(0), ptr(gamma), ptr(mean), ptr(rstd),
                                  ptr(dx), dx.stride(0), ptr(dgamma), ptr(dbeta), ptr(ws), ptr(cast_out), ptr(rowscale), rows_per_sample,
                                  stream()), "gemm_nt_lnbwd")
    return dx


def mlp192_fwd(x1, ln_w, ln_b, fc1_w, fc1_b, fc2_wt, fc2_b, *, eps=1e-6, rowscale=None, rows_per_sample=0, want_tap=False, save=True,
               out=None, next_ln=None):
    """The fused MLP branch of a D = 192 block (include/dkd.h, dkd_mlp192_fwd).  x1 f32 [M, 192]; fc1_w / fc2_wt bf16 [hidden, 192].
    -> dict(x2, tap, y2, pre, h, mean, rstd): y2 / h padded to a multiple of 16 rows, ``pre`` opaque (fragment-native, for mlp192_bwd).
    ``next_ln`` = (weight, bias) of the next block's norm1: also returns next_y (bf16 LayerNorm of x2), next_mean, next_rstd."""
    M, D = x1.shape
    Hd = fc1_w.shape[0]
    assert D == 192 and x1.dtype == F32 and x1.is_contiguous() and fc1_w.dtype == BF16 and fc2_wt.dtype == BF16
    assert fc1_w.shape == (Hd, D) and fc2_wt.shape == (Hd, D) and fc1_w.is_contiguous() and fc2_wt.is_contiguous()
    dev, Mp = x1.device, (M + 15) // 16 * 16
    r = {"x2": torch.empty_like(x1) if out is None else out, "tap": torch.empty(M, D, device=dev, dtype=BF16) if want_tap else None,
         "y2": None, "pre": None, "h": None, "mean": None, "rstd": None}
    if save:
        r.update(y2=torch.empty(Mp, D, device=dev, dtype=BF16), pre=torch.empty(Mp * Hd, device=dev, dtype=BF16),
                 h=torch.empty(Mp, Hd, device=dev, dtype=BF16), mean=torch.empty(M, device=dev, dtype=F32),
                 rstd=torch.empty(M, device=dev, dtype=F32))
    nw = nb = None
    if next_ln is not None:
        nw, nb = next_ln
        r.update(next_y=torch.empty(M, D, device=dev, dtype=BF16), next_mean=torch.empty(M, device=dev, dtype=F32),
                 next_rstd=torch.empty(M, device=dev, dtype=F32))
    check(lib().dkd_mlp192_fwd(ptr(x1), ptr(ln_w), ptr(ln_b), eps, ptr(fc1_w), ptr(fc1_b), ptr(fc2_wt), ptr(fc2_b), ptr(rowscale),
                               rows_per_sample, ptr(r["x2"]), ptr(r["tap"]), ptr(r["y2"]), ptr(r["pre"]), ptr(r["h"]), ptr(r["mean"]),
                               ptr(r["rstd"]), ptr(nw), ptr(nb), ptr(r.get("next_y")), ptr(r.get("next_mean")), ptr(r.get("next_rstd")), M, Hd,
                               stream()), "mlp192_fwd")
    return r


def mlp192_bwd(g, pre, fc2_wt, fc1_w, x1, ln_w, mean, rstd, d_ln_w, d_ln_b, *, gtap=None, s2=None, s1=None, rows_per_sample=0,
               want_cast=True):
    """Backward of mlp192_fwd up to (not including) the two weight gradients (include/dkd.h, dkd_mlp192_bwd).  g f32 [M, 192] is updated in
    place (+= LayerNorm backward); d_ln_w / d_ln_b are accumulated.  -> (dF bf16 [Mp, 192], dH bf16 [Mp, hidden], cast_out bf16 [M, 192])."""
    M, D = g.shape
    Hd = fc1_w.shape[0]
    assert D == 192 and g.dtype == F32 and g.is_contiguous() and x1.is_contiguous()
    dev, Mp = g.device, (M + 15) // 16 * 16
    dF = torch.empty(Mp, D, device=dev, dtype=BF16)
    dH = torch.empty(Mp, Hd, device=dev, dtype=BF16)
    cast = torch.empty(M, D, device=dev, dtype=BF16) if want_cast else None
    ws = torch.empty(lib().dkd_layernorm_bwd_workspace_bytes(M, D), device=dev, dtype=torch.uint8)
    check(lib().dkd_mlp192_bwd(ptr(g), ptr(gtap), ptr(s2), ptr(s1), rows_per_sample, ptr(pre), ptr(fc2_wt), ptr(fc1_w), ptr(x1), ptr(ln_w),
                               ptr(mean), ptr(rstd), ptr(dF), ptr(dH), ptr(cast), ptr(d_ln_w), ptr(d_ln_b), ptr(ws), M, Hd, stream()),
          "mlp192_bwd")
    return dF, dH, cast


def im2col_patches(img, p):
    assert img.dtype == F32 and img.is_contiguous() and img.dim() == 4
    B, Cc, H, W = img.shape
    out = torch.empty(B * (H // p) * (W // p), Cc * p * p, device=img.device, dtype=BF16)
    check(lib().dkd_im2col_patches(ptr(img), ptr(out), B, Cc, H, W, p, stream()), "im2col")
    return out


def prefix_tokens_fwd(x, tok, pos, B, N, D, npre):
    check(lib().dkd_prefix_tokens_fwd(ptr(x), ptr(tok), ptr(pos), B, N, D, npre, stream()), "prefix_tokens")


def embed_bwd(dx, dtok, dpos, B, N, D, npre):
    check(lib().dkd_embed_bwd(ptr(dx), ptr(dtok), ptr(dpos), B, N, D, npre, stream()), "embed_bwd")


def scale_cast_bf16(x, *, M=None, xmap=IDENT, rowscale=None, rows_per_sample=0, add=None, ld_out=None):
    assert x.dtype == F32 and x.dim() == 2
    M = x.shape[0] if M is None else M
    D = x.shape[1]
    ld = ld_out or D
    y = torch.empty(M, ld, device=x.device, dtype=BF16) if ld == D else torch.zeros(M, ld, device=x.device, dtype=BF16)
    check(lib().dkd_scale_cast_bf16(ptr(x), x.stride(0), xmap, ptr(rowscale), rows_per_sample, ptr(add),
                                    _is_f32(add) if add is not None else 1, add.stride(0) if add is not None else 0, ptr(y), ld,
                                    M, D, stream()), "scale_cast")
    return y


def scale_bf16_(x, scalar):
    """x (bf16, contiguous, numel % 8 == 0) *= scalar (0-dim / 1-element f32 DEVICE tensor), in place, product formed in f32."""
    assert x.dtype == BF16 and x.is_contiguous() and scalar.dtype == F32 and scalar.numel() == 1
    check(lib().dkd_scale_bf16(ptr(x), ptr(scalar), x.numel(), stream()), "scale_bf16")
    return x


def cast_pad_bf16(src, Cp, scalar=None):
    """src f32 [rows, C] (unit column stride) -> bf16 [rows, Cp], columns C.. zero; optionally times a 1-element f32 device tensor."""
    assert src.dtype == F32 and src.dim() == 2 and src.stride(1) == 1
    rows, Cc = src.shape
    out = torch.empty(rows, Cp, device=src.device, dtype=BF16)
    check(lib().dkd_cast_pad_bf16(ptr(src), src.stride(0), ptr(scalar), ptr(out), rows, Cc, Cp, stream()), "cast_pad_bf16")
    return out


def droppath_scales(keep_prob, B, seed):
    """keep_prob f32 [n] (device) -> f32 [n, B]: Bernoulli(keep_prob[i]) / keep_prob[i] per sample, one launch for all branches."""
    n = keep_prob.numel()
    out = torch.empty(n, B, device=keep_prob.device, dtype=F32)
    check(lib().dkd_droppath_scales(ptr(out), ptr(keep_prob), n, B, int(seed) & (2 ** 64 - 1), stream()), "droppath_scales")
    return out


def cast_weight(w, w_bf16=None, w_t_bf16=None):
    """w f32 [rows, cols] (contiguous) -> bf16 copy and/or bf16 transpose, written into the given buffers."""
    assert w.dtype == F32 and w.is_contiguous()
    rows = w.shape[0]
    cols = w.numel() // rows
    check(lib().dkd_cast_weight(ptr(w), ptr(w_bf16), ptr(w_t_bf16), rows, cols, stream()), "cast_weight")


def cast_weight_table(pairs):
    """Device-side table for ``cast_weight_group``: pairs = [(w f32 [rows, cols] contiguous, wt bf16 [cols, rows])] ->
    (int64 tensor of DkdCastItem records, n, total_tiles).  Keep it as long as the tensors in ``pairs`` live."""
    rec, tiles = [], 0
    for w, wt in pairs:
        rows = w.shape[0]
        cols = w.numel() // rows
        assert w.dtype == F32 and w.is_contiguous() and wt.dtype == BF16 and wt.is_contiguous() and wt.numel() == w.numel()
        rec += [w.data_ptr(), wt.data_ptr(), rows | (cols << 32), tiles]           # {w, wt, rows, cols, first_tile, pad}
        tiles += ((rows + 31) // 32) * ((cols + 31) // 32)
    return torch.tensor(rec, dtype=torch.int64, device=pairs[0][0].device), len(pairs), tiles


def cast_weight_group(table):
    tab, n, tiles = table
    check(lib().dkd_cast_weight_group(ptr(tab), n, tiles, stream()), "cast_weight_group")


def colsum(x, out, *, M=None, N=None, xmap=IDENT):
    M = x.shape[0] if M is None else M
    N = x.shape[1] if N is None else N
    check(lib().dkd_colsum(ptr(x), _is_f32(x), x.stride(0), xmap, ptr(out), M, N, stream()), "colsum")
    return out


def add_rows(x, y, *, ymap=IDENT, accumulate=True, M=None):
    M = x.shape[0] if M is None else M
    check(lib().dkd_add_rows(ptr(x), _is_f32(x), x.stride(0), ptr(y), y.stride(0), ymap, M, x.shape[1], int(accumulate), stream()),
          "add_rows")
    return y


def logit_loss(z, target, *, smoothing=0.1, kd_mode=0, z_kd=None, z_t=None, tau=1.0, w_base=1.0, w_kd=0.0):
    """-> (losses f32 [5] = (base, distill, w_base * base + w_kd * distill, w_base * base, w_kd * distill), dz, dz_kd | None).
    target: f32 [B, C] soft targets or int64 [B] labels."""
    assert z.dtype == F32 and z.is_contiguous()
    B, Cc = z.shape
    soft = target if target.dtype == F32 else None
    labels = target if target.dtype == torch.int64 else None
    if soft is None and labels is None:
        raise TypeError("target must be f32 soft targets or int64 labels")
    losses = torch.zeros(5, device=z.device, dtype=F32)
    dz = torch.empty_like(z)
    dz_kd = torch.empty_like(z) if kd_mode else None
    if kd_mode:
        assert z_kd.is_contiguous() and z_t.is_contiguous() and z_kd.dtype == F32 and z_t.dtype == F32
    check(lib().dkd_logit_loss(ptr(z), ptr(soft), ptr(labels), smoothing, kd_mode, ptr(z_kd), ptr(z_t), tau, w_base, w_kd,
                               ptr(losses), ptr(dz), ptr(dz_kd), B, Cc, stream()), "logit_loss")
    return losses, dz, dz_kd


def topk_correct(z, labels, ks):
    """z f32 [B, C], labels int64 [B] -> f32 [len(ks)]: percentage of rows whose label is among the top ks[i] logits."""
    import ctypes
    assert z.dtype == F32 and z.is_contiguous() and labels.dtype == torch.int64 and labels.is_contiguous() and 1 <= len(ks) <= 4
    out = torch.zeros(len(ks), device=z.device, dtype=F32)
    arr = (ctypes.c_int32 * len(ks))(*[int(k) for k in ks])
    check(lib().dkd_topk_correct(ptr(z), ptr(labels), z.shape[0], z.shape[1], ctypes.cast(arr, ctypes.c_void_p), len(ks), ptr(out),
                                 stream()), "topk_correct")
    return out


def mse_loss(a, t, loss, w_over_denom, *, M=None, tmap=IDENT, mask=None, grad=True, grad_f32=False, grad_out=None):
    """loss[0] += w/denom * sum(mask (a - t)^2); returns d loss / d a ([M, D], bf16 unless grad_f32) or None.

    ``grad_out``: preallocated gradient buffer (its dtype and row stride are used; lets the caller K-pad it).
    """
    M = a.shape[0] if M is None else M
    D = a.shape[1]
    assert a.stride(1) == 1 and t.stride(1) == 1, "mse_loss operands must be row-major"
    if grad_out is not None:
        da = grad_out
    else:
        da = torch.empty(M, D, device=a.device, dtype=F32 if grad_f32 else BF16) if grad else None
    check(lib().dkd_mse_loss(ptr(a), _is_f32(a), a.stride(0), ptr(t), _is_f32(t), t.stride(0), tmap, ptr(mask), w_over_denom,
                             ptr(loss), ptr(da), _is_f32(da) if da is not None else 0, da.stride(0) if da is not None else 0,
                             M, D, stream()), "mse_loss")
    return da


def mask_select(x, mask_token, mask):
    out = torch.empty_like(x)
    check(lib().dkd_mask_select(ptr(x), ptr(mask_token), ptr(mask), ptr(out), x.shape[0], x.shape[1], stream()), "mask_select")
    return out


def mask_select_bwd(dout, mask, dmask_token):
    dx = torch.empty_like(dout)
    check(lib().dkd_mask_select_bwd(ptr(dout), ptr(mask), ptr(dx), ptr(dmask_token), dout.shape[0], dout.shape[1], stream()),
          "mask_select_bwd")
    return dx


def adamw_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    check(lib().dkd_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), p.numel(), lr, beta1, beta2, eps, wd, step, grad_scale,
                               stream()), "adamw")


def adamw_step_gated(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, state):
    """AdamW on a flat segment that is skipped when its unit received no gradient this step (include/dkd.h, dkd_adamw_step_gated)."""
    assert state.dtype == F32 and state.numel() == 2
    check(lib().dkd_adamw_step_gated(ptr(p), ptr(g), ptr(m), ptr(v), ptr(p_bf16), p.numel(), lr, beta1, beta2, eps, wd, ptr(state), stream()),
          "adamw_step_gated")


def jacobi_eigh(A, sweeps=10):
    """A f32 [batch, n, n] symmetric (n <= 128) -> (evals [batch, n] descending, evecs [batch, n, n], columns sorted alike)."""
    assert A.dtype == F32 and A.dim() == 3 and A.shape[1] == A.shape[2]
    A = A.contiguous()
    bt, n, _ = A.shape
    ev = torch.empty(bt, n, device=A.device, dtype=F32)
    vec = torch.empty(bt, n, n, device=A.device, dtype=F32)
    check(lib().dkd_jacobi_eigh(ptr(A), ptr(ev), ptr(vec), bt, n, sweeps, stream()), "jacobi_eigh")
    ev, order = torch.sort(ev, dim=1, descending=True)
    vec = torch.gather(vec, 2, order[:, None, :].expand(bt, n, n))
    return ev, vec


def lowrank_workspace(L, Dt, device):
    return torch.empty(lib().dkd_lowrank_workspace_bytes(L, Dt), device=device, dtype=torch.uint8)


def lowrank_step(G, V, mode, ws, rank=0, hi=None, lo=None, evals=None, ritz_sweeps=2):
    """One block-subspace-iteration step on G f32 [L, Dt, Dt] (upper 128-tiles valid), V f32 [L, Dt, 96] in place (include/dkd.h):
    mode 0 power step, 1 tracking step (at most ``ritz_sweeps`` Jacobi sweeps of Rayleigh-Ritz, then a power step orthonormalised in Ritz
    order; fills hi / lo bf16 [L, rank, Dt] and evals f32 [L, 96] when given), 2 orthonormalise V, 3 converged Rayleigh-Ritz in span(V)
    (outputs as mode 1)."""
    assert G.dtype == F32 and V.dtype == F32 and G.is_contiguous() and V.is_contiguous() and V.shape[2] == 96
    L, Dt = G.shape[0], G.shape[1]
    assert G.shape == (L, Dt, Dt) and V.shape[:2] == (L, Dt)
    if hi is not None:
        assert hi.dtype == BF16 and lo.dtype == BF16 and hi.is_contiguous() and lo.is_contiguous() and hi.shape == lo.shape == (L, rank, Dt)
    check(lib().dkd_lowrank_step(ptr(G), ptr(V), L, Dt, mode, ritz_sweeps, rank, ptr(hi), ptr(lo), ptr(evals), ptr(ws), stream()),
          "lowrank_step")
    return V


def lowrank_chain_workspace(L, Dt, device):
    """Scratch of ``lowrank_chain`` (zero-initialised: its Gram accumulators must be zero on the first call; every call leaves them so)."""
    return torch.zeros(lib().dkd_lowrank_chain_workspace_bytes(L, Dt), device=device, dtype=torch.uint8)


def lowrank_chain(G, V, n_mult, ritz_sweeps, ws, rank=0, hi=None, lo=None, evals=None):
    """``n_mult`` power steps from the orthonormal basis V f32 [L, Dt, 96] (in place) and a converged Rayleigh-Ritz step, all layers in
    the same short launches (include/dkd.h: dkd_lowrank_chain) -- the per-batch solve that stands for model/loss.py:321's svd."""
    assert G.dtype == F32 and V.dtype == F32 and G.is_contiguous() and V.is_contiguous() and V.shape[2] == 96
    L, Dt = G.shape[0], G.shape[1]
    assert G.shape == (L, Dt, Dt) and V.shape[:2] == (L, Dt)
    if hi is not None:
        assert hi.dtype == BF16 and lo.dtype == BF16 and hi.is_contiguous() and lo.is_contiguous() and hi.shape == lo.shape == (L, rank, Dt)
    check(lib().dkd_lowrank_chain(ptr(G), ptr(V), L, Dt, n_mult, ritz_sweeps, rank, ptr(hi), ptr(lo), ptr(evals), ptr(ws), stream()),
          "lowrank_chain")
    return V


def lowrank_chain_info(ws, L, Dt):
    """Jacobi sweeps the last ``lowrank_chain`` ran per layer: int32 [L, 2] (column 1) -- diagnostics."""
    off = lib().dkd_lowrank_chain_workspace_bytes(L, Dt) - (L * 8 + 255) // 256 * 256
    return ws[off:off + L * 8].view(torch.int32).view(L, 2)


def lowrank_info(ws, L, Dt):
    """Jacobi sweeps of the last ``lowrank_step`` per layer: int32 [L, 2] (orthonormalisation, Rayleigh-Ritz) -- diagnostics."""
    off = lib().dkd_lowrank_workspace_bytes(L, Dt) - (L * 8 + 255) // 256 * 256
    return ws[off:off + L * 8].view(torch.int32).view(L, 2)


def im2col3x3(x, B, hw):
    """x bf16 [B*hw*hw, C] (token grid) -> cols bf16 [B*hw*hw, 9*C]."""
    assert x.dtype == BF16 and x.is_contiguous()
    Cc = x.shape[1]
    cols = torch.empty(x.shape[0], 9 * Cc, device=x.device, dtype=BF16)
    check(lib().dkd_im2col3x3(ptr(x), ptr(cols), B, hw, Cc, stream()), "im2col3x3")
    return cols


def col2im3x3(dcols, B, hw, relu_gate=None):
    assert dcols.dtype == BF16 and dcols.is_contiguous()
    Cc = dcols.shape[1] // 9
    dx = torch.empty(dcols.shape[0], Cc, device=dcols.device, dtype=BF16)
    check(lib().dkd_col2im3x3(ptr(dcols), ptr(relu_gate), ptr(dx), B, hw, Cc, stream()), "col2im3x3")
    return dx


def conv3x3_wgrad(dy, x, dw, dbias, B, hw):
    """dw f32 [Cout, 9 * Cin] (= [Cout, ky, kx, cin]) += weight gradient of the 3 x 3 / pad 1 convolution y = conv(x); dbias f32 [Cout] +=
    column sums of dy.  dy bf16 [B*hw*hw, Cout], x bf16 [B*hw*hw, Cin]: no im2col matrix (include/dkd.h: dkd_conv3x3_wgrad)."""
    assert dy.dtype == BF16 and x.dtype == BF16 and dy.is_contiguous() and x.is_contiguous() and dw.dtype == F32 and dw.is_contiguous()
    Cout, Cin = dy.shape[1], x.shape[1]
    assert dw.shape == (Cout, 9 * Cin) and dy.shape[0] == x.shape[0] == B * hw * hw
    check(lib().dkd_conv3x3_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(dbias), B, hw, Cin, Cout, stream()), "conv3x3_wgrad")
    return dw


def sort_l1_loss(s, t, loss, w, *, B, P, tmap=IDENT, grad_f32=False):
    """loss[0] += w * sum |sort_tokens(s) - sort_tokens(t)|; returns d loss / d s (same layout as s: [B*P, D])."""
    D = s.shape[1]
    assert s.is_contiguous() and s.shape[0] == B * P and t.stride(1) == 1
    ds = torch.empty(B * P, D, device=s.device, dtype=F32 if grad_f32 else BF16)
    check(lib().dkd_sort_l1_loss(ptr(s), _is_f32(s), ptr(t), _is_f32(t), t.stride(0), tmap, w, ptr(loss), ptr(ds), int(grad_f32), B, P, D,
                                 stream()), "sort_l1_loss")
    return ds


def normalize_mse(s, t_hat, loss, w_over_denom, w_scalar=None, ld_grad=None):
    assert s.dtype == F32 and s.is_contiguous() and t_hat.dtype == BF16 and t_hat.is_contiguous()
    M, D = s.shape
    ld = ld_grad or D
    ds = torch.empty(M, ld, device=s.device, dtype=BF16) if ld == D else torch.zeros(M, ld, device=s.device, dtype=BF16)
    check(lib().dkd_normalize_mse(ptr(s), ptr(t_hat), ptr(w_scalar), w_over_denom, ptr(loss), ptr(ds), ld, M, D, stream()), "normalize_mse")
    return ds


def diffkd_prepare(t, noise, sigma, temb, *, M, rows_per_sample, tmap=IDENT):
    """-> (t_hat bf16 [M, D], nz f32 [M, D], x_in bf16 [M, D])."""
    D = t.shape[1]
    assert t.dtype == BF16 and noise.dtype == F32 and noise.is_contiguous() and temb.is_contiguous() and sigma.dtype == F32
    t_hat = torch.empty(M, D, device=t.device, dtype=BF16)
    nz = torch.empty(M, D, device=t.device, dtype=F32)
    x_in = torch.empty(M, D, device=t.device, dtype=BF16)
    check(lib().dkd_diffkd_prepare(ptr(t), t.stride(0), tmap, ptr(noise), ptr(sigma), ptr(temb), rows_per_sample, ptr(t_hat), ptr(nz),
                                   ptr(x_in), M, D, stream()), "diffkd_prepare")
    return t_hat, nz, x_in


def dropout_mse(a, t, keep, keep_scale, loss, w_over_denom):
    """loss[0] += w * sum (a*keep*scale - t)^2 -> gradient w.r.t. a (bf16, same shape)."""
    assert a.dtype == F32 and t.dtype == F32 and a.is_contiguous() and t.is_contiguous()
    da = torch.empty(a.shape, device=a.device, dtype=BF16)
    check(lib().dkd_dropout_mse(ptr(a), ptr(t), ptr(keep), keep_scale, w_over_denom, ptr(loss), ptr(da), a.numel(), stream()), "dropout_mse")
    return da


PROBE_SYMBOLS = ("gemm_nt_kernel<128>", "gemm_nt_kernel<64>", "gemm_nt256_kernel")
PROBE_FAMILIES = PROBE_SYMBOLS + ("student_block_bwd", "loss_kernels", "student_block_fwd")


def saliency_scores(q, k, *, B, L, H, q_rows_per_sample, k_rows_per_sample, q_first=0, k_first=0, diagonal=True, extra_key_row=-1):
    """Head-averaged softmax attention weights of the saliency scorer (include/dkd.h, dkd_saliency_scores): q, k f32 2-D projections
    (row stride = stride(0); k may be a column slice of the same matrix as q) -> scores f32 [B, L]."""
    assert q.dtype == F32 and k.dtype == F32 and q.dim() == 2 and k.dim() == 2 and q.stride(1) == 1 and k.stride(1) == 1
    hd = k.shape[1] // H
    assert k.shape[1] == H * hd and q.shape[1] == H * hd
    out = torch.empty(B, L, device=q.device, dtype=F32)
    check(lib().dkd_saliency_scores(ptr(q), ptr(k), ptr(out), B, L, H, hd, q.stride(0), k.stride(0), q_rows_per_sample, k_rows_per_sample,
                                    q_first, k_first, 1 if diagonal else 0, extra_key_row, stream()), "saliency_scores")
    return out


def probe_begin():
    """bench.py: start bracketing every NT-GEMM launch with HIP events (recorded inside the library, on the launch stream)."""
    check(lib().dkd_probe_begin(), "probe_begin")


def probe_end():
    """-> {symbol: (flops, ms, launches)} for the launches since probe_begin()."""
    fl, ms, n = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_int32 * 3)()
    check(lib().dkd_probe_end(fl, ms, n), "probe_end")
    return {PROBE_SYMBOLS[i]: (fl[i], ms[i], n[i]) for i in range(3) if n[i]}


def probe_end_ex():
    """-> {family: (flops, bytes, ms, launches)} over PROBE_FAMILIES for the launches since probe_begin()."""
    n = len(PROBE_FAMILIES)
    fl, by, ms, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_int32 * n)()
    check(lib().dkd_probe_end_ex(n, fl, by, ms, cnt), "probe_end_ex")
    return {PROBE_FAMILIES[i]: (fl[i], by[i], ms[i], cnt[i]) for i in range(n) if cnt[i]}


def mixup_(x, lam, box=None):
    """In-place Mixup (box=None) or CutMix (box=(yl, yh, xl, xh)) of a device batch f32 [B, C, H, W] with its flip."""
    assert x.dtype == F32 and x.is_contiguous() and x.dim() == 4
    B, Cc, H, W = x.shape
    yl, yh, xl, xh = box if box is not None else (0, 0, 0, 0)
    check(lib().dkd_mixup(ptr(x), B, Cc, H, W, lam, int(box is not None), int(yl), int(yh), int(xl), int(xh), stream()), "mixup")
    return x


def mixup(x, lam, box=None):
    """Mixup / CutMix of a device batch into a NEW tensor (x is left as it is): the same bytes moved as ``mixup_``."""
    assert x.dtype == F32 and x.is_contiguous() and x.dim() == 4
    B, Cc, H, W = x.shape
    yl, yh, xl, xh = box if box is not None else (0, 0, 0, 0)
    out = torch.empty_like(x)
    check(lib().dkd_mixup_to(ptr(x), ptr(out), B, Cc, H, W, lam, int(box is not None), int(yl), int(yh), int(xl), int(xh), stream()), "mixup_to")
    return out


def mixup_with_patches(x, lam, box, p):
    """``mixup`` that also returns the mix as the bf16 patch matrix of ``im2col_patches(out, p)`` (same launch)."""
    assert x.dtype == F32 and x.is_contiguous() and x.dim() == 4
    B, Cc, H, W = x.shape
    yl, yh, xl, xh = box if box is not None else (0, 0, 0, 0)
    out = torch.empty_like(x)
    patches = torch.empty(B * (H // p) * (W // p), Cc * p * p, device=x.device, dtype=BF16)
    check(lib().dkd_mixup_to_patches(ptr(x), ptr(out), ptr(patches), p, B, Cc, H, W, lam, int(box is not None), int(yl), int(yh), int(xl),
                                     int(xh), stream()), "mixup_to_patches")
    return out, patches


def mixup_targets(labels, num_classes, lam, smoothing):
    assert labels.dtype == torch.int64 and labels.is_contiguous()
    out = torch.empty(labels.shape[0], num_classes, device=labels.device, dtype=F32)
    check(lib().dkd_mixup_targets(ptr(labels), ptr(out), labels.shape[0], num_classes, lam, smoothing, stream()), "mixup_targets")
    return out


def ema_update(ema, p, decay):
    assert ema.dtype == F32 and p.dtype == F32 and ema.numel() == p.numel()
    check(lib().dkd_ema_update(ptr(ema), ptr(p), ema.numel(), decay, stream()), "ema_update")
