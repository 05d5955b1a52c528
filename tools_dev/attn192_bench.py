"""Fused qkv + attention forward (dkd_attn192_fwd) against the two launches it replaces, at the headline student shape.
usage: python tools_dev/attn192_bench.py [B=256] [N=197]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
dev = torch.device("cuda", 0)
BF = torch.bfloat16
y1 = torch.randn(B * N, 192, device=dev).to(BF)
w = (torch.randn(576, 192, device=dev) * 192 ** -0.5).to(BF)
bias = torch.randn(576, device=dev) * 0.1


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def unfused():
    qkv = ops.gemm_nt(y1, w, bias=bias)
    return ops.attn_fwd(qkv, B, N, 3)


print(f"B {B} N {N}: fused {timeit(lambda: ops.attn192_fwd(y1, w, bias, B, N)):.1f} us, gemm_nt + attn_fwd {timeit(unfused):.1f} us")
wp = (torch.randn(192, 192, device=dev) * 192 ** -0.5).to(BF)
bp = torch.randn(192, device=dev) * 0.1
x = torch.randn(B * N, 192, device=dev)
s1 = torch.ones(B, device=dev)
qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
x1 = torch.empty_like(x)
from deltakd_amd import ffi
L = ffi.lib()


def fused_proj():
    ffi.check(L.dkd_attn192_fwd_proj(ffi.ptr(y1), ffi.ptr(w), ffi.ptr(bias), ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(lse), ffi.ptr(wp), ffi.ptr(bp), ffi.ptr(x),
                                     ffi.ptr(s1), ffi.ptr(x1), B, N, ffi.stream()), "fwd proj")


def fused_only():
    ffi.check(L.dkd_attn192_fwd(ffi.ptr(y1), ffi.ptr(w), ffi.ptr(bias), ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(lse), B, N, ffi.stream()), "fwd")


def proj_launch():
    ops.gemm_nt(out, wp, out=x1, bias=bp, resid=x, rowscale=s1, rows_per_sample=N, out_f32=True)


t_fp, t_f, t_p = timeit(fused_proj), timeit(fused_only), timeit(proj_launch)
print(f"with proj + residual: one launch {t_fp:.1f} us; qkv + attention {t_f:.1f} + proj launch {t_p:.1f} = {t_f + t_p:.1f} us")
