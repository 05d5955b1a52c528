"""Fused proj dgrad + attention backward (dkd_attn192_bwd) against the two launches it replaces, at the headline student shape.
usage: python tools_dev/attn192_bwd_bench.py [B=256] [N=197]      (preallocated outputs: allocator time is not the kernels')"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ffi, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
dev = torch.device("cuda", 0)
BF = torch.bfloat16
y1 = torch.randn(B * N, 192, device=dev).to(BF)
w = (torch.randn(576, 192, device=dev) * 192 ** -0.5).to(BF)
bias = torch.randn(576, device=dev) * 0.1
wpt = (torch.randn(192, 192, device=dev) * 192 ** -0.5).to(BF)
dy = torch.randn(B * N, 192, device=dev).to(BF)
qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
dqkv = torch.empty_like(qkv)
d_o = torch.empty_like(out)
L = ffi.lib()


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


def fused():
    ffi.check(L.dkd_attn192_bwd(ffi.ptr(dy), ffi.ptr(wpt), ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(lse), ffi.ptr(dqkv), B, N, ffi.stream()), "bwd")


def proj_dgrad():
    ops.gemm_nt(dy, wpt, out=d_o)


def attn_bwd():
    ffi.check(L.dkd_attn_bwd(ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(d_o), ffi.ptr(lse), ffi.ptr(dqkv), B, N, 3, ffi.stream()), "attn_bwd")


proj_dgrad()
t_f, t_p, t_a = timeit(fused), timeit(proj_dgrad), timeit(attn_bwd)
print(f"B {B} N {N}: fused {t_f:.1f} us; proj dgrad {t_p:.1f} + attention backward {t_a:.1f} = {t_p + t_a:.1f} us")
