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


wq = (torch.randn(192, 576, device=dev) * 192 ** -0.5).to(BF)
x = torch.randn(B * N, 192, device=dev)
gam = torch.ones(192, device=dev)
mean, rstd = x.mean(1).contiguous(), torch.rsqrt(x.var(1, unbiased=False) + 1e-6).contiguous()
g = torch.zeros(B * N, 192, device=dev)
dg, db = torch.zeros(192, device=dev), torch.zeros(192, device=dev)
ws = torch.empty(L.dkd_layernorm_bwd_workspace_bytes(B * N, 192) // 4, device=dev)
NONE9 = [None] * 9


def fused():
    ffi.check(L.dkd_attn192_bwd(ffi.ptr(dy), ffi.ptr(wpt), ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(lse), ffi.ptr(dqkv), *NONE9, B, N, ffi.stream()), "bwd")


def fused_ln():
    ffi.check(L.dkd_attn192_bwd(ffi.ptr(dy), ffi.ptr(wpt), ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(lse), ffi.ptr(dqkv), ffi.ptr(wq), ffi.ptr(x), ffi.ptr(gam),
                                ffi.ptr(mean), ffi.ptr(rstd), ffi.ptr(g), ffi.ptr(dg), ffi.ptr(db), ffi.ptr(ws), B, N, ffi.stream()), "bwd ln")


def lnbwd():
    ops.gemm_nt_lnbwd(dqkv, wq, x, gam, mean, rstd, g, dg, db, ws)


def proj_dgrad():
    ops.gemm_nt(dy, wpt, out=d_o)


def attn_bwd():
    ffi.check(L.dkd_attn_bwd(ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(d_o), ffi.ptr(lse), ffi.ptr(dqkv), B, N, 3, ffi.stream()), "attn_bwd")


proj_dgrad()
t_f, t_p, t_a, t_fl, t_l = timeit(fused), timeit(proj_dgrad), timeit(attn_bwd), timeit(fused_ln), timeit(lnbwd)
print(f"B {B} N {N}: fused {t_f:.1f} us; proj dgrad {t_p:.1f} + attention backward {t_a:.1f} = {t_p + t_a:.1f} us | with the qkv dgrad + LayerNorm "
      f"backward: fused {t_fl:.1f} us; separate launch {t_l:.1f} us (+ reduction) -> {t_f + t_l:.1f} / {t_p + t_a + t_l:.1f} us")
