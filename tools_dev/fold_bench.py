#!/usr/bin/env python3
"""Standalone timing of the teacher's wide GEMMs with and without the LayerNorm fold (same shapes, back to back, HIP events)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ops
BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device("cuda:0")
M, D = 256 * 198, 768
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {}
x = torch.randn(M, D, device=dev, generator=g)
xb = x.to(BF16)
stats = torch.stack([x.sum(1), (x * x).sum(1)], 1).contiguous()
for name, N, gelu in (("qkv", 2304, False), ("fc1", 3072, True)):
    w = (torch.randn(N, D, device=dev, generator=g) * 0.03).to(BF16)
    bias = torch.randn(N, device=dev, generator=g)
    c = w.float().sum(1).contiguous()
    out = torch.empty(M, N, device=dev, dtype=BF16)
    res[name + "_plain_us"] = timeit(lambda: ops.gemm_nt(xb, w, out=out, bias=bias, gelu=gelu))
    res[name + "_fold_us"] = timeit(lambda: ops.gemm_nt(xb, w, out=out, bias=bias, gelu=gelu, ln_stats=stats, ln_c=c))
    res[name + "_plain2_us"] = timeit(lambda: ops.gemm_nt(xb, w, out=out, bias=bias, gelu=gelu))
# producer: fc2 (K = 3072) and proj (K = 768) with and without xb / rowstats
for name, K in (("fc2", 3072), ("proj", 768)):
    a = (torch.randn(M, K, device=dev, generator=g)).to(BF16)
    w = (torch.randn(D, K, device=dev, generator=g) * 0.03).to(BF16)
    bias = torch.randn(D, device=dev, generator=g)
    xr = torch.randn(M, D, device=dev, generator=g)
    xo = torch.empty(M, D, device=dev, dtype=BF16)
    st = torch.zeros(M, 2, device=dev)
    res[name + "_plain_us"] = timeit(lambda: ops.gemm_nt(a, w, out=xr, bias=bias, resid=xr))
    res[name + "_emit_us"] = timeit(lambda: ops.gemm_nt(a, w, out=xr, bias=bias, resid=xr, xb=xo, rowstats=st))
    res[name + "_ln_us"] = timeit(lambda: ops.layernorm_fwd(xr, bias, bias, save_stats=False))
print(json.dumps(res))
