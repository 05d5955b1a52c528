"""dev: time of the wide NT kernel vs K at fixed M, N (slope = per-K-step cost, intercept = per-tile fixed cost)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ops
dev = "cuda:0"
M, N = 50688, 2304
for K in (768, 3072):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = ops.gemm_nt(a, b, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_nt(a, b, out=out, bias=bias)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    if os.environ.get("DKD_NT256_ABL", "0") != "0":
        torch.cuda.synchronize()
        c = out.view(torch.int64).flatten()[:2].tolist()
        print(f"   block0: {c[0]} shader cycles in {c[1]/100:.1f} us -> {c[0]/max(c[1],1)*0.1:.3f} GHz")
    print(f"K={K:5d} {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  per-tile {us/6.96:6.2f} us  per-kstep {us/6.96/(K/64):5.2f} us")
