"""dev: the teacher's proj GEMM alone (M = 256 x 197, N = K = 768, f32 residual epilogue in place, bf16 copy + row sums for the LayerNorm fold),
timed with events over a rotating set of buffers (cold operands, like in the model).  usage: python tools_dev/proj_bench.py [K=768] [fold=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops

K = int(sys.argv[1]) if len(sys.argv) > 1 else 768
fold = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
M, N = 256 * 197, 768
NB = 6
A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NB)]
X = [torch.randn(M, N, device=dev) for _ in range(NB)]
XB = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
ST = [torch.zeros(M, 2, device=dev) for _ in range(NB)]
W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
b = torch.randn(N, device=dev)


def run(i):
    j = i % NB
    kw = dict(xb=XB[j], rowstats=ST[j]) if fold else {}
    ops.gemm_nt(A[j], W, out=X[j], bias=b, resid=X[j], **kw)


for i in range(6):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 48
e0.record()
for i in range(n):
    run(i)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"K={K} fold={fold} DKD_NT_RING={os.environ.get('DKD_NT_RING', '(default 1)')}: {us:.1f} us per launch, {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
