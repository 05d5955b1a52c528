"""dev: run-to-run determinism / linearity of dkd_attn192_bwd at B > CU count: which samples / rows / parts differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = int(sys.argv[2]) if len(sys.argv) > 2 else 197
dev = torch.device("cuda", 0); BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
y1 = r(B * N, 192).to(BF); w = r(576, 192, scale=192 ** -0.5).to(BF); bias = r(576, scale=0.5)
wpt = r(192, 192, scale=192 ** -0.5).to(BF); dy = r(B * N, 192).to(BF)
qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
runs = [ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N) for _ in range(4)]
torch.cuda.synchronize()
for i, x in enumerate(runs[1:], 1):
    d = (x.float() - runs[0].float()).view(B, N, 3, 192).abs()
    bad = d.amax((1, 2, 3)) > 0
    print(f"run {i} vs 0: {int(bad.sum())} samples differ", torch.nonzero(bad).flatten()[:16].tolist())
    if bad.any():
        b = int(torch.nonzero(bad)[0])
        rows = torch.nonzero(d[b].amax((1, 2)) > 0).flatten()
        parts = d[b].amax((0, 2)).tolist()
        print(f"   sample {b}: {len(rows)} rows differ, first {rows[:12].tolist()}, max per part (dq, dk, dv) {parts}, per head {d[b].view(N, 3, 3, 64).amax((0, 1, 3)).tolist()}")
d_o = ops.gemm_nt(dy, wpt)
unf = ops.attn_bwd(qkv, out, d_o, lse.view(B, 3, N), B, N, 3)
e = (runs[0].float() - unf.float()).view(B, N, 576).abs().amax((1, 2))
print("vs unfused: worst samples", torch.topk(e, 5).indices.tolist(), torch.topk(e, 5).values.tolist(), "median", e.median().item())
