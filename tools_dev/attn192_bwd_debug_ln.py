"""dev: where does the LayerNorm part of dkd_attn192_bwd go wrong?  d_ln_b isolates the GEMM (sum over rows of dT), dx by rows / features."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 197)
dev = torch.device("cuda", 0); BF = torch.bfloat16; D = 192
g_ = torch.Generator().manual_seed(1)
r = lambda *s, scale=1.0: (torch.randn(*s, generator=g_) * scale).to(dev)
y1 = r(B * N, D).to(BF); w = r(576, D, scale=D ** -0.5).to(BF); bias = r(576, scale=0.5)
wpt = r(D, D, scale=D ** -0.5).to(BF); dy = r(B * N, D).to(BF)
x = r(B * N, D, scale=1.5) + 0.2; gamma = torch.ones(D, device=dev); g0 = torch.zeros(B * N, D, device=dev)
mean = torch.zeros(B * N, device=dev); rstd = torch.ones(B * N, device=dev)      # xhat = x: dx = gy - mean(gy) - x mean(gy x)
qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
wqt = w.t().contiguous()
g = g0.clone(); dgam = torch.zeros(D, device=dev); dbet = torch.zeros(D, device=dev)
dqkv = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N, qkv_wt=wqt, x=x, ln_w=gamma, mean=mean, rstd=rstd, g=g, d_ln_w=dgam, d_ln_b=dbet)
torch.cuda.synchronize()
dT = dqkv.float() @ w.float()
print("d_ln_b (= column sums of dT): max err per 32-feature block", [(dbet - dT.sum(0))[32 * j:32 * j + 32].abs().max().item() for j in range(6)], "scale", dT.sum(0).abs().max().item())
for c in range(3):
    part = dqkv.float()[:, 192 * c:192 * c + 192] @ w.float()[192 * c:192 * c + 192]
    print(f"   if only chunk {c} were summed: err {(dbet - part.sum(0)).abs().max().item():.3e}")
ref = dT - dT.mean(1, keepdim=True) - x * (dT * x).mean(1, keepdim=True)
e = (g - ref).abs().view(B, N, D)
print("dx err by sample:", e.amax((1, 2)).tolist())
print("dx err by row group of 16 (sample 0):", [round(e[0, 16 * k:16 * k + 16].max().item(), 3) for k in range((N + 15) // 16)])
print("dx err by feature block of 8 (sample 0):", [round(e[0, :, 8 * k:8 * k + 8].max().item(), 3) for k in range(24)])
# is g - g0 perhaps LN'(something simpler)?  e.g. dT itself
print("||g - dT|| / ||dT||:", ((g - dT).norm() / dT.norm()).item(), " ||g - ref|| / ||ref||:", ((g - ref).norm() / ref.norm()).item())
