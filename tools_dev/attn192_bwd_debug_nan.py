"""dev: poison LDS (and freed device memory) with NaNs, then run dkd_attn192_bwd: where do non-finite values appear?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops
B, N, D = 256, 197, 192
dev = torch.device("cuda", 0); BF = torch.bfloat16
g = torch.Generator().manual_seed(1)
r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
y1 = r(B * N, D).to(BF); w = r(576, D, scale=D ** -0.5).to(BF); bias = r(576, scale=0.5)
wpt = r(D, D, scale=D ** -0.5).to(BF); dy = r(B * N, D).to(BF)
qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
ref = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)
a = torch.full((256 * 256, 768), float("nan"), device=dev, dtype=BF); b = torch.full((2304, 768), float("nan"), device=dev, dtype=BF)
keep = ops.gemm_nt(a, b)                       # (kept alive: the allocator must not hand its NaN-filled memory to the next call)
got = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)
torch.cuda.synchronize()
bad = ~torch.isfinite(got.float()).view(B, N, 3, 3, 64)
print("non-finite elements:", int(bad.sum()), "of", bad.numel(), "| equal to the clean run:", bool(torch.equal(got, ref)))
if bad.any():
    print("by part (dq, dk, dv):", bad.sum((0, 1, 3, 4)).tolist(), " by head:", bad.sum((0, 1, 2, 4)).tolist())
    print("samples affected:", int(bad.any(1).any(1).any(1).any(1).sum()), " rows affected per sample (first bad sample):",
          torch.nonzero(bad[int(torch.nonzero(bad.flatten(1).any(1))[0])].flatten(1).any(1)).flatten()[:20].tolist())
d = (got.float() - ref.float()).abs().view(B, N, 3, 3, 64)
d = torch.nan_to_num(d, nan=1e9)
print("differs from the clean run: elements", int((d > 0).sum()), "| by part", (d > 0).sum((0, 1, 3, 4)).tolist(), "| by head", (d > 0).sum((0, 1, 2, 4)).tolist(),
      "| samples", int((d.flatten(1).amax(1) > 0).sum()), "| max diff", d.max().item())
bs = torch.nonzero(d.flatten(1).amax(1) > 0).flatten()[:6].tolist()
for b_ in bs:
    rows = torch.nonzero(d[b_].flatten(1).amax(1) > 0).flatten()
    print(f"  sample {b_}: {len(rows)} rows differ, first {rows[:10].tolist()} last {rows[-3:].tolist()}; per (part, head) max", d[b_].amax((0, 3)).tolist())
# a third run, nothing in between: is the kernel deterministic after the poisoning?
got3 = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)
torch.cuda.synchronize()
print("third run == second:", bool(torch.equal(got3, got)), " third == clean:", bool(torch.equal(got3, ref)))
