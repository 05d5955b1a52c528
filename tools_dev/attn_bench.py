"""Micro-benchmark of the attention kernels on the headline shapes (teacher: B*H = 3072 heads, student: 768)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ops
dev = "cuda:0"
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, B, N, H in (("teacher", 256, 198, 12), ("student", 256, 197, 3)):
    if os.environ.get("ONLY", name) != name: continue
    qkv = (torch.randn(B * N, 3 * H * 64, device=dev)).to(torch.bfloat16)
    out, lse = ops.attn_fwd(qkv, B, N, H)
    dout = torch.randn_like(out)
    fl = 4.0 * N * N * 64 * B * H
    tf = t(lambda: ops.attn_fwd(qkv, B, N, H))
    tb = t(lambda: ops.attn_bwd(qkv, out, dout, lse, B, N, H))
    print(f"{name}: fwd {tf:.1f} us ({fl/tf/1e6:.0f} TFLOP/s)  bwd {tb:.1f} us ({2.5*fl/tb/1e6:.0f} TFLOP/s useful)")
