"""dev: LayerNorm fwd/bwd timing on the student / teacher shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ops
def t(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, M, D in (("student", 50432, 192), ("teacher", 50688, 768)):
    x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, g, b)
    dy = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    dx = torch.zeros(M, D, device="cuda"); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    tf = t(lambda: ops.layernorm_fwd(x, g, b))
    tb = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, accumulate=True))
    ws = torch.empty(2 * D * ((M + 63) // 64), device="cuda")
    tw = t(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dg, db, accumulate=True, ws=ws))
    print(f"{name}: fwd {tf:.1f} us ({M*D*6/tf/1e6:.2f} TB/s)  bwd {tb:.1f} us ({M*D*14/tb/1e6:.2f} TB/s)  bwd+ws {tw:.1f} us")
