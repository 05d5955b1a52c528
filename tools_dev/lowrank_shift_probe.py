"""dev: the LRKD tracker on never-repeating, shifting batches of 256 (tests/test_fullsize_gpu.py::_shifting_batches) for several accuracy
settings: captured energy / singular-value error / residual per call, and the cost of a call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_fullsize_gpu as T
from deltakd_amd import vit
from deltakd_amd.losses import LowRankTargets

torch.manual_seed(42)
t = vit.create_model("deit_base_distilled_patch16_224", num_classes=1000).to(T.DEV).eval()
for p in t.parameters():
    p.requires_grad = False
k, npre = 64, 2
settings = [tuple(int(v) for v in s.split(",")) for s in os.environ.get("PROBE_SETTINGS", "1,2;2,2;2,4;3,4;4,6;8,12").split(";")]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14
taps_seq = []
with torch.no_grad():
    for x in T._shifting_batches(n, seed=123, jump_at=n - 3):
        _, taps = t.forward_with_taps(x, (0, 1, 11))
        taps_seq.append([taps[0].clone(), taps[1].clone(), taps[11].clone()])
exact = [[T._exact_lowrank(tp, npre, k)[1] for tp in sel] for sel in taps_seq]
for wi, rs in settings:
    solver = LowRankTargets(warm_iters=wi, ritz_sweeps=rs)
    rows = []
    for call, sel in enumerate(taps_seq):
        with torch.no_grad():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tg = solver(sel, npre, k)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            res = max(solver.residual(T._gram_of(sel, npre), k))
        energy, sv = 1.0, 0.0
        for got, S in zip(tg, exact[call]):
            got = got.double()
            energy = min(energy, ((got ** 2).sum() / (S[:k] ** 2).sum()).item())
            sv = max(sv, ((got.norm(dim=0) - S[:k]).abs() / S[0]).max().item())
        from deltakd_amd import ops as _ops
        sweeps = int(_ops.lowrank_chain_info(solver._cws, 3, 768)[:, 1].max()) if getattr(solver, "_cws", None) is not None else -1
        rows.append((energy, sv, res, dt, sweeps))
    print(f"warm_iters {wi} ritz_sweeps {rs}: ms/call (host-synced) {sum(r[3] for r in rows[2:]) / (n - 2):.2f}")
    for call, r in enumerate(rows):
        print(f"   call {call:2d}{' (jump)' if call >= n - 3 else ''}: energy {r[0]:.5f}  sv {r[1]:.2e}  residual {r[2]:.2e}  jacobi sweeps {r[4]}")
    S = exact[1][0]
print("spectrum of tap 0, call 1: sigma_1..4, 32, 64, 65, 96 / sigma_1:", [round((S[i] / S[0]).item(), 4) for i in (0, 1, 2, 3, 31, 63, 64, 95)])
