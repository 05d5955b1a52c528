#!/usr/bin/env python3
"""Diagnostics of the LRKD target chain on real teacher taps: Jacobi sweeps per tracking step and time per step.
   python tools_dev/lowrank_probe.py [batch] [calls] [same]   ("same": feed the same batch every call, as an idle bench would)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops, vit
from deltakd_amd.losses import LowRankTargets

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
same = len(sys.argv) > 3 and sys.argv[3] == "same"
sweeps = int(os.environ.get("RITZ_SWEEPS", "2"))
dev = "cuda:0"
torch.manual_seed(42)
t = vit.create_model("deit_base_distilled_patch16_224", num_classes=1000).to(dev).eval()
for p in t.parameters():
    p.requires_grad = False
solver = LowRankTargets(ritz_sweeps=sweeps)
g = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn(B, 3, 224, 224, device=dev, generator=g)
for c in range(calls):
    x = x0 if same else torch.randn(B, 3, 224, 224, device=dev, generator=g)
    with torch.no_grad():
        _, taps = t.forward_with_taps(x, (0, 1, 11))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tg = solver([taps[0], taps[1], taps[11]], 2, 64)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    info = ops.lowrank_info(solver._ws, 3, 768).cpu().tolist()
    # quality vs the exact SVD of this batch (layer 0): captured rank-64 energy and worst singular-value error
    T = taps[0][:, 2:].reshape(-1, 768).float()
    S = torch.linalg.svdvals(T.double())
    got = tg[0].double()
    energy = (got ** 2).sum().item() / (S[:64] ** 2).sum().item()
    sverr = ((got.norm(dim=0) - S[:64]).abs() / S[0]).max().item()
    print(f"call {c}: {dt:7.2f} ms  energy {energy:.5f} sv_err {sverr:.2e}  sweeps (orth, ritz) per layer {info}  ritz[0,:3] {solver.ritz[0,:3].tolist()} ritz[0,60:66] {[round(v,1) for v in solver.ritz[0,60:66].tolist()]} ritz[0,90:] {[round(v,1) for v in solver.ritz[0,90:].tolist()]}", flush=True)
