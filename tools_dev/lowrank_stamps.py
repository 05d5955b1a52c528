#!/usr/bin/env python3
"""dev: phase durations inside the LRKD chain's kernels (workgroup 0, s_memrealtime stamps of a -DDKD_LR_STAMPS build):
    FILE=lowrank MACRO=DKD_LR_STAMPS bash tools_dev/build_abl.sh 1 && DKD_LIB=tools_dev/bin/libdkd_abl1.so python tools_dev/lowrank_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ffi, ops  # noqa: E402

L, Dt, b = 3, 768, 96
g = torch.Generator(device="cuda").manual_seed(0)
T = torch.randn(L, 4096, Dt, device="cuda", generator=g) * torch.logspace(0, -1.5, Dt, device="cuda")
G = (T.transpose(1, 2) @ T).contiguous()
V = torch.linalg.qr(torch.randn(L, Dt, b, device="cuda", generator=g))[0].contiguous()
ws = ops.lowrank_chain_workspace(L, Dt, G.device)
fn = ffi.lib().dkd_lr_read_stamps
fn.restype, fn.argtypes = C.c_int, [C.c_void_p]
names = {0: ["start", "panel+C+first chunk", "K loop", "C apply + Y out", "-", "gram (MFMA) + atomics"],
         1: ["start", "load+scale", "cholesky", "inverse", "write C"],
         2: ["start", "load + H + symmetrise", "jacobi", "-", "sort + S' (2 mm96)", "cholesky", "inverse", "C = W D^-1 L^-T + write"]}
for n_mult in (8, 8, 8):
    ops.lowrank_chain(G, V, n_mult, 12, ws)
    torch.cuda.synchronize()
    buf = np.zeros(3 * 64, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    st = buf.reshape(3, 64).astype(np.int64)
    sweeps = int(ops.lowrank_chain_info(ws, L, Dt)[0, 1])
    for k, nm in names.items():
        n = len(nm)
        d = [(st[k, i] - st[k, i - 1]) / 100.0 for i in range(1, n if k != 2 else 7)]
        if k != 2:
            d = [v for a, v in zip(nm[1:], d) if a != "-"]
        lab = [x for x in nm[1:] if x != "-"]
        print(("lr_mult", "lr_orth", "lr_ritz")[k], f"total {(st[k, (n - 1) if k != 2 else 6] - st[k, 0]) / 100.0:.1f} us:",
              ", ".join(f"{a} {v:.1f}" for a, v in zip(lab, d)), f"(jacobi sweeps {sweeps})" if k == 2 else "")
    print()
