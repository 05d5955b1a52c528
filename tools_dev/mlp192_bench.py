#!/usr/bin/env python3
"""A/B timing of the fused MLP-branch kernels against the launch sequences they replace, at the headline shape (bs 256 x 197 tokens,
D = 192, hidden 768).  HIP events on torch's current stream; prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ops  # noqa: E402
from deltakd_amd.ffi import IDENT  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3        # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dev = torch.device("cuda:0")
    M, D, Hd, rps = B * 197, 192, 768, 197
    Mp = (M + 15) // 16 * 16
    g = torch.Generator(device=dev).manual_seed(0)
    x1 = torch.randn(M, D, device=dev, generator=g)
    ln_w, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    w1 = (torch.randn(Hd, D, device=dev, generator=g) * 0.05).to(BF16)
    w2 = (torch.randn(D, Hd, device=dev, generator=g) * 0.05).to(BF16)
    w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
    b1, b2 = torch.zeros(Hd, device=dev), torch.zeros(D, device=dev)
    sc = (torch.rand(B, device=dev, generator=g) < 0.9).float() / 0.9
    x2 = torch.empty_like(x1)
    tap = torch.empty(M, D, device=dev, dtype=BF16)
    res = {"M": M}

    # ---- forward
    def fused_fwd():
        return ops.mlp192_fwd(x1, ln_w, ln_b, w1, b1, w2t, b2, rowscale=sc, rows_per_sample=rps, want_tap=True, out=x2)
    y2 = torch.empty(M, D, device=dev, dtype=BF16)
    pre = torch.empty(M, Hd, device=dev, dtype=BF16)
    h = torch.empty(M, Hd, device=dev, dtype=BF16)

    def unfused_fwd():
        y, mean, rstd = ops.layernorm_fwd(x1, ln_w, ln_b)
        ops.gemm_nt(y, w1, out=h, bias=b1, gelu=True, preact=pre)
        ops.gemm_nt(h, w2, out=x2, bias=b2, resid=x1, rowscale=sc, rows_per_sample=rps, tap=tap)
    res["fwd_fused_us"] = timeit(fused_fwd)
    only = bool(os.environ.get("MLP_BENCH_FUSED_ONLY"))
    res["fwd_unfused_us"] = None if only else timeit(unfused_fwd)
    fwd_bytes = M * (D * 4 * 2 + D * 2 * 2 + Hd * 2 * 2)       # x1 in, x2 out, y2 + tap, pre + h
    res["fwd_fused_algorithmic_GBps"] = fwd_bytes / res["fwd_fused_us"] / 1e3
    res["fwd_fused_TFLOPs"] = 4.0 * M * D * Hd / res["fwd_fused_us"] / 1e6

    # ---- backward
    fw = fused_fwd()
    torch.cuda.synchronize()
    gbuf = torch.randn(M, D, device=dev, generator=g)
    gtap = torch.randn(M, D, device=dev, generator=g).to(BF16)
    d_w, d_b = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    s1 = sc

    def fused_bwd():
        return ops.mlp192_bwd(gbuf, fw["pre"], w2t, w1, x1, ln_w, fw["mean"], fw["rstd"], d_w, d_b, gtap=gtap, s2=sc, s1=s1, rows_per_sample=rps)
    unfused_fwd()
    _, mean_u, rstd_u = ops.layernorm_fwd(x1, ln_w, ln_b)
    ws = torch.empty(ops.lib().dkd_layernorm_bwd_workspace_bytes(M, D) // 4, device=dev, dtype=F32)
    dH = torch.empty(M, Hd, device=dev, dtype=BF16)
    cast = torch.empty(M, D, device=dev, dtype=BF16)

    def unfused_bwd():
        dF = ops.scale_cast_bf16(gbuf, rowscale=sc, rows_per_sample=rps, add=gtap)
        ops.gemm_nt(dF, w2t, out=dH, dgelu=True, preact=pre)
        ops.gemm_nt_lnbwd(dH, w1t, x1, ln_w, mean_u, rstd_u, gbuf, d_w, d_b, ws, cast_out=cast, rowscale=s1, rows_per_sample=rps)
    res["bwd_fused_us"] = timeit(fused_bwd)
    gbuf.normal_(generator=g)
    res["bwd_unfused_us"] = None if only else timeit(unfused_bwd)
    res["lib"] = os.environ.get("DKD_LIB", "default")
    bwd_bytes = M * (D * 4 * 3 + D * 2 * 3 + Hd * 2 * 2)       # g in/out, x1; gtap, dF, cast; pre in, dH out
    res["bwd_fused_algorithmic_GBps"] = bwd_bytes / res["bwd_fused_us"] / 1e3
    res["bwd_fused_TFLOPs"] = 4.0 * M * D * Hd / res["bwd_fused_us"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
