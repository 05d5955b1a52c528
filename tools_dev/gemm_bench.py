"""Micro-benchmark of dkd_gemm_nt / dkd_gemm_tn on the shapes of the headline step (teacher DeiT-base, student DeiT-tiny)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ops

dev = "cuda:0"
BF16 = torch.bfloat16
shapes = [  # (name, M, N, K, epilogue)
    ("t_qkv", 50688, 2304, 768, "bias"), ("t_proj", 50688, 768, 768, "resid"), ("t_fc1", 50688, 3072, 768, "gelu"),
    ("t_fc2", 50688, 768, 3072, "resid"), ("t_projb", 50688, 768, 768, "bias"), ("t_fc2b", 50688, 768, 3072, "bias"), ("t_fc1b", 50688, 3072, 768, "bias"), ("s_qkv", 50432, 576, 192, "bias"), ("s_fc1", 50432, 768, 192, "gelu"), ("s_fc1p", 50432, 768, 192, "gelu_pre"), ("s_fc2d", 50432, 768, 192, "dgelu"),
    ("s_fc2", 50432, 192, 768, "resid"), ("s_proj", 50432, 192, 192, "resid"),
]
only = sys.argv[1:] 
reps = 20
res = {}
for name, M, N, K, epi in shapes:
    if only and name not in only: continue
    pad = int(os.environ.get("GB_PAD", "0"))      # extra elements per row of both operands (row stride K + pad): L2 channel spread
    a = torch.randn(M, K + pad, device=dev).to(BF16)[:, :K]
    b = (torch.randn(N, K + pad, device=dev) * 0.05).to(BF16)[:, :K]
    bias = torch.randn(N, device=dev)
    kw = dict(bias=bias)
    if epi == "gelu":
        kw.update(gelu=True)
    if epi == "gelu_pre":
        kw.update(gelu=True, preact=torch.empty(M, N, device=dev, dtype=BF16))
    if epi == "dgelu":
        kw = dict(dgelu=True, preact=torch.randn(M, N, device=dev).to(BF16))
    if epi == "resid":
        kw.update(resid=torch.randn(M, N, device=dev), out_f32=True)
    out = ops.gemm_nt(a, b, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_nt(a, b, out=out, **kw)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * M * N * K / ms / 1e9
    res[name] = (ms * 1e3, tf)
    last = (epi, kw, a, b, bias, out)
    print(f"{name:8s} M={M} N={N} K={K} {epi:6s} {ms*1e3:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
# correctness spot check of the last shape
if res:
    epi, kw, a_, b_, bias, out_ = last
    ref = a_[-512:].float() @ b_.float().t() + (0 if epi == "dgelu" else bias)
    if epi in ("gelu", "gelu_pre"):
        ref = torch.nn.functional.gelu(ref)
    if epi == "dgelu":
        p_ = kw["preact"][-512:].float().requires_grad_(True)
        torch.nn.functional.gelu(p_).sum().backward()
        ref = ref * p_.grad
    if epi == "resid":
        ref = ref + kw["resid"][-512:]
    err = (out_[-512:].float() - ref).abs().max().item() / ref.abs().max().item()
    print("spot-check rel err", err)

print("-- wgrad (gemm_tn) on the student shapes")
for name, M, N1, N2 in (("s_fc2_w", 50432, 192, 768), ("s_fc1_w", 50432, 768, 192), ("s_proj_w", 50432, 192, 192), ("s_qkv_w", 50432, 576, 192),
                        ("gram", 50176, 768, 768)):
    if only and name not in only: continue
    a = torch.randn(M, N1, device=dev).to(BF16)
    b = torch.randn(M, N2, device=dev).to(BF16)
    out = torch.zeros(N1, N2, device=dev)
    cs = torch.zeros(N1, device=dev)
    ops.gemm_tn(a, b, out, colsum=cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm_tn(a, b, out, colsum=cs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:9s} M={M} N1={N1} N2={N2} {ms*1e3:8.1f} us  {2.0*M*N1*N2/ms/1e9:7.1f} TFLOP/s  {(M*(N1+N2)*2)/ms/1e6:7.1f} GB/s min-traffic", flush=True)
ref = a.float().t() @ b.float() if a.shape[0] == b.shape[0] else None
if ref is not None and out.shape == ref.shape:
  print("tn spot-check rel err", ((out / (reps + 1)) - ref).abs().max().item() / ref.abs().max().item())
