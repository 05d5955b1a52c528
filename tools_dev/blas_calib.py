"""Dev calibration only: what the vendor GEMM (hipBLASLt through torch.matmul) reaches on the teacher shapes, plain bf16 out."""
import torch
for name, M, N, K in [("t_qkv", 50688, 2304, 768), ("t_fc1", 50688, 3072, 768), ("t_fc2", 50688, 768, 3072), ("t_proj", 50688, 768, 768),
                      ("big", 8192, 8192, 8192)]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): c = a @ b.t()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:7s} M={M} N={N} K={K}  {us:8.1f} us  {2.0*M*N*K/us*1e-6:7.1f} TFLOP/s")
