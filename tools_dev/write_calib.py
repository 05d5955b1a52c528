"""dev: streaming write / copy rates for the GEMM output sizes."""
import torch
for mb in (233, 311, 155):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    y = torch.randn(n, device="cuda").to(torch.bfloat16)
    for name, fn in (("fill", lambda: x.fill_(1.0)), ("copy", lambda: x.copy_(y))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name} {mb} MB: {us:7.1f} us  write {mb/us*1e-6*1e6/1e0:6.2f} TB/s" .replace("TB/s", "MB/us = TB/s"))
