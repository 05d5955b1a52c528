#!/usr/bin/env python3
"""Teacher forward time per image as a function of the images per call (tile-round quantisation of the N = 768 GEMMs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import vit
dev = "cuda:0"
torch.manual_seed(0)
t = vit.create_model("deit_base_distilled_patch16_224", num_classes=1000).to(dev).eval()
for p in t.parameters():
    p.requires_grad = False
for B in (256, 512, 768, 1024):
    x = torch.randn(B, 3, 224, 224, device=dev)
    with torch.no_grad():
        for _ in range(2):
            t.forward_with_taps(x, (0, 1, 11))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            t.forward_with_taps(x, (0, 1, 11))
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"B={B}: {ms:.2f} ms per call, {ms / B * 256:.2f} ms per 256 images", flush=True)
