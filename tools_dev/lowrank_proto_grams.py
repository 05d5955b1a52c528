"""CPU side of the LRKD-solver prototyping (round 5): Gram matrices of REAL teacher taps on never-repeating, shifting batches.

Runs the oracle's deit_base_distilled (oracle/vit_ref.py, seed 42 = the GPU tests' teacher) on the shifting batches of
tests/test_fullsize_gpu.py::_shifting_batches (same generator, CPU, batch 64 instead of 256), takes the block.mlp taps of blocks
0, 1, 11, strips the two prefix tokens, rounds to bf16 (what the device taps are) and writes G = T^T T (float64 accumulate, stored
f32 like dkd_gram's output) to an .npz.  tools_dev/lowrank_proto_algo.py iterates on solver variants against these in numpy float32.

    python tools_dev/lowrank_proto_grams.py /tmp/lrkd_grams.npz [n_batches] [batch]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_ref  # noqa: E402


def shifting_batches(n, seed, B, jump_at=None):
    g = torch.Generator().manual_seed(seed)

    def bank():
        p = torch.randn(12, 3, 28, 28, generator=g)
        return torch.nn.functional.interpolate(p, size=224, mode="bilinear", align_corners=False)
    protos = bank()
    jumped = bank() * 3.0
    for t in range(n):
        pb = jumped if jump_at is not None and t >= jump_at else protos
        temp = 0.5 + 2.5 * ((t * 7) % 10) / 10.0
        mix = torch.softmax(torch.randn(B, 12, generator=g) * temp, 1)
        contrast = 0.6 + 0.25 * (t % 5)
        noise = 0.3 + 0.15 * ((t * 3) % 7)
        x = contrast * torch.einsum("bc,cdhw->bdhw", mix, pb)
        x += noise * torch.randn(B, 3, 224, 224, generator=g)
        yield x


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    torch.manual_seed(42)
    t = vit_ref.create_model_ref("deit_base_distilled_patch16_224", num_classes=1000).eval()
    taps = {}
    hooks = [t.blocks[i].mlp.register_forward_hook(lambda m, i_, o, k=i: taps.__setitem__(k, o)) for i in (0, 1, 11)]
    grams = []
    for call, x in enumerate(shifting_batches(n, seed=123, B=B, jump_at=n - 3)):
        t0 = time.time()
        with torch.no_grad():
            t(x)
        gl = []
        for i in (0, 1, 11):
            T = taps[i][:, 2:].reshape(-1, 768).to(torch.bfloat16).double()
            gl.append((T.t() @ T).float().numpy())
        grams.append(np.stack(gl))
        print(f"batch {call}: {time.time() - t0:.1f} s", flush=True)
    for h in hooks:
        h.remove()
    np.savez(out, G=np.stack(grams))


if __name__ == "__main__":
    main()
