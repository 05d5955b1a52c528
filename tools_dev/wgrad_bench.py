"""Time the grouped weight-gradient launch of `nb` student blocks (4 problems each) at the headline shape.
usage: python tools_dev/wgrad_bench.py [blocks=6] [iters=20]      (knobs: DKD_TN_GROUP_WIDE / DKD_TN_GROUP_SLOTS / DKD_TN_GROUP_BLOCKS)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deltakd_amd import ffi
from deltakd_amd.ffi import IDENT

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 6
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
M, D, Hd = 256 * 197, 192, 768
BF = torch.bfloat16
keep, probs = [], (ffi.TnProblem * (4 * nb))()
k = 0
for b in range(nb):
    for n1, n2 in ((D, Hd), (Hd, D), (D, D), (3 * D, D)):
        a = torch.randn(M, n1, device=dev).to(BF)
        bb = torch.randn(M, n2, device=dev).to(BF)
        c = torch.zeros(n1, n2, device=dev)
        cs = torch.zeros(n1, device=dev)
        keep += [a, bb, c, cs]
        q = probs[k]
        q.A, q.B, q.C, q.a_colsum, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc = a.data_ptr(), bb.data_ptr(), c.data_ptr(), cs.data_ptr(), M, n1, n2, n1, n2, n2
        q.amap = q.bmap = IDENT
        k += 1
lib = ffi.lib()


def run():
    ffi.check(lib.dkd_block_wgrad_group(ffi.C.cast(probs, ffi.C.c_void_p), 4 * nb, ffi.stream()), "wgrad")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
flop = 2.0 * M * D * (Hd + Hd + D + 3 * D) * nb
print(f"blocks {nb}: {us:.1f} us per launch = {us / nb:.1f} us per block, {flop / us / 1e6:.0f} TFLOP/s "
      f"[WIDE={os.environ.get('DKD_TN_GROUP_WIDE', '1')} SLOTS={os.environ.get('DKD_TN_GROUP_SLOTS', '-')}]")
# correctness of the last launch against torch on one problem of the first block
ref = (keep[0].float().t() @ keep[1].float()) * (iters + 3)
got = keep[2]
print("rel err fc2 wgrad", ((got - ref).norm() / ref.norm()).item())
