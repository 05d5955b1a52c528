#!/usr/bin/env python3
"""Reads the in-kernel stamps of an ablation build with bit 256 set (csrc/mlp192.hip) after ONE launch of the fused forward / backward at
the headline shape and prints the phase durations (median over waves), in shader cycles and microseconds."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deltakd_amd import ffi, ops  # noqa: E402

BF16 = torch.bfloat16


def read(nblk):
    buf = np.zeros(1024 * 8 * 64, dtype=np.uint64)
    fn = ffi.lib().dkd_mlp192_read_stamps
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int64]
    assert fn(buf.ctypes.data, buf.nbytes) == 0
    return buf.reshape(1024, 8, 64)[:nblk].astype(np.int64)


def report(st, ksn, tag):
    dur = st[:, :, 61] - st[:, :, 0]
    rt = (st[:, :, 63] - st[:, :, 62]).astype(np.float64) / 100.0          # us
    clk = np.median(dur / np.maximum(rt, 1e-9)) / 1e3                       # GHz
    pro = np.median(st[:, :, 1] - st[:, :, 0])
    fill = np.median(st[:, :, 2] - st[:, :, 1])
    steps = st[:, :, 3:2 + ksn] - st[:, :, 2:1 + ksn]
    loop = np.median(st[:, :, 60] - st[:, :, 2])
    epi = np.median(st[:, :, 61] - st[:, :, 60])
    t0 = st[:, :, 0].min()
    span = (st[:, :, 61].max() - t0)
    out = {"tag": tag, "clock_GHz": round(float(clk), 3), "prologue_cyc": int(pro), "ring_fill_cyc": int(fill), "loop_cyc": int(loop),
           "kstep_cyc_median": int(np.median(steps)), "kstep_cyc_p90": int(np.percentile(steps, 90)), "epilogue_cyc": int(epi),
           "wave_total_cyc_median": int(np.median(dur)), "first_start_to_last_end_cyc": int(span),
           "us": {k: round(v / clk / 1e3, 2) for k, v in (("prologue", pro), ("ring_fill", fill), ("loop", loop), ("epilogue", epi),
                                                          ("span", span))},
           "start_skew_cyc_p90": int(np.percentile(st[:, :, 0] - t0, 90)),
           "kstep_by_index": [int(np.median(steps[:, :, i])) for i in range(steps.shape[2])]}
    print(json.dumps(out))


def main():
    dev = torch.device("cuda:0")
    B = 256
    M, D, Hd, rps = B * 197, 192, 768, 197
    g = torch.Generator(device=dev).manual_seed(0)
    x1 = torch.randn(M, D, device=dev, generator=g)
    ln_w, ln_b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    w1 = (torch.randn(Hd, D, device=dev, generator=g) * 0.05).to(BF16)
    w2t = (torch.randn(Hd, D, device=dev, generator=g) * 0.05).to(BF16)
    b1, b2 = torch.zeros(Hd, device=dev), torch.zeros(D, device=dev)
    sc = (torch.rand(B, device=dev, generator=g) < 0.9).float() / 0.9
    for _ in range(3):
        fw = ops.mlp192_fwd(x1, ln_w, ln_b, w1, b1, w2t, b2, rowscale=sc, rows_per_sample=rps, want_tap=True)
    torch.cuda.synchronize()
    report(read(256), Hd // 32, "fwd")
    gbuf = torch.randn(M, D, device=dev, generator=g)
    gtap = torch.randn(M, D, device=dev, generator=g).to(BF16)
    d_w, d_b = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    for _ in range(3):
        ops.mlp192_bwd(gbuf, fw["pre"], w2t, w1, x1, ln_w, fw["mean"], fw["rstd"], d_w, d_b, gtap=gtap, s2=sc, s1=sc, rows_per_sample=rps)
    torch.cuda.synchronize()
    report(read(256), Hd // 32, "bwd")


if __name__ == "__main__":
    main()
