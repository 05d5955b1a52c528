"""The fused qkv projection + attention kernel of the D = 192 student (csrc/attn192.hip, dkd_attn192_fwd) against fp32 torch and against
the two launches it replaces (dkd_gemm_nt with bias + dkd_attn_fwd), through the C ABI."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def ops():
    from deltakd_amd import ops as o
    o.lib()
    return o


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev())


def close(got, ref, rel, what=""):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-12
    assert math.isfinite(err) and err <= rel * scale, f"{what}: max abs err {err:.4e} vs scale {scale:.4e} (rel {err/scale:.3e} > {rel})"


@pytest.mark.parametrize("B,N", [(2, 17), (3, 65), (5, 197), (4, 198), (3, 208), (2, 16), (300, 197), (1, 1)])
def test_attn192_fwd(ops, B, N):
    """qkv (bf16, rounded once from the fp32 accumulator like the GEMM's epilogue), the attention output and the log-sum-exp:
    against fp32 torch on the bf16 operands (qkv 1e-2: bf16 output rounding; out 2e-2: it also sees the rounded q, k, v and P) and
    against the unfused launches (same arithmetic, different summation order).  B = 300 exceeds the CU count (the persistent loop over
    samples); N = 208 fills the last key tile, N = 16 / 17 / 65 leave most key tiles to the padding mask, N = 1 is a single token."""
    H, D = 3, 192
    y1 = rnd(B * N, D, seed=1).to(BF16)
    w = rnd(3 * D, D, scale=D ** -0.5, seed=2).to(BF16)
    bias = rnd(3 * D, scale=0.5, seed=3)
    qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
    ref_qkv = y1.float() @ w.float().t() + bias
    close(qkv, ref_qkv, 1e-2, "qkv")
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)          # attention reference on the kernel's own (rounded) qkv
    s = (q @ k.transpose(-1, -2)) * 0.125
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * N, D)
    close(out, ref, 1.5e-2, "attention out")
    close(lse, torch.logsumexp(s, -1), 1e-3, "lse")
    # the launches it replaces
    qkv_u = ops.gemm_nt(y1, w, bias=bias)
    out_u, lse_u = ops.attn_fwd(qkv_u, B, N, H)
    close(qkv, qkv_u, 8e-3, "qkv vs the GEMM launch")                            # (one bf16 ulp where the fp32 sums round differently)
    close(out, out_u, 1.5e-2, "out vs the attention launch")
    close(lse, lse_u.view(B, H, N), 1e-3, "lse vs the attention launch")


def test_attn192_refuses_what_it_does_not_take(ops):
    from deltakd_amd import ffi
    y1 = rnd(209, 192).to(BF16)
    w = rnd(576, 192).to(BF16)
    b = rnd(576)
    qkv = torch.empty(209, 576, device=dev(), dtype=BF16)
    o = torch.empty(209, 192, device=dev(), dtype=BF16)
    rc = ffi.lib().dkd_attn192_fwd(ffi.ptr(y1), ffi.ptr(w), ffi.ptr(b), ffi.ptr(qkv), ffi.ptr(o), None, 1, 209, ffi.stream())
    assert rc != 0 and b"208" in ffi.lib().dkd_last_error()


@pytest.mark.parametrize("B,N", [(2, 17), (3, 65), (5, 197), (4, 198), (3, 208), (2, 16), (300, 197), (2, 8)])
def test_attn192_bwd(ops, B, N):
    """proj dgrad + attention backward in one launch (csrc/attn192_bwd.hip, dkd_attn192_bwd) against fp32 torch autograd through
    o = softmax(q k^T / 8) v, y = o Wproj^T on the kernel's own saved (bf16) q, k, v, and against the two launches it replaces
    (dkd_gemm_nt(dy, proj_wt) -> bf16 dO -> dkd_attn_bwd).  dq / dk / dv are compared per part, relative to the part's largest element:
    3e-2 against fp32 (bf16 operands: q, k, v, P, dS, dO all rounded), 2e-2 against the unfused launches (which round dO to bf16 in
    memory; here it is rounded once on its way into LDS -- the same rounding -- so the two differ by summation order only).  Shapes as
    in the forward test: B = 300 runs the persistent loop over samples; N = 208 fills the last tile; N = 8 / 16 / 17 are mostly padding."""
    H, D = 3, 192
    y1 = rnd(B * N, D, seed=1).to(BF16)
    w = rnd(3 * D, D, scale=D ** -0.5, seed=2).to(BF16)
    bias = rnd(3 * D, scale=0.5, seed=3)
    wp = rnd(D, D, scale=D ** -0.5, seed=4).to(BF16)                 # proj.weight [out, in]
    wpt = wp.t().contiguous()                                        # what the dgrad reads: proj.weight^T
    dy = rnd(B * N, D, seed=5).to(BF16)
    qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
    got = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)
    torch.cuda.synchronize()
    # fp32 autograd on the saved bf16 q, k, v
    qkv32 = qkv.float().clone().requires_grad_(True)
    q, k, v = qkv32.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    o32 = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v
    yy = o32.transpose(1, 2).reshape(B * N, D) @ wp.float().t()
    yy.backward(dy.float())
    ref = qkv32.grad
    for part, name in enumerate(("dq", "dk", "dv")):
        close(got[:, part * D:(part + 1) * D], ref[:, part * D:(part + 1) * D], 3e-2, f"{name} vs fp32 autograd")
    # the launches it replaces
    d_o = ops.gemm_nt(dy, wpt)
    unf = ops.attn_bwd(qkv, out, d_o, lse.view(B, H, N), B, N, H)
    for part, name in enumerate(("dq", "dk", "dv")):
        close(got[:, part * D:(part + 1) * D], unf[:, part * D:(part + 1) * D], 2e-2, f"{name} vs proj dgrad + attention backward launches")
    # linearity in dy: twice the upstream gradient, twice the result (exact in bf16: a power of two)
    got2 = ops.attn192_bwd((dy.float() * 2).to(BF16), wpt, qkv, out, lse, B, N)
    assert torch.equal(got2.float(), got.float() * 2)


@pytest.mark.parametrize("B,N", [(2, 17), (5, 197), (4, 198), (3, 208), (300, 197), (2, 8)])
def test_attn192_bwd_with_the_layernorm_backward(ops, B, N):
    """The same launch carried through to the branch's input (dkd_attn192_bwd with qkv_wt): dT = dqkv Wqkv stays on chip and the kernel's
    epilogue is norm1's backward -- g += LN'(dT), d_ln_w += sum dT xhat, d_ln_b += sum dT.  Checked against fp32 torch on the kernel's own
    (bf16) dqkv: dx to 2e-3 of the largest element (dT is an fp32 accumulation of bf16 products on both sides), the parameter gradients to
    2e-3; and against the launch it replaces (dkd_gemm_nt_lnbwd on that dqkv): same arithmetic, other summation order.  dqkv itself must be
    what the launch without the LayerNorm part writes (bit for bit: the head loop is the same code)."""
    H, D = 3, 192
    y1 = rnd(B * N, D, seed=1).to(BF16)
    w = rnd(3 * D, D, scale=D ** -0.5, seed=2).to(BF16)                  # qkv.weight [576, 192]
    bias = rnd(3 * D, scale=0.5, seed=3)
    wpt = rnd(D, D, scale=D ** -0.5, seed=4).to(BF16)
    dy = rnd(B * N, D, seed=5).to(BF16)
    x = rnd(B * N, D, seed=6, scale=1.5) + 0.2
    gamma = 1.0 + 0.2 * rnd(D, seed=7)
    g0 = rnd(B * N, D, seed=8, scale=0.5)
    mean = x.mean(1).contiguous()
    rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6).contiguous()
    qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
    wqt = w.t().contiguous()                                             # [192, 576] = qkv.weight^T
    g = g0.clone()
    dgam, dbet = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    dqkv = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N, qkv_wt=wqt, x=x, ln_w=gamma, mean=mean, rstd=rstd, g=g, d_ln_w=dgam, d_ln_b=dbet)
    torch.cuda.synchronize()
    assert torch.equal(dqkv, ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)), "dqkv must not depend on the LayerNorm part"
    # fp32 reference on the kernel's own dqkv
    dT = dqkv.float() @ w.float()                                        # [M, 192]
    xh = (x - mean[:, None]) * rstd[:, None]
    gy = dT * gamma
    dx = rstd[:, None] * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True))
    close(g - g0, dx, 2e-3, "dx vs fp32")
    close(dgam, (dT * xh).sum(0), 2e-3, "d_ln_w vs fp32")
    close(dbet, dT.sum(0), 2e-3, "d_ln_b vs fp32")
    # the launch it replaces
    g_u = g0.clone()
    dgam_u, dbet_u = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    ws = torch.empty(ops.lib().dkd_layernorm_bwd_workspace_bytes(B * N, D) // 4, device=dev())
    ops.gemm_nt_lnbwd(dqkv, wqt, x, gamma, mean, rstd, g_u, dgam_u, dbet_u, ws)
    torch.cuda.synchronize()
    close(g, g_u, 1e-3, "dx vs dkd_gemm_nt_lnbwd")
    close(dgam, dgam_u, 1e-3, "d_ln_w vs dkd_gemm_nt_lnbwd")
    close(dbet, dbet_u, 1e-3, "d_ln_b vs dkd_gemm_nt_lnbwd")


def test_attn192_bwd_refuses_what_it_does_not_take(ops):
    from deltakd_amd import ffi
    t = torch.zeros(209 * 576, device=dev(), dtype=BF16)
    lse = torch.zeros(3 * 209, device=dev())
    none = [None] * 9
    rc = ffi.lib().dkd_attn192_bwd(ffi.ptr(t), ffi.ptr(t), ffi.ptr(t), ffi.ptr(t), ffi.ptr(lse), ffi.ptr(t), *none, 1, 209, ffi.stream())
    assert rc != 0 and b"208" in ffi.lib().dkd_last_error()
    rc = ffi.lib().dkd_attn192_bwd(ffi.ptr(t), ffi.ptr(t), ffi.ptr(t), ffi.ptr(t), ffi.ptr(lse), ffi.ptr(t), *none, 1, 4, ffi.stream())
    assert rc != 0
    # the LayerNorm part needs all of its operands
    rc = ffi.lib().dkd_attn192_bwd(ffi.ptr(t), ffi.ptr(t), ffi.ptr(t), ffi.ptr(t), ffi.ptr(lse), ffi.ptr(t), ffi.ptr(t), *([None] * 8), 1, 197, ffi.stream())
    assert rc != 0 and b"qkv_wt" in ffi.lib().dkd_last_error()


def test_attn192_bwd_does_not_read_lds_it_never_wrote(ops):
    """Regression (round 4): the statistics rows of the padding query tile were never written, so whatever an EARLIER kernel had left in
    that part of LDS was read -- a NaN there reached dK through 0 x NaN.  Poison the CUs' LDS with NaNs (a kernel that fills 160 KiB per
    workgroup), then run the backward: finite, and equal to the run before the poisoning."""
    B, N, H, D = 256, 197, 3, 192
    y1 = rnd(B * N, D, seed=1).to(BF16)
    w = rnd(3 * D, D, scale=D ** -0.5, seed=2).to(BF16)
    bias = rnd(3 * D, scale=0.5, seed=3)
    wpt = rnd(D, D, scale=D ** -0.5, seed=4).to(BF16)
    dy = rnd(B * N, D, seed=5).to(BF16)
    qkv, out, lse = ops.attn192_fwd(y1, w, bias, B, N)
    ref = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)
    # the wide GEMM kernel owns all 160 KiB of LDS per workgroup: NaN operands leave NaN tiles behind in it on every CU
    a = torch.full((256 * 256, 768), float("nan"), device=dev(), dtype=BF16)
    b = torch.full((2304, 768), float("nan"), device=dev(), dtype=BF16)
    ops.gemm_nt(a, b)
    got = ops.attn192_bwd(dy, wpt, qkv, out, lse, B, N)
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all(), "NaN from LDS contents of an earlier kernel"
    assert torch.equal(got, ref)


@pytest.mark.parametrize("B,N,scaled", [(2, 17, True), (5, 197, True), (4, 198, False), (3, 208, True), (300, 197, True), (1, 1, False)])
def test_attn192_fwd_with_proj_and_residual(ops, B, N, scaled):
    """dkd_attn192_fwd_proj: the same launch carried through proj + DropPath scale + residual.  qkv / out / lse must be what dkd_attn192_fwd
    writes (bit for bit: the head loop is the same code); x1 against fp32 torch on the kernel's own (bf16) attention output: 2e-3 of the
    largest element (fp32 accumulation of bf16 products on both sides, fp32 residual), and against the launch it replaces."""
    H, D = 3, 192
    y1 = rnd(B * N, D, seed=1).to(BF16)
    w = rnd(3 * D, D, scale=D ** -0.5, seed=2).to(BF16)
    bias = rnd(3 * D, scale=0.5, seed=3)
    wp = rnd(D, D, scale=D ** -0.5, seed=4).to(BF16)
    bp = rnd(D, seed=5, scale=0.3)
    x = rnd(B * N, D, seed=6, scale=2.0)
    s1 = (torch.rand(B, generator=torch.Generator().manual_seed(7)) > 0.3).float().to(dev()) / 0.7 if scaled else None
    qkv0, out0, lse0 = ops.attn192_fwd(y1, w, bias, B, N)
    qkv, out, lse, x1 = ops.attn192_fwd_proj(y1, w, bias, wp, bp, x, B, N, rowscale=s1)
    torch.cuda.synchronize()
    assert torch.equal(qkv, qkv0) and torch.equal(out, out0) and torch.equal(lse, lse0)
    sc = (s1 if s1 is not None else torch.ones(B, device=dev())).repeat_interleave(N)[:, None]
    ref = x + sc * (out.float() @ wp.float().t() + bp)
    close(x1, ref, 2e-3, "x1 vs fp32")
    x1_u = ops.gemm_nt(out, wp, bias=bp, resid=x, rowscale=s1, rows_per_sample=N, out_f32=True)
    close(x1, x1_u, 1e-3, "x1 vs the proj launch")
    # in place (x1 aliases x), as the training block runs it when the residual stream is updated in place
    from deltakd_amd import ffi
    xa = x.clone()
    ffi.check(ffi.lib().dkd_attn192_fwd_proj(ffi.ptr(y1), ffi.ptr(w), ffi.ptr(bias), ffi.ptr(qkv), ffi.ptr(out), ffi.ptr(lse), ffi.ptr(wp), ffi.ptr(bp),
                                             ffi.ptr(xa), ffi.ptr(s1), ffi.ptr(xa), B, N, ffi.stream()), "in place")
    torch.cuda.synchronize()
    assert torch.equal(xa, x1)
