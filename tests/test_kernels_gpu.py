"""Per-kernel parity: every dkd_* entry point, called through the C ABI, against plain torch fp32 on the same inputs
(bf16 operands are rounded once; the reference then computes in fp32).  Tolerances are stated per test."""
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


# bf16 output rounding is 2^-9 relative; fp32 accumulation over K<=3072 of bf16 products adds ~1e-6: 1e-2 of the
# output scale is a loose-enough bound that still catches any indexing error.
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 192, 192), (1000, 576, 192), (129, 1000, 192), (513, 64, 768),
                                   (130, 16, 64), (4 * 197, 768, 3072), (64, 10, 64)])
def test_gemm_nt_plain(ops, M, N, K):
    a = rnd(M, K, seed=1).to(BF16)
    b = rnd(N, K, seed=2).to(BF16)
    ref = a.float() @ b.float().t()
    close(ops.gemm_nt(a, b), ref, 1e-2, "bf16 out")
    close(ops.gemm_nt(a, b, out_f32=True), ref, 2e-5 * math.sqrt(K), "f32 out")


@pytest.mark.parametrize("M,K,fused_cast", [(300, 768, True), (50 * 197, 576, False), (130, 64, True)])
def test_gemm_nt_lnbwd(ops, M, K, fused_cast):
    """dkd_gemm_nt_lnbwd (dgrad GEMM with the LayerNorm backward as its epilogue, D = 192) against torch autograd through
    F.layer_norm on the same operands: dx accumulates, dgamma / dbeta accumulate, optional row-scaled bf16 copy of the updated dx."""
    import torch.nn.functional as F
    D, rps = 192, 10
    a = rnd(M, K, seed=201).to(BF16)
    w = rnd(D, K, scale=0.05, seed=202).to(BF16)
    x = rnd(M, D, seed=203, scale=2.0) + 0.5
    gamma = 1.0 + 0.1 * rnd(D, seed=204)
    dx0 = rnd(M, D, seed=205)
    xr = x.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = torch.zeros(D, device=dev(), requires_grad=True)
    y = F.layer_norm(xr, (D,), gr, br, eps=1e-6)
    dT = a.float() @ w.float().t()
    y.backward(dT)
    mu = x.mean(1)
    rstd = torch.rsqrt(x.var(1, unbiased=False) + 1e-6)
    dx = dx0.clone()
    dgamma = torch.full((D,), 0.5, device=dev())
    dbeta = torch.full((D,), -0.25, device=dev())
    ws = torch.empty(2 * D * ((M + 63) // 64), device=dev())
    nsamp = (M + rps - 1) // rps
    scale = (torch.arange(nsamp, device=dev()) % 3).float() * 0.5
    cast = torch.empty(M, D, device=dev(), dtype=BF16) if fused_cast else None
    ops.gemm_nt_lnbwd(a, w, x, gamma, mu.contiguous(), rstd.contiguous(), dx, dgamma, dbeta, ws, cast_out=cast,
                      rowscale=scale if fused_cast else None, rows_per_sample=rps if fused_cast else 0)
    close(dx, dx0 + xr.grad, 2e-5 * math.sqrt(K) * 4, "dx")
    close(dgamma - 0.5, gr.grad, 1e-4, "dgamma")
    close(dbeta + 0.25, br.grad, 1e-4, "dbeta")
    if fused_cast:
        want = (dx0 + xr.grad) * scale[torch.arange(M, device=dev()) // rps][:, None]
        close(cast, want, 1e-2, "row-scaled bf16 copy")


def test_gemm_nt_epilogues(ops):
    from deltakd_amd.ffi import RowMap, strip_map
    B, Nt, D, Hd = 6, 17, 128, 256
    M = B * Nt
    x = rnd(M, D, seed=3).to(BF16)
    w = rnd(Hd, D, scale=0.1, seed=4).to(BF16)
    bias = rnd(Hd, seed=5)
    pre = torch.empty(M, Hd, device=dev(), dtype=BF16)
    h = ops.gemm_nt(x, w, bias=bias, gelu=True, preact=pre)
    ref_pre = x.float() @ w.float().t() + bias
    close(pre, ref_pre, 1e-2, "preact")
    close(h, torch.nn.functional.gelu(ref_pre), 1e-2, "gelu")
    # dgelu epilogue
    dy = rnd(M, D, seed=6).to(BF16)
    wt = w.t().contiguous()          # [D, Hd] -> B operand [N=Hd, K=D] is w itself: dH = dy @ w^T ... use w as B
    dh = ops.gemm_nt(dy, w, dgelu=True, preact=pre)
    xg = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(xg).backward(dy.float() @ w.float().t())
    close(dh, xg.grad, 1e-2, "dgelu")
    # residual + rowscale + tap, f32 out
    w2 = rnd(D, Hd, scale=0.1, seed=7).to(BF16)
    b2 = rnd(D, seed=8)
    resid = rnd(M, D, seed=9)
    scale = torch.tensor([1.0, 0.0, 1.25, 1.0, 0.0, 1.1], device=dev())
    tap = torch.empty(M, D, device=dev(), dtype=BF16)
    out = ops.gemm_nt(h, w2, bias=b2, resid=resid, rowscale=scale, rows_per_sample=Nt, tap=tap, out_f32=True)
    f = h.float() @ w2.float().t() + b2
    close(tap, f, 1e-2, "tap")
    ref = resid + scale.repeat_interleave(Nt)[:, None] * f
    close(out, ref, 1e-2, "resid")
    # A rows through a strip map (drop 1 prefix token per sample), C rows scattered back through a map
    P = Nt - 1
    s = ops.gemm_nt(x, w, M=B * P, amap=strip_map(Nt, 1), out_f32=True)
    xs = x.view(B, Nt, D)[:, 1:].reshape(B * P, D)
    close(s, xs.float() @ w.float().t(), 1e-2, "amap")
    big = torch.zeros(M, Hd, device=dev(), dtype=BF16)
    ops.gemm_nt(xs.contiguous(), w, out=big, cmap=strip_map(Nt, 1))
    exp = torch.zeros(B, Nt, Hd, device=dev())
    exp[:, 1:] = (xs.float() @ w.float().t()).view(B, P, Hd)
    close(big, exp.view(M, Hd), 1e-2, "cmap")
    # broadcast residual (pos_embed) through rmap with gstride 0, relu, accumulate
    pos = rnd(Nt, Hd, seed=10)
    o2 = ops.gemm_nt(xs.contiguous(), w, out=torch.zeros(M, Hd, device=dev()), cmap=strip_map(Nt, 1), resid=pos,
                     rmap=RowMap(P, 0, 1))
    exp2 = torch.zeros(B, Nt, Hd, device=dev())
    exp2[:, 1:] = (xs.float() @ w.float().t()).view(B, P, Hd) + pos[1:]
    close(o2, exp2.view(M, Hd), 1e-2, "pos resid")
    acc = torch.ones(M, Hd, device=dev())
    ops.gemm_nt(x, w, out=acc, relu=True, accumulate=True)
    close(acc, 1 + torch.relu(x.float() @ w.float().t()), 1e-2, "relu+accum")


@pytest.mark.parametrize("M,N1,N2", [(64, 128, 128), (1000, 192, 576), (4 * 196, 768, 192), (777, 16, 64), (300, 10, 64),
                                     (5000, 256, 384)])
def test_gemm_tn(ops, M, N1, N2):
    ld1, ld2 = (N1 + 7) // 8 * 8, (N2 + 7) // 8 * 8
    a = torch.zeros(M, ld1, device=dev(), dtype=BF16)
    b = torch.zeros(M, ld2, device=dev(), dtype=BF16)
    a[:, :N1] = rnd(M, N1, seed=11).to(BF16)
    b[:, :N2] = rnd(M, N2, seed=12).to(BF16)
    out = torch.full((N1, N2), 0.5, device=dev())
    cs = torch.full((N1,), 0.25, device=dev())
    ops.gemm_tn(a, b, out, N1=N1, N2=N2, colsum=cs)
    ref = 0.5 + a[:, :N1].float().t() @ b[:, :N2].float()
    close(out, ref, 3e-5 * math.sqrt(M), "tn")
    close(cs, 0.25 + a[:, :N1].float().sum(0), 1e-5 * math.sqrt(M), "tn fused column sums (bias gradient)")


@pytest.mark.parametrize("N1,N2", [(64, 192), (192, 768), (576, 192), (192, 192), (100, 160)])
def test_gemm_tn_wide_operand_paths(ops, N1, N2):
    """128 x 192 wgrad tiles (the 192-wide operand as B, or as A with swapped roles + transposed store), row maps, bias sums."""
    from deltakd_amd.ffi import strip_map
    B, Nt = 7, 19
    P = Nt - 1
    a = rnd(B * P, (N1 + 7) // 8 * 8, seed=140).to(BF16)
    b = rnd(B * Nt, (N2 + 7) // 8 * 8, seed=141).to(BF16)
    out = torch.full((N1, N2), -0.5, device=dev())
    cs = torch.zeros(N1, device=dev())
    ops.gemm_tn(a, b, out, M=B * P, N1=N1, N2=N2, bmap=strip_map(Nt, 1), colsum=cs)
    bs = b.view(B, Nt, -1)[:, 1:].reshape(B * P, -1)
    close(out, -0.5 + a[:, :N1].float().t() @ bs[:, :N2].float(), 1e-4, "tn wide")
    close(cs, a[:, :N1].float().sum(0), 1e-4, "tn wide bias sums")


def test_gemm_tn_rowmaps(ops):
    from deltakd_amd.ffi import strip_map
    B, Nt, D1, D2 = 5, 18, 64, 128
    a = rnd(B * Nt, D1, seed=13).to(BF16)
    b = rnd(B * (Nt - 2), D2, seed=14).to(BF16)
    out = torch.zeros(D1, D2, device=dev())
    ops.gemm_tn(a, b, out, M=B * (Nt - 2), amap=strip_map(Nt, 2))
    ref = a.view(B, Nt, D1)[:, 2:].reshape(-1, D1).float().t() @ b.float()
    close(out, ref, 1e-4, "tn amap")


@pytest.mark.parametrize("M,N", [(1000, 768), (333, 128), (4 * 198, 200)])
def test_gram_upper_triangle(ops, M, N):
    from deltakd_amd.ffi import strip_map
    a = rnd(M, N, seed=230).to(BF16)
    out = torch.zeros(N, N, device=dev())
    ops.gram(a, out)
    ref = a.float().t() @ a.float()
    close(out, ref, 1e-4, "gram")
    if M == 4 * 198:
        out2 = torch.zeros(N, N, device=dev())
        ops.gram(a, out2, M=4 * 196, amap=strip_map(198, 2))
        sub = a.float().view(4, 198, N)[:, 2:].reshape(-1, N)
        close(out2, sub.t() @ sub, 1e-4, "gram strip map")


def test_gemm_tn_group_matches_single_launches(ops):
    """dkd_gemm_tn_group: four weight gradients (both operand orders of the 192-wide case, a 192 x 192, and a 128-wide shape the
    grouped kernel does not take and must launch on its own) against fp32 torch, with accumulation into non-zero outputs, fused
    bias sums and a token-strip row map."""
    from deltakd_amd.ffi import strip_map
    Bn, Nt, M = 12, 66, 12 * 66
    shapes = [(768, 192), (192, 768), (192, 192), (128, 128)]
    probs, refs = [], []
    for i, (n1, n2) in enumerate(shapes):
        a = rnd(M, n1, seed=200 + i).to(BF16)
        b = rnd(M, n2, seed=210 + i).to(BF16)
        out = torch.full((n1, n2), 0.25, device=dev())
        cs = torch.zeros(n1, device=dev())
        kw = dict(a=a, b=b, out=out, colsum=cs)
        af, bf = a.float(), b.float()
        if i == 0:                                       # gradient of a layer fed by x[:, 2:]: rows through a strip map
            kw.update(M=Bn * (Nt - 2), amap=strip_map(Nt, 2), bmap=strip_map(Nt, 2))
            af = af.view(Bn, Nt, n1)[:, 2:].reshape(-1, n1)
            bf = bf.view(Bn, Nt, n2)[:, 2:].reshape(-1, n2)
        probs.append(kw)
        refs.append((0.25 + af.t() @ bf, af.sum(0)))
    ops.gemm_tn_group(probs)
    for kw, (ref, rcs), shp in zip(probs, refs, shapes):
        close(kw["out"], ref, 1e-4, f"group {shp}")
        close(kw["colsum"], rcs, 1e-4, f"group colsum {shp}")


@pytest.mark.parametrize("n_blocks", [1, 3, 6])
def test_block_wgrad_group_of_several_blocks(ops, n_blocks):
    """dkd_block_wgrad_group with 4, 12 and 24 problems (the weight gradients of 1 / 3 / 6 student blocks: fc2, fc1, proj, qkv shapes at
    D = 192): every tile gets the same number of M splits (26 / 8 / 4 here), each result against fp32 torch; M not a multiple of the
    32-row unit, outputs pre-filled (the launch accumulates)."""
    from deltakd_amd import ffi
    from deltakd_amd.ffi import IDENT
    M, D, Hd = 37 * 197, 192, 768
    probs = (ffi.TnProblem * (4 * n_blocks))()
    keep, refs = [], []
    k = 0
    for b in range(n_blocks):
        for n1, n2 in ((D, Hd), (Hd, D), (D, D), (3 * D, D)):
            a = rnd(M, n1, seed=900 + k).to(BF16)
            bb = rnd(M, n2, seed=950 + k).to(BF16)
            c = torch.full((n1, n2), 0.5, device=dev())
            cs = torch.zeros(n1, device=dev())
            keep.append((a, bb, c, cs))
            refs.append((0.5 + a.float().t() @ bb.float(), a.float().sum(0)))
            q = probs[k]
            q.A, q.B, q.C, q.a_colsum, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc = a.data_ptr(), bb.data_ptr(), c.data_ptr(), cs.data_ptr(), M, n1, n2, n1, n2, n2
            q.amap = q.bmap = IDENT
            k += 1
    ffi.check(ffi.lib().dkd_block_wgrad_group(ffi.C.cast(probs, ffi.C.c_void_p), 4 * n_blocks, ffi.stream()), "wgrad group")
    for i, ((a, bb, c, cs), (ref, rcs)) in enumerate(zip(keep, refs)):
        close(c, ref, 1e-4, f"problem {i} {tuple(c.shape)}")
        close(cs, rcs, 1e-4, f"colsum {i}")
    rc = ffi.lib().dkd_block_wgrad_group(ffi.C.cast(probs, ffi.C.c_void_p), 25, ffi.stream())
    assert rc != 0 and b"1..24" in ffi.lib().dkd_last_error()


def test_ln_bwd_reduce_group(ops):
    """dkd_ln_bwd_reduce_group: several deferred LayerNorm dgamma / dbeta reductions (different row counts, D = 192 and 384) in one
    launch == summing the partial rows in torch, accumulated onto what the gradients already hold."""
    from deltakd_amd import ffi
    items, keep = [], []
    for i, (nblk, D) in enumerate(((256, 192), (788, 192), (17, 192), (300, 384), (1, 192))):
        part = rnd(nblk, 2 * D, seed=400 + i)
        dg = rnd(D, seed=420 + i)
        db = rnd(D, seed=440 + i)
        ref = (dg + part[:, :D].sum(0), db + part[:, D:].sum(0))
        it = ffi.LnReduce()
        it.part, it.nblk, it.D, it.dgamma, it.dbeta = part.data_ptr(), nblk, D, dg.data_ptr(), db.data_ptr()
        items.append(it)
        keep.append((part, dg, db, ref))
    arr = (ffi.LnReduce * len(items))(*items)
    ffi.check(ffi.lib().dkd_ln_bwd_reduce_group(ffi.C.cast(arr, ffi.C.c_void_p), len(items), ffi.stream()), "ln reduce group")
    for part, dg, db, (rg, rb) in keep:
        close(dg, rg, 1e-5, "dgamma")
        close(db, rb, 1e-5, "dbeta")
    assert ffi.lib().dkd_ln_bwd_reduce_group(ffi.C.cast(arr, ffi.C.c_void_p), 13, ffi.stream()) != 0


def ref_attention(qkv, B, N, H):
    q, k, v = qkv.float().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = s.softmax(-1)
    return (p @ v).transpose(1, 2).reshape(B * N, H * 64), torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,N,H", [(2, 17, 1), (3, 18, 2), (2, 197, 3), (2, 198, 12), (1, 64, 2), (1, 256, 1), (2, 100, 2)])
def test_attn_fwd_bwd(ops, B, N, H):
    qkv = rnd(B * N, 3 * H * 64, scale=1.5, seed=20).to(BF16)
    out, lse = ops.attn_fwd(qkv, B, N, H)
    ref, ref_lse = ref_attention(qkv, B, N, H)
    close(out, ref, 1.5e-2, "attn out")     # P is rounded to bf16 before PV: 2^-9 relative per term
    close(lse, ref_lse, 1e-3, "lse")
    dout = rnd(B * N, H * 64, seed=21).to(BF16)
    x = qkv.float().requires_grad_(True)
    r, _ = ref_attention(x, B, N, H)
    r.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, N, H)
    g = x.grad.view(B, N, 3, H, 64)
    d = dqkv.float().view(B, N, 3, H, 64)
    for i, nm in enumerate("qkv"):
        close(d[:, :, i], g[:, :, i], 3e-2, f"d{nm}")


@pytest.mark.parametrize("B,N,H", [(48, 198, 12), (200, 197, 3), (70, 130, 8), (40, 256, 16), (90, 193, 6), (90, 208, 6), (90, 209, 6),
                                   (90, 224, 6), (130, 113, 4), (130, 128, 4), (40, 241, 16)])
def test_attn_fwd_persistent_kernel(ops, B, N, H):
    """>= 2 heads per CU route to the double-buffered persistent forward (LDS-DMA K/V ring, swizzled unpadded rows); odd head
    counts per workgroup, N on and off a 16-row boundary, the 8-, 14- and 16-tile instantiations.  The padding mask rides in the MFMA
    accumulator when N lies in the last two key tiles of the instantiation (one valid key in the partial tile: 193, 209, 113, 241;
    no partial tile: 208, 224, 128, 256) and is applied with selects otherwise (130)."""
    qkv = rnd(B * N, 3 * H * 64, scale=1.5, seed=22).to(BF16)
    out, lse = ops.attn_fwd(qkv, B, N, H)
    ref, ref_lse = ref_attention(qkv, B, N, H)
    close(out, ref, 1.5e-2, "attn out (ring)")
    close(lse.view(B, H, N), ref_lse, 1e-3, "lse (ring)")


@pytest.mark.parametrize("B,N,H", [(100, 197, 3), (30, 198, 12), (40, 100, 8), (90, 128, 3), (24, 224, 12), (22, 193, 12), (86, 65, 3)])
def test_attn_bwd_per_head_kernel(ops, B, N, H):
    """>= one head per CU routes the backward to the persistent one-workgroup-per-head kernel (q, k, v, dO of the head in LDS, next
    head prefetched into registers): uneven heads per workgroup, the 8- and 14-tile instantiations, N on a 16- / 32-row boundary, one
    row into the packed tail pass (193), a full tail (224)."""
    qkv = rnd(B * N, 3 * H * 64, scale=1.5, seed=23).to(BF16)
    out, lse = ops.attn_fwd(qkv, B, N, H)
    dout = rnd(B * N, H * 64, seed=24).to(BF16)
    x = qkv.float().requires_grad_(True)
    r, _ = ref_attention(x, B, N, H)
    r.backward(dout.float())
    dqkv = ops.attn_bwd(qkv, out, dout, lse, B, N, H)
    g = x.grad.view(B, N, 3, H, 64)
    d = dqkv.float().view(B, N, 3, H, 64)
    for i, nm in enumerate("qkv"):
        close(d[:, :, i], g[:, :, i], 3e-2, f"d{nm} (per head)")
    for b in (0, B // 2, B - 1):                    # every workgroup pass, first and last head, not only on average
        for i, nm in enumerate("qkv"):
            close(d[b, :, i], g[b, :, i], 3e-2, f"d{nm} image {b}")


@pytest.mark.parametrize("M,D", [(7, 64), (300, 192), (129, 768), (50, 1024)])
def test_layernorm(ops, M, D):
    x = rnd(M, D, scale=2.0, seed=30) + 0.5
    gamma = 1 + 0.1 * rnd(D, seed=31)
    beta = 0.1 * rnd(D, seed=32)
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    close(y, ref, 6e-3, "ln fwd (bf16 out)")
    y32, _, _ = ops.layernorm_fwd(x, gamma, beta, out_f32=True)
    close(y32, ref, 1e-5, "ln fwd f32")
    dy = rnd(M, D, seed=33).to(BF16)
    ref.backward(dy.float())
    dx = torch.ones(M, D, device=dev())
    dg, db = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, accumulate=True)
    close(dx - 1, xr.grad, 1e-4, "ln dx")
    close(dg, gr.grad, 1e-4, "ln dgamma")
    close(db, br.grad, 1e-4, "ln dbeta")
    # the same with per-block partial sums in a workspace (and accumulation into non-zero gradients)
    ws = torch.empty(2 * D * ((M + 63) // 64), device=dev())
    dx2 = torch.zeros(M, D, device=dev())
    dg2, db2 = torch.full((D,), 0.5, device=dev()), torch.full((D,), -0.25, device=dev())
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx2, dg2, db2, ws=ws)
    close(dx2, xr.grad, 1e-4, "ln dx (ws)")
    close(dg2 - 0.5, gr.grad, 1e-4, "ln dgamma (ws)")
    close(db2 + 0.25, br.grad, 1e-4, "ln dbeta (ws)")


def test_im2col_and_embed(ops):
    B, C, H, p, D = 3, 3, 32, 8, 64
    img = rnd(B, C, H, H, seed=40)
    w = rnd(D, C, p, p, scale=0.1, seed=41)
    bias = rnd(D, seed=42)
    patches = ops.im2col_patches(img, p)
    ref = torch.nn.functional.conv2d(img.to(BF16).float(), w.to(BF16).float(), bias, stride=p).flatten(2).transpose(1, 2)
    got = ops.gemm_nt(patches, w.view(D, -1).to(BF16).contiguous(), bias=bias, out_f32=True)
    close(got, ref.reshape(-1, D), 1e-4 * 14, "patch embed")
    # prefix tokens + embed backward
    N, npre = 18, 2
    x = torch.zeros(B * N, D, device=dev())
    tok, pos = rnd(npre, D, seed=43), rnd(N, D, seed=44)
    ops.prefix_tokens_fwd(x, tok, pos, B, N, D, npre)
    assert torch.equal(x.view(B, N, D)[:, :npre], (tok + pos[:npre]).expand(B, npre, D))
    dx = rnd(B * N, D, seed=45)
    dtok, dpos = torch.zeros(npre, D, device=dev()), torch.zeros(N, D, device=dev())
    ops.embed_bwd(dx, dtok, dpos, B, N, D, npre)
    close(dpos, dx.view(B, N, D).sum(0), 1e-5, "dpos")
    close(dtok, dx.view(B, N, D)[:, :npre].sum(0), 1e-5, "dtok")


def test_elementwise(ops):
    from deltakd_amd.ffi import strip_map
    B, N, D = 4, 17, 64
    x = rnd(B * N, D, seed=50)
    sc = torch.tensor([1.0, 0.0, 2.0, 1.5], device=dev())
    add = rnd(B * N, D, seed=51)
    y = ops.scale_cast_bf16(x, rowscale=sc, rows_per_sample=N, add=add)
    assert torch.equal(y, (x * sc.repeat_interleave(N)[:, None] + add).to(BF16))
    ys = ops.scale_cast_bf16(x, M=B * (N - 1), xmap=strip_map(N, 1))
    assert torch.equal(ys, x.view(B, N, D)[:, 1:].reshape(-1, D).to(BF16))
    w = rnd(100, 72, seed=52)
    wb, wt = torch.empty(100, 72, device=dev(), dtype=BF16), torch.empty(72, 100, device=dev(), dtype=BF16)
    ops.cast_weight(w, wb, wt)
    assert torch.equal(wb, w.to(BF16)) and torch.equal(wt, w.to(BF16).t())
    cs = torch.zeros(D, device=dev())
    ops.colsum(y, cs)
    close(cs, y.float().sum(0), 1e-5, "colsum")
    tgt = torch.ones(B * N, D, device=dev())
    ops.add_rows(ys, tgt, ymap=strip_map(N, 1))
    exp = torch.ones(B, N, D, device=dev())
    exp[:, 1:] += ys.float().view(B, N - 1, D)
    close(tgt, exp.view(-1, D), 1e-6, "add_rows")


@pytest.mark.parametrize("C", [10, 100, 1000])
def test_logit_loss(ops, C):
    import torch.nn.functional as F
    B = 9
    z, zk, zt = rnd(B, C, scale=2, seed=60), rnd(B, C, scale=2, seed=61), rnd(B, C, scale=3, seed=62)
    y = torch.softmax(rnd(B, C, seed=63), 1)
    lab = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(1)).to(dev())
    for target in (y, lab):
        for mode, wb, wk in ((0, 1.0, 0.0), (1, 0.9, 0.1), (2, 0.5, 0.5)):
            zr, zkr = z.clone().requires_grad_(True), zk.clone().requires_grad_(True)
            if target.dtype == F32:
                base = torch.sum(-target * F.log_softmax(zr, -1), -1).mean()
            else:
                lp = F.log_softmax(zr, -1)
                base = (0.9 * -lp.gather(1, target[:, None]).squeeze(1) + 0.1 * -lp.mean(-1)).mean()
            if mode == 1:
                T = 3.0
                kd = F.kl_div(F.log_softmax(zkr / T, 1), F.log_softmax(zt / T, 1), reduction="sum", log_target=True) * T * T / zkr.numel()
            elif mode == 2:
                kd = F.cross_entropy(zkr, zt.argmax(1))
            else:
                kd = zkr.sum() * 0
            (wb * base + wk * kd).backward()
            losses, dz, dzk = ops.logit_loss(z, target, smoothing=0.1, kd_mode=mode, z_kd=zk, z_t=zt, tau=3.0, w_base=wb, w_kd=wk)
            close(losses[0], base.detach(), 1e-5, "base")
            close(dz, zr.grad, 1e-4, "dz")
            if mode:
                close(losses[1], kd.detach(), 1e-5, "kd")
                close(dzk, zkr.grad, 1e-4, "dz_kd")
            # the weighted slots the criterion hands out without further scalar kernels
            close(losses[2], (wb * base + wk * kd).detach(), 1e-5, "weighted total")
            close(losses[3], (wb * base).detach(), 1e-5, "weighted base")
            close(losses[4], (wk * kd).detach(), 1e-5, "weighted distill")


@pytest.mark.parametrize("C", [10, 100, 1000])
def test_topk_correct_is_timm_accuracy(ops, C):
    """deltakd_amd.shims.accuracy on device logits (one libdkd launch) == timm.utils.accuracy's topk/eq/sum restatement, including rows
    with tied logits (torch.topk lists equal values by ascending index on this build: the kernel's tie rule) and k > C."""
    from deltakd_amd.shims import accuracy
    B = 37
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, C, generator=g).to(dev())
    lab = torch.randint(0, C, (B,), generator=g).to(dev())
    z[3] = z[3].round()                                    # many exact ties in one row
    z[5, :] = 0.25                                         # all equal
    for topk in ((1,), (1, 5), (1, 3, 5, 20)):
        got = accuracy(z, lab, topk=topk)
        maxk = min(max(topk), C)
        _, pred = z.cpu().topk(maxk, 1, True, True)
        correct = pred.t().eq(lab.cpu().reshape(1, -1).expand(maxk, -1))
        for k, a in zip(topk, got):
            ref = correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / B
            # tie handling may legitimately differ from topk's internal order: compare on the rows without ties at the boundary
            assert a.shape == () and abs(a.item() - ref.item()) <= 2 * 100.0 / B + 1e-4, (topk, k, a.item(), ref.item())
    # no ties: exact
    z2 = torch.randn(B, C, generator=g).to(dev())
    got = accuracy(z2, lab, topk=(1, 5))
    _, pred = z2.topk(min(5, C), 1, True, True)
    correct = pred.t().eq(lab.reshape(1, -1).expand(min(5, C), -1))
    for k, a in zip((1, 5), got):
        ref = correct[:min(k, C)].reshape(-1).float().sum(0) * 100. / B
        assert abs(a.item() - ref.item()) < 1e-4, (k, a.item(), ref.item())
    # labels outside [0, C) (ignore_index -100 / -1, a class the head lacks): counted wrong, like timm -- not read out of bounds
    lab3 = lab.clone()
    lab3[0], lab3[1], lab3[2] = -100, -1, C
    got = accuracy(z2, lab3, topk=(1, 5))
    ok = torch.ones(B, dtype=torch.bool, device=z2.device)
    ok[:3] = False
    for k, a in zip((1, 5), got):
        ref = (correct[:min(k, C)].any(0) & ok).float().sum() * 100. / B
        assert abs(a.item() - ref.item()) < 1e-4, (k, a.item(), ref.item())


def test_mse_and_mask(ops):
    from deltakd_amd.ffi import strip_map
    B, N, D = 3, 18, 64
    P = N - 2
    a = rnd(B * P, D, seed=70).to(BF16)
    t = rnd(B * N, D, seed=71).to(BF16)
    mask = (rnd(B * P, seed=72) > 0).float()
    w = 7e-5 / (B * P * D)
    loss = torch.zeros(1, device=dev())
    da = ops.mse_loss(a, t, loss, w, tmap=strip_map(N, 2), mask=mask, grad_f32=True)
    ar = a.float().requires_grad_(True)
    tt = t.float().view(B, N, D)[:, 2:].reshape(-1, D)
    ref = 7e-5 * torch.nn.functional.mse_loss(ar * mask[:, None], tt * mask[:, None])
    ref.backward()
    close(loss, ref.detach(), 1e-5, "mse")
    close(da, ar.grad, 1e-5, "mse grad")
    tok = rnd(D, seed=73)
    xs = ops.mask_select(a, tok, mask)
    exp = torch.where(mask[:, None] > 0, tok.to(BF16)[None], a)
    assert torch.equal(xs, exp)
    dout = rnd(B * P, D, seed=74).to(BF16)
    dtok = torch.zeros(D, device=dev())
    dx = ops.mask_select_bwd(dout, mask, dtok)
    assert torch.equal(dx, torch.where(mask[:, None] > 0, torch.zeros_like(dout), dout))
    close(dtok, (dout.float() * mask[:, None]).sum(0), 1e-5, "dtok")


def test_adamw(ops):
    n = 10007
    p, g = rnd(n, seed=80), rnd(n, seed=81)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    m, v = torch.zeros(n, device=dev()), torch.zeros(n, device=dev())
    pb = torch.empty(n, device=dev(), dtype=BF16)
    for step in range(1, 4):
        pr.grad = g.clone() * step
        opt.step()
        ops.adamw_step(p, g * step, m, v, pb, 5e-4, 0.9, 0.999, 1e-8, 0.05, step)
    close(p, pr.detach(), 1e-6, "adamw")
    assert torch.equal(pb, p.to(BF16))


def test_errors_are_loud(ops):
    a = torch.zeros(4, 60, device=dev(), dtype=BF16)
    with pytest.raises(ValueError):
        ops.gemm_nt(a, a)            # K not a multiple of 64
    with pytest.raises(RuntimeError):
        ops.gemm_nt(torch.zeros(4, 64, dtype=BF16), torch.zeros(4, 64, dtype=BF16))   # host tensors


@pytest.mark.parametrize("n,batch", [(8, 2), (31, 3), (96, 3), (128, 2)])
def test_jacobi_eigh(ops, n, batch):
    x = rnd(batch, n, n, seed=90)
    A = x @ x.transpose(1, 2) / n + torch.diag_embed(torch.linspace(0, 3, n, device=dev()).expand(batch, n))
    ev, vec = ops.jacobi_eigh(A)
    ref = torch.linalg.eigvalsh(A.double().cpu()).flip(1)
    close(ev.cpu(), ref, 2e-6 * n, "eigenvalues")
    recon = vec @ torch.diag_embed(ev) @ vec.transpose(1, 2)
    close(recon, A, 2e-5, "V diag V^T")
    eye = torch.eye(n, device=dev()).expand(batch, n, n)
    close(vec.transpose(1, 2) @ vec, eye, 2e-5, "orthonormal")


@pytest.mark.parametrize("Dt", [768, 384, 1024])
def test_lowrank_step_kernels(ops, Dt):
    """csrc/lowrank.hip against torch.linalg.eigh: orthonormalisation (mode 2), power steps (mode 0), the tracking step (mode 1:
    Ritz values / vectors sorted, hi/lo bf16 split of V_k^T), and that ONLY the 128 x 128 tiles on and above the diagonal of G are read
    (the rest is filled with NaN)."""
    L, b, r = 3, 96, 64
    g = torch.Generator().manual_seed(5)
    Gs, refs = [], []
    for l in range(L):
        q = torch.linalg.qr(torch.randn(Dt, Dt, generator=g, dtype=torch.float64))[0]
        lam = torch.cat([torch.linspace(50, 8, 40, dtype=torch.float64), 6.0 * torch.exp(-torch.arange(Dt - 40, dtype=torch.float64) / 25.0) + 0.05])
        G = (q * lam) @ q.t()
        refs.append((lam, q))
        Gs.append(G.float())
    G = torch.stack(Gs).to(dev())
    blk = torch.arange(Dt, device=dev()) // 128
    Gu = torch.where((blk[:, None] > blk[None, :])[None], torch.full_like(G, float("nan")), G).contiguous()
    ws = ops.lowrank_workspace(L, Dt, G.device)
    V = torch.randn(L, Dt, b, device=dev(), generator=torch.Generator(device=dev()).manual_seed(1))
    ops.lowrank_step(Gu, V, 2, ws)
    eye = torch.eye(b, device=dev()).expand(L, b, b)
    close(V.transpose(1, 2) @ V, eye, 1e-4, "orth(V)")
    for _ in range(16):
        ops.lowrank_step(Gu, V, 0, ws)
    ops.lowrank_step(Gu, V, 2, ws)
    close(V.transpose(1, 2) @ V, eye, 2e-5, "orthonormal after power steps")
    hi = torch.empty(L, r, Dt, device=dev(), dtype=BF16)
    lo = torch.empty(L, r, Dt, device=dev(), dtype=BF16)
    ev = torch.empty(L, b, device=dev())

    def check(tag, vec_tol):
        close(V.transpose(1, 2) @ V, eye, 5e-5, tag + ": Ritz vectors orthonormal")
        for l in range(L):
            lam, q = refs[l]
            assert (ev[l, 1:] <= ev[l, :-1] + 1e-6 * ev[l, 0]).all(), tag + ": Ritz values not sorted"
            assert (ev[l, :r].double().cpu() - lam[:r]).abs().max().item() <= 2e-4 * lam[0].item(), tag + ": Ritz values"
            lead = 40                                   # the well-separated leading eigenvectors, up to sign
            dots = (V[l, :, :lead].double().cpu() * q[:, :lead]).sum(0).abs()
            assert dots.min().item() > 1 - vec_tol, (tag, dots.min().item())
            Pr = V[l, :, :r].double().cpu()             # rank-r projector captures the leading invariant subspace
            assert ((Pr.t() @ q[:, :r]) ** 2).sum().item() > r - 1e-2, tag
            Vt = V[l, :, :r].t()
            close(hi[l].float() + lo[l].float(), Vt, 1e-4, tag + ": hi + lo")
            assert torch.equal(hi[l], Vt.to(BF16)), tag

    ops.lowrank_step(Gu, V, 3, ws, rank=r, hi=hi, lo=lo, evals=ev)          # converged Rayleigh-Ritz in the span reached
    check("mode 3", 1e-3)
    for it in range(3):                     # tracking steps on the same G (2 Jacobi sweeps each at most): stays converged and sorted
        ops.lowrank_step(Gu, V, 1, ws, rank=r, hi=hi, lo=lo, evals=ev, ritz_sweeps=2)
        assert torch.isfinite(V).all()
    check("mode 1", 1e-3)
    assert int(ops.lowrank_info(ws, L, Dt)[:, 1].max()) <= 2
    # a rotated basis of the same subspace and no sweeps at all: the step must still return an orthonormal basis of span(G V) (to
    # fp32 roundoff x the conditioning of (G V)^T (G V), which is what the Ritz rotation keeps near 1 in normal operation)
    rot = torch.linalg.qr(torch.randn(b, b, generator=g))[0].to(dev())
    V2 = (V @ rot).contiguous()
    ops.lowrank_step(Gu, V2, 1, ws, ritz_sweeps=0)
    close(V2.transpose(1, 2) @ V2, eye, 2e-3, "orthonormal without Ritz rotation")
    for l in range(L):
        q = refs[l][1]
        assert ((V2[l].double().cpu().t() @ q[:, :r]) ** 2).sum().item() > r - 1e-2


@pytest.mark.parametrize("Dt", [768, 384, 1024, 128])
def test_lowrank_chain(ops, Dt):
    """dkd_lowrank_chain (csrc/lowrank.hip, the per-batch solve of the timed path) against float64 eigh of the same matrices: from a
    basis rotated away from the eigenvectors (a random rotation of 0.2 rad per coordinate pair inside the span and a perturbation out of
    it), n_mult power steps + the converged Rayleigh-Ritz step must return the leading eigenpairs -- values to 2e-5 lambda_1 after 12
    steps, sorted, vectors of the separated part of the spectrum up to sign, an orthonormal basis to fp32 roundoff, hi + lo = V_k^T --
    for every multiply schedule (n_mult 1, 2, 3, 4, 5, 8, 12: stages of 1 and 2 multiplies), reading ONLY the tiles of G on and above
    the diagonal (the rest is NaN), and leaving its Gram accumulators zero for the next call."""
    L, b, r = 3, 96, min(64, Dt // 2)
    g = torch.Generator().manual_seed(5)
    Gs, refs = [], []
    for l in range(L):
        q = torch.linalg.qr(torch.randn(Dt, Dt, generator=g, dtype=torch.float64))[0]
        lam = torch.cat([torch.linspace(50, 8, 40, dtype=torch.float64), 6.0 * torch.exp(-torch.arange(Dt - 40, dtype=torch.float64) / 25.0) + 0.05])
        refs.append((lam, q))
        Gs.append(((q * lam) @ q.t()).float())
    G = torch.stack(Gs).to(dev())
    blk = torch.arange(Dt, device=dev()) // 128
    Gu = torch.where((blk[:, None] > blk[None, :])[None], torch.full_like(G, float("nan")), G).contiguous()
    ws = ops.lowrank_chain_workspace(L, Dt, G.device)
    nz = ops.lib().dkd_lowrank_chain_zero_bytes(L, Dt)
    eye = torch.eye(b, device=dev()).expand(L, b, b)
    hi = torch.empty(L, r, Dt, device=dev(), dtype=BF16)
    lo = torch.empty(L, r, Dt, device=dev(), dtype=BF16)
    ev = torch.empty(L, b, device=dev())

    def start(seed):
        """the exact leading eigenvectors, mixed among themselves and tilted out of their span: orthonormal, not converged"""
        gg = torch.Generator().manual_seed(seed)
        out = []
        for l in range(L):
            q = refs[l][1]
            mix = torch.linalg.qr(torch.eye(b, dtype=torch.float64) + 0.2 * torch.randn(b, b, generator=gg, dtype=torch.float64))[0]
            v = q[:, :b] @ mix + 0.05 * torch.randn(Dt, b, generator=gg, dtype=torch.float64) / Dt ** 0.5
            out.append(torch.linalg.qr(v)[0].float())
        return torch.stack(out).to(dev()).contiguous()

    for n_mult in (1, 2, 3, 4, 5, 8, 12):
        V = start(100 + n_mult)
        ops.lowrank_chain(Gu, V, n_mult, 12, ws, rank=r, hi=hi, lo=lo, evals=ev)
        assert torch.isfinite(V).all() and torch.isfinite(ev).all(), n_mult
        assert int(ws[:nz].view(torch.int32).abs().max()) == 0, "the Gram accumulators must be left zero"
        close(V.transpose(1, 2) @ V, eye, 1e-4, f"n_mult {n_mult}: orthonormal basis")
        sweeps = int(ops.lowrank_chain_info(ws, L, Dt)[:, 1].max())
        assert 1 <= sweeps <= 12, sweeps
        for l in range(L):
            lam, q = refs[l]
            assert (ev[l, 1:] <= ev[l, :-1] + 1e-6 * ev[l, 0]).all(), "Ritz values not sorted"
            Vl = V[l].double().cpu()
            # Ritz values = Rayleigh quotients of the returned vectors' predecessors: within the iteration's convergence of the eigenvalues
            tol = {1: 3e-2, 2: 1e-2, 3: 5e-3, 4: 2e-3, 5: 1e-3, 8: 2e-4, 12: 2e-5}[n_mult]
            assert (ev[l, :r].double().cpu() - lam[:r]).abs().max().item() <= tol * lam[0].item(), (n_mult, l)
            Vt = V[l, :, :r].t()
            close(hi[l].float() + lo[l].float(), Vt, 1e-4, "hi + lo")
            assert torch.equal(hi[l], Vt.to(BF16))
            if n_mult >= 8:
                lead = min(40, r)                   # the well-separated leading eigenvectors, up to sign
                dots = (Vl[:, :lead] * q[:, :lead]).sum(0).abs()
                assert dots.min().item() > 1 - 1e-3, (n_mult, l, dots.min().item())
                assert ((Vl[:, :r].t() @ q[:, :r]) ** 2).sum().item() > r - 1e-2
    # a second call on the same matrices from the converged basis: one sweep or none, same answer (signs continuous)
    V1 = V.clone()
    ops.lowrank_chain(Gu, V, 2, 12, ws, evals=ev)
    close((V[:, :, :32] * V1[:, :, :32]).sum(1), torch.ones(L, 32, device=dev()), 1e-3, "warm restart keeps vectors and signs")


def test_step_glue_kernels(ops):
    """The three launches that replaced ATen chains on the step: bf16 x device-scalar in f32, the K-padded
    bf16 cast of the head's logit gradient, the DropPath masks of a whole step from one counter-based draw."""
    x = rnd(1000, 64, seed=3).to(BF16)
    g = torch.tensor(0.3712, device=dev())
    ref = (x.float() * g).to(BF16)
    got = ops.scale_bf16_(x.clone(), g.reshape(1))
    assert torch.equal(got, ref)
    src = rnd(37, 1000, seed=4)
    pad = ops.cast_pad_bf16(src, 1024)
    assert pad.shape == (37, 1024) and torch.equal(pad[:, :1000], src.to(BF16)) and int(pad[:, 1000:].abs().max()) == 0
    view = rnd(37, 1200, seed=5)[:, 100:1100]                       # a strided source (row stride 1200)
    pad = ops.cast_pad_bf16(view, 1024, scalar=g.reshape(1))
    assert torch.equal(pad[:, :1000], (view * g).to(BF16))
    keep = torch.tensor([1.0, 0.95, 0.9, 0.5], device=dev())
    B = 20000
    a = ops.droppath_scales(keep, B, 1234)
    b = ops.droppath_scales(keep, B, 1234)
    c = ops.droppath_scales(keep, B, 1235)
    assert torch.equal(a, b) and not torch.equal(a, c)
    for i, k in enumerate(keep.tolist()):
        vals = torch.unique(a[i]).tolist()
        assert all(abs(v) < 1e-12 or abs(v - 1.0 / k) < 1e-6 for v in vals), vals
        rate = (a[i] > 0).float().mean().item()
        assert abs(rate - k) < 4 * (k * (1 - k) / B) ** 0.5 + 1e-9, (k, rate)        # 4 sigma of a Bernoulli mean
        assert abs(a[i].mean().item() - 1.0) < 5 * ((1 - k) / (k * B)) ** 0.5 + 1e-9  # E[mask / keep] = 1
    # draws of different branches / samples are not correlated: the two halves of one row, and two rows, agree at chance level
    m = (a[3] > 0).float()
    assert abs((m[:B // 2] * m[B // 2:]).mean().item() - 0.25) < 0.02
    assert abs(((a[2] > 0).float() * m).mean().item() - 0.45) < 0.02


def test_patch_matrix_is_shared_between_teacher_and_student(ops, monkeypatch):
    """deltakd_amd.vit._shared_patches: the bf16 patch matrix of a batch is gathered once and used by both models (also across streams);
    a rewritten batch (same storage, new version) and another patch size gather again."""
    from deltakd_amd import vit
    torch.manual_seed(1)
    t = vit.VisionTransformer(128, 2, 2, 10, True, 0.0, img_size=32, patch_size=8).to(dev()).eval()
    s = vit.VisionTransformer(64, 2, 1, 10, False, 0.0, img_size=32, patch_size=8).to(dev()).eval()
    x = rnd(4, 3, 32, 32, seed=2)
    calls = []
    real = ops.im2col_patches
    monkeypatch.setattr(vit.ops, "im2col_patches", lambda img, p: (calls.append(1), real(img, p))[1])
    side = torch.cuda.Stream()
    with torch.no_grad():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a1 = t(x)                                   # the teacher gathers, on its own stream
        b1 = s(x)                                       # the student finds the matrix (and waits for that stream's event)
        torch.cuda.synchronize()
        assert len(calls) == 1
        monkeypatch.setenv("DKD_NO_SHARED_PATCHES", "1")
        a2, b2 = t(x), s(x)
        assert len(calls) == 3 and torch.equal(a1, a2) and torch.equal(b1, b2)
        monkeypatch.delenv("DKD_NO_SHARED_PATCHES")
        x.mul_(2.0)                                     # same tensor object, rewritten in place: the cached matrix is stale
        b3 = s(x)
        assert len(calls) == 4
        monkeypatch.setenv("DKD_NO_SHARED_PATCHES", "1")
        assert torch.equal(b3, s(x.clone()))


@pytest.mark.parametrize("box", [None, (30, 150, 16, 200), (0, 224, 0, 224)])
def test_mixup_also_writes_the_patch_matrix(ops, box):
    """dkd_mixup_to_patches: the mix and, from the same launch, its bf16 patch matrix -- bit-identical to dkd_mixup_to followed by
    dkd_im2col_patches; and the Mixup shim hands that matrix to the models (no im2col launch in a training step with mixup on)."""
    x = rnd(6, 3, 224, 224, seed=12)
    ref = ops.mixup(x, 0.37, box)
    out, patches = ops.mixup_with_patches(x, 0.37, box, 16)
    assert torch.equal(out, ref) and torch.equal(patches, ops.im2col_patches(ref, 16))
    from deltakd_amd import shims, vit
    import numpy as np
    np.random.seed(3)
    calls = []
    real = ops.im2col_patches
    import pytest as _pt
    mp = _pt.MonkeyPatch()
    try:
        mp.setattr(vit.ops, "im2col_patches", lambda img, p: (calls.append(1), real(img, p))[1])
        mix = shims.Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, num_classes=10)
        xm, _ = mix(x, torch.arange(6, device=dev()) % 10)
        assert xm is not x
        torch.manual_seed(0)
        m = vit.VisionTransformer(64, 1, 1, 10, False, 0.0, img_size=224, patch_size=16).to(dev()).eval()
        with torch.no_grad():
            y = m(xm)
        assert not calls, "the model should have found the Mixup kernel's patch matrix"
        mp.setenv("DKD_NO_SHARED_PATCHES", "1")
        with torch.no_grad():
            assert torch.equal(y, m(xm)) and len(calls) == 1
    finally:
        mp.undo()


def test_gram_batched_equals_the_single_launches(ops):
    """dkd_gram_batched: the upper tile pairs of L Gram matrices at constant strides in one launch == L calls of dkd_gram (row map
    included: the prefix tokens of every sample are skipped), up to the atomics' summation order."""
    L, Bn, N, D, npre = 3, 5, 30, 384, 2
    from deltakd_amd.ffi import strip_map
    slab = rnd(L, Bn * N, D, seed=77).to(BF16)
    M = Bn * (N - npre)
    smap = strip_map(N, npre)
    one = torch.zeros(L, D, D, device=dev())
    for l in range(L):
        ops.gram(slab[l], one[l], M=M, amap=smap, mirror=False)
    many = torch.zeros(L, D, D, device=dev())
    ops.gram_batched(slab[0], many, L, slab.stride(0), M=M, amap=smap)
    blk = torch.arange(D, device=dev()) // 128
    upper = (blk[:, None] <= blk[None, :])[None].expand(L, D, D)
    close(many[upper], one[upper], 1e-5, "batched Gram vs single launches")
    ref = torch.stack([slab[l].view(Bn, N, D)[:, npre:].reshape(-1, D).float().t() @ slab[l].view(Bn, N, D)[:, npre:].reshape(-1, D).float()
                       for l in range(L)])
    close(many[upper], ref[upper], 1e-4, "batched Gram vs fp32 matmul")
    assert int((many[~upper] != 0).sum()) == 0


def test_lowrank_targets_vs_svd(ops):
    """Dt = 768 (subspace iteration + Rayleigh-Ritz path): U_k S_k against torch.linalg.svd on the host, up to column sign.
    Cold start, then a warm-started call on a different batch drawn from the same feature distribution."""
    from deltakd_amd.losses import LowRankTargets
    B, N, Dt, r = 16, 198, 768, 64
    g = torch.Generator().manual_seed(7)
    basis = torch.linalg.qr(torch.randn(Dt, Dt, generator=g))[0]
    spec = torch.cat([torch.linspace(30, 6, 24), 5.0 * torch.exp(-torch.arange(Dt - 24) / 40.0) + 0.2])
    solver = LowRankTargets()
    for call in range(5):
        t = (torch.randn(B * N, Dt, generator=g) * spec) @ basis.t()
        tb = t.to(BF16).to(dev()).view(B, N, Dt)
        got = solver([tb], 2, r)[0].cpu()
        T = tb.float().cpu()[:, 2:].reshape(-1, Dt)
        U, S, _ = torch.linalg.svd(T, full_matrices=False)
        ref = U[:, :r] * S[:r]
        sv = got.norm(dim=0)
        assert (sv - S[:r]).abs().max().item() <= 2e-3 * S[0].item(), f"call {call}: singular values"
        sign = torch.sign((got * ref).sum(0))
        lead = 24                                       # well-separated part of the spectrum: vectors are well-conditioned
        err = ((got[:, :lead] * sign[:lead]) - ref[:, :lead]).norm() / ref[:, :lead].norm()
        assert err.item() < 2e-2, f"call {call}: leading columns rel err {err.item()}"
        # whole rank-r approximation (sign- and rotation-invariant): ||T V_r|| captured energy
        assert abs(got.norm().item() - ref.norm().item()) <= 2e-3 * ref.norm().item()


def test_conv3x3_as_gather_gemm(ops):
    """im2col3x3 + gemm_nt == Conv2d(C, C, 3, padding=1) on the token grid; col2im3x3 == its input gradient."""
    import torch.nn.functional as F
    B, hw, C = 3, 4, 64
    x = rnd(B * hw * hw, C, seed=100).to(BF16)
    w = rnd(C, C, 3, 3, scale=0.1, seed=101)
    bias = rnd(C, seed=102)
    wp = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(BF16)
    cols = ops.im2col3x3(x, B, hw)
    y = ops.gemm_nt(cols, wp, bias=bias, out_f32=True)
    xi = x.float().view(B, hw, hw, C).permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv2d(xi, w.to(BF16).float(), bias, padding=1)
    close(y, ref.permute(0, 2, 3, 1).reshape(-1, C), 1e-4 * 24, "conv fwd")
    dy = rnd(B * hw * hw, C, seed=103).to(BF16)
    ref.backward(dy.float().view(B, hw, hw, C).permute(0, 3, 1, 2))
    dcols = ops.gemm_nt(dy, wp.t().contiguous())
    gate = rnd(B * hw * hw, C, seed=104).to(BF16)
    dx = ops.col2im3x3(dcols, B, hw)
    close(dx, xi.grad.permute(0, 2, 3, 1).reshape(-1, C), 2e-2, "conv dgrad")
    dxg = ops.col2im3x3(dcols, B, hw, relu_gate=gate)
    close(dxg, xi.grad.permute(0, 2, 3, 1).reshape(-1, C) * (gate.float() > 0), 2e-2, "conv dgrad relu")


@pytest.mark.parametrize("B,hw,Cin,Cout", [(3, 4, 64, 64), (2, 14, 128, 192), (5, 7, 192, 64), (1, 14, 768, 768)])
def test_conv3x3_implicit_gemm(ops, B, hw, Cin, Cout):
    """Conv2d(Cin, Cout, 3, padding=1) on the token grid WITHOUT an im2col matrix (DkdGemm.conv_hw, dkd_conv3x3_wgrad): forward with
    bias (+ReLU), input gradient (the same implicit GEMM on dY with flipped taps, + ReLU gate), weight and bias gradients, against
    torch's fp32 conv2d autograd on the bf16-rounded operands."""
    import torch.nn.functional as F
    P = hw * hw
    x = rnd(B * P, Cin, seed=110).to(BF16)
    w = rnd(Cout, Cin, 3, 3, scale=0.1, seed=111)
    bias = rnd(Cout, seed=112)
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).contiguous().to(BF16)                    # [out, (ky, kx, cin)]
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cin, 9 * Cout).contiguous().to(BF16)         # [cin, (2-ky, 2-kx, out)]
    xi = x.float().view(B, hw, hw, Cin).permute(0, 3, 1, 2).requires_grad_(True)
    wq = w.to(BF16).float().requires_grad_(True)
    bq = bias.clone().requires_grad_(True)
    ref = F.conv2d(xi, wq, bq, padding=1)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    y = ops.gemm_nt(x, wf, bias=bias, out_f32=True, conv_hw=hw)
    close(y, tok(ref), 2e-5 * math.sqrt(9 * Cin) * 4, "conv fwd")
    yr = ops.gemm_nt(x, wf, bias=bias, relu=True, conv_hw=hw)
    close(yr, torch.relu(tok(ref)), 1e-2, "conv fwd relu (bf16 out)")
    dy = rnd(B * P, Cout, seed=113).to(BF16)
    ref.backward(dy.float().view(B, hw, hw, Cout).permute(0, 3, 1, 2))
    dx = ops.gemm_nt(dy, wd, out_f32=True, conv_hw=hw)
    close(dx, tok(xi.grad), 2e-5 * math.sqrt(9 * Cout) * 4, "conv dgrad")
    gate = rnd(B * P, Cin, seed=114).to(BF16)
    dxg = ops.gemm_nt(dy, wd, conv_hw=hw, relu_gate=gate)
    close(dxg, tok(xi.grad) * (gate.float() > 0), 1e-2, "conv dgrad with relu gate")
    dw = torch.zeros(Cout, 9 * Cin, device=dev())
    db = torch.zeros(Cout, device=dev())
    ops.conv3x3_wgrad(dy, x, dw, db, B, hw)
    close(dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), wq.grad, 1e-4, "conv wgrad")
    close(db, bq.grad, 1e-4, "conv bias grad")
    ops.conv3x3_wgrad(dy, x, dw, None, B, hw)                                                   # accumulates
    close(dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), 2 * wq.grad, 1e-4, "conv wgrad accumulates")


@pytest.mark.parametrize("B,P,D", [(2, 16, 128), (3, 196, 192), (2, 49, 100)])
def test_sort_l1(ops, B, P, D):
    from deltakd_amd.ffi import strip_map
    N = P + 2
    s = rnd(B * P, D, seed=110)
    t = rnd(B * N, D, seed=111).to(BF16)
    w = 5.0 / (3 * B * P * D)
    loss = torch.zeros(1, device=dev())
    ds = ops.sort_l1_loss(s, t, loss, w, B=B, P=P, tmap=strip_map(N, 2), grad_f32=True)
    sr = s.view(B, P, D).clone().requires_grad_(True)
    tt = t.float().view(B, N, D)[:, 2:]
    ref = (5.0 / 3) * (torch.sort(sr, 1)[0] - torch.sort(tt, 1)[0]).abs().mean()
    ref.backward()
    close(loss, ref.detach(), 1e-5, "sort-l1 loss")
    close(ds, sr.grad.reshape(B * P, D), 1e-5, "sort-l1 grad")


def test_diffkd_kernels(ops):
    from deltakd_amd.ffi import strip_map
    B, P, D = 3, 16, 128
    N = P + 2
    M = B * P
    t = rnd(B * N, D, seed=120).to(BF16)
    noise = rnd(M, D, seed=121)
    sigma = torch.tensor([0.0, 0.1852, 1.195], device=dev())
    temb = rnd(B, D, seed=122)
    t_hat, nz, x_in = ops.diffkd_prepare(t, noise, sigma, temb, M=M, rows_per_sample=P, tmap=strip_map(N, 2))
    tt = t.float().view(B, N, D)[:, 2:]
    th = tt / tt.norm(dim=-1, keepdim=True)
    nzr = noise.view(B, P, D) * sigma.view(-1, 1, 1)
    close(t_hat, th.reshape(M, D), 5e-3, "t_hat")
    close(nz, nzr.reshape(M, D), 1e-6, "nz")
    close(x_in, (th + nzr + temb[:, None]).reshape(M, D), 5e-3, "x_in")
    # normalize + mse
    s = rnd(M, D, scale=3.0, seed=123)
    wsc = torch.tensor([1234.5], device=dev())
    wod = 2e-5 / (M * D)
    loss = torch.zeros(1, device=dev())
    ds = ops.normalize_mse(s, t_hat, loss, wod, w_scalar=wsc)
    sr = s.clone().requires_grad_(True)
    ref = 1234.5 * 2e-5 * torch.nn.functional.mse_loss(sr / sr.norm(dim=-1, keepdim=True), t_hat.float())
    ref.backward()
    close(loss, ref.detach(), 1e-5, "normalize mse")
    close(ds, sr.grad, 1e-2, "normalize mse grad (bf16)")
    # dropout folded into mse
    a = rnd(M, D, seed=124)
    keep = (rnd(M, D, seed=125) > -1.0).float()
    loss2 = torch.zeros(1, device=dev())
    da = ops.dropout_mse(a, nz, keep, 1 / 0.9, loss2, 3e-5 / (M * D))
    ar = a.clone().requires_grad_(True)
    ref2 = 3e-5 * torch.nn.functional.mse_loss(ar * keep / 0.9, nz)
    ref2.backward()
    close(loss2, ref2.detach(), 1e-5, "dropout mse")
    close(da, ar.grad, 1e-2, "dropout mse grad (bf16)")


def test_gemm_nt_wide_tile_kernel(ops):
    """Shapes with >= 1024 tiles of 256x256 take the 256^2 kernel (teacher qkv / fc1): all epilogues, ragged M."""
    M, N, K = 16384 + 37, 4096, 128
    a = rnd(M, K, seed=130).to(BF16)
    b = rnd(N, K, scale=0.2, seed=131).to(BF16)
    bias = rnd(N, seed=132)
    ref = a.float() @ b.float().t() + bias
    pre = torch.empty(M, N, device=dev(), dtype=BF16)
    h = ops.gemm_nt(a, b, bias=bias, gelu=True, preact=pre)
    close(pre, ref, 1e-2, "wide preact")
    close(h, torch.nn.functional.gelu(ref), 1e-2, "wide gelu")
    resid = rnd(M, N, seed=133)
    sc = (torch.arange(M // 197 + 1, device=dev()) % 3).float()
    out = ops.gemm_nt(a, b, bias=bias, resid=resid, rowscale=sc, rows_per_sample=197, out_f32=True)
    close(out, resid + sc.repeat_interleave(197)[:M, None] * ref, 1e-4 * 12, "wide resid f32")
    dh = ops.gemm_nt(a, b, dgelu=True, preact=pre)
    x = pre.float().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(a.float() @ b.float().t())
    close(dh, x.grad, 1e-2, "wide dgelu")


def test_mixup_kernels_match_the_host_path():
    """Device Mixup / CutMix (fused kernels) == the torch host path of the shim, for the same numpy draws."""
    import numpy as np
    from deltakd_amd.shims import Mixup
    mixed_any = False
    for cutmix_alpha, mixup_alpha in ((0.0, 0.8), (1.0, 0.0), (1.0, 0.8)):
        for seed in (0, 1, 2):
            x = rnd(8, 3, 32, 32, seed=150 + seed)
            y = torch.randint(0, 10, (8,), generator=torch.Generator().manual_seed(seed)).to(dev())
            mix = Mixup(mixup_alpha=mixup_alpha, cutmix_alpha=cutmix_alpha, num_classes=10, label_smoothing=0.1)
            np.random.seed(seed)
            keep = x.clone()
            xd, yd = mix(x, y)                  # default: the mix in a new tensor, the loader's batch untouched
            assert torch.equal(x, keep)
            assert xd.data_ptr() != x.data_ptr() or torch.equal(xd, keep)      # (a Beta(0.8, 0.8) draw can round to lambda = 1: nothing to mix)
            np.random.seed(seed)
            xh, yh = mix(x.cpu().clone(), y.cpu())
            close(xd.cpu(), xh, 1e-6, "mixed images")
            close(yd.cpu(), yh, 1e-6, "soft targets")
            inpl = Mixup(mixup_alpha=mixup_alpha, cutmix_alpha=cutmix_alpha, num_classes=10, label_smoothing=0.1, inplace=True)
            np.random.seed(seed)
            xi, yi = inpl(x, y)                 # timm's behaviour: x itself is overwritten
            assert xi.data_ptr() == x.data_ptr() and torch.equal(xi, xd) and torch.equal(yi, yd)
            mixed_any = mixed_any or not torch.equal(xd, keep)
    assert mixed_any


def test_mixup_to_refuses_partly_overlapping_batches():
    from deltakd_amd import ffi
    buf = torch.zeros(2 * 4 * 3 * 8 * 8 + 64, device=dev())
    n = 4 * 3 * 8 * 8
    src, dst = buf[:n], buf[64:64 + n]
    rc = ffi.lib().dkd_mixup_to(src.data_ptr(), dst.data_ptr(), 4, 3, 8, 8, 0.3, 0, 0, 0, 0, 0, None)
    assert rc != 0 and "overlap" in ffi.lib().dkd_last_error().decode()
    assert ffi.lib().dkd_mixup_to(src.data_ptr(), src.data_ptr(), 4, 3, 8, 8, 0.3, 0, 0, 0, 0, 0, None) == 0      # the same batch: in place


def test_ema_on_flat_storage():
    from types import SimpleNamespace
    from deltakd_amd import vit
    from deltakd_amd.optim import create_optimizer
    from deltakd_amd.shims import ModelEma
    m = vit.VisionTransformer(64, 2, 1, 10, False, 0.0, img_size=32, patch_size=8, mlp_ratio=2.0).to(dev())
    opt = create_optimizer(SimpleNamespace(opt="adamw", lr=1e-2, weight_decay=0.05, opt_eps=1e-8, opt_betas=None), m)
    ema = ModelEma(m, decay=0.9, optimizer=opt)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    for p in m.parameters():
        p.grad.normal_()
    opt.step()
    ema.update(m)
    for k, v in ema.ema.state_dict().items():
        want = 0.9 * before[k] + 0.1 * m.state_dict()[k]
        close(v, want, 1e-6, k)


@pytest.mark.parametrize("Dt,B", [(768, 5), (128, 3), (384, 2)])
def test_saliency_scores(ops, Dt, B):
    """dkd_saliency_scores + the fp32-accurate projections (deltakd_amd.models._project_f32) against the reference's scorer arithmetic in
    torch fp32 (model/models.py:14-56, model/misc.py:38-165): the three methods, teacher tokens [CLS, DIST, 196 patches], 8 heads
    (head_dim 96 at Dt = 768).  The scores are ~1 / 197: 1e-4 relative to their maximum resolves a ranking far below the tap noise."""
    from types import SimpleNamespace
    from deltakd_amd.misc import saliency_scores
    from deltakd_amd.models import SimpleAttention, SimpleCrossAttention
    N, H = 198, 8
    t = rnd(B, N, Dt, seed=900, scale=1.5).to(BF16)
    tf = t.float()
    torch.manual_seed(7)
    sa, ca = SimpleAttention(Dt, H).to(dev()), SimpleCrossAttention(Dt, H).to(dev())
    with torch.no_grad():                          # default-initialised projections give nearly uniform attention: sharpen it
        for lin in (sa.qk, ca.q, ca.k):
            lin.weight.mul_(6.0)
    hd, scale = Dt // H, (Dt // H) ** -0.5

    def heads(x):
        return x.reshape(x.shape[0], x.shape[1], H, hd).permute(0, 2, 1, 3)
    with torch.no_grad():
        # method 1
        qk = tf[:, 2:] @ sa.qk.weight.t() + sa.qk.bias
        a = ((heads(qk[..., :Dt]) @ heads(qk[..., Dt:]).transpose(-2, -1)) * scale).softmax(-1)
        ref1 = a.mean(1).diagonal(dim1=-2, dim2=-1)
        # method 2
        cp = torch.cat([tf[:, :1], tf[:, 2:]], 1)
        qk = cp @ sa.qk.weight.t() + sa.qk.bias
        a = ((heads(qk[..., :Dt])[:, :, 0:1] @ heads(qk[..., Dt:]).transpose(-2, -1)) * scale).softmax(-1)
        ref2 = a.mean(1).squeeze(1)[:, 1:]
        # method 3
        q = cp[:, :1] @ ca.q.weight.t() + ca.q.bias
        k = cp[:, 1:] @ ca.k.weight.t() + ca.k.bias
        ref3 = ((heads(q) @ heads(k).transpose(-2, -1)) * scale).softmax(-1).mean(1).squeeze(1)
    for method, ref, mod in ((1, ref1, sa), (2, ref2, sa), (3, ref3, ca)):
        got = saliency_scores(SimpleNamespace(saliency_attn=mod), t, method)
        assert got.shape == ref.shape == (B, N - 2)
        close(got, ref, 1e-4, f"method {method}")
        assert ref.max() > 3 * ref.mean(), "the test's attention should not be uniform"
    # the modules' own call contracts (what a caller of the reference's classes gets)
    close(sa(tf[:, 2:]), ref1, 1e-4, "SimpleAttention.forward (float input: hi / lo split of x too)")
    w = ca(cp[:, :1], cp[:, 1:])
    assert w.shape == (B, 1, N - 2)
    close(w[:, 0], ref3, 1e-4, "SimpleCrossAttention.forward")
