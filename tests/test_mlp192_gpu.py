"""The fused MLP-branch kernels of the D = 192 student (csrc/mlp192.hip: dkd_mlp192_fwd / dkd_mlp192_bwd), called through the C ABI,
against plain torch fp32 on the same operands (bf16 tensors rounded where the kernel rounds them), and against the unfused launch
sequence of the same library through a whole block ([3P] timm Block reached from /root/reference model/models.py:195)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32
D = 192


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


def rel_l2(got, ref):
    return ((got.float() - ref.float()).norm() / (ref.float().norm() + 1e-20)).item()


def make_problem(M, Hd, seed, rps):
    x1 = rnd(M, D, seed=seed, scale=2.0) + 0.3
    ln_w = 1.0 + 0.1 * rnd(D, seed=seed + 1)
    ln_b = 0.1 * rnd(D, seed=seed + 2)
    w1 = rnd(Hd, D, scale=0.08, seed=seed + 3).to(BF16)          # fc1.weight [hidden, 192]
    b1 = 0.2 * rnd(Hd, seed=seed + 4)
    w2 = rnd(D, Hd, scale=0.05, seed=seed + 5).to(BF16)          # fc2.weight [192, hidden]
    b2 = 0.1 * rnd(D, seed=seed + 6)
    nb = (M + rps - 1) // rps
    sc = torch.tensor([[0.0, 1.0 / 0.9, 1.0 / 0.9][i % 3] for i in range(nb)], device=dev())     # DropPath keep / keep_prob per sample
    return x1, ln_w, ln_b, w1, b1, w2, b2, sc


def forward_ref(x1, ln_w, ln_b, w1, b1, w2, b2, sc_rows):
    """fp32 torch on the operands as the kernel sees them: y2 and h are rounded to bf16 before they enter a GEMM."""
    mu = x1.mean(1, keepdim=True)
    var = ((x1 - mu) ** 2).mean(1, keepdim=True)
    rstd = torch.rsqrt(var + 1e-6)
    y2 = ((x1 - mu) * rstd * ln_w + ln_b).to(BF16)
    pre = y2.float() @ w1.float().t() + b1
    h = F.gelu(pre).to(BF16)
    f = h.float() @ w2.float().t() + b2
    x2 = x1 + sc_rows[:, None] * f
    return dict(mean=mu[:, 0], rstd=rstd[:, 0], y2=y2, pre=pre, h=h, f=f, x2=x2)


@pytest.mark.parametrize("M,Hd,rps,with_scale", [(300, 768, 10, True), (50 * 197, 768, 197, True), (17, 128, 17, False), (4 * 197, 768, 197, False),
                                                 (16 * 16 * 300 + 5, 64, 100, True)])
def test_mlp192_fwd_matches_torch(ops, M, Hd, rps, with_scale):
    """Every output of the fused forward: LayerNorm statistics, y2, h (row-major, for the weight gradients), the tap, x2.  Ragged last
    group (M % 16 != 0), 1 .. 16 groups per workgroup, more groups than 16 per CU (last case: several rounds of workgroups)."""
    x1, ln_w, ln_b, w1, b1, w2, b2, sc = make_problem(M, Hd, 300 + M % 97, rps)
    w2t = w2.t().contiguous()
    sc_rows = sc.repeat_interleave(rps)[:M] if with_scale else torch.ones(M, device=dev())
    ref = forward_ref(x1, ln_w, ln_b, w1, b1, w2, b2, sc_rows)
    nw, nb = 1.0 + 0.1 * rnd(D, seed=7), 0.1 * rnd(D, seed=8)                # the NEXT block's norm1, applied in the same epilogue
    r = ops.mlp192_fwd(x1, ln_w, ln_b, w1, b1, w2t, b2, rowscale=sc if with_scale else None, rows_per_sample=rps, want_tap=True,
                       next_ln=(nw, nb))
    torch.cuda.synchronize()
    ny = F.layer_norm(r["x2"], (D,), nw, nb, 1e-6)
    close(r["next_y"], ny, 1e-2, "next block's LayerNorm output")
    close(r["next_mean"], r["x2"].mean(1), 1e-5, "next mean")
    close(r["next_rstd"], torch.rsqrt(r["x2"].var(1, unbiased=False) + 1e-6), 1e-5, "next rstd")
    close(r["mean"], ref["mean"], 1e-5, "mean")
    close(r["rstd"], ref["rstd"], 1e-5, "rstd")
    close(r["y2"][:M], ref["y2"], 1e-2, "y2")
    close(r["h"][:M], ref["h"], 1e-2, "h")
    close(r["tap"], ref["f"], 1e-2, "tap")
    assert rel_l2(r["tap"], ref["f"]) < 6e-3, rel_l2(r["tap"], ref["f"])
    # the residual stream stays f32: error = the branch's (bf16 operands, f32 accumulation) only
    close(r["x2"] - x1, ref["x2"] - x1, 1e-2, "x2 - x1")
    assert rel_l2(r["x2"] - x1, ref["x2"] - x1) < 6e-3
    # inference form: nothing saved, in place
    x_in = x1.clone()
    r2 = ops.mlp192_fwd(x_in, ln_w, ln_b, w1, b1, w2t, b2, rowscale=sc if with_scale else None, rows_per_sample=rps, save=False, out=x_in)
    torch.cuda.synchronize()
    assert r2["y2"] is None
    close(r2["x2"], r["x2"], 1e-6, "the inference instantiation computes the same x2, in place")


@pytest.mark.parametrize("M,Hd,rps", [(300, 768, 10), (50 * 197, 768, 197), (17, 128, 17), (16 * 16 * 300 + 5, 64, 100)])
def test_mlp192_bwd_matches_torch(ops, M, Hd, rps):
    """The fused backward on the activations the fused forward saved (`pre` in its private order): dF, dH (operands of the weight-gradient
    launch), the LayerNorm backward accumulated into g, dgamma / dbeta accumulated, the scale-cast that opens the attention branch."""
    x1, ln_w, ln_b, w1, b1, w2, b2, s2 = make_problem(M, Hd, 500 + M % 89, rps)
    w2t = w2.t().contiguous()
    s2_rows = s2.repeat_interleave(rps)[:M]
    fw = ops.mlp192_fwd(x1, ln_w, ln_b, w1, b1, w2t, b2, rowscale=s2, rows_per_sample=rps, want_tap=True)
    ref = forward_ref(x1, ln_w, ln_b, w1, b1, w2, b2, s2_rows)
    g0 = rnd(M, D, seed=77)
    gtap = rnd(M, D, seed=78, scale=0.5).to(BF16)
    nb = (M + rps - 1) // rps
    s1 = torch.tensor([[1.0 / 0.95, 0.0, 1.0 / 0.95, 1.0 / 0.95][i % 4] for i in range(nb)], device=dev())
    s1_rows = s1.repeat_interleave(rps)[:M]
    d_w0, d_b0 = rnd(D, seed=79), rnd(D, seed=80)
    # ---- reference (fp32, rounding where the kernel rounds)
    dF = (s2_rows[:, None] * g0 + gtap.float()).to(BF16)
    pre16 = ref["pre"].to(BF16).float().requires_grad_(True)       # the saved pre-activation is bf16
    F.gelu(pre16).backward(dF.float() @ w2.float())
    dH = pre16.grad.to(BF16)
    dT = dH.float() @ w1.float()
    xh = (x1 - ref["mean"][:, None]) * ref["rstd"][:, None]
    gy = dT * ln_w
    dx = ref["rstd"][:, None] * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True))
    g_ref = g0 + dx
    dgam_ref, dbet_ref = d_w0 + (dT * xh).sum(0), d_b0 + dT.sum(0)
    cast_ref = s1_rows[:, None] * g_ref
    # ---- kernel
    g = g0.clone()
    d_w, d_b = d_w0.clone(), d_b0.clone()
    dF_k, dH_k, cast = ops.mlp192_bwd(g, fw["pre"], w2t, w1, x1, ln_w, fw["mean"], fw["rstd"], d_w, d_b, gtap=gtap, s2=s2, s1=s1,
                                      rows_per_sample=rps)
    torch.cuda.synchronize()
    close(dF_k[:M], dF, 1e-2, "dF")
    close(dH_k[:M], dH, 1.5e-2, "dH")
    assert rel_l2(dH_k[:M], dH) < 8e-3, rel_l2(dH_k[:M], dH)
    close(g - g0, dx, 1.5e-2, "LayerNorm backward")
    assert rel_l2(g - g0, dx) < 8e-3, rel_l2(g - g0, dx)
    close(d_w - d_w0, dgam_ref - d_w0, 1e-2, "dgamma")
    close(d_b - d_b0, dbet_ref - d_b0, 1e-2, "dbeta")
    close(cast, cast_ref, 1.5e-2, "cast_out")
    dead = s1_rows == 0
    if dead.any():
        assert cast[dead].abs().max().item() == 0.0, "rows of a dropped sample must open the attention branch with zeros"
    # a second call without the optional operands: no gtap, no scales, no cast
    g2 = g0.clone()
    d_w2, d_b2 = torch.zeros(D, device=dev()), torch.zeros(D, device=dev())
    dF2, dH2, cast2 = ops.mlp192_bwd(g2, fw["pre"], w2t, w1, x1, ln_w, fw["mean"], fw["rstd"], d_w2, d_b2, want_cast=False)
    torch.cuda.synchronize()
    assert cast2 is None
    close(dF2[:M], g0.to(BF16), 1e-6, "dF without scale / tap gradient")


def test_fast_gelu_keeps_nan(ops):
    """a NaN / +inf produced inside the fc1 GEMM must reach the loss, not be clamped to a finite activation -- in the
    GELU epilogues of the NT GEMM and in the fused MLP kernel."""
    M, K, N = 128, 64, 128
    a = rnd(M, K, seed=1).to(BF16)
    a[5, 3] = float("nan")
    a[9, 0] = float("inf")
    b = rnd(N, K, seed=2).abs().to(BF16) + 0.1                       # positive weights: row 9 is +inf everywhere
    bias = rnd(N, seed=3)
    pre = torch.empty(M, N, device=dev(), dtype=BF16)
    h = ops.gemm_nt(a, b, bias=bias, gelu=True, preact=pre)
    torch.cuda.synchronize()
    assert torch.isnan(h[5]).all() and torch.isnan(pre[5]).all(), "NaN pre-activation must stay NaN through the GELU epilogue"
    assert torch.isposinf(h[9].float()).all(), "+inf pre-activation must stay +inf"
    assert torch.isfinite(h[:5].float()).all()
    dh = ops.gemm_nt(rnd(M, K, seed=4).to(BF16), b, dgelu=True, preact=pre)
    torch.cuda.synchronize()
    assert torch.isnan(dh[5]).all(), "gelu'(NaN) must be NaN"
    # fused MLP: a NaN in the residual stream row reaches that row's output (and only that row)
    x1, ln_w, ln_b, w1, b1, w2, b2, _ = make_problem(64, 128, 900, 64)
    x1[7, 100] = float("nan")
    r = ops.mlp192_fwd(x1, ln_w, ln_b, w1, b1, w2.t().contiguous(), b2, want_tap=True)
    torch.cuda.synchronize()
    assert torch.isnan(r["x2"][7]).all() and torch.isnan(r["tap"][7].float()).all()
    ok = torch.ones(64, dtype=torch.bool, device=dev())
    ok[7] = False
    assert torch.isfinite(r["x2"][ok]).all()


@pytest.mark.parametrize("B,tap_layers,droppath", [(3, (0, 1), 0.1), (6, (), 0.0)])
def test_block_with_fused_mlp_equals_the_unfused_launch_sequence(B, tap_layers, droppath, monkeypatch):
    """Whole DeiT-tiny-width blocks (D = 192, hidden 768, N = 197) forward + backward with the fused MLP kernels and the fused qkv +
    attention kernel against the same model on the separate launches (DKD_NO_MLP_FUSION=1, DKD_NO_ATTN_FUSION=1): logits, taps, every
    parameter gradient.  Both paths round the same tensors to bf16
    at the same places, so they agree far inside the bf16-vs-fp32 parity tolerance."""
    from deltakd_amd import vit

    def run(no_fusion):
        for knob in ("DKD_NO_MLP_FUSION", "DKD_NO_ATTN_FUSION"):       # the unfused run also takes the separate qkv GEMM + attention launches
            if no_fusion:
                monkeypatch.setenv(knob, "1")
            else:
                monkeypatch.delenv(knob, raising=False)
        torch.manual_seed(0)
        m = vit.VisionTransformer(192, 2, 3, 10, False, droppath).to(dev()).train()
        with torch.no_grad():
            for blk in m.blocks:
                blk.mlp.fc2.weight.mul_(4.0)
                blk.mlp.fc1.bias.normal_(0, 0.1)
        keep = [torch.tensor([1.0, 0.0, 1.0, 1.0, 1.0, 1.0][:B]) for _ in range(4)]
        m.set_droppath_keep(keep)
        x = rnd(B, 3, 224, 224, seed=11)
        logits, taps = m.forward_with_taps(x, tap_layers)
        loss = (logits * rnd(B, 10, seed=12)).sum()
        for i in tap_layers:
            loss = loss + (taps[i].float() * rnd(B, 197, 192, seed=13 + i)).sum() * 0.01
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach(), [taps[i].detach() for i in tap_layers], {n: p.grad.detach().clone() for n, p in m.named_parameters()}

    z_f, t_f, g_f = run(False)
    z_u, t_u, g_u = run(True)
    close(z_f, z_u, 8e-3, "logits")      # two bf16 pipelines (MLP and attention branch) with different accumulation orders (parity vs fp32: 2e-2)
    for a, b in zip(t_f, t_u):
        close(a, b, 1e-2, "tap")
    worst = max((rel_l2(g_f[n], g_u[n]), n) for n in g_u if g_u[n].norm() > 0)
    assert worst[0] < 1.5e-2, f"gradient of {worst[1]} differs by {worst[0]:.3e} (relative L2)"
    assert os.environ.get("DKD_NO_MLP_FUSION") == "1"      # (the second run really took the other path)


@pytest.mark.parametrize("group", [2, 6])
def test_deferred_weight_gradients_equal_the_per_block_launches(group, monkeypatch):
    """The student defers its blocks' weight gradients and LayerNorm reductions and flushes them several blocks at a time
    (deltakd_amd.vit.flush_wgrads: DKD_WGRAD_GROUP blocks per dkd_block_wgrad_group launch).  Every parameter gradient of a 5-block
    D = 192 model must equal the one-block-at-a-time path (group 1: launched inside dkd_block_bwd) up to the atomics' summation order;
    with 5 blocks and groups of 2 the last flush holds one block.  A backward pass that stops above block 0
    (``backward(inputs=[activation after block 2])``) leaves its pending gradients to the callback queued on the autograd engine."""
    from deltakd_amd import vit

    def grads(g, partial):
        monkeypatch.setenv("DKD_WGRAD_GROUP", str(g))
        torch.manual_seed(3)
        m = vit.VisionTransformer(192, 5, 3, 10, False, 0.0, img_size=64, patch_size=16).to(dev()).train()
        img = torch.randn(6, 3, 64, 64, device=dev(), generator=torch.Generator(device=dev()).manual_seed(9))
        B, N = 6, m.num_tokens
        x = vit._EmbedFn.apply(m.pos_embed, m, img)
        xs = []
        for i in range(5):
            x, _ = vit._BlockFn.apply(x, m, i, B, N, None, None, False)
            xs.append(x)
        loss = m.forward_head(x, B).float().square().mean()
        if partial:
            loss.backward(inputs=[xs[2]])              # the engine runs the head and blocks 4, 3 only
        else:
            loss.backward()
        torch.cuda.synchronize()
        assert not vit._rt(m).get("wgrad_pending"), "nothing may stay pending after backward()"
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

    for partial in (False, True):
        ref = grads(1, partial)
        got = grads(group, partial)
        assert ref.keys() == got.keys() and "blocks.4.mlp.fc1.weight" in got and ("blocks.0.attn.qkv.weight" in got) == (not partial)
        for n in ref:
            err = (got[n] - ref[n]).norm().item() / (ref[n].norm().item() + 1e-20)
            assert err <= 2e-5, (partial, n, err)


def test_aborted_backward_does_not_leak_into_the_next_step(monkeypatch):
    """a backward pass that dies half-way leaves deferred weight gradients pending.  The next grad-mode forward must DROP
    them (it runs outside any backward pass), not flush them into .grad: a zero_grad() -> forward -> backward loop would otherwise add
    the aborted pass's partial gradients to the next step.  (A forward INSIDE a backward pass still flushes: vit._in_backward_pass.)"""
    from deltakd_amd import vit
    monkeypatch.setenv("DKD_WGRAD_GROUP", "6")
    torch.manual_seed(3)
    m = vit.VisionTransformer(192, 5, 3, 10, False, 0.0, img_size=64, patch_size=16).to(dev()).train()
    img = torch.randn(6, 3, 64, 64, device=dev(), generator=torch.Generator(device=dev()).manual_seed(9))
    real = vit._block_backward
    state = {"abort": True}

    def dying(g, gtap, model, blk, saved, idx=None):
        if state["abort"] and idx == 2:
            raise RuntimeError("boom")                  # blocks 4 and 3 have run their backward and deferred their weight gradients
        return real(g, gtap, model, blk, saved, idx)
    monkeypatch.setattr(vit, "_block_backward", dying)
    with pytest.raises(RuntimeError, match="boom"):
        m(img).float().square().mean().backward()
    torch.cuda.synchronize()
    assert vit._rt(m).get("wgrad_pending"), "the aborted pass should have left deferred weight gradients behind (else this test tests nothing)"
    state["abort"] = False
    for p in m.parameters():                            # zero_grad FIRST, then forward: the order the advisor named
        if p.grad is not None:
            p.grad.zero_()
    m(img).float().square().mean().backward()
    torch.cuda.synchronize()
    assert not vit._rt(m).get("wgrad_pending")
    got = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    torch.manual_seed(3)                                # reference: the same step on a fresh copy of the model
    m2 = vit.VisionTransformer(192, 5, 3, 10, False, 0.0, img_size=64, patch_size=16).to(dev()).train()
    m2.load_state_dict(m.state_dict())
    m2(img).float().square().mean().backward()
    torch.cuda.synchronize()
    checked = 0
    for n, p in m2.named_parameters():
        if p.grad is None:
            continue
        err = (got[n] - p.grad).norm().item() / (p.grad.norm().item() + 1e-20)
        assert err <= 2e-5, (n, err)
        checked += 1
    assert checked > 40
