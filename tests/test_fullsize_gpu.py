"""Full-size checks at BASELINE.json's headline shape (bs 256, DeiT-tiny <- DeiT-base-distilled, 224^2): several kernels only take
their large-problem paths here (persistent 256^2 GEMM tiles, LDS-DMA-ring wgrad, persistent attention), where the CPU oracle is
too slow to run end to end.  Every test is a size-independent property or a spot check against fp32 torch on a slice:

  * GEMM / wgrad rows and columns against fp32 matmul on slices taken at the start, middle and ragged end of the problem;
  * per-sample independence: the teacher's forward of 256 images equals its forward of a 24-image sub-batch on those images
    (the sub-batch runs the small-problem kernels, which the golden fixtures pin);
  * linearity of the gradient in the batch: a student step on 256 images = the mean of two steps on 128;
  * low-rank targets: orthonormal basis, idempotent projection, singular values = the leading eigenvalues of the Gram matrix
    (torch.linalg.eigvalsh of the 768 x 768 Gram is the known answer), residual energy decreasing with the rank.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
DEV = "cuda:0"
B, NT, NS_ = 256, 198, 197            # images, teacher tokens (cls + dist + 196), student tokens


@pytest.fixture(scope="module")
def ops():
    from deltakd_amd import ops as o
    o.lib()
    return o


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(*shape, generator=g, device=DEV) * scale


def rel(got, ref):
    return ((got.float() - ref.float()).abs().max() / ref.float().abs().max().clamp_min(1e-12)).item()


SLICES = ((0, 300), (25000, 25300), (-333, None))     # first rows, middle, ragged tail


@pytest.mark.parametrize("name,N,K,epi", [("qkv", 2304, 768, "bias"), ("fc1", 3072, 768, "gelu"), ("proj", 768, 768, "resid"),
                                           ("fc2", 768, 3072, "resid_tap")])
def test_teacher_gemms_full_size(ops, name, N, K, epi):
    M = B * NT
    a = rnd(M, K, seed=1).to(BF16)
    w = rnd(N, K, scale=0.05, seed=2).to(BF16)
    bias = rnd(N, seed=3)
    kw = dict(bias=bias)
    if epi == "gelu":
        kw.update(gelu=True)
    resid = tap = None
    if epi.startswith("resid"):
        resid = rnd(M, N, seed=4)
        kw.update(resid=resid, out_f32=True)
        if epi == "resid_tap":
            tap = torch.empty(M, N, device=DEV, dtype=BF16)
            kw.update(tap=tap)
    out = ops.gemm_nt(a, w, **kw)
    for lo, hi in SLICES:
        sl = slice(lo, hi)
        ref = a[sl].float() @ w.float().t() + bias
        if epi == "gelu":
            ref = torch.nn.functional.gelu(ref)
        if tap is not None:
            assert rel(tap[sl], ref) < 1e-2, (name, "tap", lo)
        if resid is not None:
            ref = ref + resid[sl]
        assert rel(out[sl], ref) < (1e-2 if out.dtype == BF16 else 1e-4), (name, lo)


@pytest.mark.parametrize("name,K,tap", [("proj", 768, False), ("fc2", 3072, True)])
def test_teacher_residual_gemms_three_batches_per_call(ops, name, K, tap):
    """The teacher's N = 768 GEMMs when three batches go through it in one call (DistillationLoss.prefetch_group): M = 3 * 50 688 rows
    are 1 782 tiles of 256 x 256 = 6.96 rounds, so the whole problem runs on the persistent kernel with the f32-residual epilogue
    (a dispatch no other test reaches).  In-place residual (C aliases resid), as the inference path uses it."""
    M, N = 3 * B * NT, 768
    a = rnd(M, K, seed=41).to(BF16)
    w = rnd(N, K, scale=0.05, seed=42).to(BF16)
    bias = rnd(N, seed=43)
    x = rnd(M, N, seed=44)
    x0 = x.clone()
    tp = torch.empty(M, N, device=DEV, dtype=BF16) if tap else None
    out = ops.gemm_nt(a, w, out=x, bias=bias, resid=x, tap=tp)
    assert out.data_ptr() == x.data_ptr()
    for lo, hi in ((0, 300), (76000, 76300), (M - 333, M)):
        ref = a[lo:hi].float() @ w.float().t() + bias
        if tap:
            assert rel(tp[lo:hi], ref) < 1e-2, (name, "tap", lo)
        assert rel(x[lo:hi], ref + x0[lo:hi]) < 1e-4, (name, lo)


@pytest.mark.parametrize("N1,N2", [(768, 192), (192, 768), (576, 192), (192, 192)])
def test_student_wgrads_full_size(ops, N1, N2):
    M = B * NS_
    a = rnd(M, N1, seed=5).to(BF16)
    b = rnd(M, N2, seed=6).to(BF16)
    out = torch.zeros(N1, N2, device=DEV)
    cs = torch.zeros(N1, device=DEV)
    ops.gemm_tn(a, b, out, colsum=cs)
    ref = a.float().t() @ b.float()
    assert rel(out, ref) < 1e-4
    assert rel(cs, a.float().sum(0)) < 1e-4
    ops.gemm_tn(a, b, out)                                   # accumulates
    assert rel(out, 2 * ref) < 1e-4


def test_attention_full_size(ops):
    H = 12
    qkv = rnd(B * NT, 3 * H * 64, scale=1.5, seed=7).to(BF16)
    out, lse = ops.attn_fwd(qkv, B, NT, H)
    for b0 in (0, 117, B - 2):
        sub = qkv.view(B, NT, -1)[b0:b0 + 2].reshape(2 * NT, -1)
        q, k, v = sub.float().view(2, NT, 3, H, 64).permute(2, 0, 3, 1, 4)
        s = (q @ k.transpose(-1, -2)) * 0.125
        ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(2 * NT, H * 64)
        assert rel(out.view(B, NT, -1)[b0:b0 + 2].reshape(2 * NT, -1), ref) < 1.5e-2
        assert rel(lse[b0:b0 + 2], torch.logsumexp(s, -1)) < 1e-3


@pytest.mark.parametrize("H,N", [(3, 197), (12, 198)])
def test_attention_backward_full_size(ops, H, N):
    """The per-head persistent backward at the headline's head counts (768 student / 3072 teacher-shaped heads, three resp. twelve
    passes per workgroup): gradients of the first, a middle and the last images against autograd through the fp32 reference, and
    linearity in dO over ALL heads (the recomputed probabilities do not depend on dO): bwd(dO1 + dO2) = bwd(dO1) + bwd(dO2)."""
    qkv = rnd(B * N, 3 * H * 64, scale=1.5, seed=8).to(BF16)
    out, lse = ops.attn_fwd(qkv, B, N, H)
    do1 = rnd(B * N, H * 64, seed=9).to(BF16)
    do2 = rnd(B * N, H * 64, seed=10).to(BF16)
    g1 = ops.attn_bwd(qkv, out, do1, lse, B, N, H)
    for b0 in (0, 131, B - 2):
        sub = qkv.view(B, N, -1)[b0:b0 + 2].reshape(2 * N, -1).float().requires_grad_(True)
        q, k, v = sub.view(2, N, 3, H, 64).permute(2, 0, 3, 1, 4)
        o = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v
        o.transpose(1, 2).reshape(2 * N, H * 64).backward(do1.view(B, N, -1)[b0:b0 + 2].reshape(2 * N, -1).float())
        got = g1.view(B, N, 3, H, 64)[b0:b0 + 2].float()
        ref = sub.grad.view(2, N, 3, H, 64)
        for i, nm in enumerate("qkv"):
            assert rel(got[:, :, i], ref[:, :, i]) < 3e-2, (nm, b0)
    g2 = ops.attn_bwd(qkv, out, do2, lse, B, N, H)
    g12 = ops.attn_bwd(qkv, out, (do1.float() + do2.float()).to(BF16), lse, B, N, H)
    assert rel(g12, g1.float() + g2.float()) < 2e-2          # three bf16 roundings apart


@pytest.fixture(scope="module")
def models():
    from deltakd_amd import vit
    torch.manual_seed(42)
    t = vit.create_model("deit_base_distilled_patch16_224", num_classes=1000).to(DEV).eval()
    for p in t.parameters():
        p.requires_grad = False
    return t


def test_teacher_forward_is_per_sample_independent(models):
    t = models
    x = rnd(B, 3, 224, 224, seed=8)
    with torch.no_grad():
        logits, taps = t.forward_with_taps(x, (0, 1, 11))
        idx = torch.tensor([0, 1, 2, 3, 60, 61, 62, 63, 100, 101, 130, 131, 180, 181, 200, 201, 220, 221, 250, 251, 252, 253, 254, 255],
                           device=DEV)
        l2, taps2 = t.forward_with_taps(x[idx].contiguous(), (0, 1, 11))
    assert rel(logits[idx], l2) < 2e-2
    for i in (0, 1, 11):
        assert rel(taps[i].view(B, NT, -1)[idx], taps2[i].view(len(idx), NT, -1)) < 2e-2, i


def test_student_gradient_is_linear_in_the_batch(models):
    """soft distillation (student distilled tiny, drop_path 0, no mixup): grad over 256 images = mean of the grads over two halves."""
    import copy
    from oracle import loss_ref
    from deltakd_amd import vit
    from deltakd_amd.losses import DistillationLoss, call_base_loss
    args = loss_ref.default_args(distillation_type="soft", dataset="imagenet", smoothing=0.1)
    torch.manual_seed(1)
    s = vit.create_model("deit_tiny_distilled_patch16_224", num_classes=1000, drop_path_rate=0.0).to(DEV).train()
    s.set_distilled_training(True)
    crit = DistillationLoss(call_base_loss(args), models, "soft", 0.1, 3.0)
    x = rnd(B, 3, 224, 224, seed=9)
    y = torch.randint(0, 1000, (B,), device=DEV, generator=torch.Generator(device=DEV).manual_seed(10))

    def grads(xb, yb):
        for p in s.parameters():
            p.grad = None
        loss = crit(xb, s(xb), s, None, yb, args)
        loss.backward()
        return loss.item(), {n: p.grad.detach().clone() for n, p in s.named_parameters() if p.grad is not None}

    l_full, g_full = grads(x, y)
    l_a, g_a = grads(x[:128].contiguous(), y[:128])
    l_b, g_b = grads(x[128:].contiguous(), y[128:])
    assert math.isfinite(l_full) and abs(l_full - 0.5 * (l_a + l_b)) < 2e-3 * abs(l_full)
    worst = 0.0
    for n, g in g_full.items():
        h = 0.5 * (g_a[n] + g_b[n])
        err = (g - h).norm() / h.norm().clamp_min(1e-12)
        worst = max(worst, err.item())
        assert err < 3e-2, (n, err.item())
    assert len(g_full) > 100 and worst > 0.0


def test_lowrank_targets_properties_full_size(ops):
    from deltakd_amd.losses import LowRankTargets
    Dt, r, npre = 768, 64, 2
    g = torch.Generator(device=DEV).manual_seed(11)
    basis = torch.randn(Dt, Dt, device=DEV, generator=g) * torch.logspace(0, -2.5, Dt, device=DEV)     # decaying spectrum
    T = (torch.randn(B * NT, Dt, device=DEV, generator=g) @ basis.t()).to(BF16).view(B, NT, Dt)
    solver = LowRankTargets()
    tg = solver([T], npre, r)[0]                               # [B*196, r] = T_patches V_r = U_r S_r
    patches = T[:, npre:].reshape(-1, Dt).float()
    G = patches.t() @ patches
    ev = torch.linalg.eigvalsh(G.double()).flip(0)[:r].float()
    s2 = (tg * tg).sum(0)                                      # squared singular values, descending
    assert rel(s2, ev) < 2e-3
    V = torch.linalg.lstsq(patches, tg).solution               # recover V_r: patches V = tg
    eye = torch.eye(r, device=DEV)
    assert (V.t() @ V - eye).abs().max() < 2e-2               # orthonormal basis (bf16 operand rounding in the projection)
    assert rel((tg @ V.t()) @ V, tg) < 2e-2                   # projection is idempotent
    res = [(patches - (patches @ V[:, :k]) @ V[:, :k].t()).pow(2).sum().item() for k in (16, 32, 64)]
    assert res[0] > res[1] > res[2] > 0


def test_lowrank_warm_start_tracks_real_teacher_taps(models):
    """The warm-started tracking step (ONE power + Rayleigh-Ritz step per batch) on REAL teacher taps over CHANGING batches: the
    deit_base_distilled teacher of the headline config, 6 different batches of 32 images, taps of blocks 0, 1, 11, each batch compared
    with torch.linalg.svd of that batch's own [32 * 196, 768] matrices (the reference's computation, model/loss.py:321).
    Sign- and rotation-invariant measures of what the loss consumes, U_k S_k = T V_k (k = 64):
      * captured energy ||T V_k||_F^2 / sum_{i<=k} sigma_i^2  (1 = the optimal rank-k subspace);
      * the singular values themselves, column by column;
      * leading columns against the SVD's, up to sign.
    The first call is a cold start (16 power steps); calls 2-6 are warm."""
    from deltakd_amd.losses import LowRankTargets
    t = models
    solver = LowRankTargets()
    k, npre = 64, 2
    worst_energy, worst_sv = 1.0, 0.0
    for call in range(6):
        x = rnd(32, 3, 224, 224, seed=100 + call)
        with torch.no_grad():
            _, taps = t.forward_with_taps(x, (0, 1, 11))
        sel = [taps[0], taps[1], taps[11]]
        tg = solver(sel, npre, k)
        for li, (tap, got) in enumerate(zip(sel, tg)):
            T = tap[:, npre:].reshape(-1, tap.shape[-1]).float().cpu().double()
            U, S, Vh = torch.linalg.svd(T, full_matrices=False)
            ref = (U[:, :k] * S[:k])
            got = got.cpu().double()
            energy = (got ** 2).sum().item() / (S[:k] ** 2).sum().item()
            worst_energy = min(worst_energy, energy)
            sv = got.norm(dim=0)
            rel_sv = ((sv - S[:k]).abs() / S[0]).max().item()
            worst_sv = max(worst_sv, rel_sv)
            print(f"call {call} layer {li}: captured energy {energy:.5f}  max |sigma err| / sigma_1 {rel_sv:.2e}  "
                  f"sigma_1 {S[0].item():.1f} sigma_64 {S[63].item():.1f} sigma_97 {S[96].item():.1f}")
            assert energy > 0.99, (call, li, energy)
            assert rel_sv < 2e-2, (call, li, rel_sv)
            # columns whose singular value is separated from its neighbours by > 5 % are well conditioned: compare up to sign
            gap = torch.minimum((S[:k] - S[1:k + 1]), torch.cat([torch.tensor([1e9], dtype=S.dtype), S[:k - 1] - S[1:k]])) / S[:k]
            well = torch.nonzero(gap > 0.05).flatten()[:8]
            for j in well.tolist():
                sgn = torch.sign((got[:, j] * ref[:, j]).sum())
                err = (got[:, j] * sgn - ref[:, j]).norm() / ref[:, j].norm()
                assert err < 5e-2, (call, li, j, err.item(), gap[j].item())
    assert worst_energy <= 1.0 + 1e-3


def _exact_lowrank(tap, npre, k):
    """The reference's U_k S_k (model/loss.py:318-326) of one teacher tap at the headline batch, in float64: T = U S V^T  =>  U_k S_k = T V_k
    with V the eigenvectors of the Gram matrix T^T T (the [50 176, 768] LAPACK SVD takes 3 s per matrix on the host; in float64 the
    Gram route loses nothing at these condition numbers, and the bs-32 test above pins the same quantities against torch.linalg.svd
    itself).  -> (targets [M, k] f64, singular values [Dt] f64)."""
    T = tap[:, npre:].reshape(-1, tap.shape[-1]).double()
    ev, V = torch.linalg.eigh((T.t() @ T).cpu())
    ev, V = ev.flip(0).clamp_min(0), V.flip(1).to(T.device)
    return T @ V[:, :k], ev.sqrt().to(T.device)


def test_lowrank_tracking_at_the_headline_batch(models):
    """The LRKD targets of the TIMED path: ``LowRankTargets`` with its defaults (round 5: 8 power steps + a converged Rayleigh-Ritz step per
    batch; the bounds below were set for round 4's one-step tracker and are kept) against
    the reference's exact SVD per batch (model/loss.py:318-326), at the benchmarked size -- batches of 256 through the deit_base_distilled
    teacher, taps of blocks 0, 1, 11, rank 64 -- over a sequence of 24 calls on 4 rotating batches (what bench.py feeds it).  Every call:
    the invariant-subspace residual of the tracked basis (``LowRankTargets.residual``; < 5e-2, measured 7e-3).  Calls 0-5 and every 8th after: against the exact
    float64 decomposition of that batch's own matrices --
      * captured energy ||T V_k||_F^2 / sum_{i<=k} sigma_i^2 >= 0.998 (1 = the optimal rank-k subspace; measured 0.9992-0.9995);
      * every singular value to 2e-3 sigma_1 (measured 2e-4);
      * columns whose singular value is separated from both neighbours by > 5 % against the exact ones up to sign: 1e-2 (measured 3e-5);
      * the LRKD loss  sum_i w_i mse(targets_i, align_i(student tap_i))  computed with the tracked and with the exact targets, signs aligned:
        for a randomly initialised student (the state the benchmark runs in) to 2e-3 (measured 4e-4), and for a surrogate of a TRAINED
        student -- aligned features equal to half the exact target on the well-separated columns -- to 5e-3 (measured 5e-4).  (Inside a cluster of nearly equal singular
        values the individual vectors are ill-conditioned: LAPACK's own choice there is as arbitrary as its signs, SURVEY.md section 0
        item 9, so no bound on those columns' contribution is claimed; the random teacher's flat spectrum is the worst case for this.)"""
    from types import SimpleNamespace
    from deltakd_amd import vit
    from deltakd_amd.losses import LowRankTargets
    from deltakd_amd.models import attach_aux
    t = models
    k, npre, w = 64, 2, (0.2, 0.2, 0.2)
    torch.manual_seed(5)
    stu = vit.create_model("deit_tiny_patch16_224", num_classes=1000)
    attach_aux(stu, t, "lrkd", SimpleNamespace(lrkd_rank=k, dataset="imagenet-1k"))
    stu.to(DEV).eval()
    gen = torch.Generator(device=DEV).manual_seed(77)
    batches = [torch.randn(B, 3, 224, 224, device=DEV, generator=gen) for _ in range(4)]
    solver = LowRankTargets()
    worst = dict(energy=1.0, sv=0.0, col=0.0, loss_random=0.0, loss_trained=0.0, residual=0.0)
    checked = 0
    for call in range(24):
        x = batches[call % 4]
        with torch.no_grad():
            _, taps = t.forward_with_taps(x, (0, 1, 11))
            sel = [taps[0], taps[1], taps[11]]
            tg = solver(sel, npre, k)
            res = solver.residual(_gram_of(sel, npre), k)
        worst["residual"] = max(worst["residual"], max(res))
        assert max(res) < 5e-2, (call, res)                 # a basis drifting out of the invariant subspace would show here first
        if not (call < 6 or call % 8 == 0):
            continue
        with torch.no_grad():
            _, staps = stu.forward_with_taps(x, (0, 1, 11))
            feats = [stu.align[i](staps[b][:, 1:]).reshape(-1, k).double() for i, b in enumerate((0, 1, 11))]
        loss = {"tracked_random": 0.0, "exact_random": 0.0, "tracked_trained": 0.0, "exact_trained": 0.0}
        for li, (tap, got, sf) in enumerate(zip(sel, tg, feats)):
            ref, S = _exact_lowrank(tap, npre, k)
            got = got.double()
            energy = ((got ** 2).sum() / (S[:k] ** 2).sum()).item()
            rel_sv = ((got.norm(dim=0) - S[:k]).abs() / S[0]).max().item()
            worst["energy"], worst["sv"] = min(worst["energy"], energy), max(worst["sv"], rel_sv)
            assert energy > 0.998, (call, li, energy)
            assert rel_sv < 2e-3, (call, li, rel_sv)
            sgn = torch.sign((got * ref).sum(0))
            ref = ref * sgn                                  # the exact targets with the tracker's column signs
            gap = torch.minimum(S[:k] - S[1:k + 1], torch.cat([S[:1] * 1e9, S[:k - 1] - S[1:k]])) / S[:k]
            well = gap > 0.05
            for j in torch.nonzero(well).flatten()[:8].tolist():
                err = ((got[:, j] - ref[:, j]).norm() / ref[:, j].norm()).item()
                worst["col"] = max(worst["col"], err)
                assert err < 1e-2, (call, li, j, err, gap[j].item())
            trained = 0.5 * ref * well                       # a student that has learned the well-conditioned columns
            loss["tracked_random"] += w[li] * ((got - sf) ** 2).mean().item()
            loss["exact_random"] += w[li] * ((ref - sf) ** 2).mean().item()
            loss["tracked_trained"] += w[li] * ((got - trained) ** 2).mean().item()
            loss["exact_trained"] += w[li] * ((ref - trained) ** 2).mean().item()
        d_r = abs(loss["tracked_random"] - loss["exact_random"]) / loss["exact_random"]
        d_t = abs(loss["tracked_trained"] - loss["exact_trained"]) / loss["exact_trained"]
        worst["loss_random"], worst["loss_trained"] = max(worst["loss_random"], d_r), max(worst["loss_trained"], d_t)
        print(f"call {call}: lrkd loss tracked {loss['tracked_random']:.6f} exact {loss['exact_random']:.6f} (rel {d_r:.2e}); trained-student surrogate "
              f"rel {d_t:.2e}; residual {max(res):.2e}")
        assert d_r < 2e-3, (call, loss)
        assert d_t < 5e-3, (call, loss)
        checked += 1
    print("worst over the sequence:", worst)
    assert checked >= 8 and solver.reconverged == 0


def _shifting_batches(n, seed, jump_at=None):
    """n never-repeating batches of 256 whose statistics move from call to call (what a real loader feeds the tracker; i.i.d. Gaussian
    noise -- what bench.py rotates -- has the same covariance every batch): every image is a random mixture of 12 smooth 'class
    prototypes' plus noise; the sharpness of the mixture, the contrast and the noise level drift with the call index; from ``jump_at``
    on the prototypes are a different set at 3 x the contrast (a deliberate distribution jump)."""
    g = torch.Generator(device=DEV).manual_seed(seed)

    def bank():
        p = torch.randn(12, 3, 28, 28, device=DEV, generator=g)
        return torch.nn.functional.interpolate(p, size=224, mode="bilinear", align_corners=False)          # smooth: low-frequency content
    protos = bank()
    jumped = bank() * 3.0
    for t in range(n):
        pb = jumped if jump_at is not None and t >= jump_at else protos
        temp = 0.5 + 2.5 * ((t * 7) % 10) / 10.0
        mix = torch.softmax(torch.randn(B, 12, device=DEV, generator=g) * temp, 1)
        contrast = 0.6 + 0.25 * (t % 5)
        noise = 0.3 + 0.15 * ((t * 3) % 7)
        x = contrast * torch.einsum("bc,cdhw->bdhw", mix, pb)
        x += noise * torch.randn(B, 3, 224, 224, device=DEV, generator=g)
        yield x


def _tracking_errors(sel, tg, npre, k):
    """(captured energy, worst singular-value error / sigma_1) of tracked targets against the exact float64 decomposition, worst layer."""
    energy, sv = 1.0, 0.0
    for tap, got in zip(sel, tg):
        _, S = _exact_lowrank(tap, npre, k)
        got = got.double()
        energy = min(energy, ((got ** 2).sum() / (S[:k] ** 2).sum()).item())
        sv = max(sv, ((got.norm(dim=0) - S[:k]).abs() / S[0]).max().item())
    return energy, sv


def _student_features(models_teacher, k):
    """A randomly initialised DeiT-tiny with the lrkd align modules: what the LRKD term compares the targets with."""
    from types import SimpleNamespace
    from deltakd_amd import vit
    from deltakd_amd.models import attach_aux
    torch.manual_seed(5)
    stu = vit.create_model("deit_tiny_patch16_224", num_classes=1000)
    attach_aux(stu, models_teacher, "lrkd", SimpleNamespace(lrkd_rank=k, dataset="imagenet-1k"))
    return stu.to(DEV).eval()


def _lrkd_term_errors(sel, tg, feats, npre, k, w=(0.2, 0.2, 0.2)):
    """The LRKD addend  sum_i w_i mse(targets_i, student_i)  (model/loss.py:326-329) computed with the product's targets and with the exact
    float64 U_k S_k of the same matrices, column signs aligned: relative difference for (a) the given student features, (b) a surrogate
    of a TRAINED student (features = half the exact target on the columns whose singular value is > 5 % away from both neighbours)."""
    loss = {"got_random": 0.0, "exact_random": 0.0, "got_trained": 0.0, "exact_trained": 0.0}
    for li, (tap, got, sf) in enumerate(zip(sel, tg, feats)):
        ref, S = _exact_lowrank(tap, npre, k)
        got = got.double()
        ref = ref * torch.sign((got * ref).sum(0))
        gap = torch.minimum(S[:k] - S[1:k + 1], torch.cat([S[:1] * 1e9, S[:k - 1] - S[1:k]])) / S[:k]
        trained = 0.5 * ref * (gap > 0.05)
        loss["got_random"] += w[li] * ((got - sf) ** 2).mean().item()
        loss["exact_random"] += w[li] * ((ref - sf) ** 2).mean().item()
        loss["got_trained"] += w[li] * ((got - trained) ** 2).mean().item()
        loss["exact_trained"] += w[li] * ((ref - trained) ** 2).mean().item()
    return (abs(loss["got_random"] - loss["exact_random"]) / loss["exact_random"],
            abs(loss["got_trained"] - loss["exact_trained"]) / loss["exact_trained"])


TERM_TOL = 1e-2          # tests/test_parity_gpu.py: every addend of a loss within 1 % of the reference's


@pytest.mark.parametrize("mode,min_energy,max_sv,max_term", [("default", 0.9999, 5e-4, 1e-3), ("fast", 0.98, 3e-2, 3e-2)])
def test_lowrank_tracking_on_fresh_shifting_batches(models, mode, min_energy, max_sv, max_term):
    """The LRKD targets where a real run lives: 24 calls on batches of 256 that NEVER repeat and whose statistics shift between calls
    (``_shifting_batches``: prototype mixtures of drifting sharpness / contrast / noise), then a deliberate distribution jump (other
    prototypes, 3 x the contrast) with the residual monitor on (DKD_LRKD_MONITOR=1's setting).  Against the exact float64 decomposition
    of each batch's own matrices (what model/loss.py:318-326 computes):

      * ``default`` = ``LowRankTargets()`` as bench.py and tools/train.py build it (round 5: 8 power steps + a converged Rayleigh-Ritz
        step on every batch, ``LowRankTargets.EXACT``): energy >= 0.9999 of the optimal rank-64 subspace's (measured 0.99994), singular
        values to 5e-4 sigma_1 (2.2e-4), and -- VERDICT round 4, item 1(a) -- the LRKD ADDEND ITSELF, computed with these targets and
        with the exact ones (signs aligned) for a randomly initialised student and for a trained-student surrogate, within 1e-3
        (measured 6e-5), i.e. ten times inside ``TERM_TOL`` = 1e-2, the bound every addend of every loss has to meet;
      * ``fast`` = ``--lrkd-fast`` (round 4's default: one tracking step, <= 2 Jacobi sweeps): energy 0.986-0.999, singular values to
        2.1e-2 sigma_1, LRKD addend off by up to 8e-3 on this sequence (1.3e-2 in the numpy model of a harsher one) -- at the edge of
        ``TERM_TOL`` instead of well inside it, which is why it is no longer the default; asserted at its measured class so that a
        regression of the opt-in mode still shows.
    After the jump the basis the call RETURNS is inside the residual bound whatever the first residual on the new distribution was."""
    from deltakd_amd.losses import LowRankTargets
    t = models
    k, npre = 64, 2
    solver = LowRankTargets() if mode == "default" else LowRankTargets(**LowRankTargets.FAST)
    if mode == "default":
        assert (solver.warm_iters, solver.ritz_sweeps) == (LowRankTargets.EXACT["warm_iters"], LowRankTargets.EXACT["ritz_sweeps"]), \
            "the default mode must be the converge-every-batch setting"
    stu = _student_features(t, k)
    worst = dict(energy=1.0, sv=0.0, residual=0.0, term_random=0.0, term_trained=0.0)
    n, jump = 28, 24
    for call, x in enumerate(_shifting_batches(n, seed=123, jump_at=jump)):
        if call == jump:
            solver.monitor_every, solver.monitor_bound = 1, 5e-2
        with torch.no_grad():
            _, taps = t.forward_with_taps(x, (0, 1, 11))
            sel = [taps[0], taps[1], taps[11]]
            tg = solver(sel, npre, k)
            res = max(solver.residual(_gram_of(sel, npre), k))
        if call >= jump:
            print(f"call {call} (after the jump): residual before the monitor acted {max(solver.last_residual):.2e}, returned basis {res:.2e}, "
                  f"re-converged so far {solver.reconverged}")
            assert res < 5e-2, (call, res, solver.last_residual)
            continue
        worst["residual"] = max(worst["residual"], res)
        if 1 <= call <= 7 or call % 4 == 0 and call > 0:
            energy, sv = _tracking_errors(sel, tg, npre, k)
            with torch.no_grad():
                _, staps = stu.forward_with_taps(x, (0, 1, 11))
                feats = [stu.align[i](staps[b][:, 1:]).reshape(-1, k).double() for i, b in enumerate((0, 1, 11))]
            d_r, d_t = _lrkd_term_errors(sel, tg, feats, npre, k)
            worst["energy"], worst["sv"] = min(worst["energy"], energy), max(worst["sv"], sv)
            worst["term_random"], worst["term_trained"] = max(worst["term_random"], d_r), max(worst["term_trained"], d_t)
            print(f"call {call}: energy {energy:.5f}, singular values to {sv:.2e} sigma_1, residual {res:.2e}; LRKD addend vs exact targets: "
                  f"random student {d_r:.2e}, trained-student surrogate {d_t:.2e}")
    print(f"{mode} (warm_iters {solver.warm_iters}, ritz_sweeps {solver.ritz_sweeps}): worst over the never-repeating, shifting sequence:", worst,
          "re-converged after the jump:", solver.reconverged)
    assert worst["residual"] < 5e-2 and worst["energy"] > min_energy and worst["sv"] < max_sv, worst
    assert worst["term_random"] < max_term and worst["term_trained"] < max_term, worst
    if mode == "default":
        assert max_term <= TERM_TOL / 10
    assert solver.reconverged <= n - jump


def test_lowrank_converge_every_batch_is_the_reference_exact_mode(models):
    """Item 4(b): ``--lrkd-exact`` (= --lrkd-warm-iters 8 --lrkd-ritz-sweeps 12, ``LowRankTargets.EXACT``) converges the basis on EVERY
    batch instead of tracking it -- the setting that stands for the reference's per-batch ``torch.linalg.svd`` (model/loss.py:318-326).
    On fresh shifting batches of 256 it must reproduce the exact decomposition an order of magnitude tighter than the tracking bounds:
    energy >= 0.9998 (measured 0.99994), singular values to 5e-4 sigma_1 (2.2e-4), residual < 1e-2 (7e-4).  Since round 5 this IS the
    default of bench.py / tools/train.py (dkd_lowrank_chain: +1.9 % step time against --lrkd-fast on the same box; round 4: +14 %)."""
    from deltakd_amd.losses import LowRankTargets
    t = models
    k, npre = 64, 2
    solver = LowRankTargets(**LowRankTargets.EXACT)
    assert solver.warm_iters == 8 and solver.ritz_sweeps == 12
    worst = dict(energy=1.0, sv=0.0, residual=0.0)
    for call, x in enumerate(_shifting_batches(5, seed=321)):
        with torch.no_grad():
            _, taps = t.forward_with_taps(x, (0, 1, 11))
            sel = [taps[0], taps[1], taps[11]]
            tg = solver(sel, npre, k)
            res = max(solver.residual(_gram_of(sel, npre), k))
        energy, sv = _tracking_errors(sel, tg, npre, k)
        worst = dict(energy=min(worst["energy"], energy), sv=max(worst["sv"], sv), residual=max(worst["residual"], res))
        print(f"exact mode, call {call}: energy {energy:.6f}, singular values to {sv:.2e} sigma_1, residual {res:.2e}")
        assert energy > 0.9998 and sv < 5e-4 and res < 1e-2, (call, energy, sv, res)
    print("exact mode, worst:", worst)


def _gram_of(taps, npre):
    """f32 Gram matrices [L, Dt, Dt] of the prefix-stripped taps (torch; only the upper 128-tiles are read by ``residual``)."""
    out = []
    for tp in taps:
        T = tp[:, npre:].reshape(-1, tp.shape[-1]).float()
        out.append(T.t() @ T)
    return torch.stack(out)


@pytest.mark.parametrize("blocks", [3, 6])
def test_grouped_weight_gradients_one_xcd_per_split(ops, blocks):
    """dkd_gemm_tn_group at the student's backward shapes, 3 and 6 blocks per launch: 57 / 114 tiles = 8 / 4 M splits, the launches that
    take the one-XCD-per-(problem, split) placement of round 5 (csrc/gemm.hip: TnGroup.xsplits; every block of the launch gets its
    (problem, tile, split) from the XCD table, nothing from the per-problem remap).  Every gradient and every bias gradient of the launch
    against fp32 matmul of the same bf16 operands -- a tile or a split that the table skipped or handed out twice shows as a wrong
    block of some output."""
    M, D, Hd = B * NS_, 192, 768
    g = torch.Generator(device=DEV).manual_seed(50 + blocks)
    probs, refs = [], []
    for b in range(blocks):
        for n1, n2 in ((D, Hd), (Hd, D), (D, D), (3 * D, D)):
            a = (torch.randn(M, n1, device=DEV, generator=g) * 0.1).to(BF16)
            bb = (torch.randn(M, n2, device=DEV, generator=g) * 0.1).to(BF16)
            out = torch.zeros(n1, n2, device=DEV)
            cs = torch.zeros(n1, device=DEV)
            probs.append(dict(a=a, b=bb, out=out, colsum=cs))
    ops.gemm_tn_group(probs)
    torch.cuda.synchronize()
    for i, q in enumerate(probs):
        ref = q["a"].float().t() @ q["b"].float()
        err = ((q["out"] - ref).norm() / ref.norm()).item()
        assert err < 2e-3, (i, err)
        # no tile missing or doubled: the worst 128 x 192 block of the output is as good as the whole
        blk = (q["out"] - ref).abs().amax().item() / ref.abs().amax().item()
        assert blk < 1e-2, (i, blk)
        rs = q["a"].float().sum(0)
        assert ((q["colsum"] - rs).norm() / rs.norm()).item() < 2e-3, i


@pytest.mark.parametrize("epi", ["bias", "gelu"])
def test_wide_gemm_ragged_rows(ops, epi):
    """The persistent 256 x 256 kernel on an M that is odd and not a multiple of 256 (edge tiles: clamped A rows, masked stores, the
    even/odd row-pair exchange of the full-line store with its odd partner missing)."""
    M, N, K = 30001, 2304, 768
    a = rnd(M, K, seed=21).to(BF16)
    w = rnd(N, K, scale=0.05, seed=22).to(BF16)
    bias = rnd(N, seed=23)
    canary = torch.full((M + 8, N), 7.0, device=DEV, dtype=BF16)
    out = canary[:M]
    ops.gemm_nt(a, w, out=out, bias=bias, gelu=(epi == "gelu"))
    assert torch.all(canary[M:] == 7.0), "rows past M were written"
    for lo, hi in ((0, 257), (15000, 15300), (M - 300, M)):
        ref = a[lo:hi].float() @ w.float().t() + bias
        if epi == "gelu":
            ref = torch.nn.functional.gelu(ref)
        assert rel(out[lo:hi], ref) < 1e-2, (epi, lo)


def test_wide_gemm_column_groups(ops):
    """W larger than an XCD's L2 keeps beside the A panels (N K 2 B > 3 MiB, an even number of column tiles): the persistent kernel
    splits the column tiles over two XCD groups and the row panels over four bands.  66 panels on 4 bands = uneven bands, M not a
    multiple of 256, every row checked through the row / column checksums of the whole output, slices through the product itself."""
    M, N, K = 16801, 4096, 768
    a = rnd(M, K, seed=31).to(BF16)
    w = rnd(N, K, scale=0.05, seed=32).to(BF16)
    bias = rnd(N, seed=33)
    canary = torch.full((M + 8, N), 7.0, device=DEV, dtype=BF16)
    out = canary[:M]
    ops.gemm_nt(a, w, out=out, bias=bias)
    assert torch.all(canary[M:] == 7.0), "rows past M were written"
    for lo, hi in ((0, 300), (4100, 4400), (8300, 8500), (12500, 12900), (M - 300, M)):      # every row band, both column halves
        ref = a[lo:hi].float() @ w.float().t() + bias
        assert rel(out[lo:hi], ref) < 1e-2, lo
    # a checksum of checksums over ALL tiles: sum_n out[m, n] = a[m] . sum_n w[n] + sum(bias)
    rows = out.float().sum(1)
    ref_rows = a.float() @ w.float().sum(0) + bias.sum()
    assert rel(rows, ref_rows) < 2e-2
    cols = out.float().sum(0)
    ref_cols = a.float().sum(0) @ w.float().t() + M * bias
    assert rel(cols, ref_cols) < 2e-2


@pytest.mark.parametrize("M,N,K", [(B * NT, 768, 3072), (B * NT, 768, 768), (1000, 256, 128)])
def test_layernorm_fold_producer(ops, M, N, K):
    """The f32-residual GEMM that also leaves bf16(x) and the rows' sum x, sum x^2 for the LayerNorm folded into the next GEMM
    (DkdGemm.xb / rowstats): the wide kernel's whole rounds + the 128-row kernel's tail (fc2's split), the 128-row kernel alone (proj),
    a small problem.  A row's N columns come from several tiles (and two launches): the sums are accumulated atomically."""
    a = rnd(M, K, seed=41).to(BF16)
    w = rnd(N, K, scale=0.05, seed=42).to(BF16)
    bias = rnd(N, seed=43)
    x0 = rnd(M, N, seed=44, scale=3.0) + 0.5
    x = x0.clone()
    xb = torch.empty(M, N, device=DEV, dtype=BF16)
    stats = torch.zeros(M, 2, device=DEV)
    tap = torch.empty(M, N, device=DEV, dtype=BF16)
    ops.gemm_nt(a, w, out=x, bias=bias, resid=x, tap=tap, xb=xb, rowstats=stats)
    torch.cuda.synchronize()
    for lo in (0, M // 2 - 100, M - 300):
        hi = min(M, lo + 300)
        ref = x0[lo:hi] + a[lo:hi].float() @ w.float().t() + bias
        assert rel(x[lo:hi], ref) < 1e-4, lo
    assert torch.equal(xb, x.to(BF16)), "xb must be the bf16 rounding of the x this launch wrote"
    assert rel(stats[:, 0], x.sum(1)) < 1e-5 and rel(stats[:, 1], (x * x).sum(1)) < 1e-5
    assert rel(tap[:300], a[:300].float() @ w.float().t() + bias) < 1e-2


@pytest.mark.parametrize("gelu", [False, True])
def test_layernorm_fold_consumer(ops, gelu):
    """The Linear behind a folded LayerNorm on the wide kernel: C = rstd (xb W'^T - mean c) + b' from the producer's row sums equals
    Linear(LayerNorm(x)) computed in fp32 -- to the same tolerance as the unfolded bf16 path (LN output rounded to bf16, then GEMM)."""
    M, N, K = 16384 + 37, 4096, 768
    x = rnd(M, K, seed=51, scale=2.0) + 0.3
    gamma, beta = 1.0 + 0.2 * rnd(K, seed=52), 0.1 * rnd(K, seed=53)
    w = rnd(N, K, scale=0.04, seed=54)
    bias = rnd(N, seed=55)
    ref = torch.nn.functional.layer_norm(x, (K,), gamma, beta, 1e-6) @ w.t() + bias
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    wf = (w * gamma[None, :]).to(BF16).contiguous()
    stats = torch.stack([x.sum(1), (x * x).sum(1)], 1).contiguous()
    got = ops.gemm_nt(x.to(BF16), wf, bias=(w @ beta + bias).contiguous(), gelu=gelu, ln_stats=stats, ln_c=wf.float().sum(1).contiguous())
    y, _, _ = ops.layernorm_fwd(x, gamma, beta)
    unfolded = ops.gemm_nt(y, w.to(BF16), bias=bias, gelu=gelu)
    torch.cuda.synchronize()
    e_fold, e_std = rel(got, ref), rel(unfolded, ref)
    print(f"folded vs fp32 {e_fold:.3e}; unfolded bf16 path vs fp32 {e_std:.3e}")
    assert e_fold < 1e-2 and e_fold < 2.0 * e_std + 1e-3
    # shapes the fold does not serve are refused, not silently computed without it
    with pytest.raises(RuntimeError, match="LayerNorm fold"):
        ops.gemm_nt(x[:512].to(BF16), wf, bias=bias, ln_stats=stats[:512].contiguous(), ln_c=bias)


def rel2(got, ref):
    return ((got.float() - ref.float()).norm() / ref.float().norm().clamp_min(1e-20)).item()


@pytest.mark.parametrize("offset", [0.0, 3.0, 10.0, 50.0])
def test_layernorm_fold_rows_with_a_common_offset(ops, offset):
    """what the fold costs on rows a pretrained teacher produces and a random-init one does not -- a COMMON offset
    (|row mean| = ``offset`` x the row's spread) plus a few massive channels (100 x).  The fold rounds x to bf16 before it is centred,
    so an element's rounding error is 2^-9 |x_i| instead of 2^-9 |x_i - mu|: relative to the unfolded bf16 path the error grows like
    sqrt(1 + offset^2); massive channels alone cost nothing (they are rounded relative to themselves either way).  Measured here
    (relative L2 against fp32 LayerNorm -> Linear) and bounded by that model; the product keeps the fold only while the rows'
    |mu| / sigma <= 3 (vit._ln_fold_check), i.e. within ~3.2 x the unfolded path's rounding noise."""
    M, N, K = 16384 + 37, 4096, 768
    x = rnd(M, K, seed=81)
    x[:, [7, 300, 611]] *= 100.0                                  # massive channels
    sigma = x.std(1, keepdim=True)
    x = x + offset * sigma * (1.0 + 0.1 * rnd(M, 1, seed=82))     # the common offset, in units of each row's own spread
    gamma, beta = 1.0 + 0.2 * rnd(K, seed=83), 0.1 * rnd(K, seed=84)
    w = rnd(N, K, scale=0.04, seed=85)
    bias = rnd(N, seed=86)
    ref = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-6).float() @ w.t() + bias
    wf = (w * gamma[None, :]).to(BF16).contiguous()
    stats = torch.stack([x.sum(1), (x * x).sum(1)], 1).contiguous()
    got = ops.gemm_nt(x.to(BF16), wf, bias=(w @ beta + bias).contiguous(), ln_stats=stats, ln_c=wf.float().sum(1).contiguous())
    y, _, _ = ops.layernorm_fwd(x, gamma, beta)
    unfolded = ops.gemm_nt(y, w.to(BF16), bias=bias)
    torch.cuda.synchronize()
    e_fold, e_std = rel2(got, ref), rel2(unfolded, ref)
    mu = x.mean(1)
    ratio = (mu.abs() / x.var(1, unbiased=False).sqrt()).max().item()
    model = math.sqrt(1.0 + ratio * ratio)
    print(f"offset {offset}: worst |mu|/sigma {ratio:.2f}; folded {e_fold:.3e}, unfolded {e_std:.3e}, ratio {e_fold / e_std:.2f} (model {model:.2f})")
    assert e_std < 5e-3
    assert e_fold < 1.5 * model * e_std + 5e-4, (e_fold, e_std, model)
    if offset <= 3.0:
        assert e_fold < 2e-2                                      # inside the guard: the tolerance the tap comparisons use
    # the guard itself, on these rows' statistics: keeps the fold up to its bound, drops it (and asks for a redo) beyond
    from deltakd_amd import vit
    guard = {"calls": 0, "pending": None, "off": False, "worst": 0.0}
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        redo = vit._ln_fold_check(guard, stats[None], K)
    assert abs(guard["worst"] - ratio) < 2e-2 * max(ratio, 1.0)
    assert redo == (ratio > vit.LN_FOLD_MAX_OFFSET) and guard["off"] == redo and bool(caught) == redo


def test_layernorm_fold_guard_on_a_teacher_with_offset_rows(monkeypatch):
    """The guard end to end: a deit_base teacher whose residual stream carries a common offset (position embedding + 6: rows with
    |mu| / sigma far above 3 from the first block on) must notice on its FIRST folded call, redo that call with the separate
    LayerNorm kernels and stay there -- bit-identical to a DKD_NO_LN_FOLD=1 run; the same weights without the offset keep the fold."""
    from deltakd_amd import vit
    torch.manual_seed(4)
    t = vit.create_model("deit_base_distilled_patch16_224", num_classes=1000).to(DEV).eval()
    for p in t.parameters():
        p.requires_grad = False
    x = rnd(B, 3, 224, 224, seed=91)
    with torch.no_grad():
        t.forward_with_taps(x, (0, 11))
        g0 = vit._rt(t)["ln_fold_guard"]
        assert not g0["off"] and g0["calls"] == 1 and g0["worst"] < 1.0, g0      # random init: near-zero row means
        t.pos_embed.add_(6.0)
        vit._rt(t).pop("ln_fold_guard")
        with pytest.warns(RuntimeWarning, match="LayerNorm fold switched off"):
            z_g, taps_g = t.forward_with_taps(x, (0, 11))
        assert vit._rt(t)["ln_fold_guard"]["off"]
        z_g2, taps_g2 = t.forward_with_taps(x, (0, 11))
        monkeypatch.setenv("DKD_NO_LN_FOLD", "1")
        z_u, taps_u = t.forward_with_taps(x, (0, 11))
    torch.cuda.synchronize()
    for a, b in ((z_g, z_u), (z_g2, z_u), (taps_g[0], taps_u[0]), (taps_g[11], taps_u[11]), (taps_g2[11], taps_u[11])):
        assert torch.equal(a, b)


def test_teacher_forward_with_folded_layernorms(models, monkeypatch):
    """The whole deit_base_distilled teacher at the training batch with its LayerNorms folded into the GEMMs (24 of 25 LayerNorm launches
    gone) against the same forward with the separate LayerNorm kernels (DKD_NO_LN_FOLD=1): logits and the taps of blocks 0, 1, 11.  Both
    are bf16-operand pipelines that round at different places, each ~0.7 % from fp32 in relative L2 (tools_dev/ln_fold_probe.py): 2e-2 of the largest element between them;
    the per-sample independence test above compares the folded batch-256 path with the unfolded small-batch path as well."""
    from deltakd_amd import vit
    t = models
    assert vit.ln_fold_supported(B * NT, 768, 3072) and not vit.ln_fold_supported(24 * NT, 768, 3072)
    x = rnd(B, 3, 224, 224, seed=61)
    with torch.no_grad():
        z_f, taps_f = t.forward_with_taps(x, (0, 1, 11))
        monkeypatch.setenv("DKD_NO_LN_FOLD", "1")
        z_u, taps_u = t.forward_with_taps(x, (0, 1, 11))
    torch.cuda.synchronize()
    assert rel(z_f, z_u) < 2e-2, rel(z_f, z_u)
    for i in (0, 1, 11):
        assert rel(taps_f[i], taps_u[i]) < 2e-2, (i, rel(taps_f[i], taps_u[i]))
    assert not torch.equal(taps_f[11], taps_u[11]), "the two runs were meant to take different kernels"


def test_vit_large_teacher_with_folded_layernorms(monkeypatch):
    """The same on BASELINE config 5's teacher (vit_large_patch16_224: D = 1024, hidden 4096, depth 24, ONE prefix token) at 128 images:
    fc2 (N = 1024, K = 4096) takes the whole-rounds + tail split with both kernels emitting, proj the 128-row kernel, qkv / fc1 the
    folded wide kernel; taps of blocks 0, 1, 2 (what wasskd reads) and the logits against the unfolded path."""
    from deltakd_amd import vit
    Bv, Nv = 128, 197
    assert vit.ln_fold_supported(Bv * Nv, 1024, 4096)
    torch.manual_seed(9)
    t = vit.create_model("vit_large_patch16_224", num_classes=1000).to(DEV).eval()
    x = rnd(Bv, 3, 224, 224, seed=71)
    with torch.no_grad():
        z_f, taps_f = t.forward_with_taps(x, (0, 1, 2))
        monkeypatch.setenv("DKD_NO_LN_FOLD", "1")
        z_u, taps_u = t.forward_with_taps(x, (0, 1, 2))
    torch.cuda.synchronize()
    assert rel(z_f, z_u) < 2e-2, rel(z_f, z_u)
    for i in (0, 1, 2):
        assert rel(taps_f[i], taps_u[i]) < 2e-2, (i, rel(taps_f[i], taps_u[i]))
    assert not torch.equal(z_f, z_u)


REAL = {  # kind -> (student, teacher, batch): BASELINE.json configs 2, 4, 3 and 5
    "soft": ("deit_tiny_distilled_patch16_224", "deit_small_distilled_patch16_224", 4),
    "lrkd": ("deit_tiny_patch16_224", "deit_base_distilled_patch16_224", 4),
    "mgd": ("deit_tiny_patch16_224", "deit_base_distilled_patch16_224", 4),
    "wasskd": ("deit_small_patch16_224", "vit_large_patch16_224", 2),
    "hard": ("deit_tiny_distilled_patch16_224", "deit_small_distilled_patch16_224", 4),
    "diffkd": ("deit_tiny_patch16_224", "deit_base_distilled_patch16_224", 4),
}


@pytest.mark.parametrize("kind", ["soft", "lrkd", "mgd", "wasskd", "hard", "diffkd"])
def test_real_architecture_step_matches_oracle(kind):
    """The BASELINE architectures themselves (224 x 224, 1000 classes) at a small batch: loss, both of its addends and a spread of
    gradients against the CPU oracle (fp32 torch restatement of the reference path), drop_path 0.  The golden fixtures use toy
    widths; this pins the 192 / 384 / 768 / 1024-wide kernel paths:
      soft    config 2  tiny-distilled <- small-distilled
      lrkd    config 4  tiny <- base-distilled (the oracle's SVD targets are shared: their column signs are arbitrary)
      mgd     config 3  tiny <- base-distilled: align 192 -> 768, mask token, the two 768-channel 3x3 generation convs as GEMMs with
                        K = 6912, masked MSE; mgd_alpha raised to 2.0 so that the term is O(1) of the loss (7e-5 in the script)
      wasskd  config 5  small <- ViT-L/16: D = 1024, depth 24, 16 heads and a teacher with ONE prefix token (the reference hard-codes
                        [:, 2:] and cannot run this pair: model/loss.py:190-193; the num_prefix_tokens superset of DESIGN.md section 1)
      hard    model/loss.py:66-67 (the reference's own exp/hard-deit-tiny.sh pair): cross-entropy of the distillation head against the
                        teacher's argmax, alpha 0.5
      diffkd  model/loss.py:105-155 + model/models.py:103-127 at the real width: tiny <- base-distilled, Dt = 768 denoiser
                        ([784, 768] x [768, 1536] x [1536, 768]), align 3 x Linear(192 -> 768); diffusion steps t injected with a
                        t = 0 sample in the batch (w_t = 1e8: the matching half is then ~1e6 x the noise-prediction half), Gaussian
                        noise and Dropout(0.1) keep masks injected; alpha 0.5 and distill_scale so that the term matters next to
                        the base loss
    Gradients are bounded per tensor relative to their own norm (6e-2), never by a global slack."""
    from oracle import loss_ref, vit_ref
    from deltakd_amd import vit
    from deltakd_amd.losses import DistillationLoss, call_base_loss
    from deltakd_amd.models import attach_aux, forward_with_features
    s_name, t_name, Bs = REAL[kind]
    args = loss_ref.default_args(distillation_type=kind, dataset="imagenet-1k", lrkd_rank=64, alpha=0.5 if kind in ("hard", "diffkd") else 0.1,
                                 tau=3.0, smoothing=0.1, mgd_alpha=2.0, mgd_mask_ratio=0.5, wasskd_type="l1")
    torch.manual_seed(3)
    o_t = vit_ref.create_model_ref(t_name, 1000, 0.0).eval()
    o_s = vit_ref.create_model_ref(s_name, 1000, 0.0).train()
    loss_ref.attach_aux_ref(o_s, o_t, kind, 64)
    with torch.no_grad():                                  # randomly initialised fc2 outputs are tiny: give the taps some scale
        for net in (o_s, o_t):
            for blk in net.blocks:
                blk.mlp.fc2.weight.mul_(8.0)
        if kind == "mgd":
            o_s.mask_token.normal_(0, 0.02)
    for p in o_t.parameters():
        p.requires_grad = False
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(Bs, 3, 224, 224, generator=g), torch.randint(0, 1000, (Bs,), generator=g)
    noise = torch.rand(Bs, 196, generator=g)
    ocrit = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), o_t, kind, args.alpha, args.tau)
    draws = {}
    if kind in ("soft", "hard"):
        oloss = ocrit(x, o_s(x), o_s, None, y, args, {})
    else:
        out, feats = loss_ref.forward_with_features_ref(o_s, x)
        if kind == "lrkd":
            with torch.no_grad():
                _, tf = loss_ref.forward_with_features_ref(o_t, x)
            draws = {"lrkd_targets": [loss_ref.lrkd_targets_ref(tf[i][:, 2:], 64) for i in (0, 1, 11)]}
        elif kind == "mgd":
            draws = {"noise": noise}
        elif kind == "diffkd":
            draws = {"t": torch.tensor([0, 3, 5, 7][:Bs]), "noise": [torch.randn(Bs, 196, 768, generator=g) for _ in range(3)],
                     "drop": [(torch.rand(Bs, 196, 768, generator=g) >= 0.1).float() for _ in range(3)]}
        oloss = ocrit(x, out, o_s, feats, y, args, draws)
    oloss.backward()
    with torch.no_grad():                                  # the base addend on its own (same logits)
        o_logits = o_s(x)
        o_base = loss_ref.call_base_loss_ref(args)(o_logits[0] if isinstance(o_logits, tuple) else o_logits, y).item()
    w_b = (1.0 - args.alpha) if kind in ("soft", "lrkd", "hard", "diffkd") else 1.0
    o_dist = oloss.item() - w_b * o_base

    t = vit.create_model(t_name, num_classes=1000, drop_path_rate=0.0)
    s = vit.create_model(s_name, num_classes=1000, drop_path_rate=0.0)
    attach_aux(s, t, kind, args)
    if kind in ("soft", "hard"):
        s.set_distilled_training(True)
    t.load_state_dict(o_t.state_dict())
    s.load_state_dict(o_s.state_dict())
    t.to(DEV).eval()
    s.to(DEV).train()
    for p in t.parameters():
        p.requires_grad = False
    assert t.num_prefix_tokens == (1 if kind == "wasskd" else 2)
    crit = DistillationLoss(call_base_loss(args), t, kind, args.alpha, args.tau)
    if kind == "lrkd":
        crit.injected["lrkd_targets"] = [tg.to(DEV) for tg in draws["lrkd_targets"]]     # share the oracle's SVD column signs
    if kind == "mgd":
        crit.injected["noise"] = noise.to(DEV)
    if kind == "diffkd":
        crit.injected["t"] = draws["t"].to(DEV)
        crit.injected["noise"] = [n.to(DEV) for n in draws["noise"]]
        crit.injected["drop"] = [d.to(DEV) for d in draws["drop"]]
    if kind in ("soft", "hard"):
        hloss = crit(x.to(DEV), s(x.to(DEV)), s, None, y.to(DEV), args)
    else:
        hout, hfeats = forward_with_features(s, x.to(DEV))
        hloss = crit(x.to(DEV), hout, s, hfeats, y.to(DEV), args)
    hloss.backward()
    assert abs(hloss.item() - oloss.item()) <= 1e-2 * abs(oloss.item()), (hloss.item(), oloss.item())
    assert abs(float(crit.last_base_loss) - w_b * o_base) <= 1e-2 * abs(w_b * o_base)
    if abs(o_dist) > 1e-3 * abs(oloss.item()):             # (soft: 1e-4 of the loss -- fp32 cancellation in total - base on the oracle side)
        assert abs(float(crit.last_distill_loss) - o_dist) <= 1.5e-2 * abs(o_dist), (float(crit.last_distill_loss), o_dist)
    if kind in ("mgd", "wasskd", "hard", "diffkd"):
        assert o_dist > 0.05 * oloss.item(), "the distillation term was meant to matter in this test"
    ref = dict(o_s.named_parameters())
    checked, bad = 0, []
    for n, p in s.named_parameters():
        if p.grad is None or ref[n].grad is None or ref[n].grad.abs().max() == 0 or n.endswith("attn.qkv.bias"):
            continue
        if any(k in n for k in ("blocks.0.", "blocks.1.", "blocks.2.", "blocks.5.", "blocks.11.", "patch_embed", "head", "align", "pos_embed",
                                "cls_token", "generation", "mask_token", "denoise_fn")):
            gr, go = p.grad.detach().cpu(), ref[n].grad
            err = ((gr - go).norm() / go.norm().clamp_min(1e-30)).item()
            if err >= 6e-2:
                bad.append((n, err, go.norm().item()))
            checked += 1
    assert not bad, bad[:8]
    assert checked >= 30


def test_generation_conv_as_gemm_full_size(ops):
    """MGD's Conv3x3(768, 768) at the headline shape (B = 256: M = 50 176 rows, K = 9 * 768 = 6 912) as implicit GEMMs (no im2col
    matrix): forward, input gradient and weight gradient, by slices against torch's fp32 conv2d on whole images (first, middle, last)."""
    Bc, hw, Cc = 256, 14, 768
    xg = rnd(Bc * hw * hw, Cc, seed=31).to(BF16)
    w = rnd(Cc, Cc, 3, 3, scale=0.02, seed=32)
    bias = rnd(Cc, seed=33)
    wf = w.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc).contiguous().to(BF16)            # [out, (ky, kx, cin)]
    wd = w.flip(2, 3).permute(1, 2, 3, 0).reshape(Cc, 9 * Cc).contiguous().to(BF16)  # [cin, (2-ky, 2-kx, out)]
    y = ops.gemm_nt(xg, wf, bias=bias, relu=True, conv_hw=hw)
    imgs = (0, 131, Bc - 1)
    P = hw * hw
    wq = w.to(BF16).float()

    def conv_ref(b):
        xi = xg[b * P:(b + 1) * P].float().t().reshape(1, Cc, hw, hw)
        return torch.nn.functional.conv2d(xi, wq, bias, padding=1).reshape(Cc, P).t()
    for b in imgs:
        assert rel(y[b * P:(b + 1) * P], torch.relu(conv_ref(b))) < 1e-2, ("fwd", b)
    dy = rnd(Bc * P, Cc, seed=34).to(BF16)
    dx = ops.gemm_nt(dy, wd, conv_hw=hw)
    for b in imgs:
        dyi = dy[b * P:(b + 1) * P].float().t().reshape(1, Cc, hw, hw)
        ref = torch.nn.functional.conv_transpose2d(dyi, wq, padding=1).reshape(Cc, P).t()
        assert rel(dx[b * P:(b + 1) * P], ref) < 1.5e-2, ("dgrad", b)
    # weight gradient over all 50 176 rows against torch on 32-image slabs accumulated in fp64
    dwp = torch.zeros(Cc, 9 * Cc, device=DEV)
    db = torch.zeros(Cc, device=DEV)
    ops.conv3x3_wgrad(dy, xg, dwp, db, Bc, hw)
    ref_dw = torch.zeros(Cc, Cc, 3, 3, device=DEV, dtype=torch.float64)
    for b0 in range(0, Bc, 32):
        xi = xg[b0 * P:(b0 + 32) * P].float().reshape(32, P, Cc).transpose(1, 2).reshape(32, Cc, hw, hw)
        dyi = dy[b0 * P:(b0 + 32) * P].float().reshape(32, P, Cc).transpose(1, 2).reshape(32, Cc, hw, hw)
        ref_dw += torch.nn.grad.conv2d_weight(xi, w.shape, dyi, padding=1).double()
    assert rel(dwp.view(Cc, 3, 3, Cc).permute(0, 3, 1, 2), ref_dw.float()) < 1e-3
    assert rel(db, dy.float().sum(0)) < 1e-4
