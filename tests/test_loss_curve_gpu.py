"""Loss-curve equivalence over many optimizer steps (BASELINE.json north_star: "loss-curve equivalent to reference").

The reference's training loop is /root/reference/tools/train.py:318-334 -> tools/engine.py:8-76: per epoch ``train_one_epoch`` then
``lr_scheduler.step(epoch)``.  Here the PRODUCT's loop (deltakd_amd.engine.train_one_epoch on the HIP models, fused losses, FusedAdamW,
teacher lookahead on a side stream, device-side mixup) and the ORACLE's loop (oracle/engine_ref.py on the fp32 torch restatement with
torch.optim.AdamW over timm's parameter groups) each run 4 epochs x 30 batches = 120 optimizer steps FREE-RUNNING from the same weights,
with the same data order, the same numpy mixup / cutmix draws, the same DropPath keep masks and masking noise per step and the same
cosine schedule (1 warm-up epoch) -- nothing is re-synchronised on the way, so bf16 rounding noise is free to accumulate through
AdamW for 120 steps.  What must agree is what a user compares between two training runs: the loss curve and the model at the end.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOY = dict(img_size=32, patch_size=8, mlp_ratio=2.0)


class Recorder:
    def __init__(self, crit):
        self.crit, self.losses = crit, []
        if hasattr(crit, "prefetch"):
            self.prefetch = crit.prefetch

    def __call__(self, *a):
        loss = self.crit(*a)
        self.losses.append(loss.detach())
        return loss


CASES = [  # kind, student width (0 = the real architectures), epochs, batches per epoch, window
    pytest.param("mgd", 64, 3, 20, 10, id="mgd-64-60steps"),
    pytest.param("lrkd", 64, 4, 30, 10, id="lrkd-64-120steps"),      # (back at 120 steps: the headline branch -- ADVICE round 4)
    pytest.param("mgd", 192, 4, 30, 10, id="mgd-192-120steps"),
    # round 5: the headline branch FREE-RUNNING -- the product computes its own LRKD targets (LowRankTargets defaults: dkd_lowrank_chain on a
    # 256-wide teacher; nothing injected), the oracle uses torch.linalg.svd of its own teacher's taps with each column's sign set to the
    # product's (LAPACK's sign is arbitrary: SURVEY section 0 item 9)
    pytest.param("lrkd-free", 64, 3, 20, 10, id="lrkd-free-running-chain-60steps"),
    # BASELINE config 2 at its real width: deit_tiny_distilled <- deit_small_distilled, soft, 224 x 224, 1000 classes, through the fused
    # D = 192 kernels; 3 epochs x 10 batches of 8 = 30 free-running steps (tools/train.py:318-334).  `-m "gpu and not real_curve"` skips it.
    pytest.param("soft", 0, 3, 10, 10, id="soft-real-30steps", marks=pytest.mark.real_curve),
]


@pytest.mark.parametrize("kind,width,epochs,n_batches,win", CASES)
def test_loss_curve_tracks_the_oracle_loop(kind, width, epochs, n_batches, win):
    """(60 steps for the 64-wide mgd pair, 120 for the 64-wide lrkd pair -- the headline branch -- and for the 192-wide student that takes
    the fused kernels, 30 for the real architectures.)  Asserted (the measured values are in the assertion messages / printed):
      * the loss averaged over windows of 10 steps, at all 12 windows: product within 0.5 % of the oracle's (measured: 0.05 %);
      * the curves are curves: the oracle's last window is well below its first (the toy problem is learning, so an all-constant
        loss could not pass), and the product's per-epoch ``train_loss`` (what train_one_epoch returns) follows the oracle's to 0.5 %;
      * the learning rate the loop reports per epoch is the schedule's, identical on both sides;
      * the trained models agree: eval-mode logits on 64 held-out images, relative L2 <= 0.05 (measured 0.008), and >= 95 % of the top-1
        decisions (measured: all).
    lrkd: the exact-SVD targets of every step are computed once from the (frozen) oracle teacher on the mixed batches and replayed to
    both loops, as in tests/test_engine_gpu.py (the SVD's column signs are arbitrary).
    lrkd-free (round 5): NOTHING is injected into the product -- its criterion computes the LRKD targets itself with the defaults of the
    timed path (``LowRankTargets.EXACT`` through dkd_lowrank_chain; the teacher is 256 wide so that the chain runs, not the small-matrix
    Jacobi path) from its own bf16 teacher taps, warm-started from batch to batch.  The product loop runs first; the oracle then trains on
    torch.linalg.svd of ITS teacher's fp32 taps with every column's sign set to the product's.  The LRKD term is half of this loss (1.1 of
    2.26), so a target chain that drifted would show in the curve: measured 1e-4 at all windows.  (Individual columns inside clusters of
    nearly equal singular values differ between the two sides by up to 0.16 in relative norm -- bf16 taps decide LAPACK's arbitrary
    choice there differently -- without moving the loss: printed, not asserted.)
    width 192 (3 heads): the student takes the fused kernels of the headline path -- dkd_attn192_fwd, dkd_mlp192_fwd / _bwd, weight gradients
    and LayerNorm reductions deferred six blocks at a time -- for all 120 free-running steps (at a third of the learning rate: at 1e-3
    this 12-block, 192-wide model's loss on 240 images starts to oscillate after ~80 steps, in the oracle too, and two runs that agreed to
    1e-4 until then separate by 0.6 % over the next 40 -- sensitivity of the trajectory, not of the arithmetic)."""
    from oracle import engine_ref, loss_ref, vit_ref
    from deltakd_amd import vit
    from deltakd_amd.engine import train_one_epoch
    from deltakd_amd.losses import DistillationLoss, call_base_loss
    from deltakd_amd.models import attach_aux
    from deltakd_amd.optim import CosineLRScheduler, create_optimizer, param_groups_weight_decay
    from deltakd_amd.shims import Mixup, NativeScaler
    free = kind == "lrkd-free"
    if free:
        kind = "lrkd"
    t_width = 256 if free else 128
    torch.manual_seed(21)
    real = width == 0
    C, B, depth, size = (1000, 8, 12, 224) if real else (10, 8, 12, 32)
    args = loss_ref.default_args(distillation_type=kind, dataset="imagenet-1k" if real else "cifar-10", mgd_alpha=2.0, mgd_mask_ratio=0.5,
                                 alpha=0.5, tau=3.0, lrkd_rank=16, opt="adamw", lr=1e-3 if width == 64 else 3e-4, weight_decay=0.05,
                                 opt_eps=1e-8, opt_betas=None, mixup=0.8, cutmix=1.0, smoothing=0.1, epochs=epochs, print_freq=100000, rank=1)
    if real:
        t_name, s_name = "deit_small_distilled_patch16_224", "deit_tiny_distilled_patch16_224"
        o_t = vit_ref.create_model_ref(t_name, C, 0.0).eval()
        o_s = vit_ref.create_model_ref(s_name, C, 0.1).train()
        t, s = vit.create_model(t_name, num_classes=C, drop_path_rate=0.0), vit.create_model(s_name, num_classes=C, drop_path_rate=0.1)
    else:
        o_t = vit_ref.VisionTransformerRef(t_width, depth, t_width // 64, C, True, 0.0, **TOY).eval()
        heads = width // 64
        o_s = vit_ref.VisionTransformerRef(width, depth, heads, C, False, 0.1, **TOY).train()
        t = vit.VisionTransformer(t_width, depth, t_width // 64, C, True, 0.0, **TOY)
        s = vit.VisionTransformer(width, depth, heads, C, False, 0.1, **TOY)
    loss_ref.attach_aux_ref(o_s, o_t, kind, args.lrkd_rank)
    attach_aux(s, t, kind, args)
    with torch.no_grad():
        for net in (o_s, o_t):
            for blk in net.blocks:
                blk.mlp.fc2.weight.mul_(8.0)
    t.load_state_dict(o_t.state_dict())
    s.load_state_dict(o_s.state_dict())
    for p in list(t.parameters()) + list(o_t.parameters()):
        p.requires_grad = False
    t.to(DEV).eval()
    s.to(DEV).train()

    # a small fixed dataset with structure (class-dependent mean pattern + noise), revisited every epoch
    g = torch.Generator().manual_seed(5)
    n_cls = min(C, 16)                      # (the real case uses 16 of its 1000 classes: a 1000-prototype table would be 600 MB)
    n_held = 16 if real else 64
    proto = torch.randn(n_cls, 3, size, size, generator=g)
    labels = [torch.randint(0, n_cls, (B,), generator=g) for _ in range(n_batches)]
    data = [(0.7 * proto[y] + torch.randn(B, 3, size, size, generator=g), y) for y in labels]
    held_y = torch.randint(0, n_cls, (n_held,), generator=g)
    held_x = 0.7 * proto[held_y] + torch.randn(n_held, 3, size, size, generator=g)
    n_steps = epochs * n_batches
    keeps = [[(torch.rand(B, generator=g) > 0.1).float() for _ in range(2 * depth)] for _ in range(n_steps)]
    noises = [torch.rand(B, (size // (16 if real else 8)) ** 2, generator=g) for _ in range(n_steps)]
    sched_kw = dict(t_initial=epochs, lr_min=1e-5, warmup_t=1, warmup_lr_init=2e-4 if width == 64 else 1e-4)
    o_mix = engine_ref.MixupRef(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)

    draws = [{"noise": n} for n in noises]
    product_targets = []
    worst_target_err = [0.0]

    def run_product():
        opt = create_optimizer(args, s)
        sched = CosineLRScheduler(opt, **sched_kw)
        crit = DistillationLoss(call_base_loss(args), t, kind, args.alpha, args.tau, teacher_stream=torch.cuda.Stream())
        crit.injected["noise"] = iter([n.to(DEV) for n in noises])
        if kind == "lrkd" and not free:
            crit.injected["lrkd_targets"] = iter([[a.to(DEV) for a in d["lrkd_targets"]] for d in draws])
        if free:
            solver = crit.lowrank
            assert solver.mode["name"].startswith("exact"), solver.mode

            class Recording:                 # the criterion's own solver, its outputs copied out per call (calls come in batch order)
                def __call__(self, taps, npre, rank):
                    assert taps[0].shape[-1] == t_width and t_width > 128      # (the chain, not the small-matrix Jacobi path)
                    out = solver(taps, npre, rank)
                    product_targets.append([o.detach().float().cpu() for o in out])
                    return out

                def __getattr__(self, k):
                    return getattr(solver, k)
            crit.lowrank = Recording()
        s.set_droppath_keep(iter(keeps))
        rec = Recorder(crit)
        mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=C)
        np.random.seed(99)
        h_ep = []
        dev_data = [(x.to(DEV), y.to(DEV)) for x, y in data]
        for e in range(epochs):
            st = train_one_epoch(s, t, [(x.clone(), y) for x, y in dev_data], rec, opt, NativeScaler(), None, mix, None, torch.device(DEV), e, args)
            sched.step(e)
            h_ep.append({k: float(v) for k, v in st.items()})
        torch.cuda.synchronize()
        return h_ep, [float(v) for v in rec.losses]

    if free:
        h_epochs, h_curve = run_product()
        assert len(product_targets) == n_steps
    if kind == "lrkd":
        np.random.seed(99)
        with torch.no_grad():
            for e in range(epochs):
                for i, (x, y) in enumerate(data):
                    xm, _ = o_mix(x.clone(), y.clone())
                    _, tf = loss_ref.forward_with_features_ref(o_t, xm)
                    tg = [loss_ref.lrkd_targets_ref(tf[b][:, 2:], args.lrkd_rank) for b in (0, 1, 11)]
                    if free:                 # the exact targets, each column with the sign the product's has; and how close the product's are
                        got = product_targets[e * n_batches + i]
                        for li in range(3):
                            sgn = torch.sign((tg[li] * got[li]).sum(0))
                            sgn[sgn == 0] = 1.0
                            tg[li] = tg[li] * sgn
                            err = ((got[li] - tg[li]).norm() / tg[li].norm()).item()
                            worst_target_err[0] = max(worst_target_err[0], err)
                    draws[e * n_batches + i]["lrkd_targets"] = tg

    # ---- oracle
    o_opt = torch.optim.AdamW(param_groups_weight_decay(o_s, args.weight_decay, o_s.no_weight_decay()), lr=args.lr, weight_decay=0.0)
    o_sched = CosineLRScheduler(o_opt, **sched_kw)
    o_crit = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), o_t, kind, args.alpha, args.tau)
    np.random.seed(99)
    o_curve, o_epochs = [], []
    for e in range(epochs):
        st, per = engine_ref.train_one_epoch_ref(o_s, o_t, [(x.clone(), y.clone()) for x, y in data], o_crit, o_opt, None, o_mix, e, args,
                                                 keep_per_step=keeps[e * n_batches:(e + 1) * n_batches],
                                                 draws_per_step=draws[e * n_batches:(e + 1) * n_batches])
        o_sched.step(e)
        o_curve += [p[0] for p in per]
        o_epochs.append(st)

    # ---- product
    if not free:
        h_epochs, h_curve = run_product()
    else:
        print(f"lrkd free-running: the product's targets vs torch.linalg.svd of the same taps (signs aligned), worst relative error over "
              f"{n_steps} steps x 3 layers: {worst_target_err[0]:.2e}")
    assert len(h_curve) == len(o_curve) == n_steps

    o_w = np.array(o_curve).reshape(-1, win).mean(1)
    h_w = np.array(h_curve).reshape(-1, win).mean(1)
    rel = np.abs(h_w - o_w) / o_w
    msg = (f"window-averaged loss  oracle {np.round(o_w, 4).tolist()}  product {np.round(h_w, 4).tolist()}  rel {np.round(rel, 4).tolist()}; "
           f"per-epoch oracle {[round(e['train_loss'], 4) for e in o_epochs]} product {[round(e['train_loss'], 4) for e in h_epochs]}")
    print(msg)
    assert o_w[-1] < (0.97 if real else 0.9) * o_w[0], "the problem should be learning: " + msg
    assert rel.max() <= 5e-3, msg
    for he, oe in zip(h_epochs, o_epochs):
        assert abs(he["train_loss"] - oe["train_loss"]) <= 5e-3 * abs(oe["train_loss"]), msg
        assert abs(he["train_lr"] - oe["train_lr"]) < 1e-12, (he, oe)
    assert len({round(e["train_lr"], 9) for e in o_epochs}) >= epochs - 1, "the schedule should move the learning rate (epoch 1 repeats the warm-up start: the reference passes `epoch` to step())"

    # ---- the trained models
    s.eval()
    o_s.eval()
    with torch.no_grad():
        z_h = s(held_x.to(DEV)).float().cpu()
        z_o = o_s(held_x)
    err = ((z_h - z_o).norm() / z_o.norm()).item()
    agree = (z_h.argmax(1) == z_o.argmax(1)).float().mean().item()
    acc_o = (z_o.argmax(1) == held_y).float().mean().item()
    print(f"final eval-mode logits: relative L2 {err:.4f}, top-1 agreement {agree:.3f}, oracle held-out accuracy {acc_o:.3f}")
    assert err <= 0.05 and agree >= 0.95, (err, agree, msg)
