"""The step loop itself on the HIP path against the oracle loop ("train_one_epoch -> per-step (loss, acc1, acc5)
and post-step student weights for 2 steps at B=4 with DropPath masks / masking noise / mixup lambdas supplied as fixture inputs"),
``validate``, the reference-compatible free-function surface, and data parallel on the real HIP model with two ranks.

The oracle loop is oracle/engine_ref.py, the de-duplicated restatement of /root/reference/tools/engine.py:8-76 (the reference file
itself cannot be imported); tests/test_host_logic.py shows on CPU that the product's loop and the oracle loop are the same loop when
driven with the same models.  Here the product side runs deltakd_amd's HIP models, fused losses, FusedAdamW and the teacher-lookahead
order; the oracle side runs the fp32 torch restatement with torch.optim.AdamW over timm's parameter groups.
"""
import copy
import json
import os
import re
import socket
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOY = dict(img_size=32, patch_size=8, mlp_ratio=2.0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel_l2(got, ref):
    return (got.double() - ref.double()).norm().item() / (ref.double().norm().item() + 1e-30)


class Recorder:
    """criterion wrapper that keeps the per-step losses (train_one_epoch only returns epoch averages)."""

    def __init__(self, crit):
        self.crit, self.losses = crit, []
        if hasattr(crit, "prefetch"):
            self.prefetch = crit.prefetch

    def __call__(self, *a):
        loss = self.crit(*a)
        self.losses.append(loss.detach())
        return loss


def _pair(kind, real):
    """-> (oracle teacher, oracle student, hip teacher, hip student, args, image size, classes), same weights on both sides."""
    from oracle import loss_ref, vit_ref
    from deltakd_amd import vit
    from deltakd_amd.models import attach_aux
    torch.manual_seed(31)
    if real:
        t_name = "deit_base_distilled_patch16_224" if kind in ("mgd", "lrkd") else "deit_small_distilled_patch16_224"
        s_name = "deit_tiny_patch16_224" if kind in ("mgd", "lrkd") else "deit_tiny_distilled_patch16_224"
        C, size = 1000, 224
        o_t = vit_ref.create_model_ref(t_name, C, 0.0).eval()
        o_s = vit_ref.create_model_ref(s_name, C, 0.1).train()
        t, s = vit.create_model(t_name, num_classes=C, drop_path_rate=0.0), vit.create_model(s_name, num_classes=C, drop_path_rate=0.1)
    else:
        C, size = 10, 32
        o_t = vit_ref.VisionTransformerRef(128, 12, 2, C, True, 0.0, **TOY).eval()
        o_s = vit_ref.VisionTransformerRef(64, 12, 1, C, kind in ("soft", "hard"), 0.1, **TOY).train()
        t = vit.VisionTransformer(128, 12, 2, C, True, 0.0, **TOY)
        s = vit.VisionTransformer(64, 12, 1, C, kind in ("soft", "hard"), 0.1, **TOY)
    args = loss_ref.default_args(distillation_type=kind, dataset="imagenet-1k" if real else "cifar-10", mgd_alpha=2.0, mgd_mask_ratio=0.5,
                                 alpha=0.5, tau=3.0, opt="adamw", lr=2e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=None, mixup=0.8,
                                 cutmix=1.0, smoothing=0.1, epochs=1, print_freq=1000, rank=1)
    loss_ref.attach_aux_ref(o_s, o_t, kind)
    attach_aux(s, t, kind, args)
    with torch.no_grad():                 # trunc_normal(.02) fc2 outputs are tiny: give the taps (and the deep logits) some scale
        for net in (o_s, o_t):
            for blk in net.blocks:
                blk.mlp.fc2.weight.mul_(8.0)
    t.load_state_dict(o_t.state_dict())
    s.load_state_dict(o_s.state_dict())
    for p in list(t.parameters()) + list(o_t.parameters()):
        p.requires_grad = False
    return o_t, o_s, t.to(DEV).eval(), s.to(DEV).train(), args, size, C


@pytest.mark.parametrize("kind,real,lr", [("soft", False, 2e-4), ("mgd", False, 2e-4), ("mgd", False, 1e-3), ("mgd", True, 2e-4),
                                          ("lrkd", False, 1e-3), ("lrkd", True, 2e-4)])
def test_train_one_epoch_matches_the_oracle_loop(kind, real, lr):
    """2 steps, B = 4, mixup/cutmix on (numpy draws replayed from the same seed), DropPath keep masks and masking noise injected
    per step.  Both loops run ``train_one_epoch`` end to end; the oracle's weights are set to the product's after the first optimizer
    step, so that the second step is compared AT THE SAME WEIGHTS and every bound below is the single-step bound of the kernels, not a
    constant tuned to how far two AdamW trajectories drift (round 2 had 8e-2 at lr 2e-4 after 0.0814 was measured at lr 1e-3; the
    cases now run at both learning rates with the same bounds).  Checked:
      * per-step loss (1e-2) and the epoch averages of loss / acc1 / acc5 / lr as train_one_epoch returns them;
      * the gradients of the SECOND step, per tensor relative to the tensor's own norm: 6e-2, the bound every single-step parity test of
        this repo uses (bf16 operands, fp32 accumulation, against fp32);
      * both optimizer steps through their updates.  An early Adam step is  -lr g / (|g| + eps)  per element (bias-corrected m / sqrt(v) at
        step 1), i.e. ~ lr sign(g): elements whose gradient is inside the bf16 noise take a step of random sign on either side, elements
        well above it must agree.  Noise model: the per-tensor bound says the elementwise error has rms <= 6e-2 rms(g); an element with
        |g_oracle| >= 0.5 rms(g) is >= 8 sigma from a sign flip, so on those elements >= 99.5 % of the first-step updates must have the
        oracle's sign (0.5 % slack for heavy-tailed errors), and -- the bound that replaces round 2's cosine >= 0.9 -- the update
        restricted to them must have cosine >= 0.98 with the oracle's.
    lrkd (the headline branch, model/loss.py:80-103 via tools/engine.py:47-48; real = DeiT-tiny <- DeiT-base-distilled): the reference
    takes an exact SVD of each step's teacher taps (model/loss.py:318-326); the oracle's per-step targets are computed first (the
    teacher is frozen and sees the same mixed batches on both sides) and handed to BOTH loops -- to the product through
    ``DistillationLoss.injected`` as an iterator, one list of three targets per step -- because the SVD's column signs are arbitrary and
    the loss is not invariant to them.  How well the product's own tracker approximates these targets is
    bounded separately (tests/test_fullsize_gpu.py::test_lowrank_tracking_at_the_headline_batch)."""
    from oracle import engine_ref, loss_ref
    from deltakd_amd.engine import train_one_epoch
    from deltakd_amd.losses import DistillationLoss, call_base_loss
    from deltakd_amd.optim import create_optimizer, param_groups_weight_decay
    from deltakd_amd.shims import Mixup, NativeScaler
    o_t, o_s, t, s, args, size, C = _pair(kind, real)
    args.lr = lr
    B, steps, depth = 4, 2, 12
    g = torch.Generator().manual_seed(77)
    data = [(torch.randn(B, 3, size, size, generator=g), torch.randint(0, C, (B,), generator=g)) for _ in range(steps)]
    keeps = [[(torch.rand(B, generator=g) > 0.15).float() for _ in range(2 * depth)] for _ in range(steps)]
    P = (size // (16 if real else 8)) ** 2
    noises = [torch.rand(B, P, generator=g) for _ in range(steps)]
    w0 = {n: p.detach().clone() for n, p in o_s.named_parameters()}
    o_mix = lambda: engine_ref.MixupRef(mixup_alpha=0.8, cutmix_alpha=1.0, label_smoothing=0.1, num_classes=C)   # noqa: E731

    svd_targets = None
    if kind == "lrkd":                    # the exact-SVD targets of every step, from the mixed batches the loops will see
        np.random.seed(123)
        mixer, svd_targets = o_mix(), []
        with torch.no_grad():
            for x, y in data:
                xm, _ = mixer(x.clone(), y.clone())
                _, tf = loss_ref.forward_with_features_ref(o_t, xm)
                svd_targets.append([loss_ref.lrkd_targets_ref(tf[b][:, 2:], args.lrkd_rank) for b in (0, 1, 11)])

    # ---- product loop (HIP models, fused losses, FusedAdamW, teacher lookahead on a side stream); weights snapshotted after each step
    opt = create_optimizer(args, s)
    crit = DistillationLoss(call_base_loss(args), t, kind, args.alpha, args.tau, teacher_stream=torch.cuda.Stream())
    crit.injected["noise"] = iter([n.to(DEV) for n in noises])
    if kind == "lrkd":
        crit.injected["lrkd_targets"] = iter([[a.to(DEV) for a in tg] for tg in svd_targets])
    s.set_droppath_keep(iter(keeps))
    rec = Recorder(crit)
    snaps, grads = [], []
    hip_step = opt.step

    def step_and_snapshot(*a, **k):
        torch.cuda.synchronize()
        grads.append({n: p.grad.detach().cpu().clone() for n, p in s.named_parameters() if p.grad is not None})
        out = hip_step(*a, **k)
        torch.cuda.synchronize()
        snaps.append({n: p.detach().cpu().clone() for n, p in s.named_parameters()})
        return out
    opt.step = step_and_snapshot
    np.random.seed(123)
    mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=C)
    stats = train_one_epoch(s, t, [(x.clone().to(DEV), y.clone().to(DEV)) for x, y in data], rec, opt, NativeScaler(), None, mix, None,
                            torch.device(DEV), 0, args)
    torch.cuda.synchronize()
    assert len(snaps) == steps and len(grads) == steps

    # ---- oracle loop (CPU fp32); after ITS first step the weights become the product's
    class SyncedAdamW(torch.optim.AdamW):
        calls = 0

        def step(self, closure=None):
            self.o_grads.append({n: p.grad.detach().clone() for n, p in o_s.named_parameters() if p.grad is not None})
            super().step(closure)
            self.o_after.append({n: p.detach().clone() for n, p in o_s.named_parameters()})
            if self.calls < steps - 1:
                with torch.no_grad():
                    for n, p in o_s.named_parameters():
                        p.copy_(snaps[self.calls][n])
            self.calls += 1

    o_opt = SyncedAdamW(param_groups_weight_decay(o_s, args.weight_decay, o_s.no_weight_decay()), lr=args.lr, weight_decay=0.0)
    o_opt.o_grads, o_opt.o_after = [], []
    o_crit = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), o_t, kind, args.alpha, args.tau)
    np.random.seed(123)
    draws = [{"noise": n} for n in noises]
    if kind == "lrkd":
        for d, tg in zip(draws, svd_targets):
            d["lrkd_targets"] = tg
    o_stats, o_steps = engine_ref.train_one_epoch_ref(o_s, o_t, [(x.clone(), y.clone()) for x, y in data], o_crit, o_opt, None, o_mix(), 0, args,
                                                      keep_per_step=keeps, draws_per_step=draws)

    hip_steps = [float(v) for v in rec.losses]
    assert len(hip_steps) == steps
    for a, (b, _, _) in zip(hip_steps, o_steps):
        assert abs(a - b) <= 1e-2 * abs(b), (hip_steps, o_steps)
    assert abs(float(stats["train_loss"]) - o_stats["train_loss"]) <= 1e-2 * abs(o_stats["train_loss"])
    assert abs(float(stats["train_lr"]) - o_stats["train_lr"]) < 1e-12
    tol_acc = 100.0 / B / steps + 1e-6                      # one near-tie flipping in one step
    assert abs(float(stats["train_acc1"]) - o_stats["train_acc1"]) <= tol_acc, (stats, o_stats)
    assert abs(float(stats["train_acc5"]) - o_stats["train_acc5"]) <= tol_acc, (stats, o_stats)

    bad, checked, worst_g, worst_cos, worst_sign = [], 0, 0.0, 1.0, 1.0
    before = [w0, snaps[0]]
    for step in range(steps):
        for n, go in o_opt.o_grads[step].items():
            if go.norm() == 0 or n.endswith("attn.qkv.bias") or n not in grads[step]:     # key third of qkv.bias: zero in exact arithmetic
                continue
            err = (grads[step][n] - go).norm().item() / go.norm().item()
            worst_g = max(worst_g, err)
            if err > 6e-2:
                bad.append((n, f"step-{step + 1} gradient", err))
            # the update of this step, on the elements whose gradient is well above the noise floor
            live = go.abs() >= 0.5 * go.pow(2).mean().sqrt()
            if step == 0 and live.sum() >= 16:
                dw_h, dw_o = (snaps[step][n] - before[step][n])[live], (o_opt.o_after[step][n] - before[step][n])[live]
                agree = (torch.sign(dw_h) == torch.sign(dw_o)).float().mean().item()
                cos = torch.nn.functional.cosine_similarity(dw_h.flatten().double(), dw_o.flatten().double(), dim=0).item()
                worst_sign, worst_cos = min(worst_sign, agree), min(worst_cos, cos)
                if agree < 0.995:
                    bad.append((n, "first-step update sign agreement", agree))
                if cos < 0.98:
                    bad.append((n, "first-step update direction", cos))
            checked += 1
    assert not bad, (len(bad), bad[:8], worst_g, worst_cos, worst_sign)
    assert checked > 200


def test_validate_matches_the_oracle_loop():
    """tools/engine.py:78-104 on the HIP student (eval mode, distilled student -> averaged heads) vs oracle.engine_ref.validate_ref."""
    from oracle import engine_ref
    from deltakd_amd.engine import validate
    o_t, o_s, t, s, args, size, C = _pair("soft", False)
    g = torch.Generator().manual_seed(3)
    data = [(torch.randn(16, 3, size, size, generator=g), torch.randint(0, C, (16,), generator=g)) for _ in range(3)]
    got = validate(s, [(x.to(DEV), y.to(DEV)) for x, y in data], torch.device(DEV), SimpleNamespace(rank=1))
    want = engine_ref.validate_ref(o_s, data)
    assert set(got) == {"val_loss", "val_acc1", "val_acc5"}
    assert abs(float(got["val_loss"]) - want["val_loss"]) <= 1e-2 * want["val_loss"], (got, want)
    for k in ("val_acc1", "val_acc5"):
        assert abs(float(got[k]) - want[k]) <= 100.0 / 16 / 3 + 1e-6, (k, got, want)     # at most one near-tie
    assert not s.training


def test_free_functions_and_aux_modules_with_the_reference_signatures():
    """The reference's own calling pattern, positional, on the HIP side (model/loss.py:86-103, :314-330; model/models.py:90-94):
        student_features = [student_model.align[i](feat[:, 1:]) ...];  teacher_features = [feat[:, 2:] ...]
        lrkd_loss(teacher_features, student_features, rank, alpha=, beta=, gamma=)
    against the oracle's lrkd term (SVD column signs aligned by handing both sides the oracle's targets is NOT possible through this
    signature, so the comparison is sign-invariant: rank-k reconstruction is compared through the projector).  Also checks
    ``align`` gradients flow, and mgd_loss / vitkd_loss positional calls."""
    from oracle import loss_ref
    from model.loss import lrkd_loss, mgd_loss          # the drop-in import path
    from deltakd_amd.models import forward_with_features
    o_t, o_s, t, s, args, size, C = _pair("lrkd", False)
    # _pair attached mgd-sized aux for kind != lrkd; for lrkd it attaches align = 3 x Linear(64 -> rank)
    rank = s.align[0].out_features
    g = torch.Generator().manual_seed(8)
    x = torch.randn(8, 3, size, size, generator=g)
    s.set_droppath_keep(None)
    s.eval()
    o_s.eval()
    out, feats = forward_with_features(s, x.to(DEV))
    with torch.no_grad():
        _, tfeats = t.forward_with_taps(x.to(DEV))
    sf = [s.align[0](feats[0][:, 1:]), s.align[1](feats[1][:, 1:]), s.align[2](feats[-1][:, 1:])]
    tf = [tfeats[0][:, 2:], tfeats[1][:, 2:], tfeats[11][:, 2:]]
    assert sf[0].shape == (8, 16, rank) and sf[0].dtype == torch.float32
    loss = lrkd_loss(tf, sf, rank, 0.2, 0.3, 0.5)
    loss.backward()
    # oracle: same features in fp32, reference arithmetic
    o_out, o_feats = loss_ref.forward_with_features_ref(o_s, x)
    with torch.no_grad():
        _, o_tf = loss_ref.forward_with_features_ref(o_t, x)
    o_sf = [o_s.align[0](o_feats[0][:, 1:]), o_s.align[1](o_feats[1][:, 1:]), o_s.align[2](o_feats[-1][:, 1:])]
    for a, b in zip(sf, o_sf):
        assert rel_l2(a.detach().cpu(), b.detach()) < 2e-2                     # Linear.forward on the MFMA GEMM
    # the loss with the ORACLE's column signs, through the reference signature with targets injected (keyword extension)
    targets = [loss_ref.lrkd_targets_ref(o_tf[b][:, 2:], rank) for b in (0, 1, 11)]
    for p in s.parameters():
        p.grad = None
    out, feats = forward_with_features(s, x.to(DEV))           # a block's activations are released by its first backward
    sf = [s.align[0](feats[0][:, 1:]), s.align[1](feats[1][:, 1:]), s.align[2](feats[-1][:, 1:])]
    loss2 = lrkd_loss(tf, sf, rank, 0.2, 0.3, 0.5, targets=[a.to(DEV) for a in targets])
    o_loss = sum(w * torch.nn.functional.mse_loss(a, f.reshape(-1, rank)) for w, a, f in zip((0.2, 0.3, 0.5), targets, o_sf))
    assert abs(loss2.item() - o_loss.item()) <= 1e-2 * abs(o_loss.item()), (loss2.item(), o_loss.item())
    loss2.backward()
    o_loss.backward()
    for i in range(3):
        for pn in ("weight", "bias"):
            a, b = getattr(s.align[i], pn).grad.cpu(), getattr(o_s.align[i], pn).grad
            assert rel_l2(a, b) < 6e-2, (i, pn, rel_l2(a, b))
    gb = s.blocks[0].mlp.fc2.weight.grad
    assert gb is not None and gb.abs().max() > 0                                # gradients reach the backbone through the taps
    assert rel_l2(gb.cpu(), o_s.blocks[0].mlp.fc2.weight.grad) < 8e-2
    # the self-computed targets give the same loss up to the SVD's column signs: compare through sign-aligned columns
    with torch.no_grad():
        from deltakd_amd.losses import LowRankTargets
        mine = LowRankTargets()([f.to(torch.bfloat16).contiguous() for f in tf], 0, rank)
    for a, b in zip(mine, targets):
        sign = torch.sign((a.cpu() * b).sum(0))
        lead = 4
        assert rel_l2(a.cpu()[:, :lead] * sign[:lead], b[:, :lead]) < 3e-2
    assert abs(loss.item() - loss2.item()) <= 0.25 * abs(loss2.item())          # sign choice moves the loss


def test_mgd_free_function_positional_call():
    from oracle import loss_ref
    from model.loss import mgd_loss
    from deltakd_amd.models import forward_with_features
    o_t, o_s, t, s, args, size, C = _pair("mgd", False)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 3, size, size, generator=g)
    noise = torch.rand(4, 16, generator=g)
    s.eval(); o_s.eval()
    out, feats = forward_with_features(s, x.to(DEV))
    with torch.no_grad():
        _, tfeats = t.forward_with_taps(x.to(DEV))
    torch.manual_seed(0)
    d = mgd_loss(s, feats, tfeats, args, noise=noise.to(DEV))                     # (student_model, student_features, teacher_features, args)
    o_out, o_feats = loss_ref.forward_with_features_ref(o_s, x)
    with torch.no_grad():
        _, o_tf = loss_ref.forward_with_features_ref(o_t, x)
    o_d = loss_ref.mgd_ref(o_s, o_feats, o_tf, 1, 2, args.mgd_mask_ratio, args.mgd_alpha, noise)
    assert abs(d.item() - o_d.item()) <= 1e-2 * abs(o_d.item()), (d.item(), o_d.item())


# ---------------------------------------------------------------------------------------------------- data parallel, real HIP model
def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_dp(world, clip, out_dir, backend="gloo"):
    """Launch tests/dp_worker.py with ``world`` ranks (gloo, all on cuda:0) -> dict name -> weights after 2 steps (rank 0's)."""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", DKD_DP_BACKEND=backend)
        out = os.path.join(out_dir, f"w{world}_c{clip}_{backend}.pt")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), out, str(clip)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return torch.load(out, weights_only=True), logs


def test_data_parallel_collectives_through_rccl_in_a_world_of_one(tmp_path):
    """The data-parallel path on the "nccl" backend (= RCCL on ROCm), the library an 8-GPU run uses: one rank, wrapper forced active
    (`DataParallel(force=True)`), so the parameter broadcast, the three bucket all-reduces launched from the block-backward callback
    on the comm stream and the tail sync all go through RCCL on the real HIP model (reference: tools/train.py:307-308,
    tools/utils.py:52-63).  In a world of one every collective is the identity: the weights after 2 steps must equal the unwrapped
    model's (same bound as the 2-rank test: f32 atomics order differs between runs)."""
    one, _ = _run_dp(1, 0.05, str(tmp_path))
    rc, logs = _run_dp(1, 0.05, str(tmp_path), backend="nccl")
    log = logs[0]
    assert "backend: nccl" in log and "rccl mapped: True" in log, log[-2000:]
    assert "overlap buckets: [0, 4, 8]" in log, log[-2000:]
    m = re.search(r"allreduce calls: (\d+) bytes: (\d+) comm stream used: (\w+)", log)
    assert m and int(m.group(1)) >= 2 * 4 and int(m.group(2)) > 0 and m.group(3) == "True", log[-2000:]
    assert rc["clipped_steps"] == 2
    for n in one["weights"]:
        d0 = one["weights"][n] - one["init"][n]
        d1 = rc["weights"][n] - rc["init"][n]
        if d0.norm() == 0:
            continue
        if n.endswith("attn.qkv.bias"):
            D = d0.numel() // 3
            d0, d1 = torch.cat([d0[:D], d0[2 * D:]]), torch.cat([d1[:D], d1[2 * D:]])
        cos = torch.nn.functional.cosine_similarity(d0.flatten().double(), d1.flatten().double(), dim=0).item()
        assert cos > 0.98, (n, cos)
    assert abs(one["losses"][0] - rc["losses"][0]) <= 1e-5 * abs(one["losses"][0])


@pytest.mark.parametrize("clip", [0.0, 0.05])
def test_two_ranks_times_half_batch_equal_one_rank_times_full_batch(clip, tmp_path):
    """Data parallel with the REAL pieces: HIP student + FusedAdamW flat buffers + bucket all-reduces launched from the block-backward
    callback on the comm stream + tail sync (+ gradient clipping on the averaged gradients when clip > 0).  Two gloo ranks share the one
    GPU of the box, each with half of a batch of 8; after 2 steps their weights must equal a single process's on the whole batch
    (mean-reduced losses: the average of the half-batch gradients IS the full-batch gradient).  Both ranks must also agree bit for bit."""
    one, _ = _run_dp(1, clip, str(tmp_path))
    two, logs = _run_dp(2, clip, str(tmp_path))
    assert "overlap buckets: [0, 4, 8]" in logs[0], logs[0][-2000:]
    assert two["ranks_identical"], "rank 0 and rank 1 hold different weights"
    if clip:
        assert one["clipped_steps"] == 2 and two["clipped_steps"] == 2, "the clip threshold was meant to bite"
    worst = 0.0
    for n in one["weights"]:
        d0 = one["weights"][n] - one["init"][n]
        d1 = two["weights"][n] - two["init"][n]
        assert torch.equal(one["init"][n], two["init"][n])
        if d0.norm() == 0:
            continue
        if n.endswith("attn.qkv.bias"):        # key third: gradient is rounding noise (zero in exact arithmetic), Adam turns it into
            D = d0.numel() // 3                # +-lr steps of random sign -- compare the live query / value thirds
            d0, d1 = torch.cat([d0[:D], d0[2 * D:]]), torch.cat([d1[:D], d1[2 * D:]])
        # updates (not weights) compared: ~lr-sized; atomics order and bf16 noise on near-zero gradients flip a few Adam signs
        cos = torch.nn.functional.cosine_similarity(d0.flatten().double(), d1.flatten().double(), dim=0).item()
        assert cos > 0.98, (n, cos)
        worst = max(worst, 1 - cos)
    for a, b in zip(one["losses"], two["losses"]):
        assert abs(a - b) <= 2e-3 * abs(a), (one["losses"], two["losses"])
