"""Pure-torch fp32 restatement of the reference's distillation losses.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Follows /root/reference/model/loss.py (DistillationLoss.forward :29-242,
lrkd_loss :314-330, mgd_loss :422-452), /root/reference/model/misc.py
(random_masking :5-32), the aux modules of /root/reference/model/models.py
(:59-178) and timm's SoftTargetCrossEntropy / LabelSmoothingCrossEntropy.
Pinned against the reference's own code by oracle/gen_golden.py.

Every random draw is an explicit input (``draws`` dict):
  mgd     : draws["noise"]  [B, P]         (torch.rand in random_masking, misc.py:14)
  diffkd  : draws["t"] [B] int64, draws["noise"][i] [B,P,Dt] standard normal (i=0..2),
            draws["drop"][i] [B,P,Dt] 0/1 keep mask of Dropout(0.1)
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- base criteria (timm.loss restated; call_base_loss model/loss.py:244-249) ----
class SoftTargetCrossEntropyRef(nn.Module):
    def forward(self, x, target):
        return torch.sum(-target * F.log_softmax(x, dim=-1), dim=-1).mean()


class LabelSmoothingCrossEntropyRef(nn.Module):
    def __init__(self, smoothing=0.1):
        super().__init__()
        self.smoothing = smoothing

    def forward(self, x, target):
        logp = F.log_softmax(x, dim=-1)
        nll = -logp.gather(-1, target.unsqueeze(1)).squeeze(1)
        smooth = -logp.mean(dim=-1)
        return ((1.0 - self.smoothing) * nll + self.smoothing * smooth).mean()


def call_base_loss_ref(args):
    if args.mixup > 0 or args.cutmix > 0. or args.cutmix_minmax:
        return SoftTargetCrossEntropyRef()
    return LabelSmoothingCrossEntropyRef(args.smoothing)


# ---- feature tap (model/models.py:181-199) ----
def forward_with_features_ref(model, x):
    taps = []
    hooks = [blk.mlp.register_forward_hook(lambda m, i, o: taps.append(o)) for blk in model.blocks]
    try:
        out = model(x)
    finally:
        for h in hooks:
            h.remove()
    return out, taps


# ---- aux modules bolted onto the student (model/models.py:76-176) ----
class DenoisingNetworkRef(nn.Module):
    """model/models.py:103-121 with the Dropout(0.1) draw made explicit."""

    def __init__(self, dims):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dims, dims * 2), nn.GELU(), nn.Linear(dims * 2, dims), nn.Dropout(0.1))
        self.time_embed = nn.Sequential(nn.Linear(1, dims), nn.GELU(), nn.Linear(dims, dims))

    def forward(self, x, t, drop_keep=None):
        x = x + self.time_embed(t.float().view(-1, 1)).unsqueeze(1)
        y = self.net[2](self.net[1](self.net[0](x)))
        if drop_keep is None:
            return self.net[3](y)
        return y * drop_keep / 0.9


class SimpleAttentionRef(nn.Module):
    """model/models.py:38-56: per-head softmax(q k^T / sqrt(d)) averaged over heads; returns its diagonal [B, N]."""

    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.qk = nn.Linear(dim, dim * 2, bias=True)

    def forward(self, x):
        B, N, C = x.shape
        qk = self.qk(x).reshape(B, N, 2, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        attn = ((qk[0] @ qk[1].transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return attn.mean(dim=1).diagonal(dim1=-2, dim2=-1)


class SimpleCrossAttentionRef(nn.Module):
    """model/models.py:14-35: softmax(q(x_query) k(x_key)^T / sqrt(d)) averaged over heads -> [B, Nq, Nk]."""

    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=True)
        self.k = nn.Linear(dim, dim, bias=True)

    def forward(self, xq, xk):
        B, Nq, C = xq.shape
        Nk = xk.shape[1]
        q = self.q(xq).reshape(B, Nq, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3)
        k = self.k(xk).reshape(B, Nk, self.num_heads, C // self.num_heads).permute(0, 2, 1, 3)
        return ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1).mean(dim=1)


def saliency_scores_ref(attn_mod, t_feat, method, pre_t=2):
    """Per-patch saliency of model/misc.py:38-165 (the quantity that is argsorted; low = kept). t_feat [B, N_t, Dt] with prefix."""
    if method == 1:
        return attn_mod(t_feat[:, pre_t:])
    cls_patch = torch.cat([t_feat[:, :1], t_feat[:, pre_t:]], dim=1)
    if method == 2:
        B, L, D = cls_patch.shape
        H = attn_mod.num_heads
        q, k = torch.chunk(attn_mod.qk(cls_patch), 2, dim=-1)
        q = q.reshape(B, L, H, D // H).permute(0, 2, 1, 3)
        k = k.reshape(B, L, H, D // H).permute(0, 2, 1, 3)
        attn = ((q[:, :, 0:1] @ k.transpose(-2, -1)) * (D // H) ** -0.5).softmax(dim=-1)
        return attn.mean(dim=1).squeeze(1)[:, 1:]
    if method == 3:
        return attn_mod(cls_patch[:, :1], cls_patch[:, 1:]).squeeze(1)
    raise ValueError(f"Invalid saliency masking method: {method}")


def saliency_mgd_ref(student, s_feats, t_feats, pre_s, pre_t, ratio, method, scores=None):
    """model/loss.py:335-360: MGD whose mask keeps the LOWEST-saliency tokens; loss = 4 * mean((G(x~) m - t m)^2)."""
    x = student.align(s_feats[-1][:, pre_s:])
    t = t_feats[-1]
    if scores is None:
        scores = saliency_scores_ref(student.saliency_attn, t, method, pre_t)
    xg, m = _masked_generation_ref(student, x, None, ratio, scores)       # argsort(scores) plays the role of argsort(noise)
    return F.mse_loss(xg * m, t[:, pre_t:] * m) * 4


def attach_aux_ref(student, teacher, kind, lrkd_rank=64, saliency_method=1):
    ds, dt = student.embed_dim, teacher.embed_dim

    def gen():
        return nn.Sequential(nn.Conv2d(dt, dt, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(dt, dt, 3, padding=1))

    if kind == "lrkd":
        student.align = nn.ModuleList([nn.Linear(ds, lrkd_rank) for _ in range(3)])
    elif kind == "diffkd":
        student.denoise_fn = DenoisingNetworkRef(dt)
        student.align = nn.ModuleList([nn.Linear(ds, dt) for _ in range(3)])
    elif kind == "mgd":
        student.align = nn.Linear(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = gen()
    elif kind == "wasskd":
        student.align_wasskd = nn.ModuleList([nn.Linear(ds, dt) for _ in range(3)])
    elif kind == "vitkd":                                   # model/models.py:76-88
        student.align2 = nn.ModuleList([nn.Linear(ds, dt) for _ in range(2)])
        student.align = nn.Linear(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = gen()
    elif kind == "curkd":                                   # model/models.py:153-167
        student.curkd_align_early = nn.ModuleList([nn.Linear(ds, dt) for _ in range(3)])
        student.curkd_align_mid = nn.ModuleList([nn.Linear(ds, dt) for _ in range(4)])
        student.curkd_align_last = nn.Linear(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = gen()
    elif kind == "saliency_mgd":                            # model/models.py:129-143
        student.align = nn.Linear(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = gen()
        student.saliency_attn = SimpleCrossAttentionRef(dt, 8) if saliency_method == 3 else SimpleAttentionRef(dt, 8)
    elif kind in ("soft", "hard"):
        if hasattr(student, "set_distilled_training"):
            student.set_distilled_training(True)
    return student


# ---- masking (model/misc.py:5-32) ----
def random_masking_ref(x, mask_ratio, noise):
    B, L, D = x.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :len_keep]
    x_keep = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    mask = torch.ones(B, L)
    mask[:, :len_keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return x_keep, mask, ids_restore, ids_shuffle[:, len_keep:]


# ---- per-branch distillation terms ----
def soft_ref(z_kd, z_t, tau):
    return F.kl_div(F.log_softmax(z_kd / tau, 1), F.log_softmax(z_t / tau, 1), reduction="sum",
                    log_target=True) * (tau * tau) / z_kd.numel()


def hard_ref(z_kd, z_t):
    return F.cross_entropy(z_kd, z_t.argmax(1))


def lrkd_targets_ref(t_feat, rank):
    """U_k S_k of the [B*P, Dt] teacher matrix (model/loss.py:318-324). No grad."""
    t2 = t_feat.reshape(-1, t_feat.size(-1))
    U, S, _ = torch.linalg.svd(t2, full_matrices=False)
    return U[:, :rank] * S[:rank]


def lrkd_ref(student, s_feats, t_feats, pre_s, pre_t, rank, w, targets=None):
    sel_s = [s_feats[0], s_feats[1], s_feats[-1]]
    sel_t = [t_feats[0], t_feats[1], t_feats[11]]
    total = 0.0
    for i in range(3):
        s = student.align[i](sel_s[i][:, pre_s:]).reshape(-1, rank)
        a = targets[i] if targets is not None else lrkd_targets_ref(sel_t[i][:, pre_t:], rank)
        total = total + w[i] * F.mse_loss(a, s)
    return total


def mgd_ref(student, s_feats, t_feats, pre_s, pre_t, mask_ratio, mgd_alpha, noise):
    x = student.align(s_feats[-1][:, pre_s:])
    t = t_feats[-1][:, pre_t:]
    B, N, D = x.shape
    keep, mask, ids_restore, _ = random_masking_ref(x, mask_ratio, noise)
    x_ = torch.cat([keep, student.mask_token.repeat(B, N - keep.shape[1], 1)], 1)
    x = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, D))
    hw = int(N ** 0.5)
    x = student.generation(x.reshape(B, hw, hw, D).permute(0, 3, 1, 2)).flatten(2).transpose(1, 2)
    m = mask.unsqueeze(-1)
    return F.mse_loss(x * m, t * m) * mgd_alpha


def wasskd_l1_ref(student, s_feats, t_feats, pre_s, pre_t):
    tot = 0.0
    for i in range(3):
        a = student.align_wasskd[i](s_feats[i][:, pre_s:])
        t = t_feats[i][:, pre_t:]
        tot = tot + (torch.sort(a, 1)[0] - torch.sort(t, 1)[0]).abs().mean()
    return tot / 3.0


def diffkd_ref(student, s_feats, t_feats, pre_s, pre_t, t, noises, drops):
    T = 8
    sigma_max = torch.where(t < T // 2, torch.tensor(0.3), torch.tensor(0.7))
    sigma = (1 - torch.cos(math.pi * t.float() / T)) * sigma_max
    sel_s = [s_feats[0], s_feats[1], s_feats[-1]]
    sel_t = [t_feats[0], t_feats[1], t_feats[-1]]
    feat = 0.0
    for i in range(3):
        sf = student.align[i](sel_s[i][:, pre_s:])
        tf = sel_t[i][:, pre_t:]
        tf = tf / tf.norm(p=2, dim=-1, keepdim=True)
        sf = sf / sf.norm(p=2, dim=-1, keepdim=True)
        nz = noises[i] * sigma.view(-1, 1, 1)
        pred = student.denoise_fn(tf + nz, t, drops[i])
        feat = feat + F.mse_loss(pred, nz)
        feat = feat + (1 / (sigma ** 2 + 1e-8)).mean() * F.mse_loss(sf, tf)
    return feat / 3 * 5e-5


def _masked_generation_ref(student, x, t, ratio, noise):
    """gather-keep -> cat(mask_token) -> restore -> Conv3x3-ReLU-Conv3x3; returns (generated tokens, mask [B, N, 1])."""
    B, N, D = x.shape
    keep, mask, ids_restore, _ = random_masking_ref(x, ratio, noise)
    x_ = torch.cat([keep, student.mask_token.repeat(B, N - keep.shape[1], 1)], 1)
    x = torch.gather(x_, 1, ids_restore.unsqueeze(-1).repeat(1, 1, D))
    hw = int(N ** 0.5)
    x = student.generation(x.reshape(B, hw, hw, D).permute(0, 3, 1, 2)).flatten(2).transpose(1, 2)
    return x, mask.unsqueeze(-1)


def vitkd_ref(student, s_feats, t_feats, pre_s, pre_t, noise, alpha_vitkd=0.00003, beta_vitkd=0.000003, lambda_vitkd=0.5):
    """model/loss.py:251-311: sum-reduced mimicking on blocks 0,1 + masked generation on the last block."""
    B = s_feats[0].shape[0]
    lr = 0.0
    for i in range(2):
        lr = lr + ((student.align2[i](s_feats[i][:, pre_s:]) - t_feats[i][:, pre_t:]) ** 2).sum()
    x, m = _masked_generation_ref(student, student.align(s_feats[-1][:, pre_s:]), None, lambda_vitkd, noise)
    gen = ((x * m - t_feats[-1][:, pre_t:] * m) ** 2).sum()
    return lr / B * alpha_vitkd + gen / B * beta_vitkd / lambda_vitkd


def curkd_ref(student, s_feats, t_feats, pre_s, pre_t, epoch, noise):
    """model/loss.py:362-420: epoch curriculum (early blocks 0-2, mid blocks 3-6, late masked generation on block 11)."""
    B = s_feats[0].shape[0]
    if epoch < 100:
        tot = sum(((student.curkd_align_early[i](s_feats[i][:, pre_s:]) - t_feats[i][:, pre_t:]) ** 2).sum() for i in range(3))
        return tot / 3.0 / B * 4e-5
    if epoch < 151:
        tot = sum(((student.curkd_align_mid[i - 3](s_feats[i][:, pre_s:]) - t_feats[i][:, pre_t:]) ** 2).sum() for i in range(3, 7))
        return tot / 4.0 / B * 4e-5
    x, m = _masked_generation_ref(student, student.curkd_align_last(s_feats[11][:, pre_s:]), None, 0.5, noise)
    return ((x * m - t_feats[11][:, pre_t:] * m) ** 2).sum() / B * 5e-5


class DistillationLossRef(nn.Module):
    """Same constructor/call contract as model/loss.py:19-29, plus ``draws``.

    Deliberate superset: prefix tokens stripped are
    ``num_prefix_tokens`` of each model (the reference hard-codes 1 / 2).
    """

    def __init__(self, base_criterion, teacher_model, distillation_type, alpha, tau):
        super().__init__()
        self.base_criterion = base_criterion
        self.teacher_model = teacher_model
        self.distillation_type = distillation_type
        self.alpha, self.tau = alpha, tau

    def forward(self, inputs, outputs, student_model, student_features, labels, args, draws=None):
        draws = draws or {}
        kd = None
        if not isinstance(outputs, torch.Tensor):
            outputs, kd = outputs
        base = self.base_criterion(outputs, labels)
        kind = self.distillation_type.lower()
        if kind == "none":
            return base
        if kd is None and kind in ("soft", "hard"):
            raise ValueError("soft/hard distillation needs a (logits, logits_dist) tuple from the student")
        with torch.no_grad():
            if kind in ("soft", "hard"):
                zt, tf = self.teacher_model(inputs), None
            else:
                zt, tf = forward_with_features_ref(self.teacher_model, inputs)
        ps = getattr(student_model, "num_prefix_tokens", 1)
        pt = getattr(self.teacher_model, "num_prefix_tokens", 2)
        if kind == "soft":
            d = soft_ref(kd, zt, self.tau)
        elif kind == "hard":
            d = hard_ref(kd, zt)
        elif kind == "lrkd":
            d = lrkd_ref(student_model, student_features, tf, ps, pt, args.lrkd_rank,
                         (args.lrkd_alpha, args.lrkd_beta, args.lrkd_gamma), draws.get("lrkd_targets"))
        elif kind == "diffkd":
            d = diffkd_ref(student_model, student_features, tf, ps, pt, draws["t"], draws["noise"], draws["drop"])
        elif kind == "wasskd":
            if args.wasskd_type != "l1":
                raise NotImplementedError("sinkhorn: geomloss oracle unavailable (parity unpinned)")
            return base + 5.0 * wasskd_l1_ref(student_model, student_features, tf, ps, pt)
        elif kind == "saliency_mgd":
            return base + saliency_mgd_ref(student_model, student_features, tf, ps, pt, args.saliency_mask_ratio, args.saliency_method,
                                           draws.get("scores"))
        elif kind == "vitkd":
            return base + vitkd_ref(student_model, student_features, tf, ps, pt, draws["noise"])
        elif kind == "curkd":
            return base + curkd_ref(student_model, student_features, tf, ps, pt, args.current_epoch, draws.get("noise"))
        elif kind == "mgd":
            return base + mgd_ref(student_model, student_features, tf, ps, pt, args.mgd_mask_ratio,
                                  args.mgd_alpha, draws["noise"])
        else:
            raise ValueError(f"Invalid distillation type: {self.distillation_type}")
        return base * (1 - self.alpha) + d * self.alpha


def default_args(**kw):
    a = dict(mixup=0.0, cutmix=0.0, cutmix_minmax=None, smoothing=0.1, lrkd_rank=64, lrkd_alpha=0.2, lrkd_beta=0.2,
             lrkd_gamma=0.2, wasskd_type="l1", mgd_alpha=7e-5, mgd_mask_ratio=0.5, alpha=0.1, tau=3.0,
             distillation_type="none", current_epoch=0, amp=False, rank=0, epochs=1, saliency_method=1,
             saliency_mask_ratio=0.5)
    a.update(kw)
    return SimpleNamespace(**a)
