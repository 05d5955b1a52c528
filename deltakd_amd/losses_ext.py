"""MGD, WassKD-L1, DiffKD, ViTKD and CurKD terms on libdkd.so (model/loss.py:229-236 + :422-452, :177-227, :105-155, :251-311,
:362-420 of the reference).

Each term is one autograd node whose forward runs the fused value+gradient kernels and whose backward is GEMMs only.
Random draws (masking noise, diffusion step, gaussian noise, dropout keep mask) come from torch's device generator unless
injected through ``DistillationLoss.injected`` (parity tests replay the reference's draws).
"""
import math

import torch

from . import ops
from .ffi import IDENT, RowMap, strip_map
from .losses import _AlignTermFn, _grad_buffer, _pad64
from .misc import masking_indices, saliency_scores
from .vit import BF16, F32, ensure_grad


# ----------------------------------------------------------------------------------------------- MGD
class _MgdFn(torch.autograd.Function):
    """align -> where(mask, mask_token, .) -> Conv3x3 -> ReLU -> Conv3x3 -> masked MSE vs the teacher's last tap.

    The two 3 x 3 convolutions (model/models.py:148-151) are implicit GEMMs: the activation [B*196, Dt] is the A operand as it is, the
    kernel gathers the 3 x 3 neighbourhood (zero padding included) in its LDS-DMA source addressing, so the [B*196, 9 Dt] im2col
    matrix -- 694 MB at Dt = 768, bs 256 -- is never written or read, forward or backward: the input gradient is the same convolution
    applied to dY with flipped taps, the weight gradient nine split-M GEMMs against row-shifted views of the input."""

    @staticmethod
    def forward(ctx, tap, sm, align, t_tap, mask, scale, npre_s, npre_t):
        B, N, Ds = tap.shape
        P = N - npre_s
        hw = int(P ** 0.5)
        M, Dt = B * P, align.out_features
        sh = sm._shadow
        tap2 = tap.reshape(B * N, Ds)
        c1, c2 = sm.generation[0], sm.generation[2]
        s = ops.gemm_nt(tap2, sh.get(align.weight), M=M, amap=strip_map(N, npre_s), bias=align.bias)
        xt = ops.mask_select(s, sm.mask_token.detach().reshape(-1).contiguous(), mask)
        y1 = ops.gemm_nt(xt, sh.get(c1.weight, conv3x3=True), bias=c1.bias, relu=True, conv_hw=hw)
        y2 = ops.gemm_nt(y1, sh.get(c2.weight, conv3x3=True), bias=c2.bias, out_f32=True, conv_hw=hw)
        loss = torch.zeros(1, device=tap.device, dtype=F32)
        Nt = t_tap.shape[1]
        dy2 = ops.mse_loss(y2, t_tap.reshape(B * Nt, Dt), loss, scale / (M * Dt), M=M, tmap=strip_map(Nt, npre_t), mask=mask)
        ctx.sm, ctx.align, ctx.saved, ctx.dims = sm, align, (tap2, xt, y1, dy2, mask), (B, N, Ds, npre_s, hw, M, Dt)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        sm, align = ctx.sm, ctx.align
        sh = sm._shadow
        tap2, xt, y1, dy2, mask = ctx.saved
        B, N, Ds, npre, hw, M, Dt = ctx.dims
        c1, c2 = sm.generation[0], sm.generation[2]
        dy2.mul_(g)

        def conv_wgrad(dy, x_in, conv):
            dwp = torch.zeros(Dt, 9 * Dt, device=dy.device, dtype=F32)
            ops.conv3x3_wgrad(dy, x_in, dwp, ensure_grad(conv.bias), B, hw)
            ensure_grad(conv.weight).add_(dwp.view(Dt, 3, 3, Dt).permute(0, 3, 1, 2))

        conv_wgrad(dy2, y1, c2)
        dy1 = ops.gemm_nt(dy2, sh.get(c2.weight, conv3x3="dgrad"), conv_hw=hw, relu_gate=y1)
        conv_wgrad(dy1, xt, c1)
        dxt = ops.gemm_nt(dy1, sh.get(c1.weight, conv3x3="dgrad"), conv_hw=hw)
        ds = ops.mask_select_bwd(dxt, mask, ensure_grad(sm.mask_token).view(-1))
        smap = strip_map(N, npre)
        ops.gemm_tn(ds, tap2, ensure_grad(align.weight), M=M, bmap=smap, colsum=ensure_grad(align.bias))
        dtap = torch.zeros(B * N, Ds, device=ds.device, dtype=BF16)
        ops.gemm_nt(ds, sh.get(align.weight, transposed=True), out=dtap, cmap=smap)
        ctx.saved = None
        return dtap.view(B, N, Ds), None, None, None, None, None, None, None


def mgd_loss(student_model, student_features, teacher_features, args, *, npre_s=1, npre_t=2, noise=None):
    """model/loss.py:422-452: mgd_alpha * mean((G(x~) * m - t * m)^2) with x~ = where(m, mask_token, align(s))."""
    tap = student_features[-1]
    B, N, _ = tap.shape
    P = N - npre_s
    if noise is None:
        noise = torch.rand(B, P, device=tap.device)
    mask, _, _, _ = masking_indices(noise, args.mgd_mask_ratio)
    return _MgdFn.apply(tap, student_model, student_model.align, teacher_features[-1], mask.reshape(-1).contiguous(),
                        float(args.mgd_alpha), npre_s, npre_t)


# ----------------------------------------------------------------------------------------------- WassKD (L1)
def wasskd_l1_loss(student_model, student_features, teacher_features, weight, npre_s, npre_t):
    """weight * (1/3) sum_i mean|sort_tokens(align_i(s_i)) - sort_tokens(t_i)|   (model/loss.py:187-199, x5 at :226)."""
    total = None
    for i in range(3):
        tap, t_tap = student_features[i], teacher_features[i]
        B, N, _ = tap.shape
        P = N - npre_s
        Nt, Dt = t_tap.shape[1], t_tap.shape[2]
        t2 = t_tap.reshape(B * Nt, Dt)
        w = weight / (3.0 * B * P * Dt)

        def cb(s, loss, Kp, t2=t2, w=w, B=B, P=P, Nt=Nt):
            ds = ops.sort_l1_loss(s, t2, loss, w, B=B, P=P, tmap=strip_map(Nt, npre_t))
            if Kp != ds.shape[1]:
                pad = _grad_buffer(ds.shape[0], ds.shape[1], Kp, ds.device)
                pad[:, :ds.shape[1]] = ds
                ds = pad
            return ds
        term = _AlignTermFn.apply(tap, student_model.align_wasskd[i], student_model._shadow, npre_s, cb)
        total = term if total is None else total + term
    return total


# ----------------------------------------------------------------------------------------------- DiffKD
class _DenoiseFn(torch.autograd.Function):
    """mse(denoise(t^ + noise, t), noise) of model/loss.py:138-145: noising kernel -> fc(GELU) -> fc -> dropout-MSE."""

    @staticmethod
    def forward(ctx, temb, dn, sh, t_tap, noise, sigma, keep, scale, npre_t):
        B, Nt, Dt = t_tap.shape
        P = Nt - npre_t
        M = B * P
        f0, f2 = dn.net[0], dn.net[2]
        t_hat, nz, x_in = ops.diffkd_prepare(t_tap.reshape(B * Nt, Dt), noise, sigma, temb.detach().contiguous(), M=M, rows_per_sample=P,
                                             tmap=strip_map(Nt, npre_t))
        pre = torch.empty(M, f0.out_features, device=t_tap.device, dtype=BF16)
        h = ops.gemm_nt(x_in, sh.get(f0.weight), bias=f0.bias, gelu=True, preact=pre)
        raw = ops.gemm_nt(h, sh.get(f2.weight), bias=f2.bias, out_f32=True)
        loss = torch.zeros(1, device=t_tap.device, dtype=F32)
        dpred = ops.dropout_mse(raw, nz, keep, 1.0 / 0.9 if keep is not None else 1.0, loss, scale / (M * Dt))
        ctx.dn, ctx.sh, ctx.saved, ctx.dims = dn, sh, (x_in, pre, h, dpred), (B, P, Dt)
        ctx.mark_non_differentiable(t_hat)
        return loss[0], t_hat

    @staticmethod
    def backward(ctx, g, _):
        dn, sh = ctx.dn, ctx.sh
        x_in, pre, h, dpred = ctx.saved
        B, P, Dt = ctx.dims
        f0, f2 = dn.net[0], dn.net[2]
        dpred.mul_(g)
        ops.gemm_tn(dpred, h, ensure_grad(f2.weight), colsum=ensure_grad(f2.bias))
        dh = ops.gemm_nt(dpred, sh.get(f2.weight, transposed=True), dgelu=True, preact=pre)
        ops.gemm_tn(dh, x_in, ensure_grad(f0.weight), colsum=ensure_grad(f0.bias))
        dx = ops.gemm_nt(dh, sh.get(f0.weight, transposed=True), out_f32=True)
        dtemb = dx.view(B, P, Dt).sum(1)         # the time embedding is broadcast over the P tokens of a sample
        ctx.saved = None
        return dtemb, None, None, None, None, None, None, None, None


def diffkd_loss(student_model, student_features, teacher_features, alpha, npre_s, npre_t, injected=None, terms_out=None):
    """alpha * (5e-5 / 3) * sum_i [mse(pred_noise_i, noise_i) + mean(w_t) * mse(s^_i, t^_i)]   (model/loss.py:105-155).
    ``terms_out`` (list, optional) receives the six addends (denoise_i, match_i) as detached device scalars."""
    injected = injected or {}
    sm = student_model
    sel_s = [student_features[0], student_features[1], student_features[-1]]
    sel_t = [teacher_features[0], teacher_features[1], teacher_features[-1]]
    B = sel_s[0].shape[0]
    dev = sel_s[0].device
    T = 8
    t = injected["t"] if "t" in injected else torch.randint(0, T, (B,), device=dev)
    sigma_max = torch.where(t < T // 2, torch.tensor(0.3, device=dev), torch.tensor(0.7, device=dev))
    sigma = ((1 - torch.cos(math.pi * t.float() / T)) * sigma_max).float().contiguous()
    w_mean = (1 / (sigma ** 2 + 1e-8)).mean().reshape(1).contiguous()
    scale = alpha * 5e-5 / 3.0
    dn = sm.denoise_fn
    total = None
    for i in range(3):
        t_tap = sel_t[i]
        Nt, Dt = t_tap.shape[1], t_tap.shape[2]
        P = Nt - npre_t
        M = B * P
        temb = dn.time_embed(t.float().view(-1, 1))                      # [B, Dt]: two callable vit.Linear layers (MFMA GEMMs) around a torch GELU
        noise = injected["noise"][i].reshape(M, Dt).contiguous() if "noise" in injected else torch.randn(M, Dt, device=dev)
        if "drop" in injected:
            keep = injected["drop"][i].reshape(M, Dt).contiguous()
        else:
            keep = (torch.rand(M, Dt, device=dev) < 0.9).float() if dn.training else None
        term_dn, t_hat = _DenoiseFn.apply(temb, dn, sm._shadow, t_tap, noise, sigma, keep, scale, npre_t)

        def cb(s, loss, Kp, t_hat=t_hat, M=M, Dt=Dt):
            return ops.normalize_mse(s, t_hat, loss, scale / (M * Dt), w_scalar=w_mean, ld_grad=Kp)
        term_match = _AlignTermFn.apply(sel_s[i], sm.align[i], sm._shadow, npre_s, cb)
        if terms_out is not None:
            terms_out += [term_dn.detach(), term_match.detach()]
        total = term_dn + term_match if total is None else total + term_dn + term_match
    return total


# ----------------------------------------------------------------------------------------------- ViTKD / CurKD
def _sum_mse_term(tap, align, shadow, t_tap, scale_per_batch, npre_s, npre_t):
    """scale * sum((align(tap[:, npre:]) - t[:, npre_t:])^2) / B   (nn.MSELoss(reduction='sum') / B of the reference)."""
    from .losses import align_mse_term
    B, N, _ = tap.shape
    P = N - npre_s
    Nt, Dt = t_tap.shape[1], t_tap.shape[2]
    # align_mse_term computes scale * mean over B*P*Dt elements: sum / B = mean * P * Dt
    return align_mse_term(tap, align, shadow, t_tap.reshape(B * Nt, Dt), strip_map(Nt, npre_t), scale_per_batch * P * Dt, npre_s)


def _masked_generation_term(student_model, align, tap, t_tap, ratio, scale_per_batch, npre_s, npre_t, noise):
    B, N, _ = tap.shape
    P = N - npre_s
    if noise is None:
        noise = torch.rand(B, P, device=tap.device)
    mask, _, _, _ = masking_indices(noise, ratio)
    Dt = t_tap.shape[2]
    return _MgdFn.apply(tap, student_model, align, t_tap, mask.reshape(-1).contiguous(), scale_per_batch * P * Dt, npre_s, npre_t)


def vitkd_loss(student_model, student_features, teacher_features, alpha_vitkd=0.00003, beta_vitkd=0.000003, lambda_vitkd=0.5, *,
               npre_s=1, npre_t=2, noise=None):
    """model/loss.py:251-311: sum-reduced mimicking of blocks 0, 1 through ``align2`` + masked generation on the last block."""
    sm = student_model
    total = None
    for i in range(2):
        term = _sum_mse_term(student_features[i], sm.align2[i], sm._shadow, teacher_features[i], alpha_vitkd, npre_s, npre_t)
        total = term if total is None else total + term
    return total + _masked_generation_term(sm, sm.align, student_features[-1], teacher_features[-1], lambda_vitkd,
                                           beta_vitkd / lambda_vitkd, npre_s, npre_t, noise)


def curkd_loss(student_model, student_features, teacher_features, args, *, npre_s=1, npre_t=2, noise=None):
    """model/loss.py:362-420: epoch curriculum -- blocks 0-2 (epoch < 100), blocks 3-6 (< 151), then masked generation on block 11."""
    sm = student_model
    epoch = args.current_epoch
    # tell the optimizer which stage is in use this step (an explicit device flag per lazy unit: a used stage whose gradient happens
    # to be exactly zero must still be updated -- weight decay, moment decay -- as torch.optim.AdamW does; deltakd_amd.optim)
    stage = sm.curkd_align_early if epoch < 100 else sm.curkd_align_mid if epoch < 151 else sm.curkd_align_last
    if torch.is_grad_enabled():
        flag = getattr(next(stage.parameters()), "dkd_unit_touched", None)
        if flag is not None:
            flag.fill_(1.0)
    if epoch < 100:
        layers, mods, div = range(3), sm.curkd_align_early, 3.0
    elif epoch < 151:
        layers, mods, div = range(3, 7), sm.curkd_align_mid, 4.0
    else:
        return _masked_generation_term(sm, sm.curkd_align_last, student_features[11], teacher_features[11], 0.5, 5e-5, npre_s, npre_t, noise)
    total = None
    for j, i in enumerate(layers):
        term = _sum_mse_term(student_features[i], mods[j], sm._shadow, teacher_features[i], 4e-5 / div, npre_s, npre_t)
        total = term if total is None else total + term
    return total


def saliency_mgd_loss(student_model, student_features, teacher_features, args, *, npre_s=1, npre_t=2, scores=None):
    """model/loss.py:335-360: MGD whose mask keeps the lowest-saliency tokens (scores from ``student_model.saliency_attn`` applied to
    the teacher's last tap); 4 * mean((G(x~) m - t m)^2)."""
    t_tap = teacher_features[-1]
    if scores is None:
        scores = saliency_scores(student_model, t_tap, args.saliency_method, npre_t)
    mask, _, _, _ = masking_indices(scores, args.saliency_mask_ratio)
    return _MgdFn.apply(student_features[-1], student_model, student_model.align, t_tap, mask.reshape(-1).contiguous(), 4.0, npre_s, npre_t)
