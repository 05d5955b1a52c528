"""Counterparts of the timm utilities the reference's step loop imports ([3P] timm==0.9.12; absent on the MI355X boxes):
``timm.utils.accuracy`` / ``NativeScaler`` (tools/engine.py:3, tools/train.py:11) and ``timm.data.Mixup`` (tools/train.py:7).
Host-side plumbing on torch tensors (they work on host or device tensors alike); the arithmetic that matters is in libdkd.
"""
import os

import numpy as np
import torch


def accuracy(output, target, topk=(1,)):
    """top-k accuracy in percent as 0-dim tensors (timm.utils.accuracy)."""
    if output.is_cuda and output.dim() == 2 and target.dim() == 1 and 1 <= len(topk) <= 4 and target.dtype in (torch.int64, torch.int32):
        # device logits: one libdkd launch (rank of the label's logit per row) instead of topk + sort + compare + reductions
        from . import ops
        z = output.detach()
        if z.dtype != torch.float32 or not z.is_contiguous():
            z = z.float().contiguous()
        acc = ops.topk_correct(z, target.to(torch.int64).contiguous(), [min(k, output.size(1)) for k in topk])
        return [acc[i] for i in range(len(topk))]
    maxk = min(max(topk), output.size(1))
    batch = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand(maxk, -1))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / batch for k in topk]


class NativeScaler:
    """timm.utils.NativeScaler's call contract: ``scaler(loss, optimizer, clip_grad=, parameters=, create_graph=)`` runs
    backward, optional grad-norm clipping, and the optimizer step.  The bf16 MFMA path keeps fp32's exponent range, so no
    loss scaling is applied (the reference's GradScaler is itself inert without ``--amp``)."""
    state_dict_key = "amp_scaler"

    def __call__(self, loss, optimizer, clip_grad=None, clip_mode="norm", parameters=None, create_graph=False, need_update=True):
        loss.backward(create_graph=create_graph)
        if need_update:
            if clip_grad is not None:
                assert parameters is not None
                # data parallel: the norm must be taken over the AVERAGED gradients (torch DDP has finished its all-reduce when
                # backward returns); deltakd_amd.ddp finishes its tail all-reduce in sync_grads()
                if clip_mode == "norm" and hasattr(optimizer, "clip_grad_norm_"):
                    optimizer.clip_grad_norm_(clip_grad)                 # flat buffers: syncs first, one norm per buffer
                else:
                    if hasattr(optimizer, "sync_grads"):
                        optimizer.sync_grads()
                    params = [p for p in parameters if p.grad is not None]
                    if clip_mode == "value":
                        torch.nn.utils.clip_grad_value_(params, clip_grad)
                    else:
                        torch.nn.utils.clip_grad_norm_(params, clip_grad)
            optimizer.step()

    # checkpoint["scaler"]: timm's NativeScaler saves its torch GradScaler's state; this one scales nothing, so it writes a disabled
    # GradScaler's state (an empty dict, what torch returns for enabled=False) and accepts anything on load
    def state_dict(self):
        return {}

    def load_state_dict(self, state_dict):
        if not isinstance(state_dict, dict):
            raise TypeError("scaler state must be a dict")


def _one_hot(x, num_classes, on_value, off_value):
    x = x.long().view(-1, 1)
    return torch.full((x.size(0), num_classes), off_value, device=x.device).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1., smoothing=0.0):
    off = smoothing / num_classes
    on = 1. - smoothing + off
    return _one_hot(target, num_classes, on, off) * lam + _one_hot(target.flip(0), num_classes, on, off) * (1. - lam)


class Mixup:
    """timm.data.Mixup, mode='batch' (the mode every exp/*.sh uses): one lambda ~ Beta per batch from numpy's global RNG,
    CutMix with probability ``switch_prob`` (random box, lambda corrected to the box area), label-smoothed soft targets."""

    def __init__(self, mixup_alpha=1., cutmix_alpha=0., cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch",
                 correct_lam=True, label_smoothing=0.1, num_classes=1000, inplace=False, patch_size=16):
        if mode != "batch":
            raise ValueError("deltakd_amd.shims.Mixup implements mode='batch'")
        if cutmix_minmax is not None:
            raise ValueError("cutmix_minmax is not supported")
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes, self.correct_lam = label_smoothing, num_classes, correct_lam
        self.mixup_enabled = True
        # device batches only: False (default) returns the mix in a new tensor and leaves the loader's batch alone (same bytes moved; a
        # batch kept resident in HBM can be mixed again next epoch / step without a copy); True overwrites x as timm does.  The reference's
        # loop uses the RETURNED tensor only (tools/engine.py:16-18).  Host tensors are always mixed in place, as in timm.
        self.inplace = inplace
        # out-of-place device mixes also leave the bf16 patch matrix of the mixed batch for the models' patch embeddings (same launch,
        # deltakd_amd.vit.register_patches): the gather pass over the mixed images disappears.  None disables; a model with another patch
        # size simply does not find the matrix and gathers its own.
        self.patch_size = patch_size

    def _params_per_batch(self):
        lam, use_cutmix = 1., False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand() < self.switch_prob
                a = self.cutmix_alpha if use_cutmix else self.mixup_alpha
                lam = float(np.random.beta(a, a))
            elif self.mixup_alpha > 0.:
                lam = float(np.random.beta(self.mixup_alpha, self.mixup_alpha))
            elif self.cutmix_alpha > 0.:
                use_cutmix = True
                lam = float(np.random.beta(self.cutmix_alpha, self.cutmix_alpha))
        return lam, use_cutmix

    def __call__(self, x, target):
        assert len(x) % 2 == 0, "Batch size should be even when using this"
        lam, use_cutmix = self._params_per_batch()
        box = None
        if lam != 1. and use_cutmix:
            H, W = x.shape[-2:]
            ratio = np.sqrt(1 - lam)
            ch, cw = int(H * ratio), int(W * ratio)
            cy, cx = np.random.randint(0, H), np.random.randint(0, W)
            yl, yh = np.clip(cy - ch // 2, 0, H), np.clip(cy + ch // 2, 0, H)
            xl, xh = np.clip(cx - cw // 2, 0, W), np.clip(cx + cw // 2, 0, W)
            box = (int(yl), int(yh), int(xl), int(xh))
            if self.correct_lam:
                lam = 1. - (yh - yl) * (xh - xl) / float(H * W)
        if x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 4 == 0:
            # batch already resident in HBM: one fused kernel + one soft-target kernel; the
            # lambda / box draws stay on the host's numpy RNG exactly as in timm
            from . import ops
            if lam != 1.:
                p = self.patch_size
                if self.inplace:
                    x = ops.mixup_(x, float(lam), box)
                elif p and p % 4 == 0 and x.shape[-2] % p == 0 and x.shape[-1] % p == 0 and not os.environ.get("DKD_NO_SHARED_PATCHES"):
                    from . import vit
                    x, patches = ops.mixup_with_patches(x, float(lam), box, p)
                    vit.register_patches(x, p, patches)
                else:
                    x = ops.mixup(x, float(lam), box)
            tgt = target.to(device=x.device, dtype=torch.int64).contiguous()
            return x, ops.mixup_targets(tgt, self.num_classes, float(lam), self.label_smoothing)
        if lam != 1.:                       # host tensors (the reference mixes before the H2D copy): plain torch
            if box is not None:
                yl, yh, xl, xh = box
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
            else:
                x_flipped = x.flip(0).mul_(1. - lam)
                x.mul_(lam).add_(x_flipped)
        return x, mixup_target(target, self.num_classes, lam, self.label_smoothing)


class ModelEma:
    """timm.utils.ModelEma's contract (``ModelEma(model, decay)``, ``.update(model)``, ``.ema`` holding the averaged weights) on flat
    storage: when the student's parameters live in FusedAdamW's flat buffers the whole update is one kernel per buffer."""

    def __init__(self, model, decay=0.9999, device=None, resume="", optimizer=None):
        import copy
        inner = model.module if hasattr(model, "module") else model
        self.decay = decay
        self.ema = copy.deepcopy(inner).eval()
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._shadows = []
        for m in self.ema.modules():              # the copy must not share (or trust) the student's bf16 weight shadows
            if hasattr(m, "_shadow"):
                m._shadow = type(m._shadow)()
                self._shadows.append(m._shadow)
        self._flat = None
        flats = getattr(optimizer, "_flat", None)
        if flats:
            # mirror the optimizer's flat layout so that update() is a single launch per weight-decay group
            self._flat = []
            where = optimizer._where
            own = dict(inner.named_parameters())
            ema_named = dict(self.ema.named_parameters())
            for f in flats:
                self._flat.append(None if f is None else (f["p"], f["p"].clone()))
            for name, p in own.items():
                w = where.get(id(p))
                if w is not None and name in ema_named:
                    i, s, e = w
                    ema_named[name].data = self._flat[i][1][s:s + p.numel()].view(p.shape)

    @torch.no_grad()
    def update(self, model):
        if self._flat is not None:
            from . import ops
            for pair in self._flat:
                if pair is not None:
                    ops.ema_update(pair[1], pair[0], self.decay)
            for sh in self._shadows:               # raw kernel wrote the masters: their bf16 copies are stale
                sh.optimizer_stepped(bf16_fresh=False)
            return
        inner = model.module if hasattr(model, "module") else model
        for (k, ev), (_, mv) in zip(self.ema.state_dict().items(), inner.state_dict().items()):
            if ev.dtype.is_floating_point:
                ev.mul_(self.decay).add_(mv.detach(), alpha=1. - self.decay)
