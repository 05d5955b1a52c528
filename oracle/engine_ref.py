"""Pure-torch CPU restatement of the reference's step loop.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

Follows /root/reference/tools/engine.py: ``train_one_epoch`` :8-76 and ``validate`` :78-104.  That file cannot be imported
(duplicated ``def`` line :8-9 -> IndentationError; duplicated blocks :25-34, :36-45, :60-66, :90-91), so the loop below is its
DE-DUPLICATED reading, statement for statement, each statement citing the surviving reference line.
PARITY UNPINNED against an execution of the reference loop (it cannot execute); pinned piecewise: the criterion it calls is
oracle/loss_ref.py (pinned bit-exactly against the reference's own model/loss.py by oracle/gen_golden.py), the meters restate
/root/reference/logs/logger.py:27-63 (``global_avg = total / count``), and the timm pieces (accuracy, Mixup mode='batch',
NativeScaler without AMP = backward + optional clip + step) restate timm==0.9.12 [3P, absent from the image].

Every random draw is an input: DropPath keep masks per step (``keep_per_step``), the criterion's draws per step
(``draws_per_step``), Mixup's lambda / box come from numpy's global RNG exactly as in timm (seed it before the call).
"""
import numpy as np
import torch

from .loss_ref import forward_with_features_ref


def accuracy_ref(output, target, topk=(1,)):
    """timm.utils.accuracy [3P]: top-k hits in percent of the batch, 0-dim tensors."""
    maxk = min(max(topk), output.size(1))
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / target.size(0) for k in topk]


def _one_hot(x, num_classes, on_value, off_value):
    return torch.full((x.size(0), num_classes), off_value).scatter_(1, x.long().view(-1, 1), on_value)


class MixupRef:
    """timm.data.Mixup [3P], mode='batch', correct_lam=True, as tools/train.py:288-295 of the reference configures it."""

    def __init__(self, mixup_alpha=1., cutmix_alpha=0., prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000):
        self.mixup_alpha, self.cutmix_alpha, self.mix_prob, self.switch_prob = mixup_alpha, cutmix_alpha, prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes

    def __call__(self, x, target):
        assert len(x) % 2 == 0
        lam, use_cutmix = 1., False
        if np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0. and self.cutmix_alpha > 0.:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            else:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            lam = float(lam_mix)
        if lam != 1.:
            if use_cutmix:
                H, W = x.shape[-2:]
                ratio = np.sqrt(1 - lam)
                cut_h, cut_w = int(H * ratio), int(W * ratio)
                cy, cx = np.random.randint(0, H), np.random.randint(0, W)
                yl, yh = np.clip(cy - cut_h // 2, 0, H), np.clip(cy + cut_h // 2, 0, H)
                xl, xh = np.clip(cx - cut_w // 2, 0, W), np.clip(cx + cut_w // 2, 0, W)
                lam = 1. - (yh - yl) * (xh - xl) / float(H * W)
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
            else:
                x_flipped = x.flip(0).mul_(1. - lam)
                x.mul_(lam).add_(x_flipped)
        off = self.label_smoothing / self.num_classes
        on = 1. - self.label_smoothing + off
        y1 = _one_hot(target, self.num_classes, on, off)
        y2 = _one_hot(target.flip(0), self.num_classes, on, off)
        return x, y1 * lam + y2 * (1. - lam)


class _Meter:
    """logs/logger.py:27-63: running total / count (the only statistic train_one_epoch returns)."""

    def __init__(self):
        self.total, self.count = 0.0, 0

    def update(self, v):
        self.total += float(v)
        self.count += 1

    @property
    def global_avg(self):
        return self.total / self.count


def train_one_epoch_ref(student_model, teacher_model, train_loader, criterion, optimizer, clip_grad, mixup_fn, epoch, args,
                        keep_per_step=None, draws_per_step=None):
    """-> (dict of global averages as tools/engine.py:76 returns it, list of per-step (loss, acc1, acc5))."""
    student_model.train()                                                     # :10
    teacher_model.eval()                                                      # :11
    meters = {k: _Meter() for k in ("train_loss", "train_acc1", "train_acc5", "train_lr")}   # :12
    per_step = []
    for step, (samples, targets) in enumerate(train_loader):                  # :15
        if mixup_fn is not None:
            original_targets = targets                                        # :17
            samples, targets = mixup_fn(samples, targets)                     # :18
        if keep_per_step is not None:
            student_model.set_droppath_keep(keep_per_step[step])
        if args.distillation_type.lower() in ['soft', 'hard']:                # :36-38
            student_logits, student_feats = student_model(samples), None
        else:                                                                 # :39-40
            student_logits, student_feats = forward_with_features_ref(student_model, samples)
        args.current_epoch = epoch                                            # :47
        draws = draws_per_step[step] if draws_per_step is not None else None
        loss = criterion(samples, student_logits, student_model, student_feats, targets, args, draws)   # :48
        if not isinstance(student_logits, torch.Tensor):                      # :50-51
            student_logits, _ = student_logits
        if mixup_fn is not None:                                              # :53-56
            acc1, acc5 = accuracy_ref(student_logits, original_targets, topk=(1, 5))
        else:
            acc1, acc5 = accuracy_ref(student_logits, targets, topk=(1, 5))
        optimizer.zero_grad()                                                 # :58
        loss.backward()                                                       # :61-62  timm NativeScaler without AMP [3P]:
        if clip_grad is not None:                                             #          backward, clip_grad_norm_, step
            torch.nn.utils.clip_grad_norm_([p for p in student_model.parameters() if p.grad is not None], clip_grad)
        optimizer.step()
        meters["train_loss"].update(loss.item())                              # :71-74
        meters["train_acc1"].update(acc1.item())
        meters["train_acc5"].update(acc5.item())
        meters["train_lr"].update(optimizer.param_groups[0]['lr'])
        per_step.append((loss.item(), acc1.item(), acc5.item()))
    return {k: m.global_avg for k, m in meters.items()}, per_step            # :76


@torch.no_grad()
def validate_ref(student_model, val_loader, args=None):
    """tools/engine.py:78-104: eval-mode student, plain CrossEntropyLoss, top-1 / top-5; global averages over the batches."""
    criterion = torch.nn.CrossEntropyLoss()                                   # :80
    student_model.eval()                                                      # :81
    meters = {k: _Meter() for k in ("val_loss", "val_acc1", "val_acc5")}
    for samples, targets in val_loader:                                       # :85
        student_logits = student_model(samples)                               # :90
        if not isinstance(student_logits, torch.Tensor):                      # :93-94
            student_logits, _ = student_logits
        loss = criterion(student_logits, targets)                             # :96
        acc1, acc5 = accuracy_ref(student_logits, targets, topk=(1, 5))       # :98
        meters["val_loss"].update(loss.item())                                # :100-102
        meters["val_acc1"].update(acc1.item())
        meters["val_acc5"].update(acc5.item())
    return {k: m.global_avg for k, m in meters.items()}                       # :104
