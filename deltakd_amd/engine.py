"""The step loop -- drop-in for /root/reference/tools/engine.py (``train_one_epoch`` :8-76, ``validate`` :78-104).

The reference file does not compile (duplicated lines); this follows its de-duplicated
semantics, statement for statement.  Differences, all host-side:
  * meters accumulate 0-dim device tensors: no ``.item()`` host sync per step (deltakd_amd.logger);
  * ``forward_with_features`` works through a data-parallel wrapper.
The loop itself is model-agnostic: any nn.Module student/teacher and any criterion with the reference's call contract work.
"""
import os

import torch

from .logger import MetricLogger
from .models import forward_with_features
from .shims import accuracy


def train_one_epoch(student_model, teacher_model, train_loader, criterion, optimizer, loss_scaler, clip_grad, mixup_fn, model_ema,
                    device, epoch, args):
    student_model.train()
    teacher_model.eval()
    metric_logger = MetricLogger()
    header = f'Epoch: [{epoch+1}/{args.epochs}]'
    batches = iter(metric_logger.log_every(train_loader, getattr(args, "print_freq", 10), header, getattr(args, "rank", 0)))

    def fetch():
        try:
            samples, targets = next(batches)
        except StopIteration:
            return None
        original_targets = None
        if mixup_fn is not None:
            original_targets = targets.to(device, non_blocking=True)
            samples, targets = mixup_fn(samples, targets)
        return samples.to(device, non_blocking=True), targets.to(device, non_blocking=True), original_targets

    # One batch of lookahead, so that a criterion with a ``prefetch`` hook (deltakd_amd.losses.DistillationLoss) can start the
    # frozen teacher on batch t+1 while the student's backward of batch t runs.  Batches, mixup draws (numpy RNG) and the
    # student's torch RNG draws keep their order, so the step computes what the statement-for-statement loop computes.
    prefetch = getattr(criterion, "prefetch", None)
    if args.distillation_type.lower() == "none" or os.environ.get("DKD_NO_LOOKAHEAD"):
        prefetch = None                  # nothing to start early: keep the plain order (next batch fetched after the step)
    args.current_epoch = epoch
    nxt = fetch()
    if nxt is not None and prefetch is not None:
        prefetch(nxt[0], args)
    while nxt is not None:
        samples, targets, original_targets = nxt

        # --amp only ever wrapped the student forward in the reference (tools/engine.py:23-34); the HIP path already computes
        # in bf16 with fp32 accumulation, so the flag changes nothing here.
        if args.distillation_type.lower() in ['soft', 'hard']:
            student_logits = student_model(samples)
            student_feats = None
        else:
            student_logits, student_feats = forward_with_features(student_model, samples)

        loss = criterion(samples, student_logits, student_model, student_feats, targets, args)

        if prefetch is not None:
            nxt = fetch()
            if nxt is not None:
                prefetch(nxt[0], args)

        if not isinstance(student_logits, torch.Tensor):
            student_logits, _ = student_logits
        if mixup_fn is not None:
            acc1, acc5 = accuracy(student_logits, original_targets, topk=(1, 5))
        else:
            acc1, acc5 = accuracy(student_logits, targets, topk=(1, 5))

        optimizer.zero_grad()
        is_second_order = hasattr(optimizer, 'is_second_order') and optimizer.is_second_order
        loss_scaler(loss, optimizer, clip_grad=clip_grad, parameters=student_model.parameters(), create_graph=is_second_order)

        if model_ema is not None:
            model_ema.update(student_model)

        metric_logger.update(train_loss=loss.detach())
        metric_logger.update(train_acc1=acc1.detach())
        metric_logger.update(train_acc5=acc5.detach())
        metric_logger.update(train_lr=optimizer.param_groups[0]['lr'])
        if prefetch is None:
            nxt = fetch()
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def validate(student_model, val_loader, device, args):
    criterion = torch.nn.CrossEntropyLoss()
    student_model.eval()
    metric_logger = MetricLogger()
    for samples, targets in metric_logger.log_every(val_loader, 10, 'Val:', getattr(args, "rank", 0)):
        samples = samples.to(device, non_blocking=True)
        targets = targets.to(device, non_blocking=True)
        student_logits = student_model(samples)
        if not isinstance(student_logits, torch.Tensor):
            student_logits, _ = student_logits
        loss = criterion(student_logits, targets)
        acc1, acc5 = accuracy(student_logits, targets, topk=(1, 5))
        metric_logger.update(val_loss=loss.detach())
        metric_logger.update(val_acc1=acc1.detach())
        metric_logger.update(val_acc5=acc5.detach())
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}
