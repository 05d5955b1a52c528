"""The step loop -- drop-in for /root/reference/tools/engine.py (``train_one_epoch`` :8-76, ``validate`` :78-104).

The reference file does not compile (duplicated lines); this follows its de-duplicated
semantics, statement for statement.  Differences, all host-side:
  * meters accumulate 0-dim device tensors: no ``.item()`` host sync per step (deltakd_amd.logger);
  * ``forward_with_features`` works through a data-parallel wrapper.
The loop itself is model-agnostic: any nn.Module student/teacher and any criterion with the reference's call contract work.
"""
import collections
import os

import torch

from .logger import MetricLogger
from .models import forward_with_features
from .shims import accuracy


# Blocks whose features this repo's DistillationLoss reads, per distillation type (model/loss.py: lrkd uses student_features[0], [1]
# and [-1]; no feature term for none).  A tap costs a bf16 [M, D] store per block (and its slab), so the loop asks for exactly these;
# other types and foreign criteria get every block, as the reference's forward_with_features returns them.
_STUDENT_TAPS = {"none": (), "lrkd": (0, 1, -1)}


def _narrow_student_taps(student_model, criterion):
    """Narrow the student's taps to what THIS criterion reads (its own ``distillation_type`` -- or a ``student_taps`` attribute a
    subclass may set -- not the args': the two can differ).  Returns (model, previous setting) for the caller to restore, or None."""
    from .losses import DistillationLoss
    inner = student_model
    while not hasattr(inner, "blocks") and hasattr(inner, "module"):
        inner = inner.module
    if not hasattr(inner, "tap_layers") or not isinstance(criterion, DistillationLoss):
        return None
    taps = getattr(criterion, "student_taps", None)
    if taps is None:
        taps = _STUDENT_TAPS.get(str(criterion.distillation_type).lower())
    if taps is None:
        return None
    prev = inner.tap_layers
    inner.tap_layers = tuple(taps)
    return inner, prev


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

    # Lookahead, so that a criterion with a ``prefetch`` hook (deltakd_amd.losses.DistillationLoss) can run the frozen teacher ahead of
    # the student: the next ``prefetch_group`` batches are fetched together and handed to the hook as one group (one teacher call), and
    # a new group is started as soon as the student begins to consume the previous one, so the teacher stream always has a group in
    # flight under the student's work.  Batches, mixup draws (numpy RNG) and the student's torch RNG draws keep their order, so every
    # step computes what the statement-for-statement loop computes.
    prefetch = getattr(criterion, "prefetch", None)
    if args.distillation_type.lower() == "none" or os.environ.get("DKD_NO_LOOKAHEAD"):
        prefetch = None                  # nothing to start early: keep the plain order (next batch fetched after the step)
    group_size = max(1, int(getattr(criterion, "prefetch_group", 1))) if prefetch is not None else 1
    # Groups in flight (DKD_LOOKAHEAD / criterion.prefetch_depth, default 1).  With one, the teacher's work for batch t+1 is queued behind
    # the loss of batch t, which waits for the teacher's batch t: a serial chain teacher -> loss -> teacher.  With two, the work queued
    # behind the loss of batch t is batch t+2's and the teacher stream runs back to back.  Measured at the headline config (round 3,
    # same box, alternating): 18.06 / 18.08 / 18.05 ms per step for 1 / 2 / 3 -- the two streams time-share the CUs, the chain is not
    # what bounds the step -- so the default stays at one group (one batch of taps alive).  Round 4: lrkd asks for TWO
    # (DistillationLoss.prefetch_depth): its low-rank target chain now runs on a stream of its own, and with two groups in flight the
    # next teacher forward runs beside it (16.95 -> 16.87 ms with round 4's one-step tracker, 29.3 -> 19.6 ms with its converged mode).
    # Round 5 (converged targets by default, a 1.4 ms chain): one batch of lookahead costs +1.2 ms, three buy nothing over two
    # (profiles/r05_lrkd_chain_stream_and_lookahead_ab.txt).
    depth = max(1, int(os.environ.get("DKD_LOOKAHEAD", getattr(criterion, "prefetch_depth", 1)))) if prefetch is not None else 1
    args.current_epoch = epoch
    pending = collections.deque()        # fetched batches, in order; with a prefetch hook their teacher work has been started

    def start_group():
        group = []
        while len(group) < group_size:
            b = fetch()
            if b is None:
                break
            group.append(b)
        if group and prefetch is not None:
            prefetch([b[0] for b in group] if group_size > 1 else group[0][0], args)
        pending.extend(group)

    def _run_steps():
        while pending:
            samples, targets, original_targets = pending.popleft()

            # --amp only ever wrapped the student forward in the reference (tools/engine.py:23-34); the HIP path already computes
            # in bf16 with fp32 accumulation, so the flag changes nothing here.
            if args.distillation_type.lower() in ['soft', 'hard']:
                student_logits = student_model(samples)
                student_feats = None
            else:
                student_logits, student_feats = forward_with_features(student_model, samples)

            loss = criterion(samples, student_logits, student_model, student_feats, targets, args)

            if prefetch is not None and len(pending) < depth * group_size:
                start_group()                # between the loss and the backward: the teacher's next group overlaps the student's work

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

            metric_logger.update(train_loss=loss.detach(), train_acc1=acc1.detach(), train_acc5=acc5.detach(),
                                 train_lr=optimizer.param_groups[0]['lr'])       # (one multi-tensor add for the three device meters)
            if prefetch is None:
                start_group()

    narrowed = _narrow_student_taps(student_model, criterion)     # for this epoch only: forward_with_features outside the loop
    try:                                                          # keeps returning every block, as the reference's does
        for _ in range(depth):
            start_group()
        _run_steps()
    finally:
        if narrowed is not None:
            narrowed[0].tap_layers = narrowed[1]
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
