"""Data-parallel training over RCCL/xGMI -- counterpart of ``DistributedDataParallel(student)`` at
/root/reference/tools/train.py:307-308 (the reference's only parallelism; teacher replicated and frozen, :309).

One process per GPU (torchrun); the batch is sharded by the sampler; the only hot collective is the per-step
all-reduce(SUM)/world of the trainable gradients.  MI355X-first shape:
  * gradients already live in flat fp32 buffers (deltakd_amd.optim.FusedAdamW), so a "bucket" is a contiguous slice -- no
    gradient copies, no per-parameter autograd hooks;
  * buckets are launched from the block-backward callback in reverse layer order on a dedicated comm stream, so the
    all-reduce of block i overlaps the backward of blocks < i; the optimizer waits on the comm stream before its update;
  * bucket size defaults to 8 MiB: xGMI is point-to-point (7 links x ~153 GB/s), a DeiT-tiny gradient set is 23 MB, so 3
    large messages beat many small ones (ring latency, not bandwidth, is the cost at this size).
Works with any backend torch.distributed offers ("nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class DataParallel(nn.Module):
    def __init__(self, module, optimizer=None, bucket_bytes=8 << 20, overlap=True, process_group=None):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self._opt = optimizer
        self._comm_stream = None
        self._pending = []
        if self.world > 1:
            with torch.no_grad():                      # rank 0's parameters win (DDP constructor semantics)
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
            shadow = getattr(module, "_shadow", None)
            if shadow is not None:
                shadow.optimizer_stepped(bf16_fresh=False)
        if optimizer is not None and hasattr(optimizer, "grad_sync"):
            optimizer.grad_sync = self._sync_flat
        self.overlap = overlap and self.world > 1 and optimizer is not None and hasattr(optimizer, "flat_grads")

    # -- forward: same call contract as the wrapped model; ``with_taps`` lets forward_with_features go through the wrapper
    def forward(self, x, with_taps=False):
        if with_taps:
            return self.module.forward_with_taps(x)
        return self.module(x)

    def no_weight_decay(self):
        return self.module.no_weight_decay() if hasattr(self.module, "no_weight_decay") else set()

    # -- gradient averaging
    def _buckets(self, flat):
        n = max(1, self.bucket_bytes // flat.element_size())
        return [flat[i:i + n] for i in range(0, flat.numel(), n)]

    def _sync_flat(self, flat_grads):
        """Called by FusedAdamW.step() before the update: average the flat gradient buffers across ranks."""
        if self.world == 1:
            return
        cur = torch.cuda.current_stream() if flat_grads[0].is_cuda else None
        if cur is not None and self.overlap:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(cur)          # all gradient kernels enqueued so far
            with torch.cuda.stream(self._comm_stream):
                for fg in flat_grads:
                    for b in reversed(self._buckets(fg)):   # last layers' gradients are ready first
                        dist.all_reduce(b, group=self.group)
                        b.div_(self.world)
            cur.wait_stream(self._comm_stream)
        else:
            for fg in flat_grads:
                for b in self._buckets(fg):
                    dist.all_reduce(b, group=self.group)
                    b.div_(self.world)

    def sync_gradients(self):
        """For optimizers without flat storage: coalesce ``p.grad`` into buckets, all-reduce, scatter back."""
        if self.world == 1:
            return
        grads = [p.grad for p in self.module.parameters() if p.requires_grad and p.grad is not None]
        bucket, size = [], 0
        def flush():
            if not bucket:
                return
            flat = torch.cat([g.reshape(-1) for g in bucket])
            dist.all_reduce(flat, group=self.group)
            flat.div_(self.world)
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        for g in grads:
            bucket.append(g)
            size += g.numel() * g.element_size()
            if size >= self.bucket_bytes:
                flush()
                bucket, size = [], 0
        flush()
