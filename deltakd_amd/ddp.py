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
    def __init__(self, module, optimizer=None, bucket_bytes=8 << 20, overlap=True, process_group=None, force=False):
        """``force``: issue every collective even in a world of ONE rank (broadcast, bucket all-reduces on the comm stream, tail
        sync).  A one-GPU box can then drive the whole path through the "nccl" backend (= RCCL); results equal the unwrapped model's."""
        super().__init__()
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.bucket_bytes = bucket_bytes
        self._opt = optimizer
        self._comm_stream = None
        self._pending = []
        if self.active:
            with torch.no_grad():                      # rank 0's parameters win (DDP constructor semantics)
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
            shadow = getattr(module, "_shadow", None)
            if shadow is not None:
                shadow.optimizer_stepped(bf16_fresh=False)
        self._done = {}
        self._plan = None
        self._avg_in_collective = None
        self.collectives = 0          # all-reduce calls issued / bytes they carried (bench.py reports them for N > 1)
        self.bytes_reduced = 0
        if optimizer is not None and hasattr(optimizer, "grad_sync"):
            optimizer.grad_sync = self._sync_flat
        self.overlap = overlap and self.active and optimizer is not None and hasattr(optimizer, "flat_grads")
        if self.overlap:
            self._plan = self._plan_overlap()
            if self._plan:
                module._grad_ready_hook = self._on_block_backward
                module._wgrad_group = 4         # = blocks per bucket: the deferred weight gradients of a bucket's blocks go out together

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

    def _reduce(self, t):
        # RCCL averages inside the collective (ncclAvg): no division launch behind every bucket; gloo has no
        # AVG -- the CPU / rehearsal path keeps sum + divide
        if self._avg_in_collective is None:
            self._avg_in_collective = dist.get_backend(self.group) == "nccl"
        if self._avg_in_collective:
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(t, group=self.group)
            t.div_(self.world)
        self.collectives += 1
        self.bytes_reduced += t.numel() * t.element_size()

    @property
    def n_buckets(self):
        """Buckets reduced from the block-backward callback (overlapped); the rest goes out in the tail sync."""
        return len(self._plan) if self._plan else 0

    def _plan_overlap(self, blocks_per_bucket=4):
        """Buckets of consecutive transformer blocks, as index ranges of the optimizer's flat gradient buffers.  A bucket is
        reduced from the block-backward callback as soon as its lowest block has finished (backward walks blocks 11 -> 0)."""
        blocks = getattr(self.module, "blocks", None)
        if blocks is None or not hasattr(self._opt, "grad_ranges"):
            return None
        plan = {}
        depth = len(blocks)
        for lo in range(0, depth, blocks_per_bucket):
            params = [p for b in list(blocks)[lo:lo + blocks_per_bucket] for p in b.parameters() if p.requires_grad]
            rng = self._opt.grad_ranges(params)
            # the "smallest contiguous range" must hold these blocks' parameters and nothing else: a foreign parameter inside it
            # (e.g. an aux module registered between blocks) would be reduced before its gradient is final and then skipped by the
            # tail sync.  If that ever happens, leave the bucket to the tail sync.
            covered = {}
            for p in params:
                for i, (s, e) in self._opt.grad_ranges([p]).items():
                    covered[i] = covered.get(i, 0) + (e - s)
            if any(covered.get(i, 0) != e - s for i, (s, e) in rng.items()):
                continue
            plan[lo] = rng
        return plan

    def _on_block_backward(self, idx):
        rng = self._plan.get(idx)
        if not rng:
            return
        flats = self._opt.flat_grads
        cur = torch.cuda.current_stream() if flats[0].is_cuda else None
        if cur is not None:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(cur)          # the gradient kernels of blocks >= idx are enqueued on `cur` ...
            from .vit import wgrad_stream_of
            side = wgrad_stream_of(self.module)
            if side is not None:
                self._comm_stream.wait_stream(side)     # ... and their weight gradients on the model's wgrad stream (deltakd_amd.vit)
            with torch.cuda.stream(self._comm_stream):
                for i, (s, e) in rng.items():
                    self._reduce(flats[i][s:e])
        else:
            for i, (s, e) in rng.items():
                self._reduce(flats[i][s:e])
        for i, se in rng.items():
            self._done.setdefault(i, []).append(se)

    def _sync_flat(self, flat_grads):
        """Called by FusedAdamW.step() before the update: reduce whatever the backward callbacks have not reduced yet
        (embedding, head, aux modules -- or everything when overlap is off) and join the comm stream."""
        if not self.active:
            return
        cur = torch.cuda.current_stream() if flat_grads[0].is_cuda else None
        todo = []
        for i, fg in enumerate(flat_grads):
            pos = 0
            for s, e in sorted(self._done.get(i, [])):
                if s > pos:
                    todo.append(fg[pos:s])
                pos = max(pos, e)
            if pos < fg.numel():
                todo.append(fg[pos:])
        self._done = {}
        if cur is not None:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(cur)
            with torch.cuda.stream(self._comm_stream):
                for t in todo:
                    for b in self._buckets(t):
                        self._reduce(b)
            cur.wait_stream(self._comm_stream)
        else:
            for t in todo:
                for b in self._buckets(t):
                    self._reduce(b)

    def sync_gradients(self):
        """For optimizers without flat storage: coalesce ``p.grad`` into buckets, all-reduce, scatter back."""
        if not self.active:
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
