"""One rank of the data-parallel rehearsal driven by tests/test_engine_gpu.py (not a test module itself).

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;  argv: <output file> <clip_grad or 0>

Every rank uses cuda:0 (the GPU boxes of this pool have one card) and the gloo backend; what is exercised is the product's own
data-parallel path on the real HIP model: parameter broadcast, FusedAdamW flat gradient buffers, bucket all-reduces launched from
the block-backward callback on the comm stream, the tail sync, clipping after the sync.  Two steps of the mgd branch on a batch of 8
that is split evenly over the ranks (DropPath keep masks and masking noise are drawn for the whole batch and split alike).
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TOY = dict(img_size=32, patch_size=8, mlp_ratio=2.0)


def main():
    out_path, clip = sys.argv[1], float(sys.argv[2])
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("DKD_DP_BACKEND", "gloo")          # "nccl" (= RCCL): the one-rank smoke of the real collective library
    force = backend == "nccl"                                   # a world of one still issues every collective
    if world > 1 or force:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group("gloo", init_method="env://")
    from oracle import loss_ref                      # default_args only (argument namespace); nothing of the oracle computes here
    from deltakd_amd import vit
    from deltakd_amd.ddp import DataParallel
    from deltakd_amd.engine import train_one_epoch
    from deltakd_amd.losses import DistillationLoss, call_base_loss
    from deltakd_amd.models import attach_aux
    from deltakd_amd.optim import create_optimizer
    from deltakd_amd.shims import NativeScaler

    args = loss_ref.default_args(distillation_type="mgd", dataset="cifar-10", mgd_alpha=2.0, mgd_mask_ratio=0.5, opt="adamw", lr=1e-3,
                                 weight_decay=0.05, opt_eps=1e-8, opt_betas=None, smoothing=0.1, epochs=1, print_freq=1000, rank=1)
    torch.manual_seed(5)
    t = vit.VisionTransformer(128, 12, 2, 10, True, 0.0, **TOY)
    s = vit.VisionTransformer(64, 12, 1, 10, False, 0.1, **TOY)
    attach_aux(s, t, "mgd", args)
    with torch.no_grad():
        for net in (s, t):
            for blk in net.blocks:
                blk.mlp.fc2.weight.mul_(8.0)
        if rank != 0:                                # the wrapper must overwrite this with rank 0's parameters
            s.head.weight.add_(1.0)
            s.blocks[3].attn.qkv.weight.mul_(0.5)
    for p in t.parameters():
        p.requires_grad = False
    t.to(dev).eval()
    s.to(dev).train()
    opt = create_optimizer(args, s)
    model = DataParallel(s, opt, force=force) if (world > 1 or force) else s
    if (world > 1 or force) and rank == 0:
        print("overlap buckets:", sorted(model._plan or {}), flush=True)
    init = {n: p.detach().cpu().clone() for n, p in s.named_parameters()}

    B, steps, depth = 8, 2, 12
    lo, hi = rank * B // world, (rank + 1) * B // world
    g = torch.Generator().manual_seed(99)
    data = [(torch.randn(B, 3, 32, 32, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(steps)]
    keeps = [[(torch.rand(B, generator=g) > 0.15).float()[lo:hi] for _ in range(2 * depth)] for _ in range(steps)]
    noises = [torch.rand(B, 16, generator=g)[lo:hi].to(dev) for _ in range(steps)]
    crit = DistillationLoss(call_base_loss(args), t, "mgd", args.alpha, args.tau, teacher_stream=torch.cuda.Stream())
    crit.injected["noise"] = iter(noises)
    s.set_droppath_keep(iter(keeps))

    losses, norms = [], []

    class Rec:
        prefetch = crit.prefetch

        def __call__(self, *a):
            v = crit(*a)
            losses.append(v.detach())
            return v

    if clip:
        real_clip = opt.clip_grad_norm_

        def clip_and_record(max_norm, norm_type=2.0):
            total = real_clip(max_norm, norm_type)
            norms.append(total)
            return total
        opt.clip_grad_norm_ = clip_and_record
    loader = [(x[lo:hi].contiguous().to(dev), y[lo:hi].contiguous().to(dev)) for x, y in data]
    train_one_epoch(model, t, loader, Rec(), opt, NativeScaler(), clip or None, None, None, dev, 0, args)
    torch.cuda.synchronize()

    lv = torch.stack(losses).float()
    identical = True
    if world > 1:
        dist.all_reduce(lv)
        lv /= world
        for n, p in s.named_parameters():
            ref = p.detach().clone()
            dist.broadcast(ref, src=0)
            identical = identical and bool(torch.equal(ref, p.detach()))
        flag = torch.tensor([1.0 if identical else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        identical = bool(flag.item() == 1.0)
    if rank == 0:
        torch.save({"weights": {n: p.detach().cpu() for n, p in s.named_parameters()}, "init": init, "losses": [float(v) for v in lv],
                    "ranks_identical": identical, "clipped_steps": sum(int(float(n) > clip) for n in norms)}, out_path)
    if force and rank == 0:
        maps = open("/proc/self/maps").read()
        libs = sorted({ln.split()[-1] for ln in maps.splitlines() if "rccl" in ln.lower() or "libnccl" in ln.lower()})
        print("backend:", dist.get_backend(), "| rccl mapped:", bool(libs), libs[:2], flush=True)
        print("allreduce calls:", model.collectives, "bytes:", model.bytes_reduced, "comm stream used:",
              model._comm_stream is not None, flush=True)
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
