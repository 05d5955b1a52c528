"""The data-parallel wrapper on 2 CPU ranks (gloo): the gradient-averaging contract of SURVEY.md section 8(e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class FlatOpt:
    """stand-in with FusedAdamW's hooks (grad_sync / flat_grads) on CPU tensors"""

    def __init__(self, n):
        self.flat = torch.zeros(n)
        self.grad_sync = None

    @property
    def flat_grads(self):
        return [self.flat]

    def step(self):
        self.grad_sync(self.flat_grads)


class RangedOpt(FlatOpt):
    """adds FusedAdamW.grad_ranges(): parameter i of block b lives at [64 * (2b + i), +64) of the single flat buffer"""

    def __init__(self, blocks):
        super().__init__(64 * (2 * len(blocks) + 4))
        self.where = {id(p): (2 * b + i) * 64 for b, blk in enumerate(blocks) for i, p in enumerate(blk.parameters())}

    def grad_ranges(self, params):
        offs = [self.where[id(p)] for p in params if id(p) in self.where]
        return {0: (min(offs), max(offs) + 64)} if offs else {}


class ClipOpt(FlatOpt):
    """FlatOpt + the REAL FusedAdamW.sync_grads / clip_grad_norm_ (they only touch grad_sync, _synced and flat_grads)"""
    _synced = False

    def step(self):
        self.sync_grads()
        self._synced = False


def _bind_fused_methods():
    from deltakd_amd.optim import FusedAdamW
    ClipOpt.sync_grads = FusedAdamW.sync_grads
    ClipOpt.clip_grad_norm_ = FusedAdamW.clip_grad_norm_


class Blocky(nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([nn.Linear(4, 4) for _ in range(8)])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deltakd_amd.ddp import DataParallel
    torch.manual_seed(100 + rank)                 # different init per rank: the wrapper must broadcast rank 0's
    net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4))
    opt = FlatOpt(1000)
    dp = DataParallel(net, opt, bucket_bytes=256)
    w0 = [p.detach().clone() for p in net.parameters()]
    x = torch.full((2, 8), float(rank + 1))
    dp(x).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    dp.sync_gradients()
    synced = [p.grad.clone() for p in net.parameters()]
    opt.flat.copy_(torch.arange(1000.) * (rank + 1))
    opt.step()                                    # flat path: buckets of 64 floats
    # overlapped path: buckets reduced from the block-backward callback (11 -> 0 order), the rest at step()
    net2 = Blocky()
    opt2 = RangedOpt(net2.blocks)
    dp2 = DataParallel(net2, opt2, bucket_bytes=128)
    assert dp2._plan is not None and sorted(dp2._plan) == [0, 4]
    opt2.flat.copy_(torch.arange(opt2.flat.numel(), dtype=torch.float32) * (rank + 1))
    for idx in reversed(range(8)):
        net2._grad_ready_hook(idx)
    partial = opt2.flat.clone()                   # block ranges are reduced, the 4 trailing segments are not yet
    opt2.step()
    overlap = (partial.tolist(), opt2.flat.tolist())
    # clipping under data parallel: NativeScaler must average the gradients BEFORE it takes the norm, so that
    # every rank scales by the same factor: result == clip(mean(g)), identical on both ranks
    from deltakd_amd.shims import NativeScaler
    _bind_fused_methods()
    net3 = nn.Linear(2, 2)
    opt3 = ClipOpt(300)
    DataParallel(net3, opt3, bucket_bytes=256)
    opt3.flat.copy_(torch.linspace(-1, 1, 300) * (3.0 if rank == 0 else 1.0))       # rank-dependent gradients, mean = 2 x linspace
    dummy = (net3.weight * 0).sum()
    NativeScaler()(dummy, opt3, clip_grad=0.5, parameters=net3.parameters())
    clipped = opt3.flat.tolist()
    gathered = [None] * world
    overlap = overlap + (clipped,)
    as_lists = lambda ts: [t.tolist() for t in ts]      # plain lists: no shared-memory handles through the queue
    dist.all_gather_object(gathered, (as_lists(w0), as_lists(local), as_lists(synced), opt.flat.tolist(), overlap))
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_averaging():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=100)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    T = torch.tensor
    (w0a, la, sa, fa, ova), (w0b, lb, sb, fb, ovb) = res
    n = len(ova[1])
    want = torch.arange(n, dtype=torch.float32) * 1.5
    assert torch.allclose(T(ova[1]), want) and ova[1] == ovb[1]                       # everything averaged after step()
    assert torch.allclose(T(ova[0])[:1024], want[:1024])                              # 8 blocks x 2 x 64 reduced during backward
    assert torch.allclose(T(ova[0])[1024:], torch.arange(1024, n, dtype=torch.float32))   # the tail only at step()
    for a, b in zip(w0a, w0b):
        assert torch.equal(T(a), T(b))            # parameters broadcast from rank 0
    for ga, gb, xa, xb in zip(la, lb, sa, sb):
        assert torch.allclose(T(xa), (T(ga) + T(gb)) / 2) and torch.equal(T(xa), T(xb))
    assert torch.allclose(T(fa), torch.arange(1000.) * 1.5) and fa == fb
    mean = torch.linspace(-1, 1, 300) * 2.0
    want_clip = mean * (0.5 / (mean.norm() + 1e-6))
    assert ova[2] == ovb[2], "ranks hold different gradients after clipping"
    assert torch.allclose(T(ova[2]), want_clip, atol=1e-6)


def test_forced_world_of_one_issues_every_collective():
    """`DataParallel(force=True)` in a world of ONE rank (what the GPU suite drives through the "nccl" backend): the wrapper is
    active, plans its buckets, counts its all-reduces, and leaves the gradients unchanged (mean over one rank)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from deltakd_amd.ddp import DataParallel
        net = Blocky()
        opt = RangedOpt(net.blocks)
        idle = DataParallel(Blocky(), RangedOpt(net.blocks))
        assert not idle.active and idle._plan is None
        dp = DataParallel(net, opt, bucket_bytes=128, force=True)
        assert dp.active and sorted(dp._plan) == [0, 4]
        want = torch.arange(opt.flat.numel(), dtype=torch.float32)
        opt.flat.copy_(want)
        for idx in reversed(range(8)):
            net._grad_ready_hook(idx)
        assert dp.collectives == 2
        opt.step()
        assert dp.collectives > 2 and dp.bytes_reduced == 4 * opt.flat.numel()
        assert torch.equal(opt.flat, want)
    finally:
        dist.destroy_process_group()


def test_bench_launches_its_own_ranks_or_names_the_missing_devices():
    """`python bench.py --gpus N` without a launcher: with fewer than N GPUs it must stop before any rank
    starts and NAME the missing devices (not assert on WORLD_SIZE).  No GPU here, so `--gpus 2` must name cuda:0 and cuda:1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: the launch itself is exercised by the driver")
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    have = torch.cuda.device_count()
    assert f"cuda:{have}" in r.stderr and "cuda:1" in r.stderr and "WORLD_SIZE" not in r.stderr, r.stderr[-500:]
