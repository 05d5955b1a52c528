"""dev: RCCL smoke on one GPU (world of 1): the collective calls deltakd_amd.ddp issues -- broadcast of parameters, all-reduce of
slices of a flat buffer on a side stream -- go through the "nccl" backend without error."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
flat = torch.arange(6_000_000, device="cuda", dtype=torch.float32)
ref = flat.clone()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for s, e in ((0, 2_000_000), (2_000_000, 4_100_000), (4_100_000, 6_000_000)):
        dist.all_reduce(flat[s:e])
        flat[s:e].div_(1)
torch.cuda.current_stream().wait_stream(side)
dist.broadcast(flat, src=0)
torch.cuda.synchronize()
assert torch.equal(flat, ref)
print("rccl ok", dist.get_backend())
dist.destroy_process_group()
