"""Is the host ahead of the GPU?  Per step: the host time at which optimizer.step() RETURNS (no synchronisation) against the time the
GPU reaches the same point (an event recorded there).  lag = gpu_done - host_returned: a lag of several milliseconds means the launch
queue is full of work (the host is ahead and launch overhead is hidden); a lag near zero means the GPU waits for the host.

usage: python tools_dev/host_ahead.py [config] [steps] [batch]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from deltakd_amd.engine import train_one_epoch
from deltakd_amd.losses import DistillationLoss, call_base_loss
from deltakd_amd.models import load_teacher_student_model
from deltakd_amd.optim import create_optimizer
from deltakd_amd.shims import Mixup, NativeScaler

config = sys.argv[1] if len(sys.argv) > 1 else "lrkd"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[config]
args = bench.make_args(cfg, batch)
args.rank = 0
torch.manual_seed(42)
np.random.seed(42)
teacher, student = load_teacher_student_model(cfg["teacher"], cfg["student"], args.drop_path_rate, args)
student.to(dev)
teacher.to(dev)
opt = create_optimizer(args, student)
prio = int(os.environ.get("DKD_TEACHER_PRIO", "0"))          # (A/B: -1 = high-priority hardware queue for the teacher stream)
crit = DistillationLoss(call_base_loss(args), teacher, cfg["distillation_type"], args.alpha, args.tau,
                        teacher_stream=torch.cuda.Stream(priority=prio))
mix = Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, prob=args.mixup_prob, switch_prob=args.mixup_switch_prob,
            label_smoothing=args.smoothing, num_classes=1000)
pool = [(torch.randn(batch, 3, 224, 224, device=dev), torch.randint(0, 1000, (batch,), device=dev)) for _ in range(4)]


def loader(n):
    return [(pool[i % 4][0].clone(), pool[i % 4][1]) for i in range(n)]


marks = []
orig_step = opt.step


def step(*a, **k):
    r = orig_step(*a, **k)
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((time.perf_counter(), ev))
    if os.environ.get("DKD_SYNC_EACH_STEP"):      # A/B: what a per-step host synchronisation costs
        torch.cuda.current_stream().synchronize()
    return r


opt.step = step
train_one_epoch(student, teacher, loader(5), crit, opt, NativeScaler(), None, mix, None, dev, 0, args)
torch.cuda.synchronize()
marks.clear()
t_ref = torch.cuda.Event(enable_timing=True)
t_ref.record()
torch.cuda.synchronize()
if os.environ.get("DKD_SYNC_DEBUG"):
    torch.cuda.set_sync_debug_mode("warn")       # torch-level synchronising calls (item(), pageable copies, ...) print a warning
h0 = time.perf_counter()
train_one_epoch(student, teacher, loader(steps), crit, opt, NativeScaler(), None, mix, None, dev, 0, args)
h_end = time.perf_counter()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
g_end = time.perf_counter()
host = np.array([m[0] - h0 for m in marks]) * 1e3
gpu = np.array([t_ref.elapsed_time(m[1]) for m in marks])
lag = gpu - host
print(f"{config} bs {batch}: host returned from the epoch at {(h_end - h0) * 1e3:.1f} ms, GPU drained at {(g_end - h0) * 1e3:.1f} ms")
print(f"mean gpu ms per step (steps 2..n-1): {np.diff(gpu)[1:-1].mean():.3f}")
print("host ms between optimizer steps:", np.round(np.diff(host), 2).tolist())
print("gpu  ms between optimizer steps:", np.round(np.diff(gpu), 2).tolist())
print("lag (gpu - host) ms per step:   ", np.round(lag, 2).tolist())
