"""dev: who launches the small device kernels (copyBuffer / fill / elementwise) of a training step?  A few steps of bench.py's loop under
torch.profiler; for every CPU op that directly launched a device kernel whose name matches, print the chain of enclosing CPU ops
(autograd node, ATen op ...) with the launch count per step and the kernel time.
usage: python tools_dev/small_kernel_trace.py [config=none] [steps=3]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from deltakd_amd.engine import train_one_epoch
from deltakd_amd.losses import DistillationLoss, call_base_loss
from deltakd_amd.models import load_teacher_student_model
from deltakd_amd.optim import create_optimizer
from deltakd_amd.shims import Mixup, NativeScaler

name = sys.argv[1] if len(sys.argv) > 1 else "none"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS[name]
args = bench.make_args(cfg, 256)
dev = torch.device("cuda", 0)
torch.manual_seed(42); np.random.seed(42)
teacher, student = load_teacher_student_model(cfg["teacher"], cfg["student"], args.drop_path_rate, args)
student.to(dev)
if teacher is not None:
    teacher.to(dev)
opt = create_optimizer(args, student)
crit = DistillationLoss(call_base_loss(args), teacher, cfg["distillation_type"], args.alpha, args.tau, teacher_stream=torch.cuda.Stream())
mix = Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)
batches = [(x.clone(), y) for _ in range(3 + steps)]


def run(bs):
    train_one_epoch(student, teacher, bs, crit, opt, NativeScaler(), None, mix, None, dev, 0, args)


run(batches[:3])
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(batches[3:])
    torch.cuda.synchronize()
KEYS = ("Memcpy", "Memset", "copyBuffer", "fillBuffer", "FillFunctor", "elementwise", "copy_kernel", "multi_tensor", "distribution", "reduce_kernel", "CatArray")
sites = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.kernels:
        continue
    if any(c.kernels for c in ev.cpu_children):
        continue                                   # the innermost CPU op that owns the launch
    for k in ev.kernels:
        if not any(s in k.name for s in KEYS):
            continue
        chain, p = [], ev
        while p is not None and len(chain) < 7:
            chain.append(p.name[:70])
            p = p.cpu_parent
        key = (k.name[:60], " < ".join(chain))
        sites[key][0] += 1
        sites[key][1] += k.duration
print(f"config {name}: small device kernels per step (over {steps} steps)")
tot = 0
for (kn, chain), (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][1]):
    print(f"{n / steps:6.1f}/step {us / steps:8.1f} us/step  {kn}\n        {chain}")
    tot += n
print("total launches per step:", tot / steps)
