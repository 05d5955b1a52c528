"""dev: which Python lines issue the ATen fill / copy launches of a training step?  Runs a few steps of bench.py's loop under torch.profiler
(with_stack) and prints, per ATen op that launches a device kernel, the innermost repo frame and the call count per step.
usage: python tools_dev/aten_glue_trace.py [config=lrkd] [steps=3]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "lrkd"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS[name]
args = bench.make_args(cfg, 256)
dev = torch.device("cuda", 0)
torch.manual_seed(42); np.random.seed(42)
teacher, student = load_teacher_student_model(cfg["teacher"], cfg["student"], args.drop_path_rate, args)
student.to(dev); teacher.to(dev)
opt = create_optimizer(args, student)
crit = DistillationLoss(call_base_loss(args), teacher, cfg["distillation_type"], args.alpha, args.tau, teacher_stream=torch.cuda.Stream())
mix = Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000)
x = torch.randn(256, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (256,), device=dev)


def run(n):
    train_one_epoch(student, teacher, [(x.clone(), y) for _ in range(n)], crit, opt, NativeScaler(), None, mix, None, dev, 0, args)


run(3)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run(steps)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    if ev.name not in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::zeros", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::stack",
                       "aten::mul", "aten::div", "aten::add", "aten::sub", "aten::lt", "aten::rand", "aten::uniform_", "aten::amax", "aten::abs", "aten::sum"):
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                                  # count outermost ATen calls only
    frame = next((f for f in (ev.stack or []) if ROOT in f and "aten_glue_trace" not in f), (ev.stack or ["?"])[0] if ev.stack else "?")
    sites[(ev.name, frame.replace(ROOT + "/", ""))] += 1
print(f"config {name}: outermost ATen calls per step (over {steps} steps), by innermost repo frame")
for (op, frame), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"{n / steps:7.1f}  {op:18s} {frame}")
