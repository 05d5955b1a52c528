"""dev: 120 lrkd steps (bs 256 by default -- the size at which the teacher's LayerNorm fold and the wide kernels are active --, tiny <- base)
over four repeated synthetic batches: the loss must fall and stay finite.  usage: train_sanity.py [batch]"""
import torch, numpy as np, sys
import os
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import loss_ref
from deltakd_amd.engine import train_one_epoch
from deltakd_amd.losses import DistillationLoss, call_base_loss
from deltakd_amd.models import load_teacher_student_model
from deltakd_amd.optim import create_optimizer
from deltakd_amd.shims import Mixup, NativeScaler
args = loss_ref.default_args(distillation_type="lrkd", dataset="imagenet-1k", lrkd_rank=64, opt="adamw", lr=5e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=None, mixup=0.8, cutmix=1.0, smoothing=0.1)
args.epochs, args.print_freq, args.rank = 1, 10**9, 1
torch.manual_seed(0); np.random.seed(0)
t, s = load_teacher_student_model("deit_base_distilled_patch16_224", "deit_tiny_patch16_224", 0.1, args)
t.to("cuda").eval(); s.to("cuda").train()
opt = create_optimizer(args, s)
crit = DistillationLoss(call_base_loss(args), t, "lrkd", args.alpha, args.tau, teacher_stream=torch.cuda.Stream())
mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, prob=1.0, switch_prob=0.5, label_smoothing=0.1, num_classes=1000)
g = torch.Generator(device="cuda").manual_seed(1)
data = [(torch.randn(BS, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (BS,), device="cuda", generator=g)) for _ in range(4)]
for ep in range(6):
    st = train_one_epoch(s, t, [(x.clone(), y) for x, y in data] * 5, crit, opt, NativeScaler(), None, mix, None, torch.device("cuda"), ep, args)
    print("epoch", ep, {k: round(float(v), 4) for k, v in st.items()})
