"""Host-side logic that needs no GPU: the step loop (with oracle models standing in for the HIP ones: the loop is
model-agnostic), timm-shim semantics, optimizer grouping / schedule, error behaviour, the C ABI surface."""
import math
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import loss_ref, vit_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = dict(img_size=32, patch_size=8, mlp_ratio=2.0)


def toy_pair(kind="lrkd"):
    torch.manual_seed(0)
    t = vit_ref.VisionTransformerRef(128, 12, 2, 10, True, 0.0, **TOY).eval()
    s = vit_ref.VisionTransformerRef(64, 12, 1, 10, kind in ("soft", "hard"), 0.1, **TOY)
    loss_ref.attach_aux_ref(s, t, kind, 16)
    for p in t.parameters():
        p.requires_grad = False
    return s, t


class OracleCriterion(nn.Module):
    """reference call contract criterion(inputs, outputs, student, feats, labels, args) on the CPU oracle"""

    def __init__(self, teacher, kind, args):
        super().__init__()
        self.inner = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), teacher, kind, args.alpha, args.tau)

    def forward(self, inputs, outputs, student, feats, labels, args):
        return self.inner(inputs, outputs, student, feats, labels, args)


@pytest.mark.parametrize("kind,mix", [("none", False), ("none", True), ("lrkd", True), ("soft", True)])
def test_train_one_epoch_cpu_plumbing(kind, mix, monkeypatch):
    """BASELINE config 1 ("plumbing, no GPU"): the de-duplicated step loop of SURVEY Appendix A runs end to end."""
    from deltakd_amd import engine
    from deltakd_amd.shims import Mixup, NativeScaler
    s, t = toy_pair(kind)
    args = loss_ref.default_args(distillation_type=kind, lrkd_rank=16, epochs=1, print_freq=0)
    args.mixup = 0.8 if mix else 0.0
    args.cutmix = 1.0 if mix else 0.0
    # the oracle model has no forward_with_taps: hand the loop the oracle's feature tap
    monkeypatch.setattr(engine, "forward_with_features", loss_ref.forward_with_features_ref)
    g = torch.Generator().manual_seed(1)
    loader = [(torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)) for _ in range(3)]
    opt = torch.optim.AdamW(s.parameters(), lr=1e-3)
    before = s.head.weight.detach().clone()
    np.random.seed(0)
    mixup = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, num_classes=10) if mix else None
    stats = engine.train_one_epoch(s, t, loader, OracleCriterion(t, kind, args), opt, NativeScaler(), None, mixup, None,
                                   torch.device("cpu"), 0, args)
    assert set(stats) == {"train_loss", "train_acc1", "train_acc5", "train_lr"}
    assert all(math.isfinite(v) for v in stats.values()) and stats["train_lr"] == 1e-3
    assert not torch.equal(before, s.head.weight) and args.current_epoch == 0
    assert s.training and not t.training


def test_mixup_shim_semantics():
    from deltakd_amd.shims import Mixup, accuracy, mixup_target
    np.random.seed(3)
    x = torch.arange(4 * 3 * 8 * 8, dtype=torch.float32).view(4, 3, 8, 8)
    y = torch.tensor([0, 1, 2, 3])
    mix = Mixup(mixup_alpha=0.8, cutmix_alpha=0.0, num_classes=5, label_smoothing=0.1)
    x0 = x.clone()
    xm, ym = mix(x, y)
    lam = ((xm - x0.flip(0)) / (x0 - x0.flip(0)))[0, 0, 0, 1].item()        # x = lam x + (1 - lam) flip(x)
    assert 0 < lam < 1 and torch.allclose(xm, lam * x0 + (1 - lam) * x0.flip(0), atol=1e-3)
    assert torch.allclose(ym.sum(1), torch.ones(4)) and torch.allclose(ym, mixup_target(y, 5, lam, 0.1), atol=1e-6)
    assert abs(ym[0, 0].item() - (lam * (0.9 + 0.02) + (1 - lam) * 0.02)) < 1e-6
    cut = Mixup(mixup_alpha=0.0, cutmix_alpha=1.0, num_classes=5)
    x1 = torch.randn(4, 3, 8, 8)
    x1c = x1.clone()
    xc, yc = cut(x1, y)
    changed = (xc != x1c).float().mean().item()
    lam_c = yc[0, 0].item()          # with label_smoothing 0.1: on = 0.92, off = 0.02
    assert abs((1 - changed) - (lam_c - 0.02) / 0.9) < 0.2                   # box area matches the corrected lambda
    with pytest.raises(AssertionError):
        mix(torch.zeros(3, 3, 8, 8), torch.tensor([0, 1, 2]))               # odd batch
    out = torch.tensor([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1]])
    a1, a2 = accuracy(out, torch.tensor([1, 2]), topk=(1, 2))
    assert a1.item() == 50.0 and a2.item() == 50.0


def test_optimizer_groups_and_schedule():
    from deltakd_amd.optim import CosineLRScheduler, param_groups_weight_decay
    s, t = toy_pair("mgd")
    groups = param_groups_weight_decay(s, 0.05, s.no_weight_decay())
    named = dict(s.named_parameters())
    no_decay = {id(p) for p in groups[0]["params"]}
    for n in ("cls_token", "pos_embed", "blocks.0.norm1.weight", "blocks.3.attn.qkv.bias", "align.bias", "generation.0.bias"):
        assert id(named[n]) in no_decay, n
    for n in ("blocks.0.attn.qkv.weight", "patch_embed.proj.weight", "align.weight", "generation.2.weight", "mask_token"):
        assert id(named[n]) not in no_decay, n
    assert groups[0]["weight_decay"] == 0.0 and groups[1]["weight_decay"] == 0.05
    opt = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=5e-4)
    sch = CosineLRScheduler(opt, t_initial=300, lr_min=1e-5, warmup_t=5, warmup_lr_init=1e-6)
    assert opt.param_groups[0]["lr"] == 1e-6                      # timm starts at warmup_lr_init
    sch.step(0)
    assert opt.param_groups[0]["lr"] == 1e-6                      # the reference passes `epoch`, so epoch 1 repeats it
    sch.step(3)
    assert abs(opt.param_groups[0]["lr"] - (1e-6 + 3 * (5e-4 - 1e-6) / 5)) < 1e-12
    sch.step(150)
    assert abs(opt.param_groups[0]["lr"] - (1e-5 + 0.5 * (5e-4 - 1e-5))) < 1e-9
    sch.step(300)
    assert opt.param_groups[0]["lr"] == 1e-5


def test_metric_logger_contract():
    from deltakd_amd.logger import MetricLogger
    ml = MetricLogger()
    for v in (1.0, 2.0, torch.tensor(3.0)):
        ml.update(loss=v)
    assert ml.meters["loss"].global_avg == 2.0 and ml.meters["loss"].count == 3
    seen = list(ml.log_every(range(5), 0, "x", rank=1))
    assert seen == [0, 1, 2, 3, 4]


def test_metric_logger_adds_the_tensor_meters_of_one_call_together():
    """MetricLogger.update(a=, b=, c=) with 0-dim tensors accumulates them in one multi-tensor add (deltakd_amd.logger): totals, counts and
    the window must equal the one-meter-per-call path; python numbers in the same call keep their own path."""
    import torch
    from deltakd_amd.logger import MetricLogger
    a, b = MetricLogger(), MetricLogger()
    g = torch.Generator().manual_seed(0)
    for step in range(7):
        vals = {"loss": torch.rand((), generator=g), "acc1": torch.rand((), generator=g) * 100, "acc5": torch.rand((), generator=g) * 100}
        a.update(lr=1e-3 * step, **vals)
        for k, v in vals.items():
            b.update(**{k: v})
        b.update(lr=1e-3 * step)
    for k in ("loss", "acc1", "acc5", "lr"):
        assert abs(a.meters[k].global_avg - b.meters[k].global_avg) < 1e-6 and a.meters[k].count == b.meters[k].count == 7
        assert abs(a.meters[k].value - b.meters[k].value) < 1e-7
    assert "loss" in str(a)


def test_wgrad_group_size_knob(monkeypatch):
    """deltakd_amd.vit.wgrad_group_size: default 6 blocks per deferred weight-gradient launch, a model attribute (data parallel sets its
    bucket size) or DKD_WGRAD_GROUP override it, clamped to what dkd_block_wgrad_group takes (24 problems = 6 blocks)."""
    from types import SimpleNamespace
    from deltakd_amd import vit
    monkeypatch.delenv("DKD_WGRAD_GROUP", raising=False)
    assert vit.wgrad_group_size(SimpleNamespace()) == 6
    assert vit.wgrad_group_size(SimpleNamespace(_wgrad_group=4)) == 4
    monkeypatch.setenv("DKD_WGRAD_GROUP", "1")
    assert vit.wgrad_group_size(SimpleNamespace(_wgrad_group=4)) == 1
    monkeypatch.setenv("DKD_WGRAD_GROUP", "64")
    assert vit.wgrad_group_size(SimpleNamespace()) == 6


def test_error_behaviour_matches_reference():
    from deltakd_amd.losses import DistillationLoss, LabelSmoothingCrossEntropy
    from deltakd_amd.models import forward_with_features
    z = torch.zeros(2, 10)
    args = loss_ref.default_args()
    with pytest.raises(ValueError, match="Tuple"):       # model/loss.py:39-42
        DistillationLoss(LabelSmoothingCrossEntropy(), nn.Identity(), "soft", 0.1, 3.0)(z, z, None, None, torch.zeros(2).long(), args)
    with pytest.raises(ValueError, match="Invalid distillation type"):   # model/loss.py:238-239 (aaakd is accepted by argparse only)
        DistillationLoss(LabelSmoothingCrossEntropy(), nn.Identity(), "aaakd", 0.1, 3.0)(z, z, None, None, torch.zeros(2).long(), args)
    assert forward_with_features(nn.Linear(2, 2), torch.zeros(1, 2)) == (None, None)      # model/models.py:182-183


def test_product_path_has_no_cpu_fallback():
    from deltakd_amd import ops, vit
    m = vit.VisionTransformer(64, 2, 1, 10, False, 0.0, **TOY)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(2, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.layernorm_fwd(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))
    # and nothing under deltakd_amd/ imports the oracle
    for dp, _, files in os.walk(os.path.join(ROOT, "deltakd_amd")):
        for f in files:
            if f.endswith(".py"):
                assert not re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(dp, f)).read(), re.M), f


def test_state_dict_keys_are_timm_compatible():
    from deltakd_amd import vit
    from deltakd_amd.models import attach_aux
    hip = vit.VisionTransformer(64, 12, 1, 10, True, 0.1, **TOY)
    ref = vit_ref.VisionTransformerRef(64, 12, 1, 10, True, 0.1, **TOY)
    assert set(hip.state_dict()) == set(ref.state_dict())
    assert {k: v.shape for k, v in hip.state_dict().items()} == {k: v.shape for k, v in ref.state_dict().items()}
    for kind in ("lrkd", "mgd", "diffkd", "wasskd", "vitkd", "curkd", "saliency_mgd"):
        a = attach_aux(vit.VisionTransformer(64, 12, 1, 10, False, 0.1, **TOY), hip, kind, SimpleNamespace(lrkd_rank=16, saliency_method=1))
        b = loss_ref.attach_aux_ref(vit_ref.VisionTransformerRef(64, 12, 1, 10, False, 0.1, **TOY), ref, kind, 16)
        assert {k: v.shape for k, v in a.state_dict().items()} == {k: v.shape for k, v in b.state_dict().items()}, kind


def test_c_abi_library_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly the entry points include/dkd.h declares."""
    from deltakd_amd import ffi
    header = open(os.path.join(ROOT, "include", "dkd.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(dkd_\w+)\s*\(", header, re.M))
    assert declared, "no declarations parsed"
    handle = ffi.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in dkd.h but not exported by libdkd.so"
    assert declared == set(ffi.EXPORTS), declared ^ set(ffi.EXPORTS)
    assert handle.dkd_version() >= 100
    # argument errors come back through the error channel, not as crashes (no kernel is launched for a bad call)
    rc = handle.dkd_gemm_nt(None, None)
    assert rc == -1 and b"null" in handle.dkd_last_error()
    # workspace queries are host arithmetic: the headline student block, B = 256, N = 197, D = 192, hidden 768
    M, D, Hd = 256 * 197, 192, 768
    al = lambda n: (n + 255) // 256 * 256
    assert handle.dkd_layernorm_bwd_workspace_bytes(M, D) == al(2 * D * ((M + 63) // 64) * 4)
    assert handle.dkd_block_bwd_workspace_bytes(256, 197, D, Hd) == (3 * al(M * D * 2) + al(M * Hd * 2) + al(M * 3 * D * 2)
                                                                      + 2 * handle.dkd_layernorm_bwd_workspace_bytes(M, D))
    b16, f32 = ffi.C.c_int64(), ffi.C.c_int64()
    tot = handle.dkd_block_fwd_workspace_bytes(256, 197, D, 3, Hd, 1, 1, ffi.C.byref(b16), ffi.C.byref(f32))
    assert tot == b16.value + f32.value and b16.value == 4 * al(M * D * 2) + al(M * 3 * D * 2) + 2 * al(M * Hd * 2)
    assert f32.value == 2 * al(M * D * 4) + 4 * al(M * 4) + al(256 * 3 * 197 * 4)
    gr = ffi.BlockGrads()
    assert handle.dkd_block_bwd_workspace_carve(0x10000, 256, 197, D, Hd, ffi.C.byref(gr)) == 0      # pointer arithmetic only
    assert gr.dF == 0x10000 and gr.dT == gr.dF + al(M * D * 2) and gr.ln_ws2 == gr.dF2 + al(M * D * 2) and \
        gr.ln_ws2 + handle.dkd_layernorm_bwd_workspace_bytes(M, D) - 0x10000 == handle.dkd_block_bwd_workspace_bytes(256, 197, D, Hd)
    # the LRKD chain's scratch (round 5): two 96 x 96 Gram accumulators per layer first (the part that must be zero), two Y buffers, two
    # transforms, the diagnostics; and its argument checks answer through the error channel without launching anything
    L, Dt = 3, 768
    m96 = al(L * 96 * 96 * 4)
    assert handle.dkd_lowrank_chain_zero_bytes(L, Dt) == 2 * m96
    assert handle.dkd_lowrank_chain_workspace_bytes(L, Dt) == 4 * m96 + 2 * al(L * Dt * 96 * 4) + al(L * 8)
    for bad_dt in (96, 160, 4096):        # below the minimum, not a multiple of 64, above the maximum
        assert handle.dkd_lowrank_chain(0x1000, 0x1000, L, bad_dt, 8, 12, 64, None, None, None, 0x1000, None) == -1
        assert b"multiple of 64" in handle.dkd_last_error()
    assert handle.dkd_lowrank_chain(0x1000, 0x1000, L, Dt, 0, 12, 64, None, None, None, 0x1000, None) == -1       # no multiplies
    assert handle.dkd_lowrank_chain(0x1000, 0x1000, L, Dt, 8, 12, 64, None, None, None, 0x1010, None) == -1       # misaligned scratch


def test_checkpoint_helpers_follow_the_reference_wire_format(tmp_path):
    """tools/utils.py save_checkpoint / load_model / enable_finetune_mode (reference tools/utils.py:90-160): dict layout, best copy,
    missing file, and the finetune path (head of another size dropped, pos_embed grid resized bicubically, prefix rows kept)."""
    import torch
    from deltakd_amd import vit
    from tools.utils import enable_finetune_mode, load_model, save_checkpoint
    torch.manual_seed(0)
    src = vit.VisionTransformer(64, 2, 1, 10, False, 0.0, img_size=32, patch_size=8, mlp_ratio=2.0)
    f = str(tmp_path / "ck" / "checkpoint.pth")
    save_checkpoint({"epoch": 3, "model": {"module." + k: v for k, v in src.state_dict().items()}, "optimizer": {}, "scheduler": {},
                     "scaler": {}}, is_best=True, filename=f)
    assert os.path.exists(f) and os.path.exists(f.replace("pth", "best.pth"))
    ck = torch.load(f, weights_only=True)
    assert set(ck) == {"epoch", "model", "optimizer", "scheduler", "scaler"} and ck["epoch"] == 3
    dst = vit.VisionTransformer(64, 2, 1, 10, False, 0.0, img_size=32, patch_size=8, mlp_ratio=2.0)
    load_model(dst, f)                                    # strips the DDP "module." prefix
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(FileNotFoundError):
        load_model(dst, str(tmp_path / "nope.pth"))
    # finetune: 4x4 grid -> 8x8 grid, 10 -> 7 classes
    big = vit.VisionTransformer(64, 2, 1, 7, False, 0.0, img_size=64, patch_size=8, mlp_ratio=2.0)
    head_before = big.head.weight.detach().clone()
    state = {k: v.clone() for k, v in src.state_dict().items()}
    enable_finetune_mode(big, state)
    assert big.pos_embed.shape == (1, 1 + 64, 64)
    assert torch.equal(big.pos_embed[:, :1], src.pos_embed[:, :1])               # cls position kept
    grid = src.pos_embed[:, 1:].reshape(1, 4, 4, 64).permute(0, 3, 1, 2)
    want = torch.nn.functional.interpolate(grid, size=(8, 8), mode="bicubic", align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    assert torch.allclose(big.pos_embed[:, 1:], want)
    assert torch.equal(big.head.weight, head_before)                              # mismatching head left alone
    assert torch.equal(big.blocks[1].mlp.fc1.weight, src.blocks[1].mlp.fc1.weight)


@pytest.mark.parametrize("kind,clip", [("mgd", None), ("soft", 0.5), ("none", None)])
def test_product_loop_equals_the_oracle_loop_on_cpu(kind, clip, monkeypatch):
    """deltakd_amd.engine.train_one_epoch (the product's loop; model-agnostic) and oracle.engine_ref.train_one_epoch_ref (the
    de-duplicated restatement of tools/engine.py:8-76) driven with the SAME oracle models, draws and data must produce identical
    per-epoch statistics and identical post-epoch weights: mixup order, criterion call, accuracy targets, zero_grad / backward /
    clip / step order, meters.  (The GPU test tests/test_engine_gpu.py then swaps the HIP models in.)"""
    import copy
    from deltakd_amd import engine
    from deltakd_amd.shims import Mixup, NativeScaler
    from oracle import engine_ref
    s0, t = toy_pair(kind)
    args = loss_ref.default_args(distillation_type=kind, lrkd_rank=16, epochs=1, print_freq=0, mgd_alpha=2.0, mixup=0.8, cutmix=1.0)
    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)) for _ in range(3)]
    keeps = [[(torch.rand(4, generator=g) > 0.2).float() for _ in range(24)] for _ in range(3)]
    noises = [{"noise": torch.rand(4, 16, generator=g)} for _ in range(3)]
    monkeypatch.setattr(engine, "forward_with_features", loss_ref.forward_with_features_ref)

    class Crit(nn.Module):          # reference call contract; the draws of step i are consumed in order
        def __init__(self, teacher):
            super().__init__()
            self.inner = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), teacher, kind, args.alpha, args.tau)
            self.step = 0

        def forward(self, inputs, outputs, student, feats, labels, a):
            self.step += 1
            return self.inner(inputs, outputs, student, feats, labels, a, noises[self.step - 1])

    class StepKeep:                 # the oracle model takes a list; hand it a fresh one per forward
        def __init__(self, model):
            self.model, self.i = model, 0
            self.handle = model.register_forward_pre_hook(self)

        def __call__(self, module, inp):
            self.model.set_droppath_keep(keeps[self.i])
            self.i += 1

    # product loop
    s_a = copy.deepcopy(s0)
    StepKeep(s_a)
    opt_a = torch.optim.AdamW(s_a.parameters(), lr=1e-3)
    np.random.seed(11)
    mix = Mixup(mixup_alpha=0.8, cutmix_alpha=1.0, num_classes=10)
    stats_a = engine.train_one_epoch(s_a, t, [(x.clone(), y.clone()) for x, y in data], Crit(t), opt_a, NativeScaler(), clip, mix, None,
                                     torch.device("cpu"), 0, args)
    # oracle loop
    s_b = copy.deepcopy(s0)
    opt_b = torch.optim.AdamW(s_b.parameters(), lr=1e-3)
    np.random.seed(11)
    mix_b = engine_ref.MixupRef(mixup_alpha=0.8, cutmix_alpha=1.0, num_classes=10)
    crit_b = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), t, kind, args.alpha, args.tau)
    stats_b, per_step = engine_ref.train_one_epoch_ref(s_b, t, [(x.clone(), y.clone()) for x, y in data], crit_b, opt_b, clip, mix_b, 0, args,
                                                      keep_per_step=keeps, draws_per_step=noises)
    assert len(per_step) == 3
    for k in ("train_loss", "train_acc1", "train_acc5", "train_lr"):
        assert abs(float(stats_a[k]) - stats_b[k]) <= 1e-6 * max(1.0, abs(stats_b[k])), (k, stats_a[k], stats_b[k])
    for (n, a), (_, b) in zip(s_a.named_parameters(), s_b.named_parameters()):
        assert torch.allclose(a, b, rtol=0, atol=1e-7), n


def test_validate_equals_the_oracle_loop_on_cpu():
    """tools/engine.py:78-104."""
    from deltakd_amd import engine
    from oracle import engine_ref
    s, _ = toy_pair("soft")             # distilled student: eval mode returns the averaged heads, train-tuple branch not taken
    g = torch.Generator().manual_seed(9)
    data = [(torch.randn(6, 3, 32, 32, generator=g), torch.randint(0, 10, (6,), generator=g)) for _ in range(3)]
    a = engine.validate(s, data, torch.device("cpu"), SimpleNamespace(rank=1))
    b = engine_ref.validate_ref(s, data)
    assert set(a) == {"val_loss", "val_acc1", "val_acc5"}
    for k in b:
        assert abs(float(a[k]) - b[k]) < 1e-6, (k, a[k], b[k])
    assert not s.training


def test_training_loop_narrows_student_taps_to_what_the_criterion_reads():
    """deltakd_amd.engine: this repo's DistillationLoss reads student_features[0], [1], [-1] for lrkd and none for `none`, so the
    loop asks the student for exactly those blocks -- chosen from the CRITERION's own type (not the args', which may differ), and
    only for the duration of the epoch; a foreign criterion (or another type) gets every block, as the reference's
    forward_with_features returns them."""
    from types import SimpleNamespace
    from deltakd_amd import engine
    from deltakd_amd.losses import DistillationLoss

    class Student:
        blocks = [None] * 12
        tap_layers = None

    def crit(kind, **kw):
        c = object.__new__(DistillationLoss)
        c.__dict__.update(distillation_type=kind, **kw)
        return c

    wrapped = SimpleNamespace(module=Student())                       # a data-parallel style wrapper
    for kind, want in (("lrkd", (0, 1, -1)), ("NONE", ())):
        model, prev = engine._narrow_student_taps(wrapped, crit(kind))
        assert model is wrapped.module and prev is None and wrapped.module.tap_layers == want
        wrapped.module.tap_layers = prev
    assert engine._narrow_student_taps(wrapped, crit("mgd")) is None and wrapped.module.tap_layers is None
    assert engine._narrow_student_taps(wrapped, lambda *a: 0) is None and wrapped.module.tap_layers is None
    # a subclass that reads other blocks says so
    engine._narrow_student_taps(wrapped, crit("lrkd", student_taps=(2, 3)))
    assert wrapped.module.tap_layers == (2, 3)


def test_train_one_epoch_restores_the_students_tap_setting():
    """The narrowing must not outlive the epoch: after train_one_epoch -- also when a step raises -- the model
    returns every block again to callers outside the loop."""
    from types import SimpleNamespace
    import pytest
    import torch
    from deltakd_amd import engine
    from deltakd_amd.losses import DistillationLoss

    class Student(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList()
            self.tap_layers = None
            self.seen = []

        def forward_with_taps(self, x):
            self.seen.append(self.tap_layers)
            raise RuntimeError("stop here")

    s = Student()
    crit = object.__new__(DistillationLoss)
    torch.nn.Module.__init__(crit)
    crit.distillation_type = "lrkd"
    # args say `none` (no teacher lookahead), the criterion says lrkd: the taps follow the criterion
    args = SimpleNamespace(epochs=1, distillation_type="none", print_freq=0, rank=0)
    with pytest.raises(RuntimeError, match="stop here"):
        engine.train_one_epoch(s, torch.nn.Identity(), [(torch.zeros(2, 3), torch.zeros(2, dtype=torch.long))], crit, None, None, None,
                               None, None, "cpu", 0, args)
    assert s.seen == [(0, 1, -1)] and s.tap_layers is None
