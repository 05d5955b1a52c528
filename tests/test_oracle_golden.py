"""The CPU oracle against the golden fixtures the reference's own code produced (tests/golden, oracle/gen_golden.py).
This is what pins oracle/loss_ref.py: every branch's loss and every gradient must reproduce the reference's numbers."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import loss_ref, vit_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOY = dict(img_size=32, patch_size=8, mlp_ratio=2.0)
CASES = ["none_hardlabel", "none_softlabel", "soft_softlabel", "hard_hardlabel", "lrkd_hardlabel", "lrkd_softlabel",
         "mgd_softlabel", "wasskd_softlabel", "diffkd_hardlabel", "vitkd_softlabel", "curkd_early_hardlabel", "curkd_mid_softlabel",
         "curkd_late_softlabel", "saliency1_softlabel", "saliency2_hardlabel", "saliency3_softlabel"]


def build(fx, tsd):
    args = SimpleNamespace(**json.loads(str(fx["args_json"])))
    distilled = "distilled" in str(fx["student_name"])
    student = vit_ref.VisionTransformerRef(64, 12, 1, 10, distilled, 0.1, **TOY)
    teacher = vit_ref.VisionTransformerRef(128, 12, 2, 10, True, 0.1, **TOY)
    loss_ref.attach_aux_ref(student, teacher, args.distillation_type, args.lrkd_rank, getattr(args, "saliency_method", 1))
    student.load_state_dict({k[8:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("student.")}, strict=True)
    teacher.load_state_dict({k[8:]: torch.from_numpy(v) for k, v in tsd.items()}, strict=True)
    return student.train(), teacher.eval(), args


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference(name):
    fx = dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))
    tsd = dict(np.load(os.path.join(GOLD, "toy_teacher.npz"), allow_pickle=False))
    student, teacher, args = build(fx, tsd)
    kind = args.distillation_type
    x = torch.from_numpy(fx["x"])
    target = torch.from_numpy(fx["soft_targets"] if int(fx["use_soft_label"]) else fx["labels"])
    student.set_droppath_keep([torch.from_numpy(k.astype(np.float32)) for k in fx["keep"]])
    draws = {}
    if "draw.noise" in fx:
        draws["noise"] = torch.from_numpy(fx["draw.noise"])
    if kind == "diffkd":
        draws = {"t": torch.from_numpy(fx["draw.t"]), "noise": [torch.from_numpy(fx[f"draw.noise{i}"]) for i in range(3)],
                 "drop": [torch.from_numpy(fx[f"draw.drop{i}"].astype(np.float32)) for i in range(3)]}
    if kind == "lrkd":
        draws["lrkd_targets"] = [torch.from_numpy(fx[f"lrkd_target{i}"]) for i in range(3)]
    crit = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), teacher, kind, args.alpha, args.tau)
    out, feats = (student(x), None) if kind in ("soft", "hard") else loss_ref.forward_with_features_ref(student, x)
    loss = crit(x, out, student, feats, target, args, draws)
    loss.backward()
    logits = out if isinstance(out, torch.Tensor) else out[0]
    assert torch.allclose(logits, torch.from_numpy(fx["student_logits"]), atol=1e-5)
    assert abs(loss.item() - float(fx["loss"])) <= 1e-5 * abs(float(fx["loss"]))
    params = dict(student.named_parameters())
    for n, norm in zip(fx["grad_names"], fx["grad_norms"]):
        g = params[str(n)].grad
        assert abs(g.double().norm().item() - norm) <= 1e-4 * norm + 1e-9, str(n)
        key = "grad." + str(n)
        if key in fx:
            assert torch.allclose(g, torch.from_numpy(fx[key]), rtol=1e-3, atol=1e-7 + 1e-4 * float(np.abs(fx[key]).max())), str(n)


def test_random_masking_fixture():
    fx = np.load(os.path.join(GOLD, "random_masking.npz"))
    x, noise = torch.from_numpy(fx["x"]), torch.from_numpy(fx["noise"])
    keep, mask, restore, masked = loss_ref.random_masking_ref(x, 0.5, noise)
    assert torch.equal(keep, torch.from_numpy(fx["x_keep"])) and torch.equal(mask, torch.from_numpy(fx["mask"]))
    assert torch.equal(restore, torch.from_numpy(fx["ids_restore"])) and torch.equal(masked, torch.from_numpy(fx["ids_masked"]))
    # the product's index plumbing (pure torch, runs on CPU) gives the same mask, and the closed form where(mask, tok, x)
    from deltakd_amd.misc import masking_indices, random_masking
    m2, r2, _, len_keep = masking_indices(noise, 0.5)
    assert torch.equal(m2, mask) and torch.equal(r2, restore) and len_keep == 8
    k3, m3, r3, im3 = random_masking(x, 0.5, noise)
    assert torch.equal(k3, keep) and torch.equal(m3, mask) and torch.equal(im3, masked)
    tok = torch.randn(1, 1, x.shape[2])
    ref = torch.gather(torch.cat([keep, tok.repeat(x.shape[0], x.shape[1] - keep.shape[1], 1)], 1), 1,
                       restore.unsqueeze(-1).repeat(1, 1, x.shape[2]))
    assert torch.equal(ref, torch.where(mask.unsqueeze(-1) > 0, tok.expand_as(x), x))


def test_golden_report_records_pinning():
    rep = json.load(open(os.path.join(GOLD, "REPORT.json")))
    for name in CASES:
        assert rep[name]["rel"] < 1e-5 and rep[name]["grad_max_rel"] < 1e-4
    assert float(rep["hf_deit_crosscheck_max_abs"]) < 1e-4      # oracle ViT vs HF transformers DeiT
