#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running THE REFERENCE's own loss code on CPU.

Runs ONLY in the build container (needs /root/reference, which never travels):
    python oracle/gen_golden.py
It imports the reference's model/loss.py, model/models.py, model/misc.py after
injecting in-memory stub modules for the third-party imports that are absent
here (timm, geomloss, torchvision); ``timm.create_model`` is stubbed to return
oracle/vit_ref.py models of toy size, so the reference's own
``load_teacher_student_model`` bolts its aux modules onto them and the
reference's own ``DistillationLoss`` computes the losses / gradients stored in
the fixtures.  It also asserts that oracle/loss_ref.py reproduces them
(that is what "pinned" means in oracle/__init__.py), and cross-checks
oracle/vit_ref.py against HF transformers' DeiT (built from a local config).

Fixtures are data only: inputs, injected random draws, weights, expected
outputs and gradients.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

# --- load the oracle package without putting the repo root (which has its own model/, tools/) on sys.path
spec = importlib.util.spec_from_file_location("oracle", os.path.join(HERE, "__init__.py"),
                                              submodule_search_locations=[HERE])
oracle = importlib.util.module_from_spec(spec)
sys.modules["oracle"] = oracle
spec.loader.exec_module(oracle)
from oracle import vit_ref, loss_ref  # noqa: E402

TOY = dict(img_size=32, patch_size=8, num_classes=10, mlp_ratio=2.0)
TOY_DIMS = {  # name -> (D, depth, heads, distilled)
    "deit_toy_student": (64, 12, 1, False),
    "deit_toy_student_distilled": (64, 12, 1, True),
    "deit_toy_teacher_distilled": (128, 12, 2, True),
}


def install_stubs():
    timm = types.ModuleType("timm")

    def create_model(name, pretrained=False, drop_path_rate=0.0, num_classes=1000, **kw):
        D, depth, H, dist = TOY_DIMS[name]
        return vit_ref.VisionTransformerRef(D, depth, H, num_classes, dist, drop_path_rate, img_size=TOY["img_size"],
                                            patch_size=TOY["patch_size"], mlp_ratio=TOY["mlp_ratio"])

    timm.create_model = create_model
    tl = types.ModuleType("timm.loss")
    tl.SoftTargetCrossEntropy = loss_ref.SoftTargetCrossEntropyRef
    tl.LabelSmoothingCrossEntropy = loss_ref.LabelSmoothingCrossEntropyRef
    td = types.ModuleType("timm.data")
    td.create_transform = None
    geo = types.ModuleType("geomloss")
    geo.SamplesLoss = None
    tv = types.ModuleType("torchvision")
    tvd = types.ModuleType("torchvision.datasets")
    tvt = types.ModuleType("torchvision.transforms")
    tv.datasets, tv.transforms = tvd, tvt
    for k, v in {"timm": timm, "timm.loss": tl, "timm.data": td, "geomloss": geo, "torchvision": tv,
                 "torchvision.datasets": tvd, "torchvision.transforms": tvt}.items():
        sys.modules[k] = v


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def grads_np(module, prefix="grad."):
    return {prefix + k: p.grad.detach().cpu().numpy() for k, p in module.named_parameters() if p.grad is not None}


def main():
    sys.dont_write_bytecode = True
    install_stubs()
    sys.path.insert(0, REF)
    import model.loss as ref_loss          # noqa: E402  (the reference's files)
    import model.models as ref_models      # noqa: E402
    import model.misc as ref_misc          # noqa: E402
    assert ref_loss.__file__.startswith(REF) and ref_misc.__file__.startswith(REF)
    os.makedirs(OUT, exist_ok=True)
    report = {}

    B, C = 4, TOY["num_classes"]
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, TOY["img_size"], TOY["img_size"], generator=g)
    labels = torch.randint(0, C, (B,), generator=g)
    soft_targets = torch.softmax(torch.randn(B, C, generator=g) * 2.0, dim=1)   # mixup-style soft target

    branches = [
        # (tag, distillation_type, student name, extra args, label kinds: False = int labels, True = soft targets)
        ("none", "none", "deit_toy_student", {}, (False, True)),
        ("soft", "soft", "deit_toy_student_distilled", dict(alpha=0.1, tau=3.0), (True,)),
        ("hard", "hard", "deit_toy_student_distilled", dict(alpha=0.5), (False,)),
        ("lrkd", "lrkd", "deit_toy_student", dict(alpha=0.1, lrkd_rank=16), (False, True)),
        ("mgd", "mgd", "deit_toy_student", dict(mgd_alpha=7e-5, mgd_mask_ratio=0.5), (True,)),
        ("wasskd", "wasskd", "deit_toy_student", dict(wasskd_type="l1"), (True,)),
        ("diffkd", "diffkd", "deit_toy_student", dict(alpha=0.1), (False,)),
        ("vitkd", "vitkd", "deit_toy_student", dict(), (True,)),
        ("curkd_early", "curkd", "deit_toy_student", dict(current_epoch=0), (False,)),
        ("curkd_mid", "curkd", "deit_toy_student", dict(current_epoch=120), (True,)),
        ("curkd_late", "curkd", "deit_toy_student", dict(current_epoch=200), (True,)),
        ("saliency1", "saliency_mgd", "deit_toy_student", dict(saliency_method=1, saliency_mask_ratio=0.5), (True,)),
        ("saliency2", "saliency_mgd", "deit_toy_student", dict(saliency_method=2, saliency_mask_ratio=0.5), (False,)),
        ("saliency3", "saliency_mgd", "deit_toy_student", dict(saliency_method=3, saliency_mask_ratio=0.5), (True,)),
    ]
    teacher_saved = False
    for tag, kind, sname, extra, label_kinds in branches:
        for soft_label in label_kinds:
            args = loss_ref.default_args(distillation_type=kind, dataset="cifar-10", **extra)
            if soft_label:
                args.mixup = 0.8
            torch.manual_seed(42)
            teacher, student = ref_models.load_teacher_student_model("deit_toy_teacher_distilled", sname, 0.1, args)
            # give the zero-initialised pieces non-trivial values so every gradient path is exercised
            with torch.no_grad():
                for net, seed in ((teacher, 1001), (student, 1002)):
                  torch.manual_seed(seed)
                  if net is teacher:       # same teacher in every fixture: re-draw all of it from its own seed
                      net.init_weights()
                  for m in net.modules():
                    if isinstance(m, (nn.Linear, nn.Conv2d)) and m.bias is not None:
                        m.bias.normal_(0, 0.02)
                    if isinstance(m, nn.LayerNorm):
                        m.weight.normal_(1.0, 0.05)
                        m.bias.normal_(0, 0.05)
                if hasattr(student, "mask_token"):
                    student.mask_token.normal_(0, 0.02)
                # trunc_normal(.02) weights give tiny features; scale the MLP taps up to O(1)
                for net in (student, teacher):
                    for blk in net.blocks:
                        blk.mlp.fc2.weight.mul_(8.0)
                        blk.attn.proj.weight.mul_(4.0)
            student.train()
            teacher.eval()
            depth = len(student.blocks)
            kg = torch.Generator().manual_seed(7)
            keep = []
            for i, blk in enumerate(student.blocks):
                p = blk.drop_path1.drop_prob
                for _ in range(2):
                    keep.append((torch.rand(B, generator=kg) >= p).float())
            # force at least one dropped sample in the deepest block so the path is exercised
            keep[-1][1] = 0.0
            keep[-2][2] = 0.0
            student.set_droppath_keep(keep)

            crit = ref_loss.DistillationLoss(ref_loss.call_base_loss(args), teacher, kind, args.alpha, args.tau)
            tgt = soft_targets if soft_label else labels

            def run_student():
                if kind in ("soft", "hard"):
                    return student(x), None
                return ref_models.forward_with_features(student, x)

            # --- the reference computes the loss
            out, feats = run_student()
            torch.manual_seed(99)
            loss = crit(x, out, student, feats, tgt, args)
            student.zero_grad()
            loss.backward()
            ref_grads = grads_np(student)

            # --- replay the draws the reference consumed under seed 99, for the oracle and the HIP path
            draws, draws_np = {}, {}
            with torch.no_grad():
                t_logits, t_feats = loss_ref.forward_with_features_ref(teacher, x)
            torch.manual_seed(99)
            P = student.patch_embed.num_patches
            Dt = teacher.embed_dim
            if kind in ("mgd", "vitkd") or (kind == "curkd" and args.current_epoch >= 151):
                draws["noise"] = torch.rand(B, P)
                draws_np["draw.noise"] = draws["noise"].numpy()
            if kind == "diffkd":
                draws["t"] = torch.randint(0, 8, (B,))
                draws["noise"], draws["drop"] = [], []
                for i in range(3):
                    draws["noise"].append(torch.randn(B, P, Dt))
                    draws["drop"].append(torch.empty(B, P, Dt).bernoulli_(0.9))
                draws_np["draw.t"] = draws["t"].numpy()
                for i in range(3):
                    draws_np[f"draw.noise{i}"] = draws["noise"][i].numpy()
                    draws_np[f"draw.drop{i}"] = draws["drop"][i].numpy().astype(np.uint8)
            if kind == "saliency_mgd":
                with torch.no_grad():
                    sc = loss_ref.saliency_scores_ref(student.saliency_attn, t_feats[-1], args.saliency_method)
                draws_np["draw.scores"] = sc.numpy()
            if kind == "lrkd":
                sel = [t_feats[0], t_feats[1], t_feats[11]]
                draws["lrkd_targets"] = [loss_ref.lrkd_targets_ref(f[:, 2:], args.lrkd_rank) for f in sel]
                for i in range(3):
                    draws_np[f"lrkd_target{i}"] = draws["lrkd_targets"][i].numpy()

            # --- the oracle restatement must reproduce it
            ocrit = loss_ref.DistillationLossRef(loss_ref.call_base_loss_ref(args), teacher, kind, args.alpha, args.tau)
            out2, feats2 = (student(x), None) if kind in ("soft", "hard") else loss_ref.forward_with_features_ref(student, x)
            if kind == "diffkd":
                # the reference's denoiser is the reference's own class (Dropout drawn internally):
                # wrap it so the injected keep mask is used
                dn = student.denoise_fn

                def denoise(xx, tt, keepmask, dn=dn):
                    xx = xx + dn.time_embed(tt.float().view(-1, 1)).unsqueeze(1)
                    return dn.net[2](dn.net[1](dn.net[0](xx))) * keepmask / 0.9
                student.__dict__["denoise_fn"] = denoise
            loss2 = ocrit(x, out2, student, feats2, tgt, args, draws)
            if kind == "diffkd":
                del student.__dict__["denoise_fn"]
            student.zero_grad()
            loss2.backward()
            o_grads = grads_np(student)
            rel = abs(loss2.item() - loss.item()) / max(abs(loss.item()), 1e-12)
            gmax = 0.0
            for k in ref_grads:
                den = np.abs(ref_grads[k]).max() + 1e-12
                gmax = max(gmax, float(np.abs(ref_grads[k] - o_grads[k]).max() / den))
            name = f"{tag}_{'softlabel' if soft_label else 'hardlabel'}"
            report[name] = dict(loss_reference=loss.item(), loss_oracle=loss2.item(), rel=rel, grad_max_rel=gmax)
            tol = 2e-3 if kind == "lrkd" else 1e-5   # lrkd: fresh SVD inside the reference vs stored targets
            assert rel < tol and gmax < max(tol, 1e-4), (name, report[name])

            # --- the two addends of the loss, each computed by the reference's own code WITHOUT forming total - base in fp32 (the
            # distillation term is 1e-6 of the loss for some branches): loss = w_b * base + w_d * d
            #   base : the reference's base criterion on the reference student's logits;
            #   d    : alpha-weighted branches (soft, hard, lrkd, diffkd: model/loss.py:241) -> the reference's DistillationLoss built
            #          with alpha = 1 (base * 0 + d * 1); free-function branches (mgd, vitkd, curkd, saliency_mgd) -> that function;
            #          wasskd (inline, base + 5 d, d ~ 0.3 of the loss) -> total - base.
            # Same seed (99) before every call, so every call consumes the draws of the main run.
            def grads_or_zero():
                return {"grad." + k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32))
                        for k, p in student.named_parameters()}

            out_b, feats_b = run_student()
            logits_b = out_b if isinstance(out_b, torch.Tensor) else out_b[0]
            base = crit.base_criterion(logits_b, tgt)
            student.zero_grad()
            base.backward()
            g_base = grads_or_zero()
            alpha_weighted = kind in ("soft", "hard", "lrkd", "diffkd")
            w_b = (1.0 - args.alpha) if alpha_weighted else 1.0
            d_val, g_d, w_d = 0.0, None, 0.0
            if alpha_weighted:
                crit1 = ref_loss.DistillationLoss(ref_loss.call_base_loss(args), teacher, kind, 1.0, args.tau)
                out_d, feats_d = run_student()
                torch.manual_seed(99)
                d = crit1(x, out_d, student, feats_d, tgt, args)
                w_d = args.alpha
            elif kind in ("mgd", "vitkd", "curkd", "saliency_mgd"):
                out_d, feats_d = run_student()
                with torch.no_grad():
                    _, tf_ref = ref_models.forward_with_features(teacher, x)
                torch.manual_seed(99)
                if kind == "mgd":
                    d = ref_loss.mgd_loss(student, feats_d, tf_ref, args)
                elif kind == "vitkd":
                    d = ref_loss.vitkd_loss(student, feats_d, tf_ref, alpha_vitkd=0.00003, beta_vitkd=0.000003, lambda_vitkd=0.5)
                elif kind == "curkd":
                    d = ref_loss.curkd_loss(student, feats_d, tf_ref, args)
                else:
                    d = ref_loss.saliency_mgd_loss(student, feats_d, tf_ref, args)
                w_d = 1.0
            else:
                d = None
            if d is not None:
                student.zero_grad()
                d.backward()
                g_d = grads_or_zero()
                d_val = float(d.item())
            elif kind == "wasskd":
                w_d = 5.0
                d_val = (float(loss.item()) - float(base.item())) / 5.0
                g_d = {k: (ref_grads.get(k, np.zeros_like(g_base[k])) - g_base[k]) / 5.0 for k in g_base}
            base_term, distill_term = w_b * float(base.item()), w_d * d_val
            assert abs(loss.item() - (base_term + distill_term)) <= 2e-6 * abs(loss.item()), (name, loss.item(), base_term, distill_term)
            if g_d is not None:
                for k in ref_grads:
                    comp = w_b * g_base[k] + w_d * g_d[k]
                    den = np.abs(ref_grads[k]).max() + 1e-12
                    assert np.abs(comp - ref_grads[k]).max() / den < 1e-4, (name, k)
            report[name].update(base_term=base_term, distill_term=distill_term)
            # --- "strong" variant: the same reference quantities recombined with the distillation term scaled by K so that it is
            # half of the base term (the reference's constants leave it at 1e-6 .. 1e-2 of the loss at toy size, where a loss-level
            # tolerance cannot see it):  loss_K = w_b base + K w_d d,  grad_K = w_b grad(base) + K w_d grad(d).
            strong = {}
            if g_d is not None and 0.0 < distill_term < 0.05 * base_term:
                K = float(f"{0.5 * base_term / distill_term:.2g}")
                strong["strong.scale"] = np.array(K, dtype=np.float64)
                strong["strong.loss"] = np.array(base_term + K * distill_term, dtype=np.float64)
                s_names = sorted(ref_grads)
                sg = {k: (w_b * g_base[k].astype(np.float64) + K * w_d * g_d[k].astype(np.float64)) for k in s_names}
                strong["strong.grad_norms"] = np.array([np.sqrt((sg[k] ** 2).sum()) for k in s_names])
                for k in s_names:
                    short = k[len("grad."):]
                    if not short.startswith("blocks.") or short.split(".")[1] in ("0", "5", "11"):
                        strong["strong." + k] = sg[k].astype(np.float32)
                report[name].update(strong_scale=K)

            logits = out if isinstance(out, torch.Tensor) else out[0]
            fx = dict(x=x.numpy(), labels=labels.numpy(), soft_targets=soft_targets.numpy(),
                      base_loss=np.array(base_term, dtype=np.float64), distill_loss=np.array(distill_term, dtype=np.float64),
                      use_soft_label=np.array(int(soft_label)), loss=np.array(loss.item(), dtype=np.float64),
                      student_logits=logits.detach().numpy(), teacher_logits=t_logits.numpy(),
                      keep=np.stack([k.numpy() for k in keep]).astype(np.uint8),
                      kind=np.array(kind), student_name=np.array(sname),
                      args_json=np.array(json.dumps({k: v for k, v in vars(args).items()})))
            if not isinstance(out, torch.Tensor):
                fx["student_logits_kd"] = out[1].detach().numpy()
            if feats is not None:
                for i in (0, 1, 11):
                    fx[f"student_feat{i}"] = feats[i].detach().numpy()
                    fx[f"teacher_feat{i}"] = t_feats[i].numpy()
            fx.update(draws_np)
            fx.update(strong)
            fx.update(sd_np(student, "student."))
            if not teacher_saved:     # identical in every fixture (own seed): stored once
                np.savez_compressed(os.path.join(OUT, "toy_teacher.npz"), **sd_np(teacher, "teacher."))
                teacher_saved = True
            # gradients: L2 norm of every parameter's grad + full grads of a representative subset
            names = sorted(ref_grads)
            fx["grad_names"] = np.array([n[len("grad."):] for n in names])
            fx["grad_norms"] = np.array([np.sqrt((ref_grads[n].astype(np.float64) ** 2).sum()) for n in names])
            for n in names:
                short = n[len("grad."):]
                if not short.startswith("blocks.") or short.split(".")[1] in ("0", "5", "11"):
                    fx[n] = ref_grads[n]
            np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
            print(f"{name:22s} loss {loss.item():.8f}  oracle rel {rel:.2e}  grad rel {gmax:.2e}")

    # --- random_masking: reference vs closed form where(m, mask_token, x)
    torch.manual_seed(5)
    xx = torch.randn(3, 16, 8)
    torch.manual_seed(11)
    keepx, mask, ids_restore, ids_masked = ref_misc.random_masking(xx, 0.5)
    torch.manual_seed(11)
    noise = torch.rand(3, 16)
    k2, m2, r2, im2 = loss_ref.random_masking_ref(xx, 0.5, noise)
    assert torch.equal(keepx, k2) and torch.equal(mask, m2) and torch.equal(ids_restore, r2) and torch.equal(ids_masked, im2)
    np.savez_compressed(os.path.join(OUT, "random_masking.npz"), x=xx.numpy(), noise=noise.numpy(), x_keep=keepx.numpy(),
                        mask=mask.numpy(), ids_restore=ids_restore.numpy(), ids_masked=ids_masked.numpy())
    report["random_masking"] = "bit-exact"

    # --- ViT arithmetic cross-check against HF transformers DeiT (timm itself is absent: parity unpinned)
    try:
        for k in [k for k in sys.modules if k.split(".")[0] in ("torchvision", "timm", "geomloss")]:
            del sys.modules[k]          # the stubs would confuse transformers' optional-dependency probing
        from transformers import DeiTConfig, DeiTModel
        cfg = DeiTConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=256,
                         image_size=32, patch_size=8, layer_norm_eps=1e-6, hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0, qkv_bias=True)
        hf = DeiTModel(cfg, add_pooling_layer=False).eval()
        mine = vit_ref.VisionTransformerRef(64, 2, 1, 10, True, 0.0, img_size=32, patch_size=8).eval()
        with torch.no_grad():
            e = hf.embeddings
            mine.cls_token.copy_(e.cls_token)
            mine.dist_token.copy_(e.distillation_token)
            mine.pos_embed.copy_(e.position_embeddings)
            mine.patch_embed.proj.weight.copy_(e.patch_embeddings.projection.weight)
            mine.patch_embed.proj.bias.copy_(e.patch_embeddings.projection.bias)
            for i, layer in enumerate(hf.layers):
                b = mine.blocks[i]
                a = layer.attention
                b.attn.qkv.weight.copy_(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0))
                b.attn.qkv.bias.copy_(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0))
                b.attn.proj.load_state_dict(a.o_proj.state_dict())
                b.norm1.load_state_dict(layer.layernorm_before.state_dict())
                b.norm2.load_state_dict(layer.layernorm_after.state_dict())
                b.mlp.fc1.load_state_dict(layer.mlp.fc1.state_dict())
                b.mlp.fc2.load_state_dict(layer.mlp.fc2.state_dict())
            mine.norm.load_state_dict(hf.layernorm.state_dict())
            y_hf = hf(pixel_values=x).last_hidden_state
            y_me = mine.forward_features(x)
        err = (y_hf - y_me).abs().max().item()
        report["hf_deit_crosscheck_max_abs"] = err
        assert err < 1e-4, err
        print("HF DeiT cross-check max abs err", err)
    except ImportError as e:   # transformers not importable: record, do not fail
        report["hf_deit_crosscheck_max_abs"] = f"skipped: {e}"

    with open(os.path.join(OUT, "REPORT.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
