"""Fused AdamW over flat parameter storage + timm-compatible factories.

Replaces ``timm.optim.create_optimizer`` / ``timm.scheduler.create_scheduler`` ([3P], /root/reference/tools/train.py:264-266)
and the multi-tensor torch AdamW they build.  MI355X-first layout: all trainable parameters live in ONE fp32 buffer
(the module's ``nn.Parameter``s become views of it), gradients in a second flat buffer (``p.grad`` views), so

  * one AdamW launch per weight-decay group (2 per step) instead of a foreach chain, and the same launch refreshes the
    bf16 shadow copies the MFMA GEMMs read (``Shadow.bind_flat``);
  * ``zero_grad`` is one memset and keeps the gradient buffers alive (the kernels accumulate straight into ``p.grad``);
  * data-parallel gradient averaging is an all-reduce over contiguous slices of one buffer (deltakd_amd.ddp).

Update rule = torch.optim.AdamW (decoupled weight decay, bias correction), checked against it in tests.
"""
import math

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def param_groups_weight_decay(model, weight_decay, no_weight_decay_list=()):
    """timm.optim.param_groups_weight_decay [3P]: 1-D params, biases and the model's no_weight_decay() set get wd 0."""
    no_weight_decay_list = set(no_weight_decay_list)
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if param.ndim <= 1 or name.endswith(".bias") or name in no_weight_decay_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{"params": no_decay, "weight_decay": 0.}, {"params": decay, "weight_decay": weight_decay}]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, shadows=()):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self._shadows = list(shadows)
        self._flat = []
        self._where = {}               # id(param) -> (flat index, start, end) inside the flat buffers
        self._step = 0
        self._units = {}               # lazy unit -> device f32 [2]: (largest |g| this step, the unit's own step count)
        self._unit_segs = {}           # lazy unit -> [(flat index, start, end)]
        self._touched = {}             # lazy unit -> device f32 scalar the loss sets to 1 when it uses the unit (``p.dkd_unit_touched``)
        self.grad_sync = None          # set by deltakd_amd.ddp: called with the flat grad buffers before the update
        self._synced = False           # gradients of the current iteration already averaged (sync_grads() ran ahead of step())
        self._flatten()

    def _flatten(self):
        bound, bound_params = {}, []
        for group in self.param_groups:
            # (dkd_never_grad: parameters no gradient ever reaches -- the saliency scorer -- are skipped like torch.optim.AdamW skips
            # ``p.grad is None``: no update, no weight decay; they keep their slot in the checkpoint's parameter numbering)
            ps = [p for p in group["params"] if p.requires_grad and not getattr(p, "dkd_never_grad", False)]
            if not ps:
                self._flat.append(None)
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise RuntimeError("FusedAdamW needs parameters on an MI355X device (move the model first; no CPU path)")
            # Parameters that receive a gradient only in some steps (``p.dkd_lazy_unit`` = a name shared by the parameters that are used
            # together: curkd's align stages, model/loss.py:362-420) go to the END of the group's buffer, unit by unit.  torch.optim.AdamW
            # skips a parameter whose gradient is None altogether (no moment decay, no weight decay, its own step count); the kernels here
            # write into pre-allocated zero-filled gradients, so "nothing was written" is detected per unit on the device and those
            # segments are skipped by the gated kernel (dkd_adamw_step_gated).
            ps = sorted(ps, key=lambda q: (getattr(q, "dkd_lazy_unit", None) is not None, getattr(q, "dkd_lazy_unit", None) or ""))
            sizes = [(p.numel() + 7) // 8 * 8 for p in ps]          # keep every f32 AND bf16 view 16-byte aligned
            total = sum(sizes)
            fp = torch.zeros(total, device=dev, dtype=F32)
            fg = torch.zeros(total, device=dev, dtype=F32)
            fb = torch.zeros(total, device=dev, dtype=BF16)
            off = 0
            segments = []                                           # [start, end, unit | None], consecutive
            for p, n in zip(ps, sizes):
                k = p.numel()
                fp[off:off + k].copy_(p.detach().reshape(-1))
                old_grad = p.grad
                p.data = fp[off:off + k].view(p.shape)
                p.grad = fg[off:off + k].view(p.shape)
                if old_grad is not None:
                    p.grad.copy_(old_grad)
                bound[id(p)] = fb[off:off + k]
                bound_params.append(p)
                self._where[id(p)] = (len(self._flat), off, off + n)
                unit = getattr(p, "dkd_lazy_unit", None)
                if unit is not None:
                    p.dkd_unit_touched = self._touched.setdefault(unit, torch.zeros((), device=dev, dtype=F32))
                if segments and segments[-1][2] == unit:
                    segments[-1][1] = off + n
                else:
                    segments.append([off, off + n, unit])
                off += n
            fb.copy_(fp)
            for s0, e0, unit in segments:
                if unit is not None:
                    self._units.setdefault(unit, torch.zeros(2, device=dev, dtype=F32))
                    self._unit_segs.setdefault(unit, []).append((len(self._flat), s0, e0))
            self._flat.append(dict(p=fp, g=fg, bf=fb, m=torch.zeros_like(fp), v=torch.zeros_like(fp), segments=segments))
        for sh in self._shadows:
            sh.bind_flat(bound, bound_params)
            sh.optimizer_stepped(bf16_fresh=True)

    def grad_ranges(self, params):
        """Smallest contiguous [start, end) range of each flat gradient buffer that covers ``params`` -> {flat index: (s, e)}."""
        out = {}
        for p in params:
            w = self._where.get(id(p))
            if w is None:
                continue
            i, s, e = w
            lo, hi = out.get(i, (s, e))
            out[i] = (min(lo, s), max(hi, e))
        return out

    @property
    def flat_grads(self):
        return [f["g"] for f in self._flat if f is not None]

    def zero_grad(self, set_to_none=False):
        for f in self._flat:
            if f is not None:
                f["g"].zero_()
        self._synced = False

    @torch.no_grad()
    def sync_grads(self):
        """Data parallel: finish the gradient averaging of this iteration NOW (the tail all-reduce and the join of the comm stream,
        deltakd_amd.ddp).  Anything that reads the averaged gradients before ``step()`` -- gradient clipping -- calls this first, as
        torch DDP has finished its all-reduce by the time ``backward()`` returns (tools/engine.py:61-62 of the reference clips inside
        the scaler, after backward).  Idempotent per iteration; ``step()`` calls it if nobody did."""
        if self.grad_sync is not None and not self._synced:
            self.grad_sync(self.flat_grads)
        self._synced = True

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm, norm_type=2.0):
        """torch.nn.utils.clip_grad_norm_ over the flat gradient buffers (one norm per buffer instead of one per parameter);
        -> total norm (0-dim device tensor).  Averages the gradients first when data parallel."""
        self.sync_grads()
        flats = self.flat_grads
        if norm_type == float("inf"):
            total = torch.stack([g.abs().max() for g in flats]).max()
        else:
            total = torch.stack([torch.linalg.vector_norm(g, norm_type) for g in flats]).norm(norm_type)
        coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
        for g in flats:
            g.mul_(coef)
        return total

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedAdamW does not take a closure")
        self.sync_grads()
        self._synced = False
        self._step += 1
        for unit, state in self._units.items():           # did anything write a gradient for this unit?  (device side: no host sync)
            # "a gradient was written": the loss said so (explicit flag) or some element is non-zero (losses that do not flag)
            gate = torch.stack([self._flat[i]["g"][s0:e0].abs().amax() for i, s0, e0 in self._unit_segs[unit]]
                               + [self._touched[unit]]).amax()
            self._touched[unit].zero_()
            state[0] = gate
            state[1] += (gate > 0).to(F32)
        for group, f in zip(self.param_groups, self._flat):
            if f is None:
                continue
            b1, b2 = group["betas"]
            for s0, e0, unit in f["segments"]:
                seg = [f[k][s0:e0] for k in ("p", "g", "m", "v", "bf")]
                if unit is None:
                    ops.adamw_step(*seg, group["lr"], b1, b2, group["eps"], group["weight_decay"], self._step)
                else:
                    ops.adamw_step_gated(*seg, group["lr"], b1, b2, group["eps"], group["weight_decay"], self._units[unit])
        for sh in self._shadows:
            sh.optimizer_stepped(bf16_fresh=True)
            if hasattr(sh, "refresh_transposed"):
                sh.refresh_transposed()

    # ---- checkpoint wire format: torch.optim.AdamW's (what timm's create_optimizer builds for --opt adamw and the reference
    # saves under checkpoint["optimizer"], /root/reference/tools/train.py:349-357), so checkpoints move both ways
    _TORCH_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)

    def state_dict(self):
        state, groups, idx = {}, [], 0
        unit_steps = {u: int(st[1].item()) for u, st in self._units.items()}       # (host sync: checkpoint time only)
        for group, f in zip(self.param_groups, self._flat):
            ids = []
            for p in group["params"]:
                w = self._where.get(id(p))
                unit = getattr(p, "dkd_lazy_unit", None)
                step = unit_steps[unit] if unit in unit_steps else self._step
                if w is not None and step > 0:
                    _, s, _ = w
                    k = p.numel()
                    state[idx] = {"step": torch.tensor(float(step)),
                                  "exp_avg": f["m"][s:s + k].view(p.shape).clone(),
                                  "exp_avg_sq": f["v"][s:s + k].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
            g = dict(self._TORCH_GROUP_DEFAULTS)
            g.update({k: v for k, v in group.items() if k != "params"})
            g["params"] = ids
            groups.append(g)
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        steps = set()
        # state the checkpoint does not hold (torch omits parameters that never had a gradient: a curkd stage before its epochs) is
        # RESET, not left at the live optimizer's values: zero moments, zero step count
        for f in self._flat:
            if f is not None:
                f["m"].zero_()
                f["v"].zero_()
        for st in self._units.values():
            st.zero_()
        for group, f, saved in zip(self.param_groups, self._flat, groups):
            if len(saved["params"]) != len(group["params"]):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
            for p, idx in zip(group["params"], saved["params"]):
                st = sd["state"].get(idx, sd["state"].get(str(idx)))
                w = self._where.get(id(p))
                if st is None or w is None:
                    continue
                _, s, _ = w
                k = p.numel()
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {idx}: shape {tuple(st['exp_avg'].shape)} does not match the parameter {tuple(p.shape)}")
                f["m"][s:s + k].copy_(st["exp_avg"].reshape(-1))
                f["v"][s:s + k].copy_(st["exp_avg_sq"].reshape(-1))
                unit = getattr(p, "dkd_lazy_unit", None)
                if unit in self._units:
                    self._units[unit][1] = float(st["step"])        # a lazy unit keeps its own count
                else:
                    steps.add(int(float(st["step"])))
            for key in ("lr", "betas", "eps", "weight_decay", "initial_lr"):
                if key in saved:
                    group[key] = tuple(saved[key]) if key == "betas" else saved[key]
        if len(steps) > 1:
            raise ValueError(f"FusedAdamW keeps one step count for all parameters; the checkpoint has {sorted(steps)}")
        self._step = steps.pop() if steps else 0


def _shadows_of(model):
    return [m._shadow for m in model.modules() if hasattr(m, "_shadow")]


def create_optimizer(args, model, filter_bias_and_bn=True):
    """timm.optim.create_optimizer(args, model) [3P] for ``--opt adamw`` (the only optimizer the exp/*.sh scripts use)."""
    opt = getattr(args, "opt", "adamw").lower()
    if opt != "adamw":
        raise ValueError(f"deltakd_amd.optim supports --opt adamw (got {opt!r})")
    wd = args.weight_decay
    inner = model.module if hasattr(model, "module") else model
    if wd and filter_bias_and_bn:
        skip = inner.no_weight_decay() if hasattr(inner, "no_weight_decay") else ()
        groups = param_groups_weight_decay(inner, wd, skip)
        wd = 0.
    else:
        groups = [{"params": [p for p in inner.parameters() if p.requires_grad]}]
    kw = dict(lr=args.lr, weight_decay=wd, eps=getattr(args, "opt_eps", None) or 1e-8)
    if getattr(args, "opt_betas", None):
        kw["betas"] = tuple(args.opt_betas)
    return FusedAdamW(groups, shadows=_shadows_of(inner), **kw)


class CosineLRScheduler:
    """timm.scheduler.CosineLRScheduler [3P] as ``create_scheduler`` configures it (t_in_epochs, one cycle, linear warm-up)."""

    def __init__(self, optimizer, t_initial, lr_min=0., warmup_t=0, warmup_lr_init=0.):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = t_initial, lr_min, warmup_t, warmup_lr_init
        self.base_values = [g["lr"] for g in optimizer.param_groups]
        for g, v in zip(optimizer.param_groups, self.base_values):
            g.setdefault("initial_lr", v)
        self._set([warmup_lr_init] * len(self.base_values) if warmup_t else self.base_values)

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (v - self.warmup_lr_init) / self.warmup_t for v in self.base_values]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def step(self, epoch, metric=None):
        self._set(self._get_lr(epoch))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def create_scheduler(args, optimizer):
    """timm.scheduler.create_scheduler(args, optimizer) [3P] for ``--sched cosine`` -> (scheduler, num_epochs)."""
    if getattr(args, "sched", "cosine") != "cosine":
        raise ValueError("deltakd_amd.optim supports --sched cosine")
    sched = CosineLRScheduler(optimizer, t_initial=args.epochs, lr_min=getattr(args, "min_lr", 1e-5),
                              warmup_t=getattr(args, "warmup_epochs", 5), warmup_lr_init=getattr(args, "warmup_lr", 1e-6))
    return sched, args.epochs + getattr(args, "cooldown_epochs", 10)
