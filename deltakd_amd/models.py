"""Model factory + feature tap -- drop-in for /root/reference/model/models.py.

``load_teacher_student_model`` (:59-178) and ``forward_with_features`` (:181-199) keep their signatures; the aux modules
bolted onto the student keep their attribute names (align, mask_token, generation, denoise_fn, align_wasskd) because they
define checkpoint keys.  Deliberate fixes, documented in DESIGN.md:
  * ``forward_with_features`` looks through a data-parallel wrapper for the block list but calls the WRAPPER's forward
    (the reference returns (None, None) for a DDP-wrapped student: SURVEY.md section 3.5);
  * no forward hooks: the fc2 GEMM epilogue writes the ``block.mlp`` output as a second result.
"""
import torch
import torch.nn as nn

from . import vit
from .vit import Linear

DATASET_NUM_CLASSES = {"cifar-100": 100, "cifar-10": 10, "imagenet-1k": 1000, "imagenet-21k": 21843, "stanford_cars": 196,
                       "caltech256": 256, "flowers": 102}     # dataset/datasets.py:10-46


def _linear_default(i, o):
    """Linear holder with nn.Linear's default (kaiming-uniform) init, as the reference's aux nn.Linear layers get."""
    m, ref = Linear(i, o), nn.Linear(i, o)
    with torch.no_grad():
        m.weight.copy_(ref.weight)
        m.bias.copy_(ref.bias)
    return m


class Conv3x3(nn.Module):
    """Parameter holder with nn.Conv2d(C, C, 3, padding=1)'s layout and default init (model/models.py:149-151)."""

    def __init__(self, cin, cout):
        super().__init__()
        conv = nn.Conv2d(cin, cout, kernel_size=3, padding=1)
        self.weight = nn.Parameter(conv.weight.detach().clone())
        self.bias = nn.Parameter(conv.bias.detach().clone())


class Generation(nn.Sequential):
    """Conv3x3 - ReLU - Conv3x3 with nn.Sequential's key names ("0.weight", "2.weight")."""

    def __init__(self, dim):
        super().__init__(Conv3x3(dim, dim), nn.ReLU(inplace=True), Conv3x3(dim, dim))


class DenoisingNetwork(nn.Module):
    """model/models.py:103-121 (key names net.0 / net.2 / time_embed.0 / time_embed.2)."""

    def __init__(self, dims):
        super().__init__()
        self.net = nn.Sequential(Linear(dims, dims * 2), nn.GELU(), Linear(dims * 2, dims), nn.Dropout(0.1))
        self.time_embed = nn.Sequential(Linear(1, dims), nn.GELU(), Linear(dims, dims))     # [B, 1] -> [B, dims]: callable (vit._LinearFn)
        for m in (self.net[0], self.net[2], self.time_embed[0], self.time_embed[2]):       # nn.Linear default init, as the reference's layers get
            ref = nn.Linear(m.in_features, m.out_features)
            with torch.no_grad():
                m.weight.copy_(ref.weight)
                m.bias.copy_(ref.bias)


def _project_f32(x, lin, row_map=None, M=None):
    """x W^T + b at fp32 accuracy on the bf16 MFMA GEMMs: W = hi + lo (bf16 pair: ~16 significant bits of the fp32 master), two
    accumulating passes; x likewise when it is not bf16 already (the teacher taps are).  x 2-D [rows, in] -> f32 [M or rows, out].
    The scorer weights receive no gradient (model/misc.py ranks with them under argsort only), so the split is cached per weight version."""
    from . import ops
    w = lin.weight
    key = (w._version, w.data_ptr())
    cache = getattr(lin, "_dkd_hilo", None)
    if cache is None or cache[0] != key:
        wf = w.detach().float()
        hi = wf.to(vit.BF16)
        lo = (wf - hi.float()).to(vit.BF16)
        cache = (key, hi.contiguous(), lo.contiguous())
        lin._dkd_hilo = cache
    _, w_hi, w_lo = cache
    kw = {} if row_map is None else {"amap": row_map}
    if x.dtype == vit.BF16:
        parts = [x.contiguous()]
    else:
        xf = x.float()
        x_hi = xf.to(vit.BF16)
        parts = [x_hi.contiguous(), (xf - x_hi.float()).to(vit.BF16).contiguous()]
    out = ops.gemm_nt(parts[0], w_hi, M=M, bias=lin.bias.detach(), out_f32=True, **kw)
    ops.gemm_nt(parts[0], w_lo, out=out, M=M, accumulate=True, **kw)
    if len(parts) > 1:
        ops.gemm_nt(parts[1], w_hi, out=out, M=M, accumulate=True, **kw)
    return out


def _never_trained(module):
    """Marks a scorer's parameters: the reference builds them with requires_grad = True, but no gradient ever reaches them (they only
    rank tokens under argsort), so torch.optim.AdamW skips them every step (``if p.grad is None: continue``) -- no update and NO weight
    decay.  deltakd_amd.optim.FusedAdamW leaves parameters with this mark out of its flat buffers for the same effect (its kernels
    write into pre-allocated zero-filled gradients, where 'never written' cannot be told from 'zero')."""
    for p in module.parameters():
        p.dkd_never_grad = True
    return module


class SimpleAttention(nn.Module):
    """model/models.py:38-56 (saliency scorer): head-averaged self-attention weights, their DIAGONAL per token.  Only ever used to RANK
    tokens (argsort), so it carries no gradient.  The [B N, C] x [C, 2 C] projection runs on the MFMA GEMMs at fp32 accuracy
    (``_project_f32``), the scores on libdkd's fp32 scorer kernel (csrc/saliency.hip): no torch matmul / softmax."""

    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.qk = _linear_default(dim, dim * 2)
        _never_trained(self)

    @torch.no_grad()
    def forward(self, x):
        from . import ops
        B, N, C = x.shape
        qk = _project_f32(x.reshape(B * N, C), self.qk)
        return ops.saliency_scores(qk[:, :C], qk[:, C:], B=B, L=N, H=self.num_heads, q_rows_per_sample=N, k_rows_per_sample=N, diagonal=True)


class SimpleCrossAttention(nn.Module):
    """model/models.py:14-35: head-averaged attention weights of the queries on the keys, [B, Nq, Nk]."""

    def __init__(self, dim, num_heads=8):
        super().__init__()
        self.num_heads, self.scale = num_heads, (dim // num_heads) ** -0.5
        self.q = _linear_default(dim, dim)
        self.k = _linear_default(dim, dim)
        _never_trained(self)

    @torch.no_grad()
    def forward(self, x_query, x_key):
        from . import ops
        B, Nq, C = x_query.shape
        Nk = x_key.shape[1]
        q = _project_f32(x_query.reshape(B * Nq, C), self.q)
        k = _project_f32(x_key.reshape(B * Nk, C), self.k)
        rows = [ops.saliency_scores(q, k, B=B, L=Nk, H=self.num_heads, q_rows_per_sample=Nq, k_rows_per_sample=Nk, q_first=i, diagonal=False)
                for i in range(Nq)]
        return torch.stack(rows, dim=1)


def attach_aux(student, teacher, distillation_type, args=None):
    """Bolt the method-specific trainable modules onto the student (model/models.py:76-176)."""
    kind = distillation_type.lower()
    ds, dt = student.embed_dim, teacher.embed_dim
    if kind == "lrkd":
        student.align = nn.ModuleList([_linear_default(ds, args.lrkd_rank) for _ in range(3)])
    elif kind in ("soft", "hard"):
        if hasattr(student, "set_distilled_training"):
            student.set_distilled_training(enable=True)
    elif kind == "diffkd":
        student.denoise_fn = DenoisingNetwork(dt)
        student.align = nn.ModuleList([_linear_default(ds, dt) for _ in range(3)])
    elif kind == "mgd":
        student.align = _linear_default(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = Generation(dt)
    elif kind == "wasskd":
        student.align_wasskd = nn.ModuleList([_linear_default(ds, dt) for _ in range(3)])
    elif kind == "saliency_mgd":                            # model/models.py:129-143
        student.align = _linear_default(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = Generation(dt)
        method = getattr(args, "saliency_method", 1)
        student.saliency_attn = SimpleCrossAttention(dt, num_heads=8) if method == 3 else SimpleAttention(dt, num_heads=8)
    elif kind == "vitkd":                                   # model/models.py:76-88
        student.align2 = nn.ModuleList([_linear_default(ds, dt) for _ in range(2)])
        student.align = _linear_default(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = Generation(dt)
    elif kind == "curkd":                                   # model/models.py:153-167
        student.curkd_align_early = nn.ModuleList([_linear_default(ds, dt) for _ in range(3)])
        student.curkd_align_mid = nn.ModuleList([_linear_default(ds, dt) for _ in range(4)])
        student.curkd_align_last = _linear_default(ds, dt)
        student.mask_token = nn.Parameter(torch.zeros(1, 1, dt))
        student.generation = Generation(dt)
        # each stage receives gradients only in its epochs (model/loss.py:376-420); outside them torch.optim.AdamW skips its parameters
        # (grad None): FusedAdamW does the same per unit (deltakd_amd.optim)
        for unit, mods in (("curkd_early", [student.curkd_align_early]), ("curkd_mid", [student.curkd_align_mid]),
                           ("curkd_last", [student.curkd_align_last, student.generation])):
            for m in mods:
                for p in m.parameters():
                    p.dkd_lazy_unit = unit
        student.mask_token.dkd_lazy_unit = "curkd_last"
    return student


def load_teacher_student_model(teacher_model_name, student_model_name, drop_path_rate=0.1, args=None):
    num_classes = DATASET_NUM_CLASSES[args.dataset]
    extra = getattr(args, "model_kwargs", None) or {}
    teacher_model = vit.create_model(teacher_model_name, pretrained=True, drop_path_rate=drop_path_rate, num_classes=num_classes,
                                     checkpoint_path=getattr(args, "teacher_checkpoint", None), **extra)
    student_model = vit.create_model(student_model_name, pretrained=False, drop_path_rate=drop_path_rate, num_classes=num_classes,
                                     **extra)
    teacher_model.eval()
    for param in teacher_model.parameters():
        param.requires_grad = False
    attach_aux(student_model, teacher_model, args.distillation_type, args)
    return teacher_model, student_model


def forward_with_features(model, x):
    """-> (model_output, [block.mlp output of every block]) ; (None, None) if the model has no ``blocks``."""
    inner = model
    while not hasattr(inner, "blocks") and hasattr(inner, "module"):
        inner = inner.module
    if not hasattr(inner, "blocks"):
        return None, None
    if inner is model:
        return model.forward_with_taps(x)
    return model(x, with_taps=True)     # data-parallel wrapper: its forward arms gradient sync, then taps come back
