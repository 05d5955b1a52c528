"""Masking helpers -- drop-in for /root/reference/model/misc.py:5-32 (``random_masking``).

The argsort-of-noise index plumbing stays on torch (tiny int64 tensors, [B, 196]); the heavy part of the reference's
gather -> cat(mask_token) -> gather-restore sequence (model/loss.py:433-440) is collapsed into one select kernel
(``ops.mask_select``): it equals ``where(mask, mask_token, x)`` with ``mask = ids_restore >= len_keep``.
"""
import torch


def masking_indices(noise, mask_ratio):
    """noise [B, L] -> (mask f32 [B, L] (1 = masked), ids_restore, ids_shuffle, len_keep): model/misc.py:12-30."""
    B, L = noise.shape
    len_keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    mask = (ids_restore >= len_keep).to(torch.float32)
    return mask, ids_restore, ids_shuffle, len_keep


def random_masking(x, mask_ratio, noise=None):
    """Same contract as model/misc.py:5-32: -> (x_keep, mask, ids_restore, ids_masked).  ``noise`` may be injected."""
    N, L, D = x.shape
    if noise is None:
        noise = torch.rand(N, L, device=x.device)
    mask, ids_restore, ids_shuffle, len_keep = masking_indices(noise, mask_ratio)
    ids_keep = ids_shuffle[:, :len_keep]
    x_keep = torch.gather(x, dim=1, index=ids_keep.unsqueeze(-1).repeat(1, 1, D))
    return x_keep, mask, ids_restore, ids_shuffle[:, len_keep:L]


@torch.no_grad()
def saliency_scores(student_model, teacher_feat, method, n_prefix=2):
    """The per-patch score that model/misc.py:38-165 argsorts (lowest scores are KEPT).  teacher_feat: [B, N_t, Dt] incl. prefix."""
    attn = student_model.saliency_attn
    t = teacher_feat.float()
    if method == 1:
        return attn(t[:, n_prefix:])
    cls_patch = torch.cat([t[:, :1], t[:, n_prefix:]], dim=1)
    if method == 2:
        B, L, D = cls_patch.shape
        H = attn.num_heads
        q, k = torch.chunk(attn.qk(cls_patch), 2, dim=-1)
        q = q.reshape(B, L, H, D // H).permute(0, 2, 1, 3)
        k = k.reshape(B, L, H, D // H).permute(0, 2, 1, 3)
        a = ((q[:, :, 0:1] @ k.transpose(-2, -1)) * (D // H) ** -0.5).softmax(dim=-1)
        return a.mean(dim=1).squeeze(1)[:, 1:]
    if method == 3:
        w = attn(cls_patch[:, :1], cls_patch[:, 1:])
        return w.squeeze(1) if w.dim() == 3 else w
    raise ValueError(f"Invalid saliency masking method: {method}")


def saliency_masking(student_model, teacher_feat, student_feat, mask_ratio, method):
    """Same contract as model/misc.py:38-165: -> (x_keep, mask, ids_restore)."""
    scores = saliency_scores(student_model, teacher_feat, method)
    mask, ids_restore, ids_shuffle, len_keep = masking_indices(scores, mask_ratio)
    D = student_feat.shape[-1]
    x_keep = torch.gather(student_feat, dim=1, index=ids_shuffle[:, :len_keep].unsqueeze(-1).expand(-1, -1, D))
    return x_keep, mask, ids_restore
