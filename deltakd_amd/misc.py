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
    """The per-patch score that model/misc.py:38-165 argsorts (lowest scores are KEPT).  teacher_feat: [B, N_t, Dt] incl. prefix tokens
    (the teacher tap as it is: no token is copied out -- the scorer kernel addresses CLS / patch rows of each sample itself).
    Projections and scores run on libdkd at fp32 accuracy (deltakd_amd.models._project_f32, csrc/saliency.hip)."""
    from . import ops
    from .ffi import RowMap
    from .models import _project_f32
    attn = student_model.saliency_attn
    B, N, D = teacher_feat.shape
    H, L = attn.num_heads, N - n_prefix
    t2 = teacher_feat.reshape(B * N, D)
    if method in (1, 2):
        qk = _project_f32(t2, attn.qk)                            # every token once (the prefix rows are simply not addressed)
        q, k = qk[:, :D], qk[:, D:]
        if method == 1:                                            # self-attention among the patches, diagonal
            return ops.saliency_scores(q, k, B=B, L=L, H=H, q_rows_per_sample=N, k_rows_per_sample=N, q_first=n_prefix, k_first=n_prefix,
                                       diagonal=True)
        # CLS query against [CLS | patches]; the CLS key takes part in the softmax, the patch columns are the scores
        return ops.saliency_scores(q, k, B=B, L=L, H=H, q_rows_per_sample=N, k_rows_per_sample=N, q_first=0, k_first=n_prefix, diagonal=False,
                                   extra_key_row=0)
    if method == 3:                                                # cross attention: CLS query, patch keys
        q = _project_f32(t2, attn.q, row_map=RowMap(1, N, 0), M=B)
        k = _project_f32(t2, attn.k)
        return ops.saliency_scores(q, k, B=B, L=L, H=H, q_rows_per_sample=1, k_rows_per_sample=N, q_first=0, k_first=n_prefix, diagonal=False)
    raise ValueError(f"Invalid saliency masking method: {method}")


def saliency_masking(student_model, teacher_feat, student_feat, mask_ratio, method):
    """Same contract as model/misc.py:38-165: -> (x_keep, mask, ids_restore)."""
    scores = saliency_scores(student_model, teacher_feat, method)
    mask, ids_restore, ids_shuffle, len_keep = masking_indices(scores, mask_ratio)
    D = student_feat.shape[-1]
    x_keep = torch.gather(student_feat, dim=1, index=ids_shuffle[:, :len_keep].unsqueeze(-1).expand(-1, -1, D))
    return x_keep, mask, ids_restore
