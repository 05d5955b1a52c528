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
