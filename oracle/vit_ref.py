"""Pure-torch fp32 restatement of the ViT / DeiT models the reference trains.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- never imported by the product.

The reference obtains its models from the third-party ``timm==0.9.12``
(/root/reference/requirements.txt:28) through ``timm.create_model`` at
/root/reference/model/models.py:60-68; timm is absent from this image, so the
arithmetic below restates timm 0.9.12's ``VisionTransformer`` /
``VisionTransformerDistilled``: pre-LN blocks, fused qkv,
``F.scaled_dot_product_attention``, exact-erf GELU MLP, LayerNorm eps 1e-6,
per-sample DropPath.  Parameter names equal timm's state-dict keys.

Stochastic ops take their random draws as inputs (``set_droppath_keep``) so the
HIP path can be fed the identical draws.
"""
import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

# name -> (embed_dim, depth, heads, distilled)   [timm registry, SURVEY.md App. B]
REGISTRY = {
    "deit_tiny_patch16_224": (192, 12, 3, False),
    "deit_small_patch16_224": (384, 12, 6, False),
    "deit_base_patch16_224": (768, 12, 12, False),
    "deit_tiny_distilled_patch16_224": (192, 12, 3, True),
    "deit_small_distilled_patch16_224": (384, 12, 6, True),
    "deit_base_distilled_patch16_224": (768, 12, 12, True),
    "vit_large_patch16_224": (1024, 24, 16, False),
}


def trunc_normal_(t: torch.Tensor, std: float = 0.02) -> torch.Tensor:
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


class DropPathRef(nn.Module):
    """Per-sample stochastic depth with scale-by-keep (timm ``DropPath``)."""

    def __init__(self, drop_prob: float):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.keep: Optional[torch.Tensor] = None   # injected [B] 0/1 draw

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep_prob = 1.0 - self.drop_prob
        if self.keep is not None:
            r = self.keep.to(x.dtype).view(-1, 1, 1)
        else:
            r = x.new_empty(x.shape[0], 1, 1).bernoulli_(keep_prob)
        return x * (r / keep_prob)


class MlpRef(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class AttentionRef(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.head_dim = dim // heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        s = (q * self.head_dim ** -0.5) @ k.transpose(-2, -1)
        o = s.softmax(dim=-1) @ v
        return self.proj(o.transpose(1, 2).reshape(B, N, C))


class BlockRef(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = AttentionRef(dim, heads)
        self.drop_path1 = DropPathRef(drop_path)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = MlpRef(dim, int(dim * mlp_ratio))
        self.drop_path2 = DropPathRef(drop_path)

    def forward(self, x):
        x = x + self.drop_path1(self.attn(self.norm1(x)))
        x = x + self.drop_path2(self.mlp(self.norm2(x)))
        return x


class PatchEmbedRef(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)
        self.num_patches = (img_size // patch) ** 2

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class VisionTransformerRef(nn.Module):
    def __init__(self, embed_dim=192, depth=12, num_heads=3, num_classes=1000, distilled=False,
                 drop_path_rate=0.0, img_size=224, patch_size=16, in_chans=3, mlp_ratio=4.0):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_classes = num_classes
        self.num_prefix_tokens = 2 if distilled else 1
        self.distilled = distilled
        self.distilled_training = False
        self.patch_embed = PatchEmbedRef(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if distilled:
            self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        n_tok = self.patch_embed.num_patches + self.num_prefix_tokens
        self.pos_embed = nn.Parameter(torch.zeros(1, n_tok, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[BlockRef(embed_dim, num_heads, mlp_ratio, dpr[i]) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, num_classes)
        if distilled:
            self.head_dist = nn.Linear(embed_dim, num_classes)
        self.init_weights()

    def init_weights(self):
        trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        if self.distilled:
            trunc_normal_(self.dist_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def set_distilled_training(self, enable=True):
        self.distilled_training = enable

    def set_droppath_keep(self, keep: Optional[List[torch.Tensor]]):
        """keep: list of 2*depth tensors [B] (0/1), order (blk0.dp1, blk0.dp2, blk1.dp1, ...)."""
        for i, blk in enumerate(self.blocks):
            blk.drop_path1.keep = None if keep is None else keep[2 * i]
            blk.drop_path2.keep = None if keep is None else keep[2 * i + 1]

    def forward_features(self, x):
        x = self.patch_embed(x)
        pre = [self.cls_token.expand(x.shape[0], -1, -1)]
        if self.distilled:
            pre.append(self.dist_token.expand(x.shape[0], -1, -1))
        x = torch.cat(pre + [x], dim=1) + self.pos_embed
        x = self.blocks(x)
        return self.norm(x)

    def forward_head(self, x):
        if not self.distilled:
            return self.head(x[:, 0])
        a, b = self.head(x[:, 0]), self.head_dist(x[:, 1])
        if self.distilled_training and self.training:
            return a, b
        return (a + b) / 2

    def forward(self, x):
        return self.forward_head(self.forward_features(x))


def create_model_ref(name, num_classes=1000, drop_path_rate=0.0, **kw):
    D, depth, H, dist = REGISTRY[name]
    return VisionTransformerRef(D, depth, H, num_classes, dist, drop_path_rate, **kw)
