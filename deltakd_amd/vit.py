"""ViT / DeiT on libdkd.so -- the models behind ``timm.create_model`` at /root/reference/model/models.py:60-68.

timm is absent on the MI355X boxes, so the product owns the model definitions.  Parameter names equal
timm==0.9.12's state-dict keys (cls_token, dist_token, pos_embed, patch_embed.proj.*, blocks.N.norm1.*,
blocks.N.attn.qkv.*, blocks.N.attn.proj.*, blocks.N.norm2.*, blocks.N.mlp.fc1.*, blocks.N.mlp.fc2.*, norm.*, head.*,
head_dist.*) so real checkpoints load from a local file.

Execution model (MI355X-first, not a module-per-op graph):
  * fp32 master parameters; bf16 shadow copies W and W^T feed the MFMA GEMMs (W^T turns every dgrad into the same
    K-contiguous NT kernel); the fp32 residual stream never leaves fp32.
  * ONE autograd node per transformer block (``_BlockFn``): forward = 7 kernel launches (LN, qkv GEMM+bias, attention,
    proj GEMM+bias+DropPath+residual, LN, fc1 GEMM+bias+GELU, fc2 GEMM+bias+tap+DropPath+residual); backward = 19.
    Parameter gradients are accumulated by the kernels straight into ``p.grad`` (atomics / split-M wgrad), so the
    nodes return no parameter gradients and ``zero_grad`` must zero, not drop, the grads
    (``deltakd_amd.optim`` does; ``ensure_grads`` allocates them otherwise).
  * the per-block feature tap of model/models.py:185-193 (``block.mlp`` output, before DropPath / residual) is written
    by the fc2 GEMM epilogue as a second output of the same kernel -- no forward hooks, no extra pass.
"""
import math
import os
import weakref
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import ffi, ops
from .ffi import IDENT, RowMap, strip_map

BF16, F32 = torch.bfloat16, torch.float32

REGISTRY = {
    # name: (embed_dim, depth, heads, distilled)            [timm registry, SURVEY.md Appendix B]
    "deit_tiny_patch16_224": (192, 12, 3, False),
    "deit_small_patch16_224": (384, 12, 6, False),
    "deit_base_patch16_224": (768, 12, 12, False),
    "deit_tiny_distilled_patch16_224": (192, 12, 3, True),
    "deit_small_distilled_patch16_224": (384, 12, 6, True),
    "deit_base_distilled_patch16_224": (768, 12, 12, True),
    "vit_large_patch16_224": (1024, 24, 16, False),
}


def _trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def ensure_grad(p: torch.Tensor) -> torch.Tensor:
    """Parameter gradients are written in place by the kernels: make sure the buffer exists (zero-filled)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


class Shadow:
    """bf16 copies (W, optionally W^T, optionally K-padded) of fp32 master weights, refreshed when the master changes.

    Staleness is detected from torch's version counter (in-place optimizer updates bump it) plus ``generation``, which the
    fused optimizer bumps because its kernel updates the masters behind torch's back.  When the fused optimizer is bound
    (``bind_flat``) it refreshes the plain bf16 copies itself, inside the AdamW kernel, and only W^T needs a launch here.
    """

    def __init__(self):
        self._slots = {}
        self._bound = {}          # id(param) -> bf16 view kept fresh by FusedAdamW
        self._params = {}         # id(param) -> param, for the bound ones
        self._fresh_versions = {} # id(param) -> p._version when the AdamW kernel last refreshed the bound view
        self.generation = 0
        self.bound_generation = -1

    def bind_flat(self, views, params=()):
        self._bound = dict(views)
        self._params = {id(p): p for p in params}
        self._slots.clear()

    def optimizer_stepped(self, bf16_fresh: bool):
        """The masters changed behind torch's back (raw kernel).  bf16_fresh: the same kernel re-cast the bound bf16 views; they
        are trusted only while the master's torch version stays what it is now (load_state_dict / in-place edits bump it)."""
        self.generation += 1
        if bf16_fresh:
            self.bound_generation = self.generation
            self._fresh_versions = {i: p._version for i, p in self._params.items()}

    def refresh_transposed(self):
        """Re-cast every plain W^T copy handed out so far in ONE launch (FusedAdamW calls this after its update: the ~50 student
        matrices are each far too small to fill the GPU as a launch of their own).  Copies requested for the first time, K-padded
        ones and conv weights still go through ``get``."""
        todo = [(k, v) for k, v in self._slots.items() if k[1] and not k[2] and not k[3] and len(v) > 2]
        if not todo:
            return
        keys = tuple(k for k, _ in todo)
        if getattr(self, "_t_keys", None) != keys or any(v[2].data_ptr() != v[0][2] for _, v in todo):
            pairs = [(v[2].detach().reshape(v[2].shape[0], -1), v[1]) for _, v in todo]
            self._t_table, self._t_keys = ops.cast_weight_table(pairs), keys
        ops.cast_weight_group(self._t_table)
        for k, v in todo:
            p = v[2]
            self._slots[k] = ((self.generation, p._version, p.data_ptr()), v[1], p)

    def get(self, p: torch.Tensor, transposed=False, pad_k_to=0, conv3x3=False):
        """conv3x3: p is a [out, cin, 3, 3] conv weight, served as [out, (ky, kx, cin)] -- the K order of the implicit-GEMM convolution
        (ops.gemm_nt conv_hw=) and of ops.im2col3x3.  conv3x3="dgrad": the operand of the INPUT gradient, the same convolution applied
        to dY with the taps flipped and the channel roles swapped: [cin, (2 - ky, 2 - kx, out)]."""
        key = (id(p), transposed, pad_k_to, conv3x3)
        slot = self._slots.get(key)
        stamp = (self.generation, p._version, p.data_ptr())
        if slot is not None and slot[0] == stamp:
            return slot[1]
        if conv3x3 == "dgrad":
            assert not transposed
            w2 = p.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(p.shape[1], -1).contiguous()
        elif conv3x3:
            w2 = p.detach().permute(0, 2, 3, 1).reshape(p.shape[0], -1).contiguous()
        else:
            w2 = p.detach().reshape(p.shape[0], -1)
        rows, cols = w2.shape
        if not transposed:
            bound = None if conv3x3 else self._bound.get(id(p))
            if (bound is not None and self.bound_generation == self.generation
                    and self._fresh_versions.get(id(p), p._version) == p._version):
                buf = bound.view(rows, cols)                      # already refreshed by the AdamW kernel
            else:
                buf = slot[1] if slot else (bound.view(rows, cols) if bound is not None else
                                            torch.empty(rows, cols, device=p.device, dtype=BF16))
                ops.cast_weight(w2, buf, None)
        else:
            ld = max(rows, pad_k_to)
            if ld != rows:   # K-padded transpose (dgrad whose K = out_features is not a multiple of 64)
                buf = slot[1] if slot else torch.zeros(cols, ld, device=p.device, dtype=BF16)
                tmp = torch.empty(cols, rows, device=p.device, dtype=BF16)
                ops.cast_weight(w2, None, tmp)
                buf[:, :rows].copy_(tmp)
            else:
                buf = slot[1] if slot else torch.empty(cols, rows, device=p.device, dtype=BF16)
                ops.cast_weight(w2, None, buf)
                self._slots[key] = (stamp, buf, p)       # with the parameter: refresh_transposed() re-casts these in one launch
                return buf
        self._slots[key] = (stamp, buf)
        return buf

    def clear(self):
        self._slots.clear()
        self._bound = {}
        self._params = {}
        self._fresh_versions = {}


# ----------------------------------------------------------------------------------------------- parameter holders
def _pad64(n):
    return (n + 63) // 64 * 64


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the MFMA GEMMs, for callers that use an aux Linear the way the reference does
    (``student_model.align[i](feat[:, 1:])``, model/loss.py:88-92,190,426).  x: any float dtype / strides, [..., in] -> f32 [..., out].
    The fused loss terms (deltakd_amd.losses._AlignTermFn) do not come through here: they read the tap through a row map and keep
    the bf16 shadows; this is the compatible path, with per-call casts of the weight."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K, N = weight.shape[1], weight.shape[0]
        lead = x.shape[:-1]
        Kp, Np = _pad64(K), _pad64(N)
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        if Kp == K:
            xb = x2.to(BF16).contiguous()
        else:
            xb = torch.zeros(M, Kp, device=x.device, dtype=BF16)
            xb[:, :K] = x2
        w = weight.detach().contiguous()
        wb = torch.empty(N, K, device=x.device, dtype=BF16)
        wt = torch.empty(K, N, device=x.device, dtype=BF16)
        ops.cast_weight(w, wb, wt)
        if Kp != K:
            wp = torch.zeros(N, Kp, device=x.device, dtype=BF16)
            wp[:, :K] = wb
            wb = wp
        if Np != N:                                      # dgrad contracts over out_features: K-pad W^T
            wtp = torch.zeros(K, Np, device=x.device, dtype=BF16)
            wtp[:, :N] = wt
            wt = wtp
        out = ops.gemm_nt(xb, wb, bias=None if bias is None else bias.detach(), out_f32=True)
        ctx.saved = (xb, wt)
        ctx.dims = (lead, M, K, N, Kp, Np, x.dtype, bias is not None)
        return out.view(*lead, N)

    @staticmethod
    def backward(ctx, g):
        xb, wt = ctx.saved
        lead, M, K, N, Kp, Np, xdtype, has_bias = ctx.dims
        g2 = g.reshape(M, N)
        if Np == N:
            gb = g2.to(BF16).contiguous()
        else:
            gb = torch.zeros(M, Np, device=g.device, dtype=BF16)
            gb[:, :N] = g2
        dw = torch.zeros(N, Kp, device=g.device, dtype=F32)
        db = torch.zeros(N, device=g.device, dtype=F32) if has_bias else None
        ops.gemm_tn(gb, xb, dw, N1=N, colsum=db)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_nt(gb, wt, out_f32=True).view(*lead, K).to(xdtype)
        ctx.saved = None
        return dx, dw[:, :K], db


class Linear(nn.Module):
    """nn.Linear's layout (weight [out, in], bias [out]) and call contract, computed by libdkd's GEMMs (no torch matmul)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features)) if bias else None
        _trunc_normal_(self.weight, std=.02)

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)

    def extra_repr(self):
        return f"{self.in_features}, {self.out_features}"


class LayerNormP(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _ConvP(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        conv = nn.Conv2d(cin, cout, k, k)            # PyTorch's default Conv2d init, like timm's PatchEmbed
        self.weight = nn.Parameter(conv.weight.detach().clone())
        self.bias = nn.Parameter(conv.bias.detach().clone())


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch
        self.num_patches = (img_size // patch) ** 2
        self.proj = _ConvP(in_chans, dim, patch)


class Attention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        assert dim == heads * 64, "the attention kernels are built for head_dim 64 (every model in scope)"
        self.num_heads = heads
        self.qkv = Linear(dim, dim * 3)
        self.proj = Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = Linear(dim, hidden)
        self.fc2 = Linear(hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, drop_path):
        super().__init__()
        self.norm1 = LayerNormP(dim)
        self.attn = Attention(dim, heads)
        self.norm2 = LayerNormP(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.drop_prob = float(drop_path)


# ----------------------------------------------------------------------------------------------- block execution
# A block is ONE call into libdkd (dkd_blocks_fwd / dkd_block_bwd, csrc/block.hip): the library walks the kernel launches.
# Python only allocates the activation slabs (2-3 allocations per block) and fills the descriptor.
def _bytes_al(n_elems, itemsize):
    return (n_elems * itemsize + 255) // 256 * 256


def mlp_fusion_enabled(D, hidden):
    """The fused MLP kernels (csrc/mlp192.hip) take the DeiT-tiny shape; DKD_NO_MLP_FUSION=1 keeps the separate launches (A/B)."""
    return D == 192 and hidden % 64 == 0 and not os.environ.get("DKD_NO_MLP_FUSION")


def attn_fusion_enabled(D, H, N):
    """The fused qkv + attention kernel (csrc/attn192.hip) takes the DeiT-tiny shape; DKD_NO_ATTN_FUSION=1 keeps the two launches (A/B)."""
    return D == 192 and H == 3 and N <= 208 and not os.environ.get("DKD_NO_ATTN_FUSION")


def _fill_weights(bs: ffi.Block, blk: Block, sh: Shadow, B, N, backward: bool):
    D = blk.norm1.weight.numel()
    bs.B, bs.N, bs.D, bs.H, bs.hidden, bs.eps = B, N, D, blk.attn.num_heads, blk.mlp.fc1.out_features, blk.norm1.eps
    a, m = blk.attn, blk.mlp
    bs.ln1_w, bs.ln1_b, bs.ln2_w, bs.ln2_b = (blk.norm1.weight.data_ptr(), blk.norm1.bias.data_ptr(), blk.norm2.weight.data_ptr(),
                                              blk.norm2.bias.data_ptr())
    bs.qkv_b, bs.proj_b, bs.fc1_b, bs.fc2_b = a.qkv.bias.data_ptr(), a.proj.bias.data_ptr(), m.fc1.bias.data_ptr(), m.fc2.bias.data_ptr()
    bs.qkv_w, bs.proj_w = sh.get(a.qkv.weight).data_ptr(), sh.get(a.proj.weight).data_ptr()
    bs.fc1_w, bs.fc2_w = sh.get(m.fc1.weight).data_ptr(), sh.get(m.fc2.weight).data_ptr()
    if not backward:                     # (the backward runs on the descriptor the forward filled: the flag is decided once)
        bs.fuse_attn = 1 if attn_fusion_enabled(D, blk.attn.num_heads, N) else 0
        bs.fuse_mlp = 1 if mlp_fusion_enabled(D, m.fc1.out_features) else 0
        if bs.fuse_mlp:
            bs.fc2_wt = sh.get(m.fc2.weight, transposed=True).data_ptr()     # the fused forward reads fc2's weight transposed
    if backward:
        bs.qkv_wt, bs.proj_wt = sh.get(a.qkv.weight, transposed=True).data_ptr(), sh.get(a.proj.weight, transposed=True).data_ptr()
        bs.fc1_wt, bs.fc2_wt = sh.get(m.fc1.weight, transposed=True).data_ptr(), sh.get(m.fc2.weight, transposed=True).data_ptr()


def _block_forward_train(x, B, N, blk: Block, sh: Shadow, s1, s2, want_tap: bool, ln1=None, next_blk=None):
    """x f32 [B*N, D] -> (x2, tap | None, ctx tuple holding the descriptor and the slabs that back its pointers, handoff).
    ``ln1`` = (y1, mean1, rstd1) when the previous block's fused MLP kernel has already applied this block's norm1 to x;
    ``next_blk``: with the fused MLP kernels, the block whose norm1 this one applies to its output -> handoff = its (y1, mean1, rstd1)."""
    M, D = x.shape
    Hd, H = blk.mlp.fc1.out_features, blk.attn.num_heads
    dev = x.device
    # bf16 slab: y1 | qkv | o | y2 | pre | h | tap ; f32 slab: x1 | x2 | mean1 | rstd1 | mean2 | rstd2 | lse
    Mp = (M + 15) // 16 * 16             # the fused MLP kernels store y2 / pre / h in whole 16-row groups
    sizes16 = [0 if ln1 is not None else M * D, M * 3 * D, M * D, Mp * D, Mp * Hd, Mp * Hd] + ([M * D] if want_tap else [])
    off16, tot = [], 0
    for n in sizes16:
        off16.append(tot)
        tot += _bytes_al(n, 2)
    slab16 = torch.empty(tot, device=dev, dtype=torch.uint8)
    sizes32 = [M * D, M * D, M, M, M, M, B * H * N]
    off32, tot = [], 0
    for n in sizes32:
        off32.append(tot)
        tot += _bytes_al(n, 4)
    slab32 = torch.empty(tot, device=dev, dtype=torch.uint8)
    p16, p32 = slab16.data_ptr(), slab32.data_ptr()
    bs = ffi.Block()
    _fill_weights(bs, blk, sh, B, N, backward=False)
    bs.s1, bs.s2 = ffi.ptr(s1), ffi.ptr(s2)
    bs.x, bs.x1, bs.x2 = x.data_ptr(), p32 + off32[0], p32 + off32[1]
    bs.mean1, bs.rstd1, bs.mean2, bs.rstd2, bs.lse = (p32 + off32[i] for i in (2, 3, 4, 5, 6))
    bs.y1, bs.qkv, bs.o, bs.y2, bs.pre, bs.h = (p16 + off16[i] for i in range(6))
    if ln1 is not None:
        bs.y1, bs.mean1, bs.rstd1, bs.ln1_ready = ln1[0].data_ptr(), ln1[1].data_ptr(), ln1[2].data_ptr(), 1
    handoff = None
    if next_blk is not None and bs.fuse_mlp:
        handoff = (torch.empty(M, D, device=dev, dtype=BF16), torch.empty(M, device=dev, dtype=F32), torch.empty(M, device=dev, dtype=F32))
        bs.next_ln1_w, bs.next_ln1_b = next_blk.norm1.weight.data_ptr(), next_blk.norm1.bias.data_ptr()
        bs.next_y1, bs.next_mean1, bs.next_rstd1 = (t.data_ptr() for t in handoff)
    tap = None
    if want_tap:
        bs.tap = p16 + off16[6]
        tap = slab16[off16[6]:off16[6] + M * D * 2].view(BF16).view(M, D)
    ffi.check(ffi.lib().dkd_blocks_fwd(ffi.C.byref(bs), 1, ffi.stream()), "block_fwd")
    x2 = slab32[off32[1]:off32[1] + M * D * 4].view(F32).view(M, D)
    return x2, tap, (bs, slab16, slab32, x, s1, s2, ln1), handoff


_RUNTIME = weakref.WeakKeyDictionary()      # model -> dict of per-process runtime objects (workspaces, streams, events): never copied
                                            # or pickled with the module


def _rt(model):
    d = _RUNTIME.get(model)
    if d is None:
        d = _RUNTIME[model] = {}
    return d


def _backward_workspace(model, bs, dev, which=0):
    """The scratch of dkd_block_bwd, shared by all blocks of a model (size from the library: dkd_block_bwd_workspace_bytes).  Two of
    them alternate when the weight gradients run on their own stream (they read dF / dH / dF2 / dqkv of block i while block i-1 runs)."""
    pool = _rt(model).setdefault("bwd_ws", {})
    ws = pool.get(which)
    need = ffi.lib().dkd_block_bwd_workspace_bytes(bs.B, bs.N, bs.D, bs.hidden)
    if ws is None or ws.numel() < need or ws.device != dev:
        ws = torch.empty(need, device=dev, dtype=torch.uint8)
        pool[which] = ws
    return ws


def _wgrad_stream(model, dev, fused_mlp=False):
    """Side stream for the four weight gradients of a block (one grouped launch: ~1.5 short blocks per CU that spend their time in
    the ring fill and in f32 atomics of partial tiles -- latency, not throughput): issued there, the launch overlaps the next
    block's backward instead of standing between two of its kernels.  DKD_NO_WGRAD_OVERLAP=1 keeps everything on one stream.
    With the fused MLP kernels (one 8-wave workgroup per CU holding 143 KB of LDS) there is nothing to overlap with: a wgrad workgroup
    and a fused-backward workgroup cannot share a CU, the two launches only delayed each other (in-situ 165 + 177 us against 91 + 100
    alone), so the weight gradients then go out inline (measured: student-only step 6.60 -> 6.48 ms); DKD_WGRAD_OVERLAP=1 forces the
    side stream for A/B."""
    if os.environ.get("DKD_NO_WGRAD_OVERLAP") or (fused_mlp and not os.environ.get("DKD_WGRAD_OVERLAP")):
        return None
    rt = _rt(model)
    st = rt.get("wgrad_side")
    if st is None or st.device != dev:
        st = rt["wgrad_side"] = torch.cuda.Stream(device=dev)
        rt["wgrad_done"], rt["bwd_parity"] = [None, None], 0
        rt["wgrad_events"] = [torch.cuda.Event(), torch.cuda.Event()]
    return st


def wgrad_stream_of(model):
    """The stream a model's weight gradients are issued on (None: the compute stream) -- data parallel waits on it."""
    return _rt(model).get("wgrad_side")


WGRAD_GROUP_MAX = 6           # blocks per deferred weight-gradient launch (4 problems each; dkd_block_wgrad_group takes 24)


def wgrad_group_size(model):
    """Blocks whose weight gradients share one launch.  The student walks its blocks backward and keeps the operands of up to this many
    alive (one backward workspace each); the flush is ONE dkd_block_wgrad_group: 6 blocks = 114 tiles on 4 M splits instead of 26 per
    block, one ring fill / atomic tail instead of six (student-only step 6.25 -> see DESIGN).  ``model._wgrad_group`` (data parallel
    sets 4 = its bucket size, so a bucket's all-reduce starts as soon as its blocks are done) or DKD_WGRAD_GROUP; 1 = inside
    dkd_block_bwd, as in round 2."""
    g = int(os.environ.get("DKD_WGRAD_GROUP", getattr(model, "_wgrad_group", WGRAD_GROUP_MAX)))
    return max(1, min(WGRAD_GROUP_MAX, g))


# The student and the teacher of a step embed the SAME mixed batch with the same patch size (tools/engine.py:37,48: the criterion is given
# the samples the student saw): the bf16 patch matrix [B * 196, 768] is gathered once per batch and shared -- the second model to ask
# (the student, a batch of lookahead later) waits on the event of the stream that produced it.  Keyed by the tensor OBJECT and its
# version counter, held weakly: a new batch, or the same storage rewritten in place, gathers again.  DKD_NO_SHARED_PATCHES=1 disables.
_PATCH_CACHE = {}            # id(tensor) -> (weakref to the tensor, key, patches, event, stream); dropped when the tensor dies


def _shared_patches(img, p):
    if os.environ.get("DKD_NO_SHARED_PATCHES"):
        return ops.im2col_patches(img, p)
    cur = torch.cuda.current_stream(img.device)
    key = (img._version, p, img.data_ptr(), tuple(img.shape))
    hit = _PATCH_CACHE.get(id(img))
    if hit is not None and hit[0]() is img and hit[1] == key:
        _, _, patches, ev, st = hit
        if st != cur.cuda_stream:
            cur.wait_event(ev)
            patches.record_stream(cur)
        return patches
    patches = ops.im2col_patches(img, p)
    ev = torch.cuda.Event()
    ev.record(cur)
    if hit is None:
        weakref.finalize(img, _PATCH_CACHE.pop, id(img), None)
    _PATCH_CACHE[id(img)] = (weakref.ref(img), key, patches, ev, cur.cuda_stream)
    return patches


def register_patches(img, p, patches):
    """The bf16 patch matrix of ``img`` produced by somebody else (the Mixup kernel writes it beside the mix) for ``_shared_patches``."""
    cur = torch.cuda.current_stream(img.device)
    ev = torch.cuda.Event()
    ev.record(cur)
    if id(img) not in _PATCH_CACHE:
        weakref.finalize(img, _PATCH_CACHE.pop, id(img), None)
    _PATCH_CACHE[id(img)] = (weakref.ref(img), (img._version, p, img.data_ptr(), tuple(img.shape)), patches, ev, cur.cuda_stream)


def _in_backward_pass():
    """Is the autograd engine executing a graph task on this thread?  (torch._C._current_graph_task_id() is -1 outside one.)"""
    fn = getattr(torch._C, "_current_graph_task_id", None)
    if fn is None:                # a torch without the accessor: assume we might be (pending gradients are then flushed, never lost)
        return True
    return fn() != -1


def flush_wgrads(model):
    """Launch the weight gradients deferred so far (no-op when there are none) and report their blocks to the data-parallel hook."""
    rt = _rt(model)
    pend = rt.get("wgrad_pending")
    if not pend:
        return
    rt["wgrad_pending"] = []
    n = 4 * len(pend)
    probs = (ffi.TnProblem * n)()
    k = 0
    for entry in pend:
        M = entry["M"]
        for pa, pb, pc, pcs, n1, n2 in entry["problems"]:
            q = probs[k]
            q.A, q.B, q.C, q.a_colsum, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc = pa, pb, pc, pcs, M, n1, n2, n1, n2, n2
            q.amap = q.bmap = IDENT
            k += 1
    ffi.check(ffi.lib().dkd_block_wgrad_group(ffi.C.cast(probs, ffi.C.c_void_p), n, ffi.stream()), "block wgrad group")
    red = [it for entry in pend for it in entry["ln"] if it.part]       # the LayerNorm dgamma / dbeta reductions of the same blocks
    for lo in range(0, len(red), 12):
        arr = (ffi.LnReduce * len(red[lo:lo + 12]))(*red[lo:lo + 12])
        ffi.check(ffi.lib().dkd_ln_bwd_reduce_group(ffi.C.cast(arr, ffi.C.c_void_p), len(arr), ffi.stream()), "ln reduce group")
    hook = getattr(model, "_grad_ready_hook", None)             # data parallel: these blocks' gradients are final now
    if hook is not None:
        for entry in pend:
            if entry["idx"] is not None:
                hook(entry["idx"])


def _block_backward(g, gtap, model, blk: Block, saved, idx=None):
    """g: f32 [M, D] gradient w.r.t. the block output (overwritten with the input gradient and returned).  Returns (g, deferred):
    ``deferred`` says the block's weight gradients wait in the model's pending list for flush_wgrads()."""
    bs, slab16, slab32, x, s1, s2, ln1 = saved
    _fill_weights(bs, blk, model._shadow, bs.B, bs.N, backward=True)
    side = _wgrad_stream(model, g.device, bool(bs.fuse_mlp))
    par = 0
    rt = _rt(model)
    group = wgrad_group_size(model) if side is None else 1
    pend = rt.setdefault("wgrad_pending", [])
    if side is not None:
        par = rt["bwd_parity"] = 1 - rt["bwd_parity"]
        done = rt["wgrad_done"][par]
        if done is not None:                       # the weight gradients that last read this workspace (two blocks ago)
            torch.cuda.current_stream().wait_event(done)
    elif group > 1:
        par = ("group", len(pend))                 # one workspace per pending block: dF / dH / dF2 / dqkv live until the flush
    ws = _backward_workspace(model, bs, g.device, par)
    gr = ffi.BlockGrads()
    ffi.check(ffi.lib().dkd_block_bwd_workspace_carve(ws.data_ptr(), bs.B, bs.N, bs.D, bs.hidden, ffi.C.byref(gr)), "bwd_workspace")
    gr.g, gr.gtap = g.data_ptr(), ffi.ptr(gtap)
    defer = side is not None or (group > 1 and bool(gr.dF2))
    gr.defer_wgrad = 1 if defer else 0
    ln_items = None
    if defer and side is None and not os.environ.get("DKD_NO_LN_REDUCE_GROUP"):
        ln_items = (ffi.LnReduce * 2)()
        gr.ln_defer = ffi.C.cast(ln_items, ffi.C.c_void_p)
    a, m = blk.attn, blk.mlp
    gr.d_ln1_w, gr.d_ln1_b = ensure_grad(blk.norm1.weight).data_ptr(), ensure_grad(blk.norm1.bias).data_ptr()
    gr.d_ln2_w, gr.d_ln2_b = ensure_grad(blk.norm2.weight).data_ptr(), ensure_grad(blk.norm2.bias).data_ptr()
    gr.d_qkv_w, gr.d_qkv_b = ensure_grad(a.qkv.weight).data_ptr(), ensure_grad(a.qkv.bias).data_ptr()
    gr.d_proj_w, gr.d_proj_b = ensure_grad(a.proj.weight).data_ptr(), ensure_grad(a.proj.bias).data_ptr()
    gr.d_fc1_w, gr.d_fc1_b = ensure_grad(m.fc1.weight).data_ptr(), ensure_grad(m.fc1.bias).data_ptr()
    gr.d_fc2_w, gr.d_fc2_b = ensure_grad(m.fc2.weight).data_ptr(), ensure_grad(m.fc2.bias).data_ptr()
    ffi.check(ffi.lib().dkd_block_bwd(ffi.C.byref(bs), ffi.C.byref(gr), ffi.stream()), "block_bwd")
    if not defer:
        return g, False
    M, D, Hd = bs.B * bs.N, bs.D, bs.hidden
    problems = ((gr.dF, bs.h, gr.d_fc2_w, gr.d_fc2_b, D, Hd), (gr.dH, bs.y2, gr.d_fc1_w, gr.d_fc1_b, Hd, D),
                (gr.dF2, bs.o, gr.d_proj_w, gr.d_proj_b, D, D), (gr.dqkv, bs.y1, gr.d_qkv_w, gr.d_qkv_b, 3 * D, D))
    if side is None:
        # (slab16 / ln1 hold h, y2, o, y1; kept alive here until the flush has been enqueued -- same stream, so that is enough)
        pend.append({"M": M, "problems": problems, "idx": idx, "keep": (slab16, ln1, ws, bs), "ln": ln_items if ln_items is not None else ()})
        return g, True
    probs = (ffi.TnProblem * 4)()
    for q, (pa, pb, pc, pcs, n1, n2) in zip(probs, problems):
        q.A, q.B, q.C, q.a_colsum, q.M, q.N1, q.N2, q.lda, q.ldb, q.ldc = pa, pb, pc, pcs, M, n1, n2, n1, n2, n2
        q.amap = q.bmap = IDENT
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ffi.check(ffi.lib().dkd_block_wgrad_group(ffi.C.cast(probs, ffi.C.c_void_p), 4, ffi.stream()), "block wgrad")
        ev = rt["wgrad_events"][par]
        ev.record(side)
    rt["wgrad_done"][par] = ev
    slab16.record_stream(side)             # h, y2, o, y1 are read there after this function's caller drops them
    if ln1 is not None:
        ln1[0].record_stream(side)         # (y1 handed over by the previous block's fused MLP kernel lives outside the slab)
    return g, False


def ln_fold_supported(M, D, hidden):
    """Shapes for which a block's LayerNorms can be folded into the GEMMs around them (include/dkd.h, DkdGemm.xb): the consumers (qkv,
    fc1) must take libdkd's wide kernel -- N % 256 == 0 and >= 1024 tiles of 256 x 256, K = D >= 512 -- and the producers (proj, fc2) its
    f32-residual epilogues (N = D % 128 == 0).  True for the DeiT-base / ViT-L teachers at the training batch; DKD_NO_LN_FOLD=1 keeps
    the separate LayerNorm launches (A/B)."""
    if os.environ.get("DKD_NO_LN_FOLD") or D < 512 or D % 128 or D % 64:
        return False
    panels = (M + 255) // 256
    for n in (3 * D, hidden):
        if n % 256 or panels * (n // 256) < 1024 or M * D >= 2 ** 31 or n * D >= 2 ** 31:
            return False
    return True


def _folded_linear(model, norm, lin):
    """(bf16(gamma * W), W beta + b, row sums of the bf16 matrix) of a LayerNorm -> Linear pair, cached until a parameter changes."""
    cache = _rt(model).setdefault("ln_fold_w", {})
    key = (id(norm), id(lin))
    stamp = tuple((p._version, p.data_ptr()) for p in (norm.weight, norm.bias, lin.weight, lin.bias)) + (model._shadow.generation,)
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    with torch.no_grad():
        w = lin.weight.detach().float()
        wf = (w * norm.weight.detach().float()[None, :]).to(BF16).contiguous()
        bias = (w @ norm.bias.detach().float() + lin.bias.detach().float()).contiguous()
        csum = wf.float().sum(1).contiguous()
    cache[key] = (stamp, (wf, bias, csum))
    return wf, bias, csum


def _blocks_forward_infer(x, B, N, model, scales, want):
    """Inference pass over all blocks in ONE library call: in-place fp32 residual stream, activation buffers shared by all
    blocks, taps only for the blocks in ``want``.  Returns (x, taps list).  Where the shapes allow it (``ln_fold_supported``: the
    DeiT-base / ViT-L teachers at the training batch) the LayerNorms are folded into the GEMMs around them -- no LayerNorm launches
    except the first block's norm1."""
    blocks = model.blocks
    depth = len(blocks)
    M, D = x.shape
    Hd = blocks[0].mlp.fc1.out_features
    dev = x.device
    guard = _rt(model).setdefault("ln_fold_guard", {"calls": 0, "pending": None, "off": False, "worst": 0.0})
    fold = ln_fold_supported(M, D, Hd) and all(s is None for s in scales) and not guard["off"]
    # The descriptor table, the activation slab and the statistics buffers of a frozen model are the same from call to call: filling 12
    # descriptors (bf16 shadows, folded weights, ~40 ctypes fields each) cost ~0.28 ms of host time per teacher call, during which both
    # streams sat idle (kernel trace of round 3).  They are cached per (shape, taps, parameter state); a call only patches the pointers
    # of the residual stream and the taps.  Reusing the slab across calls is stream-ordered (one teacher call after the other).
    rt = _rt(model)
    plist = rt.get("infer_params")
    if plist is None or plist[0] != model._shadow.generation:
        plist = rt["infer_params"] = (model._shadow.generation, list(model.parameters()))
    key = (M, D, Hd, B, N, fold, tuple(sorted(want)), dev, plist[0], tuple((p._version, p.data_ptr()) for p in plist[1]),
           tuple(None if s is None else s.data_ptr() for s in scales))
    cached = rt.get("infer_table")
    if cached is None or cached[0] != key:
        sizes = [M * D, M * 3 * D, M * D, M * Hd]        # y (LN1 and LN2 outputs alias; with the fold: the bf16 copy of x), qkv, o, h
        off, tot = [], 0
        for n in sizes:
            off.append(tot)
            tot += _bytes_al(n, 2)
        slab = torch.empty(tot, device=dev, dtype=torch.uint8)
        p = slab.data_ptr()
        arr = (ffi.Block * depth)()
        keep = [slab]                                     # what the descriptors point to
        stats = torch.zeros(2 * depth, M, 2, device=dev, dtype=F32) if fold else None
        for i, blk in enumerate(blocks):
            bs = arr[i]
            _fill_weights(bs, blk, model._shadow, B, N, backward=False)
            bs.s1, bs.s2 = ffi.ptr(scales[2 * i]), ffi.ptr(scales[2 * i + 1])
            bs.y1 = bs.y2 = p + off[0]
            bs.qkv, bs.o, bs.h = p + off[1], p + off[2], p + off[3]
            if fold:
                bits = 2 | (1 if i > 0 else 0) | (4 if i + 1 < depth else 0)
                bs.ln_fold, bs.xb = bits, p + off[0]
                bs.stats1, bs.stats2 = stats[2 * i].data_ptr(), stats[2 * i + 1].data_ptr()
                if i + 1 < depth:
                    bs.stats_next = stats[2 * i + 2].data_ptr()
                if i > 0:
                    wf, bias, csum = _folded_linear(model, blk.norm1, blk.attn.qkv)
                    bs.qkv_w, bs.qkv_b, bs.qkv_c = wf.data_ptr(), bias.data_ptr(), csum.data_ptr()
                    keep.append((wf, bias, csum))
                wf, bias, csum = _folded_linear(model, blk.norm2, blk.mlp.fc1)
                bs.fc1_w, bs.fc1_b, bs.fc1_c = wf.data_ptr(), bias.data_ptr(), csum.data_ptr()
                keep.append((wf, bias, csum))
        cached = rt["infer_table"] = (key, arr, stats, keep)
    _, arr, stats, _ = cached
    if stats is not None:
        stats.zero_()
    taps = [None] * depth
    xp = x.data_ptr()
    # the wanted taps are slices of ONE tensor, in block order: a consumer that reduces all of them (the LRKD Gram matrices) can do so in
    # one batched launch (ops.gram_batched)
    slab = torch.empty(len(want), M, D, device=dev, dtype=BF16) if want else None
    nxt = 0
    for i in range(depth):
        bs = arr[i]
        bs.x = bs.x1 = bs.x2 = xp
        if i in want:
            taps[i] = slab[nxt]
            nxt += 1
            bs.tap = taps[i].data_ptr()
    # the activation slab / statistics buffers are reused from call to call: a call on ANOTHER stream than the previous one (a
    # teacher-stream prefetch followed by a main-stream fallback) first waits for that call's kernels
    cur = torch.cuda.current_stream(dev)
    last = rt.get("infer_done")
    if last is not None and last[0] != cur.cuda_stream:
        cur.wait_event(last[1])
    ffi.check(ffi.lib().dkd_blocks_fwd(arr, depth, ffi.stream()), "blocks_fwd")
    ev = last[1] if last is not None else torch.cuda.Event()
    ev.record(cur)
    rt["infer_done"] = (cur.cuda_stream, ev)
    redo = _ln_fold_check(guard, stats, D) if fold else False
    return x, taps, redo


LN_FOLD_MAX_OFFSET = float(os.environ.get("DKD_LN_FOLD_MAX_OFFSET", "3.0"))
LN_FOLD_CHECK_EVERY = 64


def _ln_fold_check(guard, stats, D):
    """Guard of the LayerNorm fold.  The fold rounds x to bf16 BEFORE it is centred: an element's rounding error is
    2^-9 |x_i| instead of 2^-9 |x_i - mu|, i.e. the error of the normalised row grows by sqrt(1 + (mu / sigma)^2) -- nothing for the
    near-zero-mean rows of a random-init teacher or for a few massive channels (those round relative to themselves either way), but
    a row with a COMMON offset |mu| >> sigma loses (mu / sigma) x in accuracy (tests/test_fullsize_gpu.py::
    test_layernorm_fold_rows_with_a_common_offset measures it).  The producers' row statistics are already in HBM, so the worst
    |mu| / sigma over all rows and layers costs one small reduction: it is taken on the model's first folded call (synchronously:
    that call is redone unfolded if it fails) and on every 64th call after it (read back asynchronously, acted on a few calls later).
    Above DKD_LN_FOLD_MAX_OFFSET (3: at most ~3.2x the unfolded path's rounding noise) the model drops back to the separate
    LayerNorm launches for good.  Returns True when THIS call must be redone."""
    n = guard["calls"]
    guard["calls"] = n + 1
    pend = guard["pending"]
    if pend is not None and pend[1].query():
        guard["pending"] = None
        guard["worst"] = max(guard["worst"], float(pend[0]))
        if float(pend[0]) > LN_FOLD_MAX_OFFSET:
            _ln_fold_disable(guard, float(pend[0]))
    if n % LN_FOLD_CHECK_EVERY:
        return False
    with torch.no_grad():
        mu = stats[..., 0] / D
        var = (stats[..., 1] / D - mu * mu).clamp_min(0.0)
        ratio = (mu.abs() * torch.rsqrt(var + 1e-6)).amax()
    if n == 0:
        worst = float(ratio)                       # one host sync, on the first call of the model's life
        guard["worst"] = worst
        if worst > LN_FOLD_MAX_OFFSET:
            _ln_fold_disable(guard, worst)
            return True
        return False
    host = torch.empty((), dtype=F32, pin_memory=True)
    host.copy_(ratio, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    guard["pending"] = (host, ev)
    return False


def _ln_fold_disable(guard, worst):
    import warnings
    guard["off"] = True
    warnings.warn(f"deltakd_amd: LayerNorm fold switched off for this model: rows with |mean| / std = {worst:.1f} > "
                  f"{LN_FOLD_MAX_OFFSET} (bf16 rounding before centring would cost that factor in accuracy); the separate LayerNorm "
                  f"kernels run from here on", RuntimeWarning)


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, idx, B, N, s1, s2, want_tap):
        blk = model.blocks[idx]
        ctx.set_materialize_grads(False)     # an unused tap must not cost a zero-filled [M, D] gradient
        # norm1 of this block may already have been applied by the previous block's fused MLP kernel (handed over through the runtime
        # dict: keyed by the tensor it belongs to), and this block's kernel does the same for the next one
        rt = _rt(model)
        ln1 = rt.pop("ln1_handoff", None)
        if ln1 is not None and (ln1[0] != idx or ln1[1] != x.data_ptr() or os.environ.get("DKD_NO_LN1_HANDOFF")):
            ln1 = None
        nxt = model.blocks[idx + 1] if idx + 1 < len(model.blocks) and not os.environ.get("DKD_NO_LN1_HANDOFF") else None
        x2, tap, saved, handoff = _block_forward_train(x, B, N, blk, model._shadow, s1, s2, want_tap, ln1[2] if ln1 else None, nxt)
        if handoff is not None:
            rt["ln1_handoff"] = (idx + 1, x2.data_ptr(), handoff)
        ctx.model, ctx.idx = model, idx
        ctx.saved = saved
        if tap is None:
            tap = x2.new_empty(0)
            ctx.mark_non_differentiable(tap)
        return x2, tap

    @staticmethod
    def backward(ctx, g, gtap):
        if ctx.saved is None:
            raise RuntimeError("deltakd_amd block: backward called twice (activations are released after the first pass)")
        if gtap is not None and gtap.numel() == 0:
            gtap = None
        if gtap is not None:
            gtap = gtap.contiguous()
        if g is None:
            g = torch.zeros_like(ctx.saved[3])
        g = g.contiguous()
        if g.dtype != F32:
            g = g.float()
        model = ctx.model
        gin, deferred = _block_backward(g, gtap, model, model.blocks[ctx.idx], ctx.saved, ctx.idx)
        ctx.saved = None
        side = wgrad_stream_of(model)
        if side is not None and ctx.idx == 0:       # last block of the backward walk: everything downstream (embedding backward,
            torch.cuda.current_stream().wait_stream(side)   # optimizer) sees all weight gradients
        if deferred:
            rt = _rt(model)
            if len(rt["wgrad_pending"]) >= wgrad_group_size(model) or ctx.idx == 0:
                flush_wgrads(model)                 # (calls the data-parallel hook for every block it covers)
            elif len(rt["wgrad_pending"]) == 1:
                # safety net for a backward pass that never reaches block 0 (a frozen prefix, a partial graph): whatever is still
                # pending goes out when the autograd engine finishes this pass (a no-op after a regular flush)
                torch.autograd.Variable._execution_engine.queue_callback(lambda model=model: flush_wgrads(model))
        else:
            hook = getattr(model, "_grad_ready_hook", None)     # data parallel: this block's gradients are final
            if hook is not None:
                hook(ctx.idx)
        return gin, None, None, None, None, None, None, None


class _EmbedFn(torch.autograd.Function):
    """patch-embed GEMM (+bias +pos_embed, scattered behind the prefix tokens) and prefix-token assembly."""

    @staticmethod
    def forward(ctx, anchor, model, img):
        x, patches = model._embed(img)
        ctx.model, ctx.patches, ctx.B = model, patches, img.shape[0]
        return x

    @staticmethod
    def backward(ctx, g):
        m = ctx.model
        B, N, D, npre = ctx.B, m.num_tokens, m.embed_dim, m.num_prefix_tokens
        P = N - npre
        g = g.contiguous()
        dxb = ops.scale_cast_bf16(g, M=B * P, xmap=strip_map(N, npre))
        w = m.patch_embed.proj.weight
        ops.gemm_tn(dxb, ctx.patches, ensure_grad(w).view(D, -1), colsum=ensure_grad(m.patch_embed.proj.bias))
        if npre == 1:                               # one prefix token: the kernel adds straight into cls_token.grad
            ops.embed_bwd(g, ensure_grad(m.cls_token).view(1, D), ensure_grad(m.pos_embed).view(N, D), B, N, D, npre)
        else:
            dtok = torch.zeros(npre, D, device=g.device, dtype=F32)
            ops.embed_bwd(g, dtok, ensure_grad(m.pos_embed).view(N, D), B, N, D, npre)
            ensure_grad(m.cls_token).view(-1).add_(dtok[0])
            ensure_grad(m.dist_token).view(-1).add_(dtok[1])
        ctx.patches = None
        return None, None, None


class _HeadFn(torch.autograd.Function):
    """final LayerNorm on the prefix tokens + classifier head(s); logits f32 [B, C] (x npre)."""

    @staticmethod
    def forward(ctx, x, model, B):
        m = model
        N, D, npre = m.num_tokens, m.embed_dim, m.num_prefix_tokens
        pmap = RowMap(npre, N, 0)
        ctx.set_materialize_grads(False)
        y, mean, rstd = ops.layernorm_fwd(x, m.norm.weight, m.norm.bias, M=B * npre, xmap=pmap)
        outs = []
        heads = [m.head] + ([m.head_dist] if m.distilled else [])
        for t, hd in enumerate(heads):
            outs.append(ops.gemm_nt(y, m._shadow.get(hd.weight), M=B, amap=RowMap(1, npre, t), bias=hd.bias, out_f32=True))
        ctx.model, ctx.B = m, B
        ctx.saved = (x, y, mean, rstd)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gz):
        m, B = ctx.model, ctx.B
        x, y, mean, rstd = ctx.saved
        N, D, npre, C = m.num_tokens, m.embed_dim, m.num_prefix_tokens, m.num_classes
        Cp = (C + 63) // 64 * 64
        heads = [m.head] + ([m.head_dist] if m.distilled else [])
        dy = torch.zeros(B * npre, D, device=x.device, dtype=F32)
        for t, hd in enumerate(heads):
            if gz[t] is None:
                continue
            dz = ops.cast_pad_bf16(gz[t].float() if gz[t].stride(-1) == 1 else gz[t].float().contiguous(), Cp)       # (one launch: zeros + strided cast-copy were two)
            ops.gemm_tn(dz, y, ensure_grad(hd.weight), M=B, N1=C, bmap=RowMap(1, npre, t), colsum=ensure_grad(hd.bias))
            # d y[b, t, :] = dz[b, :] @ W  (NT against the K-padded W^T shadow), scattered to row b*npre + t
            ops.gemm_nt(dz, m._shadow.get(hd.weight, transposed=True, pad_k_to=Cp), out=dy, cmap=RowMap(1, npre, t))
        g = torch.zeros_like(x)
        ops.layernorm_bwd(dy, x, m.norm.weight, mean, rstd, g, ensure_grad(m.norm.weight), ensure_grad(m.norm.bias), M=B * npre,
                          xmap=RowMap(npre, N, 0), dxmap=RowMap(npre, N, 0))
        ctx.saved = None
        return g, None, None


# ----------------------------------------------------------------------------------------------- the model
class VisionTransformer(nn.Module):
    def __init__(self, embed_dim=192, depth=12, num_heads=3, num_classes=1000, distilled=False, drop_path_rate=0.0,
                 img_size=224, patch_size=16, in_chans=3, mlp_ratio=4.0):
        super().__init__()
        self.embed_dim, self.num_classes, self.distilled = embed_dim, num_classes, distilled
        self.num_prefix_tokens = 2 if distilled else 1
        self.distilled_training = False
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_tokens = self.patch_embed.num_patches + self.num_prefix_tokens
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        if distilled:
            self.dist_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_tokens, embed_dim))
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, dpr[i]) for i in range(depth)])
        self.norm = LayerNormP(embed_dim)
        self.head = Linear(embed_dim, num_classes)
        if distilled:
            self.head_dist = Linear(embed_dim, num_classes)
        _trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        if distilled:
            _trunc_normal_(self.dist_token, std=.02)
        self._shadow = Shadow()
        self._keep = None
        self.tap_layers = None               # blocks whose mlp output forward_with_taps returns (None: all)
        # weights loaded into the fp32 masters (resume / finetune / eval-only use) must reach the bf16 copies the GEMMs read
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._shadow.optimizer_stepped(bf16_fresh=False))

    # -- timm surface
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "dist_token"}

    def set_distilled_training(self, enable=True):
        self.distilled_training = enable

    def set_droppath_keep(self, keep):
        """Inject the Bernoulli draws of DropPath (2 per block, [B] 0/1) instead of sampling them (parity tests).  ``keep``: a list used
        by every training forward, or an iterator yielding one such list per training forward (multi-step tests)."""
        if keep is None or hasattr(keep, "__next__"):
            self._keep = keep
        else:
            self._keep = [k.to(F32) for k in keep]

    def _apply(self, fn, *a, **k):
        self._shadow.clear()
        return super()._apply(fn, *a, **k)

    # -- pieces
    def _droppath_scales(self, B, device):
        if not self.training:
            return [None] * (2 * len(self.blocks))
        probs = [blk.drop_prob for blk in self.blocks for _ in range(2)]
        if self._keep is not None:
            keep = next(self._keep) if hasattr(self._keep, "__next__") else self._keep
            return [None if p == 0.0 else (keep[i].to(device=device, dtype=F32) / (1.0 - p)).contiguous() for i, p in enumerate(probs)]
        if max(probs) == 0.0:
            return [None] * len(probs)
        # (cached on the device: building it from the host list every step was a pageable host-to-device copy, which torch follows
        # with a stream synchronise -- the host could never run ahead of the GPU across a step boundary)
        kp = _rt(self).get("keep_prob")
        if kp is None or kp[0] != (tuple(probs), device):
            kp = _rt(self)["keep_prob"] = ((tuple(probs), device), (1.0 - torch.tensor(probs, device=device, dtype=F32)).contiguous())
        # ONE launch for the 2 x depth masks of a step (rand + lt + cast + div were four): a counter-based generator on the device,
        # seeded per step from torch's CPU generator (torch.manual_seed covers it; drawing the seed launches nothing)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        scale = ops.droppath_scales(kp[1], B, seed)
        return [None if p == 0.0 else scale[i] for i, p in enumerate(probs)]

    def _embed(self, img):
        if not img.is_cuda:
            raise RuntimeError("deltakd_amd.vit: input must live on an MI355X device (no CPU path)")
        img = img.contiguous().float()
        B = img.shape[0]
        N, D, npre = self.num_tokens, self.embed_dim, self.num_prefix_tokens
        P = N - npre
        patches = _shared_patches(img, self.patch_embed.patch_size)
        x = torch.empty(B * N, D, device=img.device, dtype=F32)
        pos = self.pos_embed.view(N, D)
        ops.gemm_nt(patches, self._shadow.get(self.patch_embed.proj.weight), out=x, bias=self.patch_embed.proj.bias,
                    cmap=strip_map(N, npre), resid=pos, rmap=RowMap(P, 0, npre))
        if npre == 1:
            tok = self.cls_token.view(1, D)
        else:
            # [cls; dist] as one [2, D] matrix.  For a FROZEN model (the teacher) it is built once per token version; a trainable
            # model's tokens are updated in place by the optimizer kernels (no version bump), so there it is rebuilt every call.
            frozen = not self.cls_token.requires_grad and not self.dist_token.requires_grad
            key = (self.cls_token._version, self.dist_token._version, self.cls_token.data_ptr(), self.dist_token.data_ptr())
            hit = _rt(self).get("prefix_tok") if frozen else None
            if hit is None or hit[0] != key:
                tok = torch.cat([self.cls_token.view(1, D), self.dist_token.view(1, D)], 0)
                if frozen:
                    _rt(self)["prefix_tok"] = (key, tok)
            else:
                tok = hit[1]
        ops.prefix_tokens_fwd(x, tok.detach().contiguous(), pos, B, N, D, npre)
        return x, patches

    def forward_tokens(self, img, tap_layers: Optional[Sequence[int]] = None):
        """-> (x f32 [B*N, D] after the last block, taps: list (len depth) of bf16 [B, N, D] or None)."""
        B = img.shape[0]
        N = self.num_tokens
        depth = len(self.blocks)
        want = set(range(depth)) if tap_layers is None else {i % depth for i in tap_layers}
        taps: List[Optional[torch.Tensor]] = [None] * depth
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        scales = self._droppath_scales(B, img.device)
        if grad:
            stale = _rt(self).get("wgrad_pending")
            if stale:
                # weight gradients of up to six blocks wait for their grouped launch.  A grad-mode forward in the MIDDLE of a backward
                # pass (activation-checkpoint recompute, a forward inside a hook: the autograd engine is executing a graph task on this
                # thread) must not lose them -- flush first.  Outside a backward pass they are what a pass that died half-way left
                # behind: DROPPED, not flushed -- a loop that calls zero_grad() BEFORE the forward would otherwise get the aborted
                # pass's partial weight gradients added into the next step.
                if _in_backward_pass():
                    flush_wgrads(self)
                else:
                    _rt(self)["wgrad_pending"] = []
            x = _EmbedFn.apply(self.pos_embed, self, img)
            for i in range(depth):
                x, tap = _BlockFn.apply(x, self, i, B, N, scales[2 * i], scales[2 * i + 1], i in want)
                if i in want:
                    taps[i] = tap.view(B, N, self.embed_dim)
        else:
            x, _ = self._embed(img)
            x, flat_taps, redo = _blocks_forward_infer(x, B, N, self, scales, want)
            if redo:                                  # the LayerNorm-fold guard tripped on this model's first call: again, unfolded
                x, _ = self._embed(img)
                x, flat_taps, _ = _blocks_forward_infer(x, B, N, self, scales, want)
            for i, tap in enumerate(flat_taps):
                if tap is not None:
                    taps[i] = tap.view(B, N, self.embed_dim)
        return x, taps

    def forward_head(self, x, B):
        if torch.is_grad_enabled() and x.requires_grad:
            outs = _HeadFn.apply(x, self, B)
        else:
            N, D, npre = self.num_tokens, self.embed_dim, self.num_prefix_tokens
            y, _, _ = ops.layernorm_fwd(x, self.norm.weight, self.norm.bias, M=B * npre, xmap=RowMap(npre, N, 0), save_stats=False)
            heads = [self.head] + ([self.head_dist] if self.distilled else [])
            outs = tuple(ops.gemm_nt(y, self._shadow.get(hd.weight), M=B, amap=RowMap(1, npre, t), bias=hd.bias, out_f32=True)
                         for t, hd in enumerate(heads))
        if not self.distilled:
            return outs[0]
        if self.distilled_training and self.training:
            return outs[0], outs[1]
        return (outs[0] + outs[1]) / 2

    def forward_with_taps(self, img, tap_layers=None, head=True):
        """``tap_layers`` None: the model's ``tap_layers`` attribute (default None = every block, what the reference's
        forward_with_features returns); the training loop narrows it to the blocks its criterion reads.  ``head=False``: the logits
        are not computed (returned as None) -- the feature-matching criteria never read the teacher's."""
        x, taps = self.forward_tokens(img, self.tap_layers if tap_layers is None else tap_layers)
        return (self.forward_head(x, img.shape[0]) if head else None), taps

    def forward(self, img):
        x, _ = self.forward_tokens(img, tap_layers=())
        return self.forward_head(x, img.shape[0])


def create_model(name, pretrained=False, drop_path_rate=0.0, num_classes=1000, **kw):
    """Counterpart of ``timm.create_model`` for the names the reference scripts use (exp/*.sh).

    ``pretrained=True`` cannot fetch weights (no network on the MI355X boxes): pass ``checkpoint_path=<local file>`` to load a
    timm state dict, otherwise the (frozen) teacher keeps its seeded random init -- the throughput / parity runs use that.
    """
    ckpt = kw.pop("checkpoint_path", None)
    if name not in REGISTRY:
        raise ValueError(f"unknown model {name!r}; known: {sorted(REGISTRY)}")
    D, depth, H, dist = REGISTRY[name]
    model = VisionTransformer(D, depth, H, num_classes, dist, drop_path_rate, **kw)
    if ckpt:
        sd = torch.load(ckpt, map_location="cpu", weights_only=True)
        sd = sd.get("model", sd)
        own = model.state_dict()
        sd = {k: v for k, v in sd.items() if k in own and v.shape == own[k].shape}   # head re-initialised on class mismatch
        model.load_state_dict(sd, strict=False)
    return model
