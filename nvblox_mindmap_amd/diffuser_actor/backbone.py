"""Image backbone stand-in.

The reference's default image encoder is a frozen RADIO v2.5-B (ViT-B/16-class, ~100 M parameters, 768-d tokens on a
16x16-pixel patch grid; mindmap/image_processing/feature_extraction.py:339-370), fetched from torch.hub.  There is no
network here, so the benchmark uses a randomly initialised ViT-B/16 of the same shape: identical compute and memory
profile for the training step (a frozen forward over B x ncam 512x512 images), not identical features.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import split_linear as SL
from .split_linear import split_linear


def _lin(x, layer, split):
    """``layer(x)``; ``split``: as one fp16 GEMM of split operands at f32 accuracy (split_linear.py; frozen inference only)."""
    return split_linear(x, layer) if split else layer(x)


def split_linear_rows(x):
    """split3 of a contiguous float32 [.., K] tensor (mmf_split_activations3)."""
    from .. import _lib

    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    a3 = torch.empty((x2.shape[0], 3 * K + SL.kTail), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().mmf_split_activations3(_lib.dptr(x2), x2.shape[0], K, _lib.dptr(a3), _lib.stream_ptr(x.device)), "mmf_split_activations3")
    return a3


class _Block(nn.Module):
    split_attention = True  # the fused path's attention on the split-operand matrix-core kernel (False: torch's f32 SDPA)

    def __init__(self, dim, heads, mlp_ratio=4):
        super().__init__()
        self.n1, self.n2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.qkv, self.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.fc1, self.fc2 = nn.Linear(dim, mlp_ratio * dim), nn.Linear(mlp_ratio * dim, dim)
        self.heads = heads

    def forward(self, x, split=False):
        B, L, D = x.shape
        q, k, v = _lin(self.n1(x), self.qkv, split).view(B, L, 3, self.heads, D // self.heads).permute(2, 0, 3, 1, 4)
        x = x + _lin(F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, L, D), self.proj, split)
        return x + _lin(F.gelu(_lin(self.n2(x), self.fc1, split)), self.fc2, split)

    def forward_fused(self, x, pending):
        """The same block on the fused element-wise kernels (split_linear.py: LayerNorm / GELU / head transpose write the split
        operands of the next GEMM directly; the residual adds ride in the LayerNorm passes).  ``pending``: the previous block's last
        branch, not yet added to ``x``.  Returns (x after the attention branch, this block's MLP branch -- the next ``pending``)."""
        B, L, D = x.shape
        x, a3 = SL.ln_split3(x, pending, self.n1)
        hd = D // self.heads
        qkv = SL.mm3(a3, self.qkv, (B, L, 3, self.heads, hd))
        if self.split_attention and SL.attention_split_ok(B, L, self.heads, hd):
            # f32-accurate attention on the fp16 matrix cores, output already in the [B, L, D] layout the projection reads
            a3 = SL.attention_split(qkv, B, L, self.heads, hd, split_out=True)
        else:
            q, k, v = qkv.permute(2, 0, 3, 1, 4)
            a3 = SL.split3_heads(F.scaled_dot_product_attention(q, k, v))
        x, a3 = SL.ln_split3(x, SL.mm3(a3, self.proj, (B, L, D)), self.n2)
        h = SL.mm3(a3, self.fc1, (B * L, self.fc1.out_features))
        return x, SL.mm3(SL.gelu_split3(h), self.fc2, (B, L, D))


class VitBackbone(nn.Module):
    """ViT-B/16-shaped encoder: (B,3,H,W) in [0,1] -> (B,dim,H/16,W/16)."""

    def __init__(self, dim: int = 768, depth: int = 12, heads: int = 12, patch: int = 16, max_grid: int = 32):
        super().__init__()
        self.patch, self.dim = patch, dim
        self.embed = nn.Conv2d(3, dim, patch, patch)
        self.pos = nn.Parameter(torch.zeros(1, max_grid * max_grid, dim))
        nn.init.trunc_normal_(self.pos, std=0.02)
        self.blocks = nn.ModuleList([_Block(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim)
        self.split_gemm = False  # frozen inference on a GPU: every Linear as one fp16 GEMM of split operands (split_linear.py)
        self.fused_elementwise = True  # ... and the LayerNorm / GELU / residual / transpose passes between them fused with the split

    def forward(self, x):
        B, _, H, W = x.shape
        h, w = H // self.patch, W // self.patch
        # Patch embedding as reshape + GEMM: a stride-16 16x16 convolution touches every pixel once, and MIOpen has only its
        # naive kernel for it (1 ms per 512x512 image).  Same parameters (self.embed), same sums in a different order.
        P = self.patch
        patches = x.reshape(B, 3, h, P, w, P).permute(0, 2, 4, 1, 3, 5).reshape(B, h * w, 3 * P * P)
        t = F.linear(patches, self.embed.weight.reshape(self.dim, 3 * P * P), self.embed.bias)
        t = t + self.pos[:, : h * w]
        split = self.split_gemm and t.is_cuda and t.dtype == torch.float32 and not torch.is_grad_enabled()
        if split and self.fused_elementwise and all(SL.can_fuse(t, blk) for blk in self.blocks):
            pending = None
            for blk in self.blocks:
                t, pending = blk.forward_fused(t, pending)
            t = t + pending
        else:
            for blk in self.blocks:
                t = blk(t, split)
        return self.norm(t).transpose(1, 2).reshape(B, self.dim, h, w)
