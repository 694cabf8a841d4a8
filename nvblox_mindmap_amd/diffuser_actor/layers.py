"""Attention building blocks of the policy: 3-D rotary position code, AdaLN-conditioned attention / feed-forward
blocks (counterparts of mindmap/diffuser_actor/{layers,multihead_custom_attention,position_encodings}.py;
batch-first, ``F.scaled_dot_product_attention``)."""
import math
import os
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .train_ops import adaln_modulate_train, adaln_usable, add_layer_norm
from .train_ops import linear as _tlin


# Inference fast path (DiffuserActor.enable_fused_inference): fused HIP ops instead of the composite torch ops wherever a
# block runs without autograd on CUDA float32 tensors.  Rotary / AdaLN are the same float operations; the attention core
# agrees with the SDPA math path to rounding.
FUSED_INFERENCE = False


def _fused(x: torch.Tensor) -> bool:
    return FUSED_INFERENCE and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


def _block_dims():
    from .fused_ops import BLOCK_DIMS

    return BLOCK_DIMS


FUSED_ROTARY_TRAINING = os.environ.get("MMF_FUSED_ROTARY", "1") != "0"  # apply_rotary as one kernel each way on CUDA float32


def sinusoidal_embedding(x: torch.Tensor, dim: int) -> torch.Tensor:
    """(B,) scalars -> (B,dim): [sin(x w_k) ... , cos(x w_k) ...], w_k = 10000^(-k/(dim/2-1))."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, device=x.device, dtype=torch.float32) * (-math.log(10000.0) / (half - 1)))
    arg = x.to(torch.float32)[:, None] * freq[None, :]
    return torch.cat([arg.sin(), arg.cos()], dim=-1)


@torch.no_grad()
def rotary3d(xyz: torch.Tensor, dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """3-D rotary code of points (B,N,3): the channel dimension is split into three thirds (x, y, z); inside a third
    channel pair (2k, 2k+1) rotates with frequency 10000^(-2k/(dim/3)).  Returns (cos, sin), each (B,N,dim)."""
    assert dim % 6 == 0, "embedding_dim must be divisible by 6"
    third = dim // 3
    freq = torch.exp(torch.arange(0, third, 2, device=xyz.device, dtype=torch.float32) * (-math.log(10000.0) / third))
    ang = xyz.to(torch.float32)[..., :, None] * freq  # (B,N,3,third/2)
    ang = ang.repeat_interleave(2, dim=-1).flatten(-2)  # pairs share an angle; x-third, y-third, z-third
    return ang.cos(), ang.sin()


def apply_rotary(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Rotate channel pairs (x_{2k}, x_{2k+1}) by the pair's angle."""
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.numel() > 0 and FUSED_ROTARY_TRAINING and not torch.is_autocast_enabled()
            and cos.dtype == torch.float32 and sin.dtype == torch.float32 and not (cos.requires_grad or sin.requires_grad)):
        # one kernel each way instead of seven forward + their autograd (same float operations, forward and backward)
        from .fused_ops import rotary_apply_train

        return rotary_apply_train(x, cos, sin)
    x_rot = torch.stack([-x[..., 1::2], x[..., 0::2]], dim=-1).flatten(-2)
    return x * cos + x_rot * sin


class AdaLNBatch:
    """The scale/shift projections of many AdaLN blocks from ONE GEMM: every block applies its own Linear to the same
    silu(cond), so their weights are stacked once and a forward pass computes all of them together (fused inference path)."""

    def __init__(self, modules):
        self.index = {id(m): i for i, m in enumerate(modules)}
        self.weight = torch.cat([m.proj.weight for m in modules], dim=0)
        self.bias = torch.cat([m.proj.bias for m in modules], dim=0)
        self.width = modules[0].proj.out_features
        self.weight_t = self.weight.t().contiguous()  # [D, NA]: the layout mmf_step_prologue reads coalesced
        self.all = None

    def compute(self, cond_act: torch.Tensor) -> "AdaLNBatch":
        self.all = F.linear(cond_act, self.weight, self.bias)
        return self

    def lookup(self, module) -> torch.Tensor:
        i = self.index[id(module)]
        return self.all[:, i * self.width:(i + 1) * self.width]


class AdaLN(nn.Module):
    """x * (1 + scale(c)) + shift(c), conditioning vector c: (B,dim); zero-initialised (starts as identity)."""

    def __init__(self, dim: int):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * dim)
        nn.init.zeros_(self.proj.weight)
        nn.init.zeros_(self.proj.bias)

    def forward(self, x: torch.Tensor, cond: torch.Tensor, cond_act=None) -> torch.Tensor:
        """cond_act = F.silu(cond) if the caller already has it (it is the same for every block of a forward pass), or an
        AdaLNBatch holding this block's projection already."""
        if isinstance(cond_act, AdaLNBatch):
            ss = cond_act.lookup(self)
        else:
            ss = self.proj(F.silu(cond) if cond_act is None else cond_act)
        if _fused(x) and x.dim() == 3:
            from .fused_ops import adaln_modulate

            return adaln_modulate(x, ss)
        if adaln_usable(x) and ss.dim() == 2:
            return adaln_modulate_train(x, ss)  # the training step: one kernel forward, one sweep + a small sum backward
        scale, shift = ss.chunk(2, dim=-1)
        return x * (1 + scale[:, None, :]) + shift[:, None, :]


class RelativeAttention(nn.Module):
    """Multi-head attention with rotary position codes applied to the projected queries / keys."""

    def __init__(self, dim: int, heads: int, dropout: float = 0.0):
        super().__init__()
        assert dim % heads == 0
        self.dim, self.heads, self.dropout = dim, heads, dropout
        self.q_proj = nn.Linear(dim, dim)
        self.kv_proj = nn.Linear(dim, 2 * dim)
        self.out_proj = nn.Linear(dim, dim)
        for lin in (self.q_proj, self.kv_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)
        nn.init.zeros_(self.out_proj.bias)

    def project_kv_heads(self, memory: torch.Tensor, kv_rot=None):
        """(head-major keys, transposed values, Lk) of a memory for the matrix-core attention kernel (fused inference;
        AttentionStack consumes 3-tuples in ``kv_caches`` this way)."""
        from .fused_ops import qkv_heads

        _, kh, vt = qkv_heads(memory, None, None, self.kv_proj, kv_rot, self.heads, roles=6)
        return kh, vt, memory.shape[1]

    def project_kv(self, memory: torch.Tensor, kv_rot=None):
        """Keys (rotated) and values of `memory` for the fused path: computed once per inference for a memory that does not
        change between denoising steps (the context of the cross-attention layers)."""
        from .fused_ops import BLOCK_DIMS, kv_block, rotary_apply

        D = self.dim
        if D in BLOCK_DIMS:
            return kv_block(memory, self.kv_proj, kv_rot)
        kv = self.kv_proj(memory)
        k = kv[..., :D]
        return (rotary_apply(k, *kv_rot) if kv_rot is not None else k), kv[..., D:]

    def forward(self, query: torch.Tensor, memory: torch.Tensor, q_rot=None, kv_rot=None,
                key_padding_mask: Optional[torch.Tensor] = None, need_weights: bool = False, kv_cache=None):
        """query (B,Lq,D), memory (B,Lk,D); key_padding_mask (B,Lk) True = ignore.  Returns (out, weights or None).
        kv_cache: (keys, values) from project_kv, fused inference path only."""
        B, Lq, D = query.shape
        Lk = memory.shape[1]
        if _fused(query) and not need_weights and (D // self.heads) in (8, 15, 16, 20, 24, 32):  # head dims the kernel is built for
            from .fused_ops import attention_small, rotary_apply

            q = self.q_proj(query)
            if q_rot is not None:
                q = rotary_apply(q, *q_rot)
            k, v = kv_cache if kv_cache is not None else self.project_kv(memory, kv_rot if q_rot is not None else None)
            return self.out_proj(attention_small(q, k, v, key_padding_mask, self.heads)), None
        q = _tlin(self.q_proj, query)
        k, v = _tlin(self.kv_proj, memory).chunk(2, dim=-1)
        if q_rot is not None:
            q = apply_rotary(q, *q_rot)
            k = apply_rotary(k, *kv_rot)
        h = self.heads
        if not need_weights and (self.dropout == 0.0 or not self.training):
            from . import train_attention as TA

            if TA.usable(q, D // h):
                # the training step: forward + backward on the f32 matrix cores, straight from the [B, L, D] projections
                return _tlin(self.out_proj, TA.train_attention(q, k, v, key_padding_mask, h)), None
        q = q.view(B, Lq, h, D // h).transpose(1, 2)
        k = k.view(B, Lk, h, D // h).transpose(1, 2)
        v = v.view(B, Lk, h, D // h).transpose(1, 2)
        mask = None
        if key_padding_mask is not None:
            mask = (~key_padding_mask)[:, None, None, :]  # True = attend
        weights = None
        if need_weights:
            logits = (q @ k.transpose(-1, -2)) / math.sqrt(D // h)
            if mask is not None:
                logits = logits.masked_fill(~mask, float("-inf"))
            weights = logits.softmax(dim=-1)
            out = F.dropout(weights, self.dropout, self.training) @ v
        else:
            hd = D // h
            if q.is_cuda and hd % 8 != 0 and Lq >= 64:
                # head dims that are not a multiple of 8 (15 here) fall back to SDPA's math path, which materialises the
                # [B,h,Lq,Lk] scores; zero-padding the channel axis is exact (zero products, padded output column dropped) and
                # lets the fused attention kernel run: 2.3x faster forward + backward at 616 x 616.  `scale` keeps 1/sqrt(hd).
                padc = (-hd) % 8
                out = F.scaled_dot_product_attention(F.pad(q, (0, padc)), F.pad(k, (0, padc)), F.pad(v, (0, padc)), attn_mask=mask,
                                                     dropout_p=self.dropout if self.training else 0.0, scale=1.0 / math.sqrt(hd))[..., :hd]
            else:
                out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=self.dropout if self.training else 0.0)
        out = out.transpose(1, 2).reshape(B, Lq, D)
        return self.out_proj(out), weights


class AttentionBlock(nn.Module):
    """(AdaLN on the query) -> attention -> residual -> LayerNorm  (post-norm)."""

    def __init__(self, dim: int, heads: int, dropout: float = 0.0, use_adaln: bool = False):
        super().__init__()
        self.attn = RelativeAttention(dim, heads, dropout)
        self.norm = nn.LayerNorm(dim)
        self.drop = nn.Dropout(dropout)
        self.adaln = AdaLN(dim) if use_adaln else None

    def forward(self, query, memory, cond=None, q_rot=None, kv_rot=None, key_padding_mask=None, need_weights=False, cond_act=None,
                kv_cache=None):
        A = self.attn
        if _fused(query) and not need_weights and A.dim in _block_dims() and (A.dim // A.heads) in (8, 15, 16, 20, 24, 32):
            # whole-block kernels: (modulate + q_proj + rotary) | (kv_proj + rotary) | attention | (out_proj + residual + LayerNorm)
            from . import fused_ops as FO

            ss = None
            if self.adaln is not None and cond is not None:
                ss = cond_act.lookup(self.adaln) if isinstance(cond_act, AdaLNBatch) else self.adaln.proj(F.silu(cond) if cond_act is None else cond_act)
            q = FO.q_block(query, ss, A.q_proj, q_rot)
            k, v = kv_cache if kv_cache is not None else FO.kv_block(memory, A.kv_proj, kv_rot if q_rot is not None else None)
            att = FO.attention_small(q, k, v, key_padding_mask, A.heads)
            return FO.attn_out_block(att, query, A.out_proj, self.norm), None
        q_in = self.adaln(query, cond, cond_act) if (self.adaln is not None and cond is not None) else query
        out, w = self.attn(q_in, memory, q_rot, kv_rot, key_padding_mask, need_weights, kv_cache)
        return add_layer_norm(query, self.drop(out), self.norm), w


class FeedForwardBlock(nn.Module):
    """(AdaLN) -> Linear -> ReLU -> Linear -> residual -> LayerNorm; hidden width = dim."""

    def __init__(self, dim: int, hidden: int, dropout: float = 0.0, use_adaln: bool = False):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)
        nn.init.xavier_uniform_(self.fc1.weight)
        nn.init.xavier_uniform_(self.fc2.weight)
        self.norm = nn.LayerNorm(dim)
        self.drop = nn.Dropout(dropout)
        self.adaln = AdaLN(dim) if use_adaln else None

    def forward(self, x, cond=None, cond_act=None):
        if _fused(x) and x.dim() == 3 and x.shape[-1] in _block_dims() and self.fc1.out_features == x.shape[-1]:
            from . import fused_ops as FO

            ss = None
            if self.adaln is not None and cond is not None:
                ss = cond_act.lookup(self.adaln) if isinstance(cond_act, AdaLNBatch) else self.adaln.proj(F.silu(cond) if cond_act is None else cond_act)
            return FO.ffn_block(x, ss, self.fc1, self.fc2, self.norm)
        if self.adaln is not None and cond is not None:
            x = self.adaln(x, cond, cond_act)
        return add_layer_norm(x, self.drop(_tlin(self.fc2, self.drop(F.relu(_tlin(self.fc1, x))))), self.norm)


class AttentionStack(nn.Module):
    """`num_layers` x (AttentionBlock, FeedForwardBlock).  ``self_attention=True``: the memory is the running query
    itself (un-modulated), as in the reference's FFWRelativeSelfAttentionModule."""

    def __init__(self, dim: int, heads: int, num_layers: int, dropout: float = 0.0, use_adaln: bool = True,
                 self_attention: bool = False):
        super().__init__()
        self.self_attention = self_attention
        self.attn = nn.ModuleList([AttentionBlock(dim, heads, dropout, use_adaln) for _ in range(num_layers)])
        self.ffw = nn.ModuleList([FeedForwardBlock(dim, dim, dropout, use_adaln) for _ in range(num_layers)])

    def forward(self, query, memory=None, cond=None, q_rot=None, kv_rot=None, key_padding_mask=None, need_weights=False,
                cond_act=None, kv_caches=None, key_padding_mask16=None, out_last=None, handover=None):
        """out_last: contiguous [B, L, D] destination of the last layer's output (matrix-core cross-attention path only; the
        caller checks that the returned tensor is it).
        handover: a fused_ops.CrossHandover (cross-attention stack) / SelfHandover (self-attention stack) made (zeroed) for THIS
        inference: the attention of a layer and the block kernel behind it run as one launch (fused_ops.cross_layer / self_layer).
        key_padding_mask16: fused_ops.pad_mask16(key_padding_mask) if the caller keeps it (matrix-core attention path).
        cond_act: F.silu(cond), shared by every AdaLN of the pass; kv_caches: per-layer (keys, values) of a memory that
        is constant across calls (cross-attention at inference)."""
        weights = None
        if (_fused(query) and not need_weights and query.shape[-1] in _block_dims() and len(self.attn) > 0
                and (query.shape[-1] // self.attn[0].attn.heads) in (8, 15, 16, 20, 24, 32) and self.ffw[0].fc1.out_features == query.shape[-1]):
            # three launches per layer: (q | k | v) [or q alone with cached context keys / values], attention,
            # (out_proj + residual + LayerNorm + feed-forward block)
            from . import fused_ops as FO

            def ss_of(adaln):
                if adaln is None or cond is None:
                    return None
                return cond_act.lookup(adaln) if isinstance(cond_act, AdaLNBatch) else adaln.proj(F.silu(cond) if cond_act is None else cond_act)

            if self.self_attention and (query.shape[-1], self.attn[0].attn.heads) == FO.MFMA_DIMS:
                # matrix-core forms, two launches per layer: attention | (out_proj + LN + FFN of this layer + q/k/v of the next)
                L_, n = query.shape[1], len(self.attn)
                A0 = self.attn[0].attn
                qh, kh, vt = FO.qkv_heads(query, ss_of(self.attn[0].adaln), A0.q_proj, A0.kv_proj, q_rot, A0.heads)
                for li, (blk, ffw) in enumerate(zip(self.attn, self.ffw)):
                    A = blk.attn
                    if handover is not None and FO.FUSE_SELF_LAYER and (key_padding_mask is None or key_padding_mask16 is not None):
                        # attention + block (+ the next layer's projections) in one launch
                        if li + 1 < n:
                            nb = self.attn[li + 1]
                            query, qh, kh, vt = FO.self_layer(qh, kh, vt, L_, key_padding_mask16, query, A.out_proj, blk.norm, ss_of(ffw.adaln),
                                                              ffw.fc1, ffw.fc2, ffw.norm, handover, ss_of(nb.adaln), nb.attn.q_proj,
                                                              nb.attn.kv_proj, q_rot, A.heads)
                        else:
                            query = FO.self_layer(qh, kh, vt, L_, key_padding_mask16, query, A.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1,
                                                  ffw.fc2, ffw.norm, handover, heads=A.heads)
                        continue
                    att = FO.attention_heads(qh, kh, vt, key_padding_mask, L_, L_, key_padding_mask16)
                    if li + 1 < n and FO.FUSE_OUT_FFN_QKV:
                        nb = self.attn[li + 1]
                        query, qh, kh, vt = FO.out_ffn_qkv(att, query, A.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm,
                                                           ss_of(nb.adaln), nb.attn.q_proj, nb.attn.kv_proj, q_rot, A.heads)
                    else:
                        query = FO.out_ffn_mfma(att, query, A.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm)
                        if li + 1 < n:
                            nb = self.attn[li + 1]
                            qh, kh, vt = FO.qkv_heads(query, ss_of(nb.adaln), nb.attn.q_proj, nb.attn.kv_proj, q_rot, A.heads)
                return query, None
            qh_next = None
            for li, (blk, ffw) in enumerate(zip(self.attn, self.ffw)):
                A = blk.attn
                if (not self.self_attention and kv_caches is not None and len(kv_caches[li]) == 3
                        and (query.shape[-1], A.heads) == FO.MFMA_DIMS):
                    # cross-attention over a context whose head-major keys / values were cached (project_kv_heads): per layer
                    # attention | (out_proj + LN + FFN + the NEXT layer's query projection)
                    kh, vt, Lk = kv_caches[li]
                    if li == 0 or qh_next is None:
                        qh, _, _ = FO.qkv_heads(query, ss_of(blk.adaln), A.q_proj, None, q_rot, A.heads, roles=1)
                    else:
                        qh = qh_next
                    last = li + 1 == len(self.attn)
                    if (handover is not None and FO.FUSE_CROSS_LAYER and query.shape[1] <= 16 and Lk >= 1536
                            and (key_padding_mask is None or key_padding_mask16 is not None)):
                        # ... and the block kernel behind the split attention in the same launch
                        if last:
                            query = FO.cross_layer(qh, kh, vt, query.shape[1], Lk, key_padding_mask16, query, A.out_proj, blk.norm,
                                                   ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm, handover, heads=A.heads, out=out_last)
                        else:
                            nb = self.attn[li + 1]
                            query, qh_next = FO.cross_layer(qh, kh, vt, query.shape[1], Lk, key_padding_mask16, query, A.out_proj, blk.norm,
                                                            ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm, handover, ss_of(nb.adaln),
                                                            nb.attn.q_proj, q_rot, A.heads)
                        continue
                    if query.shape[1] <= 16 and Lk >= 1536 and (key_padding_mask is None or key_padding_mask16 is not None):
                        # a handful of query rows over a long context: keys split over several workgroups, merged by the next launch
                        att = FO.attention_heads_split(qh, kh, vt, query.shape[1], Lk, key_padding_mask16)
                    else:
                        att = FO.attention_heads(qh, kh, vt, key_padding_mask, query.shape[1], Lk, key_padding_mask16)
                    last = li + 1 == len(self.attn)
                    if last:
                        query = FO.out_ffn_mfma(att, query, A.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm, out=out_last)
                    else:
                        nb = self.attn[li + 1]
                        query, qh_next, _, _ = FO.out_ffn_qkv(att, query, A.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm,
                                                              ss_of(nb.adaln), nb.attn.q_proj, None, q_rot, A.heads)
                    continue
                if self.self_attention:
                    q, k, v = FO.qkv_block(query, ss_of(blk.adaln), A.q_proj, A.kv_proj, q_rot)
                else:
                    q = FO.q_block(query, ss_of(blk.adaln), A.q_proj, q_rot)
                    k, v = kv_caches[li] if kv_caches is not None else FO.kv_block(memory, A.kv_proj, kv_rot if q_rot is not None else None)
                att = FO.attention_small(q, k, v, key_padding_mask, A.heads)
                query = FO.out_ffn_block(att, query, A.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm)
            return query, None
        for li, (attn, ffw) in enumerate(zip(self.attn, self.ffw)):
            mem = query if self.self_attention else memory
            rot = q_rot if self.self_attention else kv_rot
            cache = kv_caches[li] if (kv_caches is not None and not self.self_attention) else None
            query, weights = attn(query, mem, cond, q_rot, rot, key_padding_mask, need_weights, cond_act, cache)
            query = ffw(query, cond, cond_act)
        return query, weights
