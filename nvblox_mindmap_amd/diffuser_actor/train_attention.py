"""Attention of the trainable transformer stacks with its backward, on libmmfusion's f32 matrix-core kernels
(``mmf_train_attention_forward`` / ``_backward``, csrc/mmf_kernels_train_attn.hip).

The reference's attention layers (mindmap/diffuser_actor/layers.py, multihead_custom_attention.py) have 8 heads of 15 channels;
``F.scaled_dot_product_attention`` has no kernel for that head size (padded to 16 it takes the memory-efficient path: 0.21 ms
forward + 0.48 ms backward per layer at batch 32 x 616 tokens, plus the pad / transpose copies).  ``train_attention`` reads q / k /
v in the projections' own [B, L, heads * head_dim] layout (chunk views of a wider projection included) and returns [B, Lq, D]."""
import ctypes as C
import os
from typing import Optional

import torch

from .. import _lib

ENABLED = os.environ.get("MMF_TRAIN_ATTENTION", "1") != "0"
MAX_HEAD_DIM = 16


def usable(q: torch.Tensor, head_dim: int) -> bool:
    """CUDA float32 tensors under autograd, head size the kernels are built for (the caller checks dropout)."""
    return (ENABLED and q.is_cuda and q.dtype == torch.float32 and head_dim <= MAX_HEAD_DIM and torch.is_grad_enabled()
            and not torch.is_autocast_enabled() and q.numel() > 0)  # (raw float32 pointers: not under autocast)


def _rows(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


def _strides6(q, k, v):
    return (C.c_int64 * 6)(q.stride(1), q.stride(0), k.stride(1), k.stride(0), v.stride(1), v.stride(0))


class _TrainAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_padding_mask, heads: int, scale: float):
        q, k, v = _rows(q), _rows(k), _rows(v)
        B, Lq, D = q.shape
        Lk = k.shape[1]
        hd = D // heads
        out = torch.empty((B, Lq, D), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, heads, Lq), dtype=torch.float32, device=q.device)
        pad = None
        if key_padding_mask is not None:
            pad = key_padding_mask.contiguous()
            pad = pad.view(torch.uint8) if pad.dtype == torch.bool else pad.to(torch.uint8)
        _lib.check(_lib.lib().mmf_train_attention_forward(_lib.dptr(q), _lib.dptr(k), _lib.dptr(v), _strides6(q, k, v),
                                                          _lib.dptr(pad) if pad is not None else None, B, heads, Lq, Lk, hd, float(scale),
                                                          _lib.dptr(out), _lib.dptr(lse), _lib.stream_ptr(q.device)), "mmf_train_attention_forward")
        ctx.save_for_backward(q, k, v, out, lse, pad if pad is not None else torch.empty(0, dtype=torch.uint8, device=q.device))
        ctx.heads, ctx.scale, ctx.has_pad = heads, float(scale), pad is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, pad = ctx.saved_tensors
        B, Lq, D = q.shape
        Lk = k.shape[1]
        dout = dout.contiguous()
        dq = torch.empty((B, Lq, D), dtype=torch.float32, device=q.device)
        dk = torch.empty((B, Lk, D), dtype=torch.float32, device=q.device)
        dv = torch.empty((B, Lk, D), dtype=torch.float32, device=q.device)
        dsum = torch.empty_like(lse)
        _lib.check(_lib.lib().mmf_train_attention_backward(_lib.dptr(q), _lib.dptr(k), _lib.dptr(v), _strides6(q, k, v),
                                                           _lib.dptr(pad) if ctx.has_pad else None, B, ctx.heads, Lq, Lk, D // ctx.heads, ctx.scale,
                                                           _lib.dptr(out), _lib.dptr(dout), _lib.dptr(lse), _lib.dptr(dsum), _lib.dptr(dq), _lib.dptr(dk), _lib.dptr(dv),
                                                           _lib.stream_ptr(q.device)), "mmf_train_attention_backward")
        return dq, dk, dv, None, None, None


def train_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_padding_mask: Optional[torch.Tensor], heads: int,
                    scale: Optional[float] = None) -> torch.Tensor:
    """softmax(scale q k^T + mask) v per head; q [B, Lq, D], k / v [B, Lk, D] (head h = channels h * D / heads ...), mask [B, Lk]
    True = ignore.  Differentiable in q, k, v."""
    hd = q.shape[-1] // heads
    return _TrainAttention.apply(q, k, v, key_padding_mask, heads, (1.0 / hd ** 0.5) if scale is None else scale)
