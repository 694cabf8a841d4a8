"""Normalisation of the trainable post-norm blocks with its backward on libmmfusion kernels (csrc/mmf_kernels_train_ops.hip):
``add_layer_norm(a, b, norm)`` = ``norm(a + b)`` of mindmap/diffuser_actor/layers.py's attention / feed-forward blocks, one kernel
forward (the residual add rides in it), two backward (dx + column partials, then the partials' sum in a fixed order)."""
import os
from typing import Optional

import torch

from .. import _lib

ENABLED = os.environ.get("MMF_TRAIN_LAYERNORM", "1") != "0"


def _f32_only() -> bool:
    """the kernels behind these Functions read and write float32 through raw pointers: under autocast an incoming gradient (or an
    F.linear result) would be fp16 -- the composite ops take over there."""
    return not torch.is_autocast_enabled()


def usable(x: torch.Tensor, norm, b: Optional[torch.Tensor] = None) -> bool:
    D = x.shape[-1]
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and _f32_only() and x.numel() > 0 and D <= 128
            and D % 4 == 0 and tuple(norm.normalized_shape) == (D,) and norm.weight is not None and norm.bias is not None
            and (b is None or (b.shape == x.shape and b.dtype == torch.float32 and b.is_cuda)))


def _partials(device) -> torch.Tensor:
    # per call, from torch's caching allocator: safe inside a captured graph (its private pool) and across streams
    return torch.empty(int(_lib.lib().mmf_layernorm_train_scratch_bytes()) // 4, dtype=torch.float32, device=device)


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, weight, bias, eps: float):
        a = a.contiguous()
        D = a.shape[-1]
        rows = a.numel() // D
        y = torch.empty_like(a)
        mean = torch.empty(rows, dtype=torch.float32, device=a.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
        s = None
        if b is not None:
            b = b.contiguous()
            s = torch.empty_like(a)
        _lib.check(_lib.lib().mmf_layernorm_train_forward(_lib.dptr(a), _lib.dptr(b), _lib.dptr(weight), _lib.dptr(bias), float(eps), rows, D,
                                                          _lib.dptr(s), _lib.dptr(y), _lib.dptr(mean), _lib.dptr(rstd), _lib.stream_ptr(a.device)),
                   "mmf_layernorm_train_forward")
        ctx.save_for_backward(a if s is None else s, weight, mean, rstd)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        assert g.dtype == torch.float32, "mmf_layernorm_train_backward reads float32 gradients"
        g = g.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        dx = torch.empty_like(x)
        dgamma = torch.empty(D, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(D, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mmf_layernorm_train_backward(_lib.dptr(g), _lib.dptr(x), _lib.dptr(weight), _lib.dptr(mean), _lib.dptr(rstd), rows, D,
                                                           _lib.dptr(dx), _lib.dptr(dgamma), _lib.dptr(dbeta), _lib.dptr(_partials(x.device)),
                                                           _lib.stream_ptr(x.device)), "mmf_layernorm_train_backward")
        return dx, (dx if ctx.has_b else None), dgamma, dbeta, None


def add_layer_norm(a: torch.Tensor, b: Optional[torch.Tensor], norm) -> torch.Tensor:
    """``norm(a + b)`` (``b`` None: ``norm(a)``) for an ``nn.LayerNorm`` over the last dimension."""
    if usable(a, norm, b):
        return _AddLayerNorm.apply(a, b, norm.weight, norm.bias, norm.eps)
    return norm(a if b is None else a + b)


class _AdaLNModulate(torch.autograd.Function):
    """x * (1 + scale[:, None]) + shift[:, None]: forward = mmf_adaln_modulate (the composite's float operations), backward = one sweep
    (dx and the per-batch-element column sums of g x and g) + the partials' sum."""

    @staticmethod
    def forward(ctx, x, ss):
        from .fused_ops import adaln_modulate

        x, ss = x.contiguous(), ss.contiguous()
        ctx.save_for_backward(x, ss)
        return adaln_modulate(x, ss)

    @staticmethod
    def backward(ctx, g):
        x, ss = ctx.saved_tensors
        assert g.dtype == torch.float32, "mmf_adaln_modulate_grad reads float32 gradients"
        g = g.contiguous()
        B, L, D = x.shape
        dx, dss = torch.empty_like(x), torch.empty_like(ss)
        scratch = torch.empty(int(_lib.lib().mmf_adaln_modulate_grad_scratch_bytes(B)) // 4, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mmf_adaln_modulate_grad(_lib.dptr(g), _lib.dptr(x), _lib.dptr(ss), B, L, D, _lib.dptr(dx), _lib.dptr(dss),
                                                      _lib.dptr(scratch), _lib.stream_ptr(x.device)), "mmf_adaln_modulate_grad")
        return dx, dss


def adaln_usable(x: torch.Tensor) -> bool:
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and torch.is_grad_enabled() and _f32_only() and x.numel() > 0
            and x.shape[-1] <= 128 and x.shape[-1] % 4 == 0)


def adaln_modulate_train(x: torch.Tensor, scale_shift: torch.Tensor) -> torch.Tensor:
    return _AdaLNModulate.apply(x, scale_shift)


class _LinearTrain(torch.autograd.Function):
    """y = x W^T + b with torch's GEMMs forward and for dx; dW and db by ``mmf_linear_weight_grad`` (rows split over the chip, f32
    matrix cores, deterministic sum of the splits) -- the BLAS libraries' best kernels for these short-and-very-deep products run at a
    fifth of that."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        assert g.dtype == torch.float32, "mmf_linear_weight_grad reads float32 gradients"
        N, K = weight.shape
        g2 = g.reshape(-1, N)
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        rows = g2.shape[0]
        dx = (g2 @ weight).reshape(x.shape) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(weight)
        db = torch.empty(N, dtype=weight.dtype, device=weight.device) if ctx.has_bias else None
        scratch = torch.empty(int(_lib.lib().mmf_linear_weight_grad_scratch_bytes(rows, N, K)) // 4, dtype=torch.float32, device=weight.device)
        _lib.check(_lib.lib().mmf_linear_weight_grad(_lib.dptr(g2), _lib.dptr(x2), rows, N, K, _lib.dptr(dW), _lib.dptr(db), _lib.dptr(scratch),
                                                     _lib.stream_ptr(weight.device)), "mmf_linear_weight_grad")
        return dx, dW, db


MIN_ROWS_LINEAR = 2048


def linear(module, x: torch.Tensor) -> torch.Tensor:
    """``module(x)`` for an ``nn.Linear``; under autograd on CUDA float32 with many rows its parameter gradients come from the
    matrix-core split kernel."""
    w = module.weight
    if (ENABLED and x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and _f32_only() and w.requires_grad and w.shape[1] <= 128
            and w.shape[0] <= 256 and x.numel() // w.shape[1] >= MIN_ROWS_LINEAR and w.is_contiguous()):
        return _LinearTrain.apply(x, w, module.bias)
    return module(x)
