"""f32 Linear layers of a FROZEN module as one fp16 library GEMM each, at f32 accuracy.

Every f32 operand is split x = hi + lo / 2048 (hi = fp16(x), lo = fp16((x - hi) * 2048): 22 bits of mantissa) and

    x w^T  =  ( x_hi (2048 w_hi)^T  +  x_hi w_lo^T  +  x_lo w_hi^T ) / 2048        (the lo.lo term is 2^-22 of the product)

is ONE GEMM over the concatenated reduction axis: [x_hi | x_hi | x_lo] (mmf_split_activations3, one pass over the activations)
times [2048 w_hi | w_lo | w_hi] (made once per weight), fp16 inputs, f32 accumulation and output, alpha = 1 / 2048, the bias as
the addend.  On MI355X the f32 matrix rate is 1/16 of the fp16 rate, so three fp16 products are ~2x faster than one f32 GEMM
(32 768 x 768 x 3 072: 1.31 ms -> 0.68 ms) and closer to the f64 result than rocBLAS's f32 kernel (2.4e-6 vs 7.2e-6 on that shape).
The reference runs its frozen backbone under TF32 (mindmap/image_processing/feature_extraction.py:322), a 10-bit mantissa.

Requires |w| < 32 (2048 w_hi must stay below 65 504) -- checked when a weight is split, the layer then stays on the f32 GEMM --
and |x| < 65 504 (activations of a LayerNorm-ed transformer are; not checked per call).  Inference only: no autograd."""
import weakref

import torch
import torch.nn.functional as F

from .. import _lib

kMinRows = 4096  # (one 512 x 512 image is 1 024 tokens: 3.24 ms on the f32 GEMMs, 3.37 ms split)
_W3_CACHE = {}  # id(weight) -> (weak reference, version, [N, 3K] fp16 or None when the weight is out of range)


def _w3(linear):
    w = linear.weight
    key = id(w)
    hit = _W3_CACHE.get(key)
    if hit is None or hit[0]() is not w or hit[1] != w._version or (hit[2] is not None and hit[2].device != w.device):
        src = w.detach().float()
        w3 = None
        if float(src.abs().max()) < 31.0 and src.shape[1] % 8 == 0:
            hi = src.half()
            lo = ((src - hi.float()) * 2048.0).half()
            w3 = torch.cat([(hi.float() * 2048.0).half(), lo, hi], dim=1).contiguous()
        hit = (weakref.ref(w, lambda _r, k=key: _W3_CACHE.pop(k, None)), w._version, w3)
        _W3_CACHE[key] = hit
    return hit[2]


def supported() -> bool:
    """torch.addmm with a float32 result from half inputs (PyTorch >= 2.8) is what the single GEMM needs."""
    return "out_dtype" in (torch.addmm.__doc__ or "")


def split_linear(x: torch.Tensor, linear) -> torch.Tensor:
    """F.linear(x, linear.weight, linear.bias) for a CUDA float32 x, computed as described in the module docstring."""
    w3 = _w3(linear) if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()) else None
    if w3 is None:
        return F.linear(x, linear.weight, linear.bias)
    K = x.shape[-1]
    if x.numel() // K < kMinRows:  # a small GEMM is launch-bound: the split pass costs more than the matrix rate returns
        return F.linear(x, linear.weight, linear.bias)
    x2 = x.reshape(-1, K).contiguous()
    a3 = torch.empty((x2.shape[0], 3 * K), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().mmf_split_activations3(_lib.dptr(x2), x2.shape[0], K, _lib.dptr(a3), _lib.stream_ptr(x.device)), "mmf_split_activations3")
    if linear.bias is not None:
        y = torch.addmm(linear.bias.detach().float(), a3, w3.t(), alpha=1.0 / 2048.0, out_dtype=torch.float32)
    else:
        y = torch.mm(a3, w3.t(), out_dtype=torch.float32) * (1.0 / 2048.0)
    return y.reshape(*x.shape[:-1], w3.shape[0])
