"""f32 Linear layers of a FROZEN module as one fp16 library GEMM each, at f32 accuracy.

Every f32 operand is split into two fp16 values, x = x_hi + x_lo (x_hi = fp16(x), x_lo the remainder: 22 bits of mantissa), and

    x w^T + b  =  x_hi w_hi^T  +  x_hi w_lo^T  +  x_lo w_hi^T  +  b                     (the lo.lo term is 2^-22 of the product)

is ONE plain GEMM over a concatenated reduction axis, fp16 inputs, f32 accumulation and output:

    [ x_hi | x_hi / 2048 | x_lo | 1, 1 / 2048, 0 x 62 ]  x  [ w_hi | 2048 w_lo | w_hi | b_hi, 2048 b_lo, 0 x 62 ]^T

(the low parts of the CONSTANT operands are stored scaled by 2048 so that they are normal fp16 numbers; the activations carry the
inverse scale, exact as a power of two -- mmf_split_activations3 writes them in one pass).  On MI355X the f32 matrix rate is 1/16
of the fp16 rate, so the tripled reduction is ~2x faster than the f32 GEMM (the backbone's four shapes at 32 768 tokens: 1.0 - 1.25 ms -> 0.45 - 0.5 ms) and closer to
the f64 result than rocBLAS's f32 kernel.  The reference runs its frozen backbone under TF32
(mindmap/image_processing/feature_extraction.py:322), a 10-bit mantissa.

Requires |w|, |b| < 65 504 (checked when a weight is split; the layer otherwise stays on the f32
GEMM) and |x| < 65 504 (activations of a LayerNorm-ed transformer are; not checked per call).  Inference only: no autograd."""
import weakref

import torch
import torch.nn.functional as F

from .. import _lib

kTail = 64  # columns behind the three parts (mmf_split_activations3): the bias pair + zeros, a row stays a multiple of 128 bytes
kMinRows = 2048  # (one 512 x 512 image = 1 024 tokens: 3.2 ms on the f32 GEMMs, 3.3 ms split -- launch-bound either way; two images: 4.8 -> 3.9 ms)
_W3_CACHE = {}  # id(weight) -> (weak reference, versions of weight and bias, [N, 3 K + 64] fp16 or None when out of range)


def _split_weight(linear):
    """[N, 3 K + 64] fp16 split of the layer's weight and bias, or None when a value is out of fp16's range / the shape unsupported."""
    src = linear.weight.detach().float()
    bias = linear.bias.detach().float() if linear.bias is not None else torch.zeros(src.shape[0], device=src.device)
    if not (float(src.abs().max()) < 6.0e4 and float(bias.abs().max()) < 6.0e4 and src.shape[1] % 8 == 0 and src.shape[1] >= kTail):
        return None
    hi = src.half()
    lo = ((src - hi.float()) * 2048.0).half()
    bh = bias.half()
    bl = ((bias - bh.float()) * 2048.0).half()
    tail = torch.zeros((src.shape[0], kTail), dtype=torch.float16, device=src.device)
    tail[:, 0], tail[:, 1] = bh, bl
    return torch.cat([hi, lo, hi, tail], dim=1).contiguous()


def _versions(linear):
    return (linear.weight._version, None if linear.bias is None else linear.bias._version)


def _w3(linear):
    w = linear.weight
    key = id(w)
    hit = _W3_CACHE.get(key)
    ver = _versions(linear)
    if hit is None or hit[0]() is not w or hit[1] != ver or (hit[2] is not None and hit[2].device != w.device):
        hit = (weakref.ref(w, lambda _r, k=key: _W3_CACHE.pop(k, None)), ver, _split_weight(linear))
        _W3_CACHE[key] = hit
    return hit[2]


def stale(linear) -> bool:
    """The layer's weight / bias changed (in place: load_state_dict, copy_) since its split copy was made.  A layer that never HAD a
    copy (an unsplittable weight: ``_w3`` caches None for it) is never stale -- whoever multiplies by it (a captured graph included)
    reads the float32 weight itself, which is the changed one."""
    hit = _W3_CACHE.get(id(linear.weight))
    if hit is None or hit[0]() is not linear.weight or hit[2] is None:
        return False
    return hit[1] != _versions(linear)


def refresh_in_place(linear) -> bool:
    """Recompute the split copy of a layer whose weights changed INTO THE SAME STORAGE: a captured HIP graph that multiplies by the
    copy (training.GraphedTrainStep's backbone graph) holds its address.  Returns False when that is impossible (no copy yet, or the
    new weights cannot be split): the caller must re-capture."""
    hit = _W3_CACHE.get(id(linear.weight))
    if hit is None or hit[0]() is not linear.weight or hit[2] is None:
        return False
    new = _split_weight(linear)
    if new is None or new.shape != hit[2].shape or new.device != hit[2].device:
        return False
    hit[2].copy_(new)
    _W3_CACHE[id(linear.weight)] = (hit[0], _versions(linear), hit[2])
    return True


def supported() -> bool:
    """torch.mm with a float32 result from half inputs (PyTorch >= 2.8) is what the single GEMM needs."""
    return "out_dtype" in (torch.mm.__doc__ or "")


def split_linear(x: torch.Tensor, linear) -> torch.Tensor:
    """F.linear(x, linear.weight, linear.bias) for a CUDA float32 x, computed as described in the module docstring."""
    w3 = _w3(linear) if (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()) else None
    if w3 is None:
        return F.linear(x, linear.weight, linear.bias)
    K = x.shape[-1]
    if x.numel() // K < kMinRows:  # a small GEMM is launch-bound: the split pass costs more than the matrix rate returns
        return F.linear(x, linear.weight, linear.bias)
    x2 = x.reshape(-1, K).contiguous()
    a3 = torch.empty((x2.shape[0], 3 * K + kTail), dtype=torch.float16, device=x.device)
    _lib.check(_lib.lib().mmf_split_activations3(_lib.dptr(x2), x2.shape[0], K, _lib.dptr(a3), _lib.stream_ptr(x.device)), "mmf_split_activations3")
    y = torch.mm(a3, w3.t(), out_dtype=torch.float32)
    return y.reshape(*x.shape[:-1], w3.shape[0])


# ---- the element-wise passes between the split GEMMs, fused with the split (csrc/mmf_kernels_policy_layer.hip) -----------------------
def mm3(a3: torch.Tensor, linear, shape) -> torch.Tensor:
    """The one fp16 GEMM of ``split_linear`` on activations that are already split ([rows, 3 K + 64] fp16) -> float32 ``shape``."""
    return torch.mm(a3, _w3(linear).t(), out_dtype=torch.float32).reshape(shape)


def can_fuse(x: torch.Tensor, block) -> bool:
    """The fused per-block path: CUDA float32 inference, every Linear of the block splittable, enough rows, LayerNorm width the
    kernel is built for."""
    if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and supported()):
        return False
    K = x.shape[-1]
    if K not in (256, 512, 768, 1024) or x.numel() // K < kMinRows:
        return False
    return all(_w3(lin) is not None for lin in (block.qkv, block.proj, block.fc1, block.fc2))


def _a3(rows: int, K: int, device) -> torch.Tensor:
    return torch.empty((rows, 3 * K + kTail), dtype=torch.float16, device=device)


def ln_split3(x: torch.Tensor, residual, ln):
    """(x + residual, split3(LayerNorm(x + residual))) in one pass over the rows (``residual`` None: (x, split3(LayerNorm(x))))."""
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    a3 = _a3(x2.shape[0], K, x.device)
    s_out, y_ptr, s_ptr = x, None, None
    if residual is not None:
        y2 = residual.reshape(-1, K)
        y2 = y2 if y2.is_contiguous() else y2.contiguous()
        s_out = torch.empty_like(x2)
        y_ptr, s_ptr = _lib.dptr(y2), _lib.dptr(s_out)
    _lib.check(_lib.lib().mmf_layernorm_split_activations3(_lib.dptr(x2), y_ptr, _lib.dptr(ln.weight), _lib.dptr(ln.bias), float(ln.eps), x2.shape[0], K,
                                                          s_ptr, _lib.dptr(a3), _lib.stream_ptr(x.device)), "mmf_layernorm_split_activations3")
    return s_out.reshape(x.shape), a3


def gelu_split3(h: torch.Tensor) -> torch.Tensor:
    K = h.shape[-1]
    h2 = h.reshape(-1, K)
    a3 = _a3(h2.shape[0], K, h.device)
    _lib.check(_lib.lib().mmf_gelu_split_activations3(_lib.dptr(h2), h2.shape[0], K, _lib.dptr(a3), _lib.stream_ptr(h.device)),
               "mmf_gelu_split_activations3")
    return a3


def split3_heads(att: torch.Tensor) -> torch.Tensor:
    """att [B, heads, L, d] (contiguous: what SDPA returns) -> split3 of its [B L, heads d] view, without the transpose copy."""
    B, H, L, d = att.shape
    att = att if att.is_contiguous() else att.contiguous()
    a3 = _a3(B * L, H * d, att.device)
    _lib.check(_lib.lib().mmf_split_attention_heads3(_lib.dptr(att), B, H, L, d, _lib.dptr(a3), _lib.stream_ptr(att.device)),
               "mmf_split_attention_heads3")
    return a3


def attention_split_ok(B: int, L: int, H: int, d: int) -> bool:
    return d == 64 and L % 128 == 0 and L > 0


def attention_split(qkv: torch.Tensor, B: int, L: int, H: int, d: int, split_out: bool = False) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v of a packed projection qkv [B, L, 3, H, d] (float32, contiguous) -> [B, L, H d] float32, at
    float32 accuracy on the fp16 matrix cores (mmf_attention_split: split operands, f32 accumulation and statistics).
    ``split_out``: the result as the split operand of the next Linear's GEMM instead ([B L, 3 H d + 64] fp16, what
    mmf_split_activations3 would make of it)."""
    assert qkv.is_contiguous() and qkv.dtype == torch.float32 and qkv.numel() == B * L * 3 * H * d
    if split_out:
        out = torch.empty((B * L, 3 * H * d + kTail), dtype=torch.float16, device=qkv.device)
    else:
        out = torch.empty((B, L, H * d), dtype=torch.float32, device=qkv.device)
    base = qkv.data_ptr()
    rs = 3 * H * d
    _lib.check(_lib.lib().mmf_attention_split(base, base + 4 * H * d, base + 8 * H * d, rs, L * rs, B, H, L, d, 1.0 / (d ** 0.5), _lib.dptr(out),
                                              1 if split_out else 0, _lib.stream_ptr(qkv.device)), "mmf_attention_split")
    return out
