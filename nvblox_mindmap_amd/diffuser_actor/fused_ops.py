"""Inference-only fused HIP ops of the diffusion head (libmmfusion: mmf_rotary_apply, mmf_adaln_modulate,
mmf_attention_small).  No autograd: callers use them under torch.no_grad() and keep the composite torch ops for training."""
from typing import Optional

import torch

from .. import _lib


def rotary_apply(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """layers.apply_rotary(x, cos, sin) in one kernel (identical float operations)."""
    D = x.shape[-1]
    shape = x.shape
    if x.dim() == 3 and x.stride(-1) == 1 and x.stride(0) == x.shape[1] * x.stride(1):
        rows, stride = x.shape[0] * x.shape[1], x.stride(1)  # e.g. the key half of a fused key/value projection
    else:
        x = x.contiguous()
        rows, stride = x.numel() // D, D
    cos = cos.expand(shape).contiguous()
    sin = sin.expand(shape).contiguous()
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mmf_rotary_apply(_lib.dptr(x), stride, _lib.dptr(cos), _lib.dptr(sin), _lib.dptr(out), rows, D,
                                           _lib.stream_ptr(x.device)), "mmf_rotary_apply")
    return out


class _RotaryTrain(torch.autograd.Function):
    """apply_rotary with its gradient, one kernel each way (forward: ``mmf_rotary_apply``, the composite's own float operations;
    backward: ``mmf_rotary_apply_grad``, the floats autograd computes for it).  cos / sin carry no gradient (rotary3d)."""

    @staticmethod
    def forward(ctx, x, cos, sin):
        cos = cos.expand(x.shape).contiguous()
        sin = sin.expand(x.shape).contiguous()
        ctx.save_for_backward(cos, sin)
        return rotary_apply(x, cos, sin)

    @staticmethod
    def backward(ctx, g):
        cos, sin = ctx.saved_tensors
        g = g.contiguous()
        D = g.shape[-1]
        dx = torch.empty_like(g)
        _lib.check(_lib.lib().mmf_rotary_apply_grad(_lib.dptr(g), _lib.dptr(cos), _lib.dptr(sin), _lib.dptr(dx), g.numel() // D, D,
                                                    _lib.stream_ptr(g.device)), "mmf_rotary_apply_grad")
        return dx, None, None


def rotary_apply_train(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    return _RotaryTrain.apply(x, cos, sin)


def adaln_modulate(x: torch.Tensor, scale_shift: torch.Tensor) -> torch.Tensor:
    """x * (1 + scale[:, None]) + shift[:, None] with scale_shift = (scale | shift) [B, 2D], x [B, L, D]."""
    x = x.contiguous()
    ss = scale_shift.contiguous()
    B, L, D = x.shape
    out = torch.empty_like(x)
    _lib.check(_lib.lib().mmf_adaln_modulate(_lib.dptr(x), _lib.dptr(ss), _lib.dptr(out), B, L, D, _lib.stream_ptr(x.device)),
               "mmf_adaln_modulate")
    return out


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_padding_mask: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(d) + padding) v per head.  q [B,Lq,D]; k, v [B,Lk,D], possibly column slices of one
    [B,Lk,2D] projection (row stride 2D); key_padding_mask [B,Lk] bool, True = ignore."""
    B, Lq, D = q.shape
    Lk = k.shape[1]
    q = q.contiguous()

    def strided(t):
        if t.stride(-1) == 1 and t.stride(0) == Lk * t.stride(1):
            return t, t.stride(1)
        t = t.contiguous()
        return t, D

    k, ks = strided(k)
    v, vs = strided(v)
    pad = None if key_padding_mask is None else key_padding_mask.contiguous().view(torch.uint8)
    out = torch.empty((B, Lq, D), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().mmf_attention_small(_lib.dptr(q), _lib.dptr(k), ks, _lib.dptr(v), vs, _lib.dptr(pad), _lib.dptr(out), B, Lq,
                                              Lk, heads, D // heads, _lib.stream_ptr(q.device)), "mmf_attention_small")
    return out


def ddpm_step(traj: torch.Tensor, pred: torch.Tensor, noise: torch.Tensor, coef_pos, coef_rot, split: int = 3) -> torch.Tensor:
    """Both schedulers' reverse step on the trajectory in one launch: traj / noise [..., C], pred [..., >= C] (the head's
    output carries extra channels), coefficients from DDPMScheduler.step_coefficients."""
    import ctypes as Ct

    traj = traj.contiguous()
    noise = noise.contiguous()
    pred = pred.contiguous()
    Cc = traj.shape[-1]
    rows = traj.numel() // Cc
    out = torch.empty_like(traj)
    a = (Ct.c_float * 6)(*coef_pos)
    b = (Ct.c_float * 6)(*coef_rot)
    _lib.check(_lib.lib().mmf_ddpm_step(_lib.dptr(traj), _lib.dptr(pred), pred.shape[-1], _lib.dptr(noise), _lib.dptr(out), rows, Cc, split,
                                        Ct.cast(a, Ct.c_void_p), Ct.cast(b, Ct.c_void_p), _lib.stream_ptr(traj.device)), "mmf_ddpm_step")
    return out


BLOCK_DIMS = (120,)  # embedding dims the whole-block kernels are built for


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.contiguous()


import weakref

_WT_CACHE = {}  # id(weight Parameter) -> (weak reference to it, version, transposed copy); dropped when the parameter dies


def _wt(linear) -> torch.Tensor:
    """Transposed ([in, out]) contiguous copy of a Linear's weight, cached per weight TENSOR OBJECT until it is modified
    (version counter) or moved.  The entry holds a weak reference and is validated by identity: ids and device addresses of
    freed modules get reused, a key made of those alone can serve another module's weights."""
    w = linear.weight
    key = id(w)
    hit = _WT_CACHE.get(key)
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[2].device != w.device:
        hit = (weakref.ref(w, lambda _r, k=key: _WT_CACHE.pop(k, None)), w._version, w.detach().t().contiguous())
        _WT_CACHE[key] = hit
    return hit[2]


def ffn_block(x, scale_shift, fc1, fc2, norm) -> torch.Tensor:
    """FeedForwardBlock at inference in one launch: h = x*(1+scale)+shift (if scale_shift); LayerNorm(h + fc2(relu(fc1(h))))."""
    x = x.contiguous()
    B, L, D = x.shape
    out = torch.empty_like(x)
    ss = _c(scale_shift)
    _lib.check(_lib.lib().mmf_ffn_block(_lib.dptr(x), _lib.dptr(ss), _lib.dptr(_wt(fc1)), _lib.dptr(_c(fc1.bias)), _lib.dptr(_wt(fc2)),
                                        _lib.dptr(_c(fc2.bias)), _lib.dptr(_c(norm.weight)), _lib.dptr(_c(norm.bias)), float(norm.eps),
                                        _lib.dptr(out), B, L, D, _lib.stream_ptr(x.device)), "mmf_ffn_block")
    return out


def q_block(x, scale_shift, q_proj, rot) -> torch.Tensor:
    """rotary(q_proj(x*(1+scale)+shift)); rot = (cos, sin) [B,L,D] or None."""
    x = x.contiguous()
    B, L, D = x.shape
    out = torch.empty_like(x)
    ss = _c(scale_shift)
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    _lib.check(_lib.lib().mmf_q_block(_lib.dptr(x), _lib.dptr(ss), _lib.dptr(_wt(q_proj)), _lib.dptr(_c(q_proj.bias)), _lib.dptr(cs),
                                      _lib.dptr(sn), _lib.dptr(out), B, L, D, _lib.stream_ptr(x.device)), "mmf_q_block")
    return out


def kv_block(memory, kv_proj, rot):
    """(rotary(keys), values) of kv_proj(memory), both [B,Lk,D] contiguous."""
    memory = memory.contiguous()
    B, L, D = memory.shape
    k = torch.empty_like(memory)
    v = torch.empty_like(memory)
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    _lib.check(_lib.lib().mmf_kv_block(_lib.dptr(memory), _lib.dptr(_wt(kv_proj)), _lib.dptr(_c(kv_proj.bias)), _lib.dptr(cs), _lib.dptr(sn),
                                       _lib.dptr(k), _lib.dptr(v), B * L, D, _lib.stream_ptr(memory.device)), "mmf_kv_block")
    return k, v


def attn_out_block(att, residual, out_proj, norm) -> torch.Tensor:
    """LayerNorm(residual + out_proj(att))."""
    att = att.contiguous()
    residual = residual.contiguous()
    B, L, D = att.shape
    out = torch.empty_like(att)
    _lib.check(_lib.lib().mmf_attn_out_block(_lib.dptr(att), _lib.dptr(residual), _lib.dptr(_wt(out_proj)), _lib.dptr(_c(out_proj.bias)),
                                             _lib.dptr(_c(norm.weight)), _lib.dptr(_c(norm.bias)), float(norm.eps), _lib.dptr(out), B * L, D,
                                             _lib.stream_ptr(att.device)), "mmf_attn_out_block")
    return out


def qkv_block(x, scale_shift, q_proj, kv_proj, rot):
    """Self-attention input side in one launch: (rotary(q_proj(modulated x)), rotary(keys), values), each [B,L,D]."""
    x = x.contiguous()
    B, L, D = x.shape
    q, k, v = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    ss = _c(scale_shift)
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    _lib.check(_lib.lib().mmf_qkv_block(_lib.dptr(x), _lib.dptr(ss), _lib.dptr(_wt(q_proj)), _lib.dptr(_c(q_proj.bias)), _lib.dptr(_wt(kv_proj)),
                                        _lib.dptr(_c(kv_proj.bias)), _lib.dptr(cs), _lib.dptr(sn), _lib.dptr(q), _lib.dptr(k), _lib.dptr(v), B, L, D,
                                        _lib.stream_ptr(x.device)), "mmf_qkv_block")
    return q, k, v


def out_ffn_block(att, residual, out_proj, norm1, scale_shift, fc1, fc2, norm2) -> torch.Tensor:
    """AttentionBlock tail + FeedForwardBlock in one launch: x1 = LN1(residual + out_proj(att)); h = modulate(x1);
    LN2(h + fc2(relu(fc1(h))))."""
    att = att.contiguous()
    residual = residual.contiguous()
    B, L, D = att.shape
    out = torch.empty_like(att)
    ss = _c(scale_shift)
    _lib.check(_lib.lib().mmf_out_ffn_block(_lib.dptr(att), _lib.dptr(residual), _lib.dptr(_wt(out_proj)), _lib.dptr(_c(out_proj.bias)),
                                            _lib.dptr(_c(norm1.weight)), _lib.dptr(_c(norm1.bias)), float(norm1.eps), _lib.dptr(ss),
                                            _lib.dptr(_wt(fc1)), _lib.dptr(_c(fc1.bias)), _lib.dptr(_wt(fc2)), _lib.dptr(_c(fc2.bias)),
                                            _lib.dptr(_c(norm2.weight)), _lib.dptr(_c(norm2.bias)), float(norm2.eps), _lib.dptr(out), B, L, D,
                                            _lib.stream_ptr(att.device)), "mmf_out_ffn_block")
    return out


# ---- matrix-core forms (mmf_kernels_policy_mfma.hip): head-major q / k / v, attention over them, out_proj + LN + FFN ----------
MFMA_DIMS = (120, 8)  # (embedding dim, heads) the MFMA kernels are built for
FUSE_SELF_LAYER = True  # self-attention + the block kernel behind it (+ the next projections) in one launch (mmf_self_layer)
FUSE_CROSS_LAYER = True  # split cross-attention + the block kernel behind it in one launch (mmf_cross_layer)
FUSE_OUT_FFN_QKV = True  # layer i's tail and layer i + 1's q | k | v in one launch (mmf_out_ffn_qkv / mmf_out_ffn_qkv2)


_W16_CACHE = {}  # id(weight Parameter) -> (weak reference to it, version, split copy); dropped when the parameter dies


def _w(linear) -> torch.Tensor:
    """A Linear's weight in the form the matrix-core layer kernels read (mmf_split_linear_weight: every f32 entry as two fp16
    values hi + lo / 2048, in the order in which the kernels' waves load them, one 64 KB block per 120 output rows), made once
    per weight TENSOR OBJECT and cached until the weight is modified (version counter) or moved -- the rules of ``_wt``."""
    w = linear.weight
    key = id(w)
    hit = _W16_CACHE.get(key)
    if hit is None or hit[0]() is not w or hit[1] != w._version or hit[2].device != w.device:
        src = w.detach().contiguous()
        assert src.dtype == torch.float32 and src.dim() == 2
        assert src.shape[1] == 120 and src.shape[0] % 120 == 0
        dst = torch.empty((src.shape[0] // 120, 128 * 256), dtype=torch.float16, device=src.device)  # one 64 KB block per 120 rows
        _lib.check(_lib.lib().mmf_split_linear_weight(_lib.dptr(src), src.shape[0], src.shape[1], _lib.dptr(dst), _lib.stream_ptr(src.device)),
                   "mmf_split_linear_weight")
        hit = (weakref.ref(w, lambda _r, k=key: _W16_CACHE.pop(k, None)), w._version, dst)
        _W16_CACHE[key] = hit
    return hit[2]


def _l16(n: int) -> int:
    return (n + 15) // 16 * 16


def qkv_heads(x, scale_shift, q_proj, kv_proj, rot, heads: int, roles: int = 7):
    """The projections of ``qkv_block`` written head-major, padded to 16 channels per head (padding = 0), in the split form the
    attention kernel multiplies (``unpack_heads`` gives the float values):
    (q_heads [B,H,L16,16], k_heads [B,H,L16,16], v_heads_t [B,H,16,L16]).  roles: 7 = q|k|v, 1 = q alone (kv_proj unused),
    6 = k|v alone (q_proj / scale_shift unused): entries that are not produced are None."""
    x = x.contiguous()
    B, L, D = x.shape
    L16 = _l16(L)
    dev = x.device
    q = torch.empty((B, heads, L16, 16), dtype=torch.float32, device=dev) if roles & 1 else None
    k = torch.empty((B, heads, L16, 16), dtype=torch.float32, device=dev) if roles & 6 else None
    v = torch.empty((B, heads, 16, L16), dtype=torch.float32, device=dev) if roles & 6 else None
    ss = _c(scale_shift) if roles & 1 else None
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    wq, bq = (_w(q_proj), _c(q_proj.bias)) if roles & 1 else (None, None)
    wkv, bkv = (_w(kv_proj), _c(kv_proj.bias)) if roles & 6 else (None, None)
    _lib.check(_lib.lib().mmf_qkv_heads(_lib.dptr(x), _lib.dptr(ss), _lib.dptr(wq), _lib.dptr(bq), _lib.dptr(wkv), _lib.dptr(bkv), _lib.dptr(cs),
                                        _lib.dptr(sn), _lib.dptr(q), _lib.dptr(k), _lib.dptr(v), B, L, D, heads, roles,
                                        _lib.stream_ptr(dev)), "mmf_qkv_heads")
    return q, k, v


def unpack_heads(t: torch.Tensor) -> torch.Tensor:
    """The float32 values of a head-major operand of ``qkv_heads`` (q_heads / k_heads [B,H,L16,16] or v_heads_t [B,H,16,L16]):
    the kernels store every aligned group of four values of the LAST axis as {4 fp16 hi | 4 fp16 lo} in the 16 bytes four floats
    would take, value = hi + lo / 2048 (the split operands of the attention kernel's matrix-core products).  For tests / debugging."""
    h = t.contiguous().view(torch.float16).reshape(*t.shape[:-1], t.shape[-1] // 4, 2, 4).to(torch.float32)
    return (h[..., 0, :] + h[..., 1, :] / 2048.0).reshape(t.shape)


def pad_mask16(key_padding_mask: torch.Tensor) -> torch.Tensor:
    """[B, Lk] bool (True = ignore) -> [B, Lk16] uint8 with the keys beyond Lk marked: the form mmf_attention_heads reads
    (one aligned 32-bit word per four keys).  Step-invariant masks are converted once per inference."""
    Lk = key_padding_mask.shape[1]
    return torch.nn.functional.pad(key_padding_mask.to(torch.uint8), (0, _l16(Lk) - Lk), value=1).contiguous()


def attention_heads(q_heads, k_heads, v_heads_t, key_padding_mask: Optional[torch.Tensor], Lq: int, Lk: int,
                    mask16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """softmax(q k^T / sqrt(15) + padding) v over the head-major operands of ``qkv_heads`` -> [B, Lq, 120].
    ``mask16``: pad_mask16(key_padding_mask) when the caller has it already."""
    B, H = q_heads.shape[:2]
    assert q_heads.shape[2] == _l16(Lq) and k_heads.shape[2] == _l16(Lk) and v_heads_t.shape[3] == _l16(Lk)
    if mask16 is None and key_padding_mask is not None:
        mask16 = pad_mask16(key_padding_mask)
    assert mask16 is None or (mask16.shape == (B, _l16(Lk)) and mask16.dtype == torch.uint8 and mask16.is_contiguous())
    out = torch.empty((B, Lq, H * 15), dtype=torch.float32, device=q_heads.device)
    _lib.check(_lib.lib().mmf_attention_heads(_lib.dptr(q_heads), _lib.dptr(k_heads), _lib.dptr(v_heads_t), _lib.dptr(mask16), _lib.dptr(out), B,
                                              Lq, Lk, H, 15, _lib.stream_ptr(q_heads.device)), "mmf_attention_heads")
    return out


def attention_heads_split(q_heads, k_heads, v_heads_t, Lq: int, Lk: int, mask16: Optional[torch.Tensor]) -> torch.Tensor:
    """``attention_heads`` for Lq <= 16 query rows over a long key axis with the keys divided among several workgroups per
    (batch element, head): returns the partial results [B, H, n_split, 18, 16] that ``out_ffn_mfma(att_partials=...)`` merges."""
    import ctypes as Ct

    B, H = q_heads.shape[:2]
    assert Lq <= 16 and q_heads.shape[2] == 16 and k_heads.shape[2] == _l16(Lk)
    assert mask16 is None or (mask16.shape == (B, _l16(Lk)) and mask16.dtype == torch.uint8 and mask16.is_contiguous())
    n = Ct.c_int(0)
    dev = q_heads.device
    _lib.check(_lib.lib().mmf_attention_heads_split(_lib.dptr(q_heads), _lib.dptr(k_heads), _lib.dptr(v_heads_t), None, None, B, Lq, Lk, H, 15,
                                                    Ct.byref(n), _lib.stream_ptr(dev)), "mmf_attention_heads_split")
    part = torch.empty((B, H, n.value, 18, 16), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().mmf_attention_heads_split(_lib.dptr(q_heads), _lib.dptr(k_heads), _lib.dptr(v_heads_t), _lib.dptr(mask16),
                                                    _lib.dptr(part), B, Lq, Lk, H, 15, Ct.byref(n), _lib.stream_ptr(dev)),
               "mmf_attention_heads_split")
    return part


def out_ffn_mfma(att, residual, out_proj, norm1, scale_shift, fc1, fc2, norm2, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``out_ffn_block`` on the matrix cores (16-token tiles, three chained GEMMs).  ``att``: the attention output [B, L, D], or
    the 5-D key-split partials of ``attention_heads_split``.  ``out``: a contiguous [B, L, D] float32 destination (e.g. the
    leading rows of a larger sequence buffer) instead of a fresh tensor."""
    residual = residual.contiguous()
    B, L, D = residual.shape
    if out is None:
        out = torch.empty_like(residual)
    else:
        assert out.shape == residual.shape and out.dtype == torch.float32 and out.is_contiguous() and out.data_ptr() != residual.data_ptr()
    if att.dim() == 5:  # key-split partials of attention_heads_split: merged by the kernel while it loads its input
        part = att.contiguous()
        _lib.check(_lib.lib().mmf_out_ffn_mfma_partials(_lib.dptr(part), part.shape[2], _lib.dptr(residual), _lib.dptr(_w(out_proj)),
                                                        _lib.dptr(_c(out_proj.bias)), _lib.dptr(_c(norm1.weight)), _lib.dptr(_c(norm1.bias)),
                                                        float(norm1.eps), _lib.dptr(_c(scale_shift)), _lib.dptr(_w(fc1)),
                                                        _lib.dptr(_c(fc1.bias)), _lib.dptr(_w(fc2)), _lib.dptr(_c(fc2.bias)),
                                                        _lib.dptr(_c(norm2.weight)), _lib.dptr(_c(norm2.bias)), float(norm2.eps), _lib.dptr(out),
                                                        B, L, D, _lib.stream_ptr(residual.device)), "mmf_out_ffn_mfma_partials")
        return out
    att = att.contiguous()
    ss = _c(scale_shift)
    _lib.check(_lib.lib().mmf_out_ffn_mfma(_lib.dptr(att), _lib.dptr(residual), _lib.dptr(_w(out_proj)), _lib.dptr(_c(out_proj.bias)),
                                           _lib.dptr(_c(norm1.weight)), _lib.dptr(_c(norm1.bias)), float(norm1.eps), _lib.dptr(ss),
                                           _lib.dptr(_w(fc1)), _lib.dptr(_c(fc1.bias)), _lib.dptr(_w(fc2)), _lib.dptr(_c(fc2.bias)),
                                           _lib.dptr(_c(norm2.weight)), _lib.dptr(_c(norm2.bias)), float(norm2.eps), _lib.dptr(out), B, L, D,
                                           _lib.stream_ptr(att.device)), "mmf_out_ffn_mfma")
    return out


def out_ffn_qkv(att, residual, out_proj, norm1, scale_shift, fc1, fc2, norm2, next_scale_shift, next_q_proj, next_kv_proj, rot, heads: int,
                out: Optional[torch.Tensor] = None):
    """``out_ffn_mfma`` of a layer and ``qkv_heads`` of the NEXT layer (on the result) in one launch.  ``att``: [B,L,D] or the
    5-D key-split partials of ``attention_heads_split``; ``next_kv_proj`` None: the next layer's queries alone (it attends to
    a cached memory).  Returns (out [B,L,D], q_heads, k_heads or None, v_heads_t or None)."""
    import ctypes as Ct

    residual = residual.contiguous()
    B, L, D = residual.shape
    L16 = _l16(L)
    dev = residual.device
    roles = 7 if next_kv_proj is not None else 1
    partials = att.contiguous() if att.dim() == 5 else None
    att = None if partials is not None else att.contiguous()
    if out is None:
        out = torch.empty_like(residual)
    else:
        assert out.shape == residual.shape and out.dtype == torch.float32 and out.is_contiguous() and out.data_ptr() != residual.data_ptr()
    q = torch.empty((B, heads, L16, 16), dtype=torch.float32, device=dev)
    k = torch.empty((B, heads, L16, 16), dtype=torch.float32, device=dev) if roles == 7 else None
    v = torch.empty((B, heads, 16, L16), dtype=torch.float32, device=dev) if roles == 7 else None
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    layer = [att, residual, _w(out_proj), _c(out_proj.bias), _c(norm1.weight), _c(norm1.bias), _c(scale_shift), _w(fc1), _c(fc1.bias),
             _w(fc2), _c(fc2.bias), _c(norm2.weight), _c(norm2.bias)]
    nxt = [_c(next_scale_shift), _w(next_q_proj), _c(next_q_proj.bias), None if roles == 1 else _w(next_kv_proj),
           None if roles == 1 else _c(next_kv_proj.bias), cs, sn]
    a13 = (Ct.c_void_p * 13)(*[None if t is None else t.data_ptr() for t in layer])
    a7 = (Ct.c_void_p * 7)(*[None if t is None else t.data_ptr() for t in nxt])
    _lib.check(_lib.lib().mmf_out_ffn_qkv(Ct.cast(a13, Ct.c_void_p), float(norm1.eps), float(norm2.eps), _lib.dptr(out), Ct.cast(a7, Ct.c_void_p),
                                          _lib.dptr(q), _lib.dptr(k), _lib.dptr(v), B, L, D, heads, roles, _lib.dptr(partials),
                                          0 if partials is None else partials.shape[2], _lib.stream_ptr(dev)), "mmf_out_ffn_qkv")
    return out, q, k, v


class Handover:
    """A hand-over buffer of the one-launch layer kernels (``cross_layer`` / ``self_layer``: self-validating 64-bit
    words + a failure word) and the launch counter that tags them.  Make (= zero) one per inference and stack: a replayed HIP graph
    repeats its tags, so the zeroing must be part of what is replayed."""

    def __init__(self, nwords: int, B: int, device):
        self.words = torch.zeros(nwords + 1, dtype=torch.int64, device=device)
        self.tag = 0
        self.B = B

    def next_tag(self) -> int:
        self.tag = self.tag % 0x7FFFFFFF + 1
        return self.tag

    def failed(self) -> bool:
        return bool(self.words[-1].item())


def CrossHandover(B: int, heads: int, device) -> Handover:
    """for ``cross_layer``: B x H x 4 partials of 18 x 16 words"""
    return Handover(B * heads * 4 * 18 * 16, B, device)


def SelfHandover(B: int, L: int, D: int, device) -> Handover:
    """for ``self_layer``: the attention output rows [B, L, D] as words"""
    return Handover(B * L * D, B, device)


def self_layer(q_heads, k_heads, v_heads_t, L: int, mask16, residual, out_proj, norm1, scale_shift, fc1, fc2, norm2, handover,
               next_scale_shift=None, next_q_proj=None, next_kv_proj=None, rot=None, heads: int = 8):
    """``attention_heads`` + ``out_ffn_mfma`` (or, with ``next_q_proj`` / ``next_kv_proj``, ``out_ffn_qkv`` producing the next
    layer's q | k | v) in ONE launch (mmf_self_layer).  Returns out [B, L, D] or (out, q_heads, k_heads, v_heads_t)."""
    import ctypes as Ct

    residual = residual.contiguous()
    B, L_, D = residual.shape
    assert L_ == L and handover.B == B and handover.words.numel() == B * L * D + 1
    dev = residual.device
    L16 = _l16(L)
    out = torch.empty_like(residual)
    layer = [None, residual, _w(out_proj), _c(out_proj.bias), _c(norm1.weight), _c(norm1.bias), _c(scale_shift), _w(fc1), _c(fc1.bias),
             _w(fc2), _c(fc2.bias), _c(norm2.weight), _c(norm2.bias)]
    a13 = (Ct.c_void_p * 13)(*[None if t is None else t.data_ptr() for t in layer])
    q = k = v = a7 = None
    if next_q_proj is not None:
        q = torch.empty((B, heads, L16, 16), dtype=torch.float32, device=dev)
        k = torch.empty((B, heads, L16, 16), dtype=torch.float32, device=dev)
        v = torch.empty((B, heads, 16, L16), dtype=torch.float32, device=dev)
        cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
        nxt = [_c(next_scale_shift), _w(next_q_proj), _c(next_q_proj.bias), _w(next_kv_proj), _c(next_kv_proj.bias), cs, sn]
        a7 = (Ct.c_void_p * 7)(*[None if t is None else t.data_ptr() for t in nxt])
    q3 = (Ct.c_void_p * 3)(q_heads.data_ptr(), k_heads.data_ptr(), v_heads_t.data_ptr())
    _lib.check(_lib.lib().mmf_self_layer(Ct.cast(a13, Ct.c_void_p), float(norm1.eps), float(norm2.eps), _lib.dptr(out),
                                         None if a7 is None else Ct.cast(a7, Ct.c_void_p), _lib.dptr(q), _lib.dptr(k), _lib.dptr(v),
                                         Ct.cast(q3, Ct.c_void_p), _lib.dptr(mask16), _lib.dptr(handover.words), handover.next_tag(), B, L, D, heads,
                                         _lib.stream_ptr(dev)), "mmf_self_layer")
    return out if q is None else (out, q, k, v)


def cross_layer(q_heads, k_heads, v_heads_t, Lq: int, Lk: int, mask16, residual, out_proj, norm1, scale_shift, fc1, fc2, norm2, handover,
                next_scale_shift=None, next_q_proj=None, rot=None, heads: int = 8, out: Optional[torch.Tensor] = None):
    """``attention_heads_split`` + ``out_ffn_mfma`` (or, with ``next_q_proj``, ``out_ffn_qkv`` producing the next layer's queries)
    in ONE launch (mmf_cross_layer): Lq <= 16 query rows per batch element over a cached context.  Returns out [B, Lq, D] or
    (out, next q_heads)."""
    import ctypes as Ct

    residual = residual.contiguous()
    B, L, D = residual.shape
    assert L == Lq and Lq <= 16 and handover.B == B
    dev = residual.device
    if out is None:
        out = torch.empty_like(residual)
    else:
        assert out.shape == residual.shape and out.dtype == torch.float32 and out.is_contiguous() and out.data_ptr() != residual.data_ptr()
    layer = [None, residual, _w(out_proj), _c(out_proj.bias), _c(norm1.weight), _c(norm1.bias), _c(scale_shift), _w(fc1), _c(fc1.bias),
             _w(fc2), _c(fc2.bias), _c(norm2.weight), _c(norm2.bias)]
    a13 = (Ct.c_void_p * 13)(*[None if t is None else t.data_ptr() for t in layer])
    qn = None
    a7 = None
    if next_q_proj is not None:
        qn = torch.empty((B, heads, 16, 16), dtype=torch.float32, device=dev)
        cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
        nxt = [_c(next_scale_shift), _w(next_q_proj), _c(next_q_proj.bias), None, None, cs, sn]
        a7 = (Ct.c_void_p * 7)(*[None if t is None else t.data_ptr() for t in nxt])
    q3 = (Ct.c_void_p * 3)(q_heads.data_ptr(), k_heads.data_ptr(), v_heads_t.data_ptr())
    _lib.check(_lib.lib().mmf_cross_layer(Ct.cast(a13, Ct.c_void_p), float(norm1.eps), float(norm2.eps), _lib.dptr(out),
                                          None if a7 is None else Ct.cast(a7, Ct.c_void_p), _lib.dptr(qn), Ct.cast(q3, Ct.c_void_p), _lib.dptr(mask16),
                                          _lib.dptr(handover.words), handover.next_tag(), B, Lq, Lk, D, heads, _lib.stream_ptr(dev)),
               "mmf_cross_layer")
    return out if qn is None else (out, qn)


def _ptr_array(tensors):
    import ctypes as Ct

    arr = (Ct.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
    return Ct.cast(arr, Ct.c_void_p), arr


def qkv_heads2(x0, x1, ss01, q_proj01, kv_proj01, rot, heads: int):
    """``qkv_heads`` (roles 7) of TWO stacks in one launch: inputs x0 / x1 [B,L,D], per-stack (scale_shift, q_proj, kv_proj);
    the rotary tables are shared.  Returns stack-major (q_heads, k_heads, v_heads_t) of leading dimension 2 B."""
    x0, x1 = x0.contiguous(), x1.contiguous()
    B, L, D = x0.shape
    L16 = _l16(L)
    dev = x0.device
    q = torch.empty((2 * B, heads, L16, 16), dtype=torch.float32, device=dev)
    k = torch.empty((2 * B, heads, L16, 16), dtype=torch.float32, device=dev)
    v = torch.empty((2 * B, heads, 16, L16), dtype=torch.float32, device=dev)
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    ops = []
    for st in range(2):
        ops += [_c(ss01[st]), _w(q_proj01[st]), _c(q_proj01[st].bias), _w(kv_proj01[st]), _c(kv_proj01[st].bias), cs, sn]
    ptr, _keep = _ptr_array(ops)
    _lib.check(_lib.lib().mmf_qkv_heads2(_lib.dptr(x0), _lib.dptr(x1), ptr, _lib.dptr(q), _lib.dptr(k), _lib.dptr(v), B, L, D, heads,
                                         _lib.stream_ptr(dev)), "mmf_qkv_heads2")
    return q, k, v


def out_ffn_mfma2(att2, res0, res1, blocks01) -> torch.Tensor:
    """``out_ffn_mfma`` of TWO stacks in one launch.  att2 [2B, L, D] stack-major attention output; res0 / res1 [B,L,D] the
    stacks' residual inputs; blocks01[st] = (out_proj, norm1, scale_shift, fc1, fc2, norm2).  Returns [2B, L, D]."""
    import ctypes as Ct

    att2 = att2.contiguous()
    res0, res1 = res0.contiguous(), res1.contiguous()
    B2, L, D = att2.shape
    B = B2 // 2
    out = torch.empty_like(att2)
    ops, eps = [], []
    for st, res in enumerate((res0, res1)):
        out_proj, norm1, ss, fc1, fc2, norm2 = blocks01[st]
        ops += [att2[st * B:(st + 1) * B], res, _w(out_proj), _c(out_proj.bias), _c(norm1.weight), _c(norm1.bias), _c(ss), _w(fc1),
                _c(fc1.bias), _w(fc2), _c(fc2.bias), _c(norm2.weight), _c(norm2.bias)]
        eps += [float(norm1.eps), float(norm2.eps)]
    ptr, _keep = _ptr_array(ops)
    e4 = (Ct.c_float * 4)(*eps)
    _lib.check(_lib.lib().mmf_out_ffn_mfma2(ptr, Ct.cast(e4, Ct.c_void_p), _lib.dptr(out), B, L, D, _lib.stream_ptr(att2.device)),
               "mmf_out_ffn_mfma2")
    return out


def paired_self_attention_stacks(stack0, stack1, x, ss_of, rot, key_padding_mask2_16, heads: int):
    """Two self-attention AttentionStacks of identical shape on the same input x [B,L,D] (the rotation and position stacks
    of the diffusion head), every layer's three launches shared between them.  ss_of(adaln) -> that block's (scale | shift)
    [B, 2D] or None; key_padding_mask2_16: pad_mask16 of the key padding mask repeated for both stacks ([2B, L16]) or None.
    Returns the two outputs [B,L,D]."""
    B, L, D = x.shape
    n = len(stack0.attn)
    assert len(stack1.attn) == n and n > 0
    x0 = x1 = x
    A0, A1 = stack0.attn[0], stack1.attn[0]
    qh, kh, vt = qkv_heads2(x0, x1, (ss_of(A0.adaln), ss_of(A1.adaln)), (A0.attn.q_proj, A1.attn.q_proj), (A0.attn.kv_proj, A1.attn.kv_proj),
                            rot, heads)
    for li in range(n):
        blocks = []
        for st in (stack0, stack1):
            blk, ffw = st.attn[li], st.ffw[li]
            blocks.append((blk.attn.out_proj, blk.norm, ss_of(ffw.adaln), ffw.fc1, ffw.fc2, ffw.norm))
        att2 = attention_heads(qh, kh, vt, None, L, L, key_padding_mask2_16)
        if li + 1 < n and FUSE_OUT_FFN_QKV:
            N0, N1 = stack0.attn[li + 1], stack1.attn[li + 1]
            y2, qh, kh, vt = out_ffn_qkv2(att2, x0, x1, blocks, (ss_of(N0.adaln), ss_of(N1.adaln)), (N0.attn.q_proj, N1.attn.q_proj),
                                          (N0.attn.kv_proj, N1.attn.kv_proj), rot, heads)
            x0, x1 = y2[:B], y2[B:]
            continue
        y2 = out_ffn_mfma2(att2, x0, x1, blocks)
        x0, x1 = y2[:B], y2[B:]
        if li + 1 < n:
            N0, N1 = stack0.attn[li + 1], stack1.attn[li + 1]
            qh, kh, vt = qkv_heads2(x0, x1, (ss_of(N0.adaln), ss_of(N1.adaln)), (N0.attn.q_proj, N1.attn.q_proj),
                                    (N0.attn.kv_proj, N1.attn.kv_proj), rot, heads)
    return x0, x1


def out_ffn_qkv2(att2, res0, res1, blocks01, next_ss01, next_q_proj01, next_kv_proj01, rot, heads: int):
    """``out_ffn_mfma2`` followed, in the same launch, by ``qkv_heads2`` of the stacks' NEXT layers on the result.  Returns
    (out [2B, L, D], q_heads, k_heads, v_heads_t) -- the latter stack-major, leading dimension 2 B."""
    import ctypes as Ct

    att2 = att2.contiguous()
    res0, res1 = res0.contiguous(), res1.contiguous()
    B2, L, D = att2.shape
    B = B2 // 2
    L16 = _l16(L)
    dev = att2.device
    out = torch.empty_like(att2)
    q = torch.empty((2 * B, heads, L16, 16), dtype=torch.float32, device=dev)
    k = torch.empty((2 * B, heads, L16, 16), dtype=torch.float32, device=dev)
    v = torch.empty((2 * B, heads, 16, L16), dtype=torch.float32, device=dev)
    cs, sn = (None, None) if rot is None else (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    ops, eps, nxt = [], [], []
    for st, res in enumerate((res0, res1)):
        out_proj, norm1, ss, fc1, fc2, norm2 = blocks01[st]
        ops += [att2[st * B:(st + 1) * B], res, _w(out_proj), _c(out_proj.bias), _c(norm1.weight), _c(norm1.bias), _c(ss), _w(fc1),
                _c(fc1.bias), _w(fc2), _c(fc2.bias), _c(norm2.weight), _c(norm2.bias)]
        eps += [float(norm1.eps), float(norm2.eps)]
        nxt += [_c(next_ss01[st]), _w(next_q_proj01[st]), _c(next_q_proj01[st].bias), _w(next_kv_proj01[st]), _c(next_kv_proj01[st].bias), cs, sn]
    ptr, _keep = _ptr_array(ops)
    nptr, _keep2 = _ptr_array(nxt)
    e4 = (Ct.c_float * 4)(*eps)
    _lib.check(_lib.lib().mmf_out_ffn_qkv2(ptr, Ct.cast(e4, Ct.c_void_p), _lib.dptr(out), nptr, _lib.dptr(q), _lib.dptr(k), _lib.dptr(v), B, L, D,
                                           heads, _lib.stream_ptr(dev)), "mmf_out_ffn_qkv2")
    return out, q, k, v


# ---- head and tail of a denoising step (mmf_kernels_policy_head.hip) -----------------------------------------------------------
def step_prologue(trajectory, traj_encoder, pos_table, time_row, history, rot_freq, adaln_wt, adaln_bias, seq_cos, seq_sin):
    """One launch for everything in front of the first attention layer of a denoising step.  trajectory [B,L,G,9];
    pos_table [L*G, D]; time_row [D] (this step's time embedding); history [B, D]; adaln_wt [D, NA] / adaln_bias [NA]: the
    stacked AdaLN projections (adaln_wt None: trajectory tokens and rotary codes only); seq_cos / seq_sin [B, Ls, D]:
    sequence-wide rotary tables whose first L*G rows are (re)written.  Returns (tokens [B, L*G, D], adaln [B, NA] or None)."""
    trajectory = trajectory.contiguous()
    B = trajectory.shape[0]
    nt = trajectory.shape[1] * trajectory.shape[2]
    D = pos_table.shape[-1]
    NA = 0 if adaln_wt is None else adaln_wt.shape[1]
    assert seq_cos.is_contiguous() and seq_sin.is_contiguous() and seq_cos.shape[1] >= nt
    tokens = torch.empty((B, nt, D), dtype=torch.float32, device=trajectory.device)
    adaln = torch.empty((B, NA), dtype=torch.float32, device=trajectory.device) if NA else None
    _lib.check(_lib.lib().mmf_step_prologue(_lib.dptr(trajectory), B, nt, _lib.dptr(_wt(traj_encoder)), _lib.dptr(_c(traj_encoder.bias)),
                                            _lib.dptr(_c(pos_table)), _lib.dptr(_c(time_row) if NA else None),
                                            _lib.dptr(_c(history) if NA else None), _lib.dptr(_c(rot_freq)), _lib.dptr(adaln_wt),
                                            _lib.dptr(adaln_bias if NA else None), NA, _lib.dptr(tokens), _lib.dptr(adaln), _lib.dptr(seq_cos),
                                            _lib.dptr(seq_sin), seq_cos.stride(0), D, _lib.stream_ptr(trajectory.device)), "mmf_step_prologue")
    return tokens, adaln


def _head_weight_array(head):
    import ctypes as Ct

    lins = [head.rotation_proj, head.position_proj, head.rotation_out[0], head.rotation_out[2], head.position_out[0], head.position_out[2],
            head.openness_out[0], head.openness_out[2]]
    if head.head_yaw_out is not None:
        lins += [head.head_yaw_out[0], head.head_yaw_out[2]]
    ptrs, keep = [], []
    for lin in lins:
        w, b = _wt(lin), _c(lin.bias)
        keep += [w, b]
        ptrs += [w.data_ptr(), b.data_ptr()]
    ptrs += [0] * (20 - len(ptrs))
    return (Ct.c_void_p * 20)(*[p or None for p in ptrs]), keep


def head_outputs(head, rot_seq, pos_seq, B: int, L: int, G: int):
    """rotation_proj / position_proj + the position / rotation / openness / head-yaw MLPs of the DiffusionHead on the
    trajectory rows (the first L*G) of the two output stacks: (pred [B,L,G,10], head_yaw [B,L,1] or None)."""
    import ctypes as Ct

    assert rot_seq.is_contiguous() and pos_seq.is_contiguous() and rot_seq.shape == pos_seq.shape
    arr, _keep = _head_weight_array(head)
    dev = rot_seq.device
    pred = torch.empty((B, L, G, 10), dtype=torch.float32, device=dev)
    yaw = torch.empty((B, L, 1), dtype=torch.float32, device=dev) if head.head_yaw_out is not None else None
    _lib.check(_lib.lib().mmf_head_outputs(_lib.dptr(rot_seq), _lib.dptr(pos_seq), rot_seq.stride(0), B, L, G, Ct.cast(arr, Ct.c_void_p),
                                           _lib.dptr(pred), _lib.dptr(yaw), rot_seq.shape[-1], _lib.stream_ptr(dev)), "mmf_head_outputs")
    return pred, yaw


def step_tail(head, rot_seq, pos_seq, traj, noise, coef_pos, coef_rot, pos_table, rot_freq, seq_cos, seq_sin, last: bool):
    """The end of a denoising step and the beginning of the next in one launch: ``head_outputs``, the reverse-diffusion
    update of ``traj`` [B,L,G,9] (``ddpm_step``), and -- unless ``last`` -- the next step's trajectory tokens and rotary codes
    (the token part of ``step_prologue``).  Returns (pred, head_yaw, new traj, next tokens or None)."""
    import ctypes as Ct

    assert rot_seq.is_contiguous() and pos_seq.is_contiguous() and rot_seq.shape == pos_seq.shape
    traj, noise = traj.contiguous(), noise.contiguous()
    B, L, G, _ = traj.shape
    arr, _keep = _head_weight_array(head)
    dev = rot_seq.device
    D = rot_seq.shape[-1]
    pred = torch.empty((B, L, G, 10), dtype=torch.float32, device=dev)
    yaw = torch.empty((B, L, 1), dtype=torch.float32, device=dev) if head.head_yaw_out is not None else None
    new_traj = torch.empty_like(traj)
    tokens = None if last else torch.empty((B, L * G, D), dtype=torch.float32, device=dev)
    a, b = (Ct.c_float * 6)(*coef_pos), (Ct.c_float * 6)(*coef_rot)
    enc = head.traj_encoder
    _lib.check(_lib.lib().mmf_step_tail(_lib.dptr(rot_seq), _lib.dptr(pos_seq), rot_seq.stride(0), B, L, G, Ct.cast(arr, Ct.c_void_p),
                                        _lib.dptr(pred), _lib.dptr(yaw), _lib.dptr(traj), _lib.dptr(noise), Ct.cast(a, Ct.c_void_p),
                                        Ct.cast(b, Ct.c_void_p), _lib.dptr(new_traj), _lib.dptr(_wt(enc)), _lib.dptr(_c(enc.bias)),
                                        _lib.dptr(_c(pos_table)), _lib.dptr(_c(rot_freq)), _lib.dptr(tokens), _lib.dptr(seq_cos),
                                        _lib.dptr(seq_sin), seq_cos.stride(0), D, _lib.stream_ptr(dev)), "mmf_step_tail")
    return pred, yaw, new_traj, tokens
