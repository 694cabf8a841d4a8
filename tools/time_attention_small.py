"""Stand-alone timing of mmf_attention_small at the policy's inference shapes, against torch SDPA."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.diffuser_actor import fused_ops as FO  # noqa: E402


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (B, Lq, Lk) in ((1, 616, 616), (1, 2, 3072), (32, 616, 616)):
    D, H = 120, 8
    q, k, v = (torch.randn(B, L, D, device="cuda") for L in (Lq, Lk, Lk))
    pad = torch.zeros(B, Lk, dtype=torch.bool, device="cuda")
    qh, kh, vh = (t.reshape(B, -1, H, D // H).transpose(1, 2) for t in (q, k, v))
    with torch.no_grad():
        t_f = timed(lambda: FO.attention_small(q, k, v, pad, H))
        t_s = timed(lambda: F.scaled_dot_product_attention(qh, kh, vh, attn_mask=(~pad)[:, None, None, :]))
    print(f"B={B} Lq={Lq} Lk={Lk}: mmf_attention_small {t_f:.1f} us/call (incl. ~10 us of Python), torch SDPA {t_s:.1f} us/call")
