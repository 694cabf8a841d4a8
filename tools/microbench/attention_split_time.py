"""mmf_attention_split against torch's f32 SDPA at the backbone's shape (B 32, 12 heads, 1 024 tokens, d 64): time and accuracy."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nvblox_mindmap_amd.diffuser_actor import split_linear as SL  # noqa: E402

B, L, H, d = 32, 1024, 12, 64
qkv = torch.randn(B, L, 3, H, d, device="cuda")
q, k, v = (t.permute(0, 2, 1, 3) for t in qkv.unbind(2))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_split = timed(lambda: SL.attention_split(qkv, B, L, H, d))
t_sdpa = timed(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
flops = 4.0 * B * H * L * L * d
print(f"split-operand MFMA attention {t_split:.3f} ms ({3 * flops / t_split / 1e9:.0f} TFLOP/s of fp16 products, {flops / t_split / 1e9:.0f} useful); "
      f"torch f32 SDPA {t_sdpa:.3f} ms ({flops / t_sdpa / 1e9:.0f} TFLOP/s)")
a = SL.attention_split(qkv, B, L, H, d)
b = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 2, 1, 3).reshape(B, L, H * d)
print("max |split - f32 SDPA| =", float((a - b).abs().max()), "scale", float(b.abs().max()))
