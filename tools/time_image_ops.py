"""Times the image-side kernels alone with HIP events (upsample at the baseline and reference shapes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.image_processing import upsample_features  # noqa: E402


def timed(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (c, hw, cpad) in [(64, (480, 640), 64), (768, (512, 512), 768), (384, (512, 512), 768), (12, (96, 128), 16)]:
    low = torch.randn(c, 16, 16, device="cuda")
    ms = timed(lambda: upsample_features(low, hw, cpad))
    mb = hw[0] * hw[1] * cpad * 2 / 1e6
    print(f"upsample {c}x16x16 -> {hw[0]}x{hw[1]}x{cpad}: {ms * 1e3:.1f} us, {mb / ms / 1e3:.2f} TB/s written")
