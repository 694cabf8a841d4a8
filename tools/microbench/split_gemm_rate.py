"""Rate of the frozen backbone's split-operand GEMMs (fp16 in, f32 out: rows = 32 x 1024 tokens, K = 3 K0 + 64) as torch.mm issues
them, beside variants that bound what a better-chosen hipBLASLt solution could give: fp16 output, the plain K0 product, and
both operand layouts.  Prints TFLOP/s per shape."""
import torch

def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

M = 32 * 1024
for (K0, N) in ((768, 2304), (768, 768), (768, 3072), (3072, 768)):
    K = 3 * K0 + 64
    a = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16)   # [N, K] row-major, used as w.t()
    wt = w.t().contiguous()                                      # [K, N] row-major
    fl = 2.0 * M * N * K
    r = {}
    r["f32out w.t()"] = timed(lambda: torch.mm(a, w.t(), out_dtype=torch.float32))
    r["f32out wt"] = timed(lambda: torch.mm(a, wt, out_dtype=torch.float32))
    r["f16out w.t()"] = timed(lambda: torch.mm(a, w.t()))
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    r["f32out into out="] = timed(lambda: torch.mm(a, w.t(), out_dtype=torch.float32, out=out))
    # the transposed product: out^T [N, M] = w [N, K] @ a^T
    r["f32out transposed"] = timed(lambda: torch.mm(w, a.t(), out_dtype=torch.float32))
    print(f"K0={K0} N={N} K={K}: " + ", ".join(f"{k}: {v:.3f} ms = {fl / v / 1e9:.0f} TF" for k, v in r.items()), flush=True)
