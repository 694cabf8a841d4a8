"""Does the split GEMM's rate depend on how its output is provided?  (torch.mm(..., out_dtype=f32) allocating vs out=), repeated
in both orders, real backbone-like data scale."""
import torch

def timed(fn, n=20):
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
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.float16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.float16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    res = []
    for rep in range(3):
        res.append(("alloc", timed(lambda: torch.mm(a, w.t(), out_dtype=torch.float32))))
        res.append(("out=", timed(lambda: torch.mm(a, w.t(), out_dtype=torch.float32, out=out))))
    print(f"K0={K0} N={N}: " + ", ".join(f"{k} {v:.3f}" for k, v in res), flush=True)
