"""How fast does a captured HIP graph run its kernel nodes on this box?  N tiny dependent elementwise kernels (a chain on one
buffer) and N independent ones (N buffers), replayed as one graph: wall time per node, host time of replay()."""
import time

import torch


def bench(n, chain, size=4096):
    xs = [torch.zeros(size, device="cuda") for _ in range(1 if chain else n)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for x in xs:
            x.add_(1.0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            xs[0 if chain else i].add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    host = 0.0
    for _ in range(reps):
        h0 = time.perf_counter()
        g.replay()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the same kernels launched eagerly
    torch.cuda.synchronize()
    e0 = time.perf_counter()
    for i in range(n):
        xs[0 if chain else i].add_(1.0)
    eh = time.perf_counter() - e0
    torch.cuda.synchronize()
    ed = time.perf_counter() - e0
    print(f"n={n:5d} {'chain      ' if chain else 'independent'} size={size:8d}: graph {dt / reps / n * 1e6:6.2f} us/node (host {host / reps / n * 1e6:6.2f}); "
          f"eager {ed / n * 1e6:6.2f} us/kernel (host {eh / n * 1e6:6.2f})")


if __name__ == "__main__":
    for size in (4096, 1 << 20):
        for chain in (True, False):
            bench(2000, chain, size)
