"""Which part of the captured training step makes graph.replay() block the host?  Times replay() host time vs total for:
the FPS call alone (hipMallocAsync / hipFreeAsync / hipMemsetAsync nodes), a chain of 2 400 small kernels, and both."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling  # noqa: E402


def probe(name, fn, reps=6):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = cpu = 0.0
    for _ in range(reps):
        h0, c0 = time.perf_counter(), time.thread_time()
        g.replay()
        host += time.perf_counter() - h0
        cpu += time.thread_time() - c0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:40s}: total {dt / reps * 1e3:8.3f} ms per replay, host wall in replay() {host / reps * 1e3:8.3f} ms, host CPU {cpu / reps * 1e3:8.3f} ms")


x = torch.randn(32, 3072, 120, device="cuda")
buf = torch.zeros(1 << 16, device="cuda")
big = torch.randn(8192, 8192, device="cuda")


def small(n=2400):
    for _ in range(n):
        buf.add_(1.0)


def gemms(n=30):
    for _ in range(n):
        torch.mm(big, big)


probe("fps alone", lambda: farthest_point_sampling(x, 614, 0))
probe("2400 small kernels", small)
probe("30 big GEMMs", gemms)
probe("30 big GEMMs + 2400 small", lambda: (gemms(), small()))
probe("fps + 2400 small kernels", lambda: (farthest_point_sampling(x, 614, 0), small()))
side = torch.cuda.Stream()


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        gemms(10)
    small()
    main.wait_stream(side)


probe("fork: 10 GEMMs || 2400 small", forked)
