"""Independent checker of the HIP farthest-point-sampling kernel (test infrastructure, numpy only): the published definition
of the dgl op the reference calls (mindmap/diffuser_actor/encoder.py:366-370: ``farthest_point_sampler(features, npoints,
start_idx=0)``): start at `start_idx`; keep, for every point, its smallest squared L2 distance (in feature space) to the
points selected so far; the next point is the one where that is largest, the FIRST such index on ties.  Distances are
accumulated channel by channel in float32 (the kernel's order), so that near-ties resolve identically."""
import numpy as np


def farthest_point_sampling_numpy(x: np.ndarray, npoints: int, start_idx: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, N, C = x.shape
    out = np.empty((B, npoints), dtype=np.int64)
    for b in range(B):
        pts = x[b]
        best = np.full(N, np.inf, dtype=np.float32)
        cur = int(start_idx)
        for k in range(npoints):
            out[b, k] = cur
            if k + 1 == npoints:
                break
            acc = np.zeros(N, dtype=np.float32)
            row = pts[cur]
            for c in range(C):
                d = pts[:, c] - row[c]
                acc = acc + d * d
            best = np.minimum(best, acc)
            cur = int(np.argmax(best))  # numpy's argmax returns the first maximal index
    return out
