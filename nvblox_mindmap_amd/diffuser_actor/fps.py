"""Farthest-point sampling in feature space (HIP kernel; counterpart of the dgl CUDA op used at
mindmap/diffuser_actor/encoder.py:366-370)."""
import torch

from .. import _lib


def farthest_point_sampling(x: torch.Tensor, npoints: int, start_idx: int = 0) -> torch.Tensor:
    """x (B,N,C) float32 on the GPU -> indices (B,npoints) int64: start_idx first, then repeatedly the point whose
    squared distance to the selected set is largest (first index on ties)."""
    if not x.is_cuda:
        raise RuntimeError("farthest_point_sampling runs on the GPU only (no CPU fallback)")
    B, N, C = x.shape
    xx = x.detach().to(torch.float32).contiguous()
    out = torch.empty((B, npoints), dtype=torch.int64, device=x.device)
    # scratch from torch's allocator (inside a HIP-graph capture: from the graph's pool), not from the runtime: a captured
    # hipMallocAsync makes hipGraphLaunch run the whole graph synchronously from the host
    L = _lib.lib()
    nbytes = int(L.mmf_fps_workspace_bytes(B, N, C))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    _lib.check(L.mmf_farthest_point_sampling_ws(_lib.dptr(xx), B, N, C, int(npoints), int(start_idx), _lib.dptr(out), _lib.dptr(ws), nbytes,
                                                _lib.stream_ptr(x.device)), "mmf_farthest_point_sampling")
    return out


def farthest_point_sampling_cpu(x: torch.Tensor, npoints: int, start_idx: int = 0) -> torch.Tensor:
    """The same selection for a model that runs on the CPU (BASELINE configs[0]: "diffuser_actor single forward ... PyTorch CPU,
    plumbing only"), CPU tensors only -- a GPU tensor takes the HIP kernel, and the kernel's checker is an independent numpy
    restatement under tests/ (tests/fps_restatement.py), not this function.  O(npoints) sequential steps."""
    if x.is_cuda:
        raise RuntimeError("farthest_point_sampling_cpu is the CPU model's path; GPU tensors take farthest_point_sampling")
    B, N, C = x.shape
    xx = x.detach().to(torch.float32)
    dist = torch.full((B, N), float("inf"), device=x.device)
    idx = torch.empty((B, npoints), dtype=torch.int64, device=x.device)
    cur = torch.full((B,), int(start_idx), dtype=torch.int64, device=x.device)
    ar = torch.arange(B, device=x.device)
    for it in range(npoints):
        idx[:, it] = cur
        if it + 1 == npoints:
            break
        d = xx - xx[ar, cur][:, None, :]
        acc = torch.zeros((B, N), device=x.device)
        for c in range(C):  # sequential float32 accumulation, the kernel's order
            acc = acc + d[..., c] * d[..., c]
        dist = torch.minimum(dist, acc)
        cur = dist.argmax(dim=1)  # first maximal index
    return idx
