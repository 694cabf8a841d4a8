"""Last step between the surface mesh and the model: sample / pad to N vertices
(mirror of mindmap/data_loading/vertex_sampling.py:29-170; same RNG draws, so the same
``torch.manual_seed`` gives the same selection as the reference)."""
from enum import Enum
from typing import Optional, Tuple

import torch

from .. import _lib


def randperm_prefix(n: int, k: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """``torch.randperm(n)[:k]`` (CPU default generator, or ``generator``; int64, host) with the same draws and the same generator
    state afterwards, in O(k) swaps: ``mmf_host_randperm_prefix`` advances torch's own serialised generator state.  A generator
    whose serialised state is not the 5056-byte mt19937 layout is left to ``torch.randperm``."""
    state = torch.get_rng_state() if generator is None else generator.get_state()
    k = min(int(k), int(n))
    out = torch.empty(k, dtype=torch.int64)
    rc = _lib.lib().mmf_host_randperm_prefix(state.data_ptr(), state.numel(), int(n), k, out.data_ptr())
    if rc != 0:
        return torch.randperm(n, generator=generator)[:k]
    if generator is None:
        torch.set_rng_state(state)
    else:
        generator.set_state(state)
    return out


class VertexSamplingMethod(Enum):
    RANDOM_WITHOUT_REPLACEMENT = "random_without_replacement"
    RANDOM_WITH_REPLACEMENT = "random_with_replacement"
    LOWEST = "lowest"
    NONE = "none"


def _seed(seed: Optional[int], generator: Optional[torch.Generator]) -> None:
    if seed is None:
        return
    if generator is None:
        torch.manual_seed(seed)  # the reference's call (vertex_sampling.py:143): every default generator of the process, CUDA's included
    else:
        generator.manual_seed(seed)  # a caller-owned stream (a loader thread): the same CPU draws, nobody else's generator touched


def select_vertex_indices(n: int, desired_num_vertices: int, method: VertexSamplingMethod, device, seed: Optional[int] = None,
                          z: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Row indices ``sample_to_n_vertices`` keeps when n > desired_num_vertices (same RNG draws / same sort); ``z``: the
    vertices' z column, needed by LOWEST only.  ``generator``: draw (and seed) THIS CPU generator instead of the process-wide
    default one -- what a loader thread inside the training process passes, so that it neither reseeds nor races the generators
    the trainer's diffusion noise comes from (the values drawn for a given seed are the same either way)."""
    if method == VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT:
        _seed(seed, generator)
        # CPU default generator, exactly like the reference (vertex_sampling.py:143-145)
        return randperm_prefix(n, desired_num_vertices, generator).to(device)
    if method == VertexSamplingMethod.RANDOM_WITH_REPLACEMENT:
        _seed(seed, generator)
        return torch.randint(0, n, (desired_num_vertices,), generator=generator).to(device)
    if method == VertexSamplingMethod.LOWEST:
        # the reference sorts by -z (np.argsort(-vertices[:, 2]), vertex_sampling.py:122): i.e. it keeps the
        # HIGHEST z despite the name; a stable sort reproduces numpy's tie order
        return torch.sort(-z, stable=True).indices[:desired_num_vertices]
    raise ValueError(f"Vertex sampling method {method} is not yet implemented.")


def sample_to_n_vertices(vertices: torch.Tensor, features: torch.Tensor, desired_num_vertices: int,
                         method: VertexSamplingMethod, seed: Optional[int] = None, generator: Optional[torch.Generator] = None
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(V,3), (V,C) -> (N,3), (N,C), (N,) valid mask.  V == N or method NONE: inputs are returned unchanged."""
    assert vertices.dim() == 2 and features.dim() == 2
    assert vertices.shape[0] == features.shape[0]
    n = features.shape[0]
    dev = vertices.device
    if method == VertexSamplingMethod.NONE or n == desired_num_vertices:
        return vertices, features, torch.ones(n, device=dev, dtype=torch.bool)
    if n > desired_num_vertices:
        valid_mask = torch.ones(desired_num_vertices, device=dev, dtype=torch.bool)
        sel = select_vertex_indices(n, desired_num_vertices, method, dev, seed, vertices[:, 2], generator)
        vertices, features = vertices[sel, :], features[sel, :]
    else:
        pad = desired_num_vertices - n
        features = torch.cat([features, torch.zeros((pad, features.shape[1]), device=features.device, dtype=features.dtype)], dim=0)
        vertices = torch.cat([vertices, torch.zeros((pad, vertices.shape[1]), device=dev, dtype=vertices.dtype)], dim=0)
        valid_mask = torch.ones(desired_num_vertices, device=dev, dtype=torch.bool)
        valid_mask[n:] = False
    assert vertices.shape[0] == desired_num_vertices and features.shape[0] == desired_num_vertices
    return vertices, features, valid_mask
