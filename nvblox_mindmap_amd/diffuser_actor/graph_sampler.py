"""HIP-graph replay of the policy's denoising loop.

At inference the policy runs ``diffusion_timesteps`` (100) sequential passes of the diffusion head at batch 1: ~150 tiny
kernels each, every one far shorter than its launch overhead -- the textbook launch-bound inner loop.  ``GraphSampler``
captures the whole unrolled loop (`DiffuserActor._denoise`) once per input shape into a ``torch.cuda.CUDAGraph`` (a hipGraph
on ROCm) and replays it with a single launch.  The loop is a pure function of (encoder outputs, pre-drawn noise), which are
copied into static buffers before each replay; results are bit-identical to the eager loop.
"""
from typing import Dict, Tuple

import torch


def _flatten(enc):
    """Encoder outputs (a dict / tuple / list of tensors or None) -> list of tensors + a rebuild function."""
    leaves = []

    def walk(x):
        if isinstance(x, torch.Tensor):
            leaves.append(x)
            return ("t", len(leaves) - 1)
        if isinstance(x, dict):
            return ("d", {k: walk(v) for k, v in x.items()})
        if isinstance(x, (list, tuple)):
            return ("l" if isinstance(x, list) else "u", [walk(v) for v in x])
        return ("c", x)

    spec = walk(enc)

    def rebuild(spec, tensors):
        kind, val = spec
        if kind == "t":
            return tensors[val]
        if kind == "d":
            return {k: rebuild(v, tensors) for k, v in val.items()}
        if kind == "l":
            return [rebuild(v, tensors) for v in val]
        if kind == "u":
            return tuple(rebuild(v, tensors) for v in val)
        return val

    return leaves, spec, rebuild


class GraphSampler:
    def __init__(self, model):
        self.model = model
        self._graphs: Dict[Tuple, dict] = {}

    def _key(self, leaves, noise):
        from . import layers as L

        # the captured kernels depend on the input shapes, the timestep grid and on which inference path is switched on
        return (tuple((tuple(t.shape), t.dtype) for t in leaves), tuple(noise.shape), tuple(self.model._inference_timesteps),
                bool(L.FUSED_INFERENCE))

    def run(self, enc, noise):
        leaves, spec, rebuild = _flatten(enc)
        key = self._key(leaves, noise)
        g = self._graphs.get(key)
        if g is None:
            g = self._capture(leaves, spec, rebuild, noise)
            self._graphs[key] = g
        for dst, src in zip(g["leaves"], leaves):
            dst.copy_(src)
        g["noise"].copy_(noise)
        g["graph"].replay()
        traj, yaw = g["out"]
        return traj.clone(), (None if yaw is None else yaw.clone())

    def _capture(self, leaves, spec, rebuild, noise):
        static_leaves = [t.clone() for t in leaves]
        static_noise = noise.clone()
        static_enc = rebuild(spec, static_leaves)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):  # warm-up outside capture: lazy initialisations (cuBLAS-like handles, caches) must be done
                self.model._denoise(static_enc, static_noise)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            out = self.model._denoise(static_enc, static_noise)
        return {"graph": graph, "leaves": static_leaves, "noise": static_noise, "out": out}
