"""Deterministic weights / inputs shared by tests/golden/make_golden_policy_head.py (which feeds them to the REFERENCE
modules and stores the outputs) and the tests (which rebuild the same arrays, convert them with
nvblox_mindmap_amd.diffuser_actor.reference_weights and compare the outputs).  numpy's PCG64 stream is stable across numpy
versions by policy, so the fixture only has to hold parameter names / shapes, checksums and the reference's outputs."""
import json

import numpy as np


def seeded_state(spec, seed):
    """spec: [(name, shape), ...] in the reference state dict's order -> {name: float32 array}.  Matrices ~ N(0, 1/fan_in),
    vectors ~ N(0, 0.2^2), LayerNorm gains around 1; the zero-initialised AdaLN projections get real values too (otherwise
    the conditioning path would be an untested identity)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        a = rng.standard_normal(shape)
        if len(shape) >= 2:
            a = a / np.sqrt(shape[-1])
        else:
            a = 0.2 * a
            if "norm" in name and name.endswith("weight"):
                a = 1.0 + a
        out[name] = a.astype(np.float32)
    return out


def state_checksum(state):
    return float(sum(np.abs(v.astype(np.float64)).sum() for v in state.values()))


def spec_to_json(spec):
    return json.dumps([[n, list(s)] for n, s in spec])


def spec_from_json(text):
    return [(n, tuple(s)) for n, s in json.loads(str(text))]


def head_inputs(seed, B, N, K, D, nhist, G, L=1, empty_sample=None, n_instr=0):
    """Inputs of one DiffusionHead forward (batch-first).  Positions are normalised-workspace-like ([-1, 1]); a random ~15 %
    of the context and of the sub-sampled context is masked; ``empty_sample`` = index of a sample whose masks are all False
    (diffusion_head.py:291-297 handles that case)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    x = {
        "trajectory": rng.standard_normal((B, L, G, 9)).astype(f32),
        "timestep": rng.integers(0, 100, size=(B,)).astype(np.int64),
        "context_feats": rng.standard_normal((B, N, D)).astype(f32),
        "context_pos": rng.uniform(-1, 1, size=(B, N, 3)).astype(f32),
        "context_mask": rng.uniform(size=(B, N)) > 0.15,
        "history_feats": rng.standard_normal((B, nhist * G, D)).astype(f32),
        "fps_feats": rng.standard_normal((B, K, D)).astype(f32),
        "fps_pos": rng.uniform(-1, 1, size=(B, K, 3)).astype(f32),
        "fps_mask": rng.uniform(size=(B, K)) > 0.15,
    }
    if n_instr:
        x["instr_feats"] = rng.standard_normal((B, n_instr, D)).astype(f32)
    if empty_sample is not None:
        x["context_mask"][empty_sample] = False
        x["fps_mask"][empty_sample] = False
    # what Encoder.run_fps hands over: masked-out rows of the sub-sampled context are zero (encoder.py:352-353,392)
    x["fps_feats"] = x["fps_feats"] * x["fps_mask"][..., None]
    return x
