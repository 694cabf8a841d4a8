"""Consumer side of the nvblox pin kit (tools/dump_nvblox_golden.py): run the kit's own replay() on this repository's
implementations -- the HIP Mapper, or the CPU oracle behind a small adaptor with the same call surface -- and compare the
result with a dumped file, array by array.

Gates, in order of strictness:
  * north_star: identical set of allocated block indices; TSDF / feature values within 1e-5 abs;
  * fallback = the reference's own regression tolerances (mindmap/tests/utils/comparisons.py:95-109): vertices within
    1e-3 m for all but 1 % of the vertices (nearest neighbour), features within 0.05 for all but 1 %.
"""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def load_kit():
    name = "dump_nvblox_golden"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", "dump_nvblox_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def golden_files(real_only=True):
    """tests/golden/nvblox_*.npz; real_only: only files whose meta says they were dumped from upstream nvblox_torch."""
    out = []
    for f in sorted(os.listdir(GOLDEN_DIR)):
        if f.startswith("nvblox_") and f.endswith(".npz"):
            path = os.path.join(GOLDEN_DIR, f)
            meta = json.loads(str(np.load(path, allow_pickle=False)["meta"]))
            if not real_only or meta["backend"].startswith("nvblox_torch"):
                out.append(path)
    return out


# ---- backends -----------------------------------------------------------------------------------------------------------
def _task_params(kit):
    T = kit.TASK
    return dict(voxel_size=T["voxel_size_m"], max_integration_distance_m=T["projective_integrator_max_integration_distance_m"],
                raycast_subsampling=1, workspace_bounds_type=2, ws_min=T["aabb_min_m"], ws_max=T["aabb_max_m"],
                tsdf_decay_factor=T["tsdf_decay_factor"],
                appearance_measurement_weight=T["projective_appearance_integrator_measurement_weight"])


def mmf_backend(**overrides):
    """This repository's HIP Mapper, configured through the same parameter names as the oracle (fusion_common.make_mapper)."""
    from fusion_common import make_mapper

    kit = load_kit()
    kw = _task_params(kit)
    kw.update(overrides)
    return dict(make_mapper=lambda channels: make_mapper(channels, **kw), version="nvblox_mindmap_amd HIP " + json.dumps(overrides))


class _OracleLayerView:
    def __init__(self, orc, layer):
        self.orc, self.layer = orc, layer

    def num_allocated_blocks(self):
        return self.orc.num_blocks(self.layer)

    def get_all_blocks(self):
        import torch

        idx = torch.from_numpy(self.orc.block_indices(self.layer))
        if self.layer == 0:
            return torch.from_numpy(self.orc.all_tsdf()), idx
        f, w = self.orc.all_features()
        return torch.cat([torch.from_numpy(f).float(), torch.from_numpy(w)[..., None]], dim=-1), idx


class _OracleMesh:
    def __init__(self, v, f):
        import torch

        self._v, self._f = torch.from_numpy(v), torch.from_numpy(f)

    def vertices(self):
        return self._v

    def vertex_features(self):
        return self._f


class OracleAsMapper:
    """The nvblox_torch.Mapper calls the kit makes, on the CPU oracle (one mapper)."""

    def __init__(self, orc):
        self.orc = orc

    def decay(self):
        self.orc.decay()

    def add_depth_frame(self, depth, T, K, mask, mapper_id=0):
        self.orc.add_depth_frame(depth.numpy(), T.numpy(), K.numpy(), None if mask is None else mask.numpy())

    def add_color_frame(self, rgb, T, K, mask_frame=None, mapper_id=0):
        self.orc.add_color_frame(rgb.numpy(), T.numpy(), K.numpy(), None if mask_frame is None else mask_frame.numpy())

    def add_feature_frame(self, feat, T, K, mask, mapper_id=0):
        self.orc.add_feature_frame(feat.numpy(), T.numpy(), K.numpy(), None if mask is None else mask.numpy())

    def tsdf_layer_view(self, mapper_id=0):
        return _OracleLayerView(self.orc, 0)

    def feature_layer_view(self, mapper_id=0):
        return _OracleLayerView(self.orc, 2)

    def update_feature_mesh(self, mapper_id=0):
        self._mesh = _OracleMesh(*self.orc.feature_mesh())

    def get_feature_mesh(self, mapper_id=0):
        return self._mesh


def oracle_backend(**overrides):
    from oracle import oracle as O

    O.build()
    kit = load_kit()
    kw = _task_params(kit)
    kw.update(overrides)
    return dict(make_mapper=lambda channels: OracleAsMapper(O.OracleMapper(O.default_params(feature_channels=channels, **kw))),
                version="nvblox_mindmap_amd CPU oracle " + json.dumps(overrides))


def replay_like(gold, backend, device):
    """Re-run exactly what the golden file's meta describes on `backend` (same block / channel / vertex sampling)."""
    kit = load_kit()
    meta = json.loads(str(gold["meta"]))
    # vertex_stride 1 = the file holds every vertex: ours then too (a flipped item changes the vertex count, and two strided samples
    # of different lists are different vertices -- their distance would measure the sampling, not the flip)
    every_vertex = int(meta.get("vertex_stride", 0)) == 1
    return kit.replay(backend, meta["config"], meta["hole_mode"], meta["frames"], meta["decay"], meta["masks"],
                      n_block_samples=len(gold["tsdf_sample_idx"]) or 6, n_channel_samples=len(meta["feature_channel_sample"]),
                      device=device, frame_indices=meta["frame_indices"], **({"n_vertex_samples": 1 << 40} if every_vertex else {}))


# ---- comparison ----------------------------------------------------------------------------------------------------------
def _rowset(a):
    return {tuple(int(v) for v in r) for r in np.asarray(a).reshape(-1, 3)}


def _nn_dist(a, b):
    """For each row of a: distance to the nearest row of b (scipy KD-tree)."""
    from scipy.spatial import cKDTree

    if len(a) == 0 or len(b) == 0:
        return np.full((len(a),), np.inf)
    d, j = cKDTree(b).query(a)
    return d, j


def compare(gold, ours):
    """Report dict of the differences between a dumped file and our replay of it."""
    r = {}
    gt, ot = _rowset(gold["tsdf_indices"]), _rowset(ours["tsdf_indices"])
    gf, of = _rowset(gold["feature_indices"]), _rowset(ours["feature_indices"])
    r["tsdf_blocks"] = (len(gt), len(ot))
    r["tsdf_blocks_missing"], r["tsdf_blocks_extra"] = len(gt - ot), len(ot - gt)
    r["feature_blocks"] = (len(gf), len(of))
    r["feature_blocks_missing"], r["feature_blocks_extra"] = len(gf - of), len(of - gf)
    n = min(len(gold["blocks_per_frame"]), len(ours["blocks_per_frame"]))
    diff = np.nonzero(np.asarray(gold["blocks_per_frame"][:n]) != np.asarray(ours["blocks_per_frame"][:n]))[0]
    r["first_frame_with_different_block_count"] = int(diff[0]) if len(diff) else None

    def sampled(kind):
        gi, oi = gold[f"{kind}_sample_idx"].reshape(-1, 3), ours[f"{kind}_sample_idx"].reshape(-1, 3)
        worst_v, worst_w, hit = 0.0, 0.0, 0
        for k, key in enumerate(gi):
            j = np.nonzero((oi == key).all(axis=1))[0]
            if not len(j):
                continue
            hit += 1
            g, o = gold[f"{kind}_sample"][k], ours[f"{kind}_sample"][int(j[0])]
            seen = (g[..., -1] > 0) | (o[..., -1] > 0)  # unobserved voxels hold arbitrary distances
            if seen.any():
                worst_w = max(worst_w, float(np.abs(g[..., -1] - o[..., -1])[seen].max()))
                worst_v = max(worst_v, float(np.abs(g[..., :-1] - o[..., :-1])[seen].max()))
        return hit, worst_v, worst_w

    r["tsdf_sampled_blocks_found"], r["tsdf_max_abs_distance_diff"], r["tsdf_max_abs_weight_diff"] = sampled("tsdf")
    r["feature_sampled_blocks_found"], r["feature_max_abs_diff"], r["feature_max_abs_weight_diff"] = sampled("feature")
    r["n_vertices"] = (int(gold["n_vertices"]), int(ours["n_vertices"]))
    gv, ov = gold["vertices"], ours["vertices"]
    if len(gv) and len(ov):
        # the golden file holds a strided sample of the sorted vertices: compare each against OUR nearest sampled vertex only
        # when strides are 1; otherwise the sample positions differ and only the distance to our sample's hull is meaningful
        d, j = _nn_dist(gv, ov)
        r["vertex_nn_dist_p99"] = float(np.quantile(d, 0.99))
        r["vertex_frac_beyond_1mm"] = float((d > 1e-3).mean())
        fd = np.abs(gold["vertex_features"] - ours["vertex_features"][j]).max(axis=1)
        close = d <= 1e-3
        r["vertex_feature_frac_beyond_0.05"] = float((fd[close] > 0.05).mean()) if close.any() else 1.0
        r["vertex_feature_max_abs_diff"] = float(fd[close].max()) if close.any() else float("inf")
    return r


def passes_north_star(r, tol=1e-5):
    return (r["tsdf_blocks_missing"] == r["tsdf_blocks_extra"] == r["feature_blocks_missing"] == r["feature_blocks_extra"] == 0
            and r["tsdf_sampled_blocks_found"] > 0 and r["tsdf_max_abs_distance_diff"] <= tol and r["tsdf_max_abs_weight_diff"] <= tol
            and r["feature_max_abs_diff"] <= tol and r["n_vertices"][0] == r["n_vertices"][1])


def passes_reference_tolerances(r):
    """mindmap/tests/utils/comparisons.py:95-109."""
    return (r.get("vertex_frac_beyond_1mm", 1.0) <= 0.01 and r.get("vertex_feature_frac_beyond_0.05", 1.0) <= 0.01
            and abs(r["n_vertices"][0] - r["n_vertices"][1]) <= 0.01 * max(r["n_vertices"][0], 1))
