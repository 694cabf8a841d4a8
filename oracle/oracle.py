"""ctypes/numpy front-end of the CPU oracle (TEST INFRASTRUCTURE -- not product code).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  See the header of ``mmf_oracle.c`` for the spec and the parity status
("parity unpinned" for the nvblox integrator: its source and golden data are absent from
/root/reference).
"""
import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmmf_oracle.so")


class OrcParams(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_float),
        ("max_integration_distance_m", C.c_float),
        ("truncation_distance_vox", C.c_float),
        ("max_weight", C.c_float),
        ("weighting_mode", C.c_int),
        ("lin_interp_max_diff_vox", C.c_float),
        ("appearance_measurement_weight", C.c_float),
        ("appearance_max_weight", C.c_float),
        ("raycast_subsampling", C.c_int),
        ("workspace_bounds_type", C.c_int),
        ("ws_min", C.c_float * 3),
        ("ws_max", C.c_float * 3),
        ("tsdf_decay_factor", C.c_float),
        ("decayed_weight_threshold", C.c_float),
        ("deallocate_decayed_blocks", C.c_int),
        ("mesh_min_weight", C.c_float),
        ("st_subsampling", C.c_int),
        ("st_max_steps", C.c_int),
        ("st_max_ray_length_m", C.c_float),
        ("st_surface_eps_vox", C.c_float),
        ("feature_channels", C.c_int),
        ("raycast_to_truncation", C.c_int),
        ("decay_appearance_layers", C.c_int),
        ("raycast_walk_from_camera", C.c_int),
        ("appearance_blend_division", C.c_int),
        ("fma_contraction", C.c_int),
        ("block_index_by_division", C.c_int),
        ("view_truncation_band_marking", C.c_int),
        ("bilinear_four_weight_sum", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "mmf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmmf_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        assert L.orc_params_size() == C.sizeof(OrcParams)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(OrcParams)]
        L.orc_h2f.restype = C.c_float
        L.orc_h2f.argtypes = [C.c_uint16]
        L.orc_f2h.restype = C.c_uint16
        L.orc_f2h.argtypes = [C.c_float]
        for name in [
            "orc_destroy", "orc_clear", "orc_decay", "orc_add_depth_frame", "orc_add_color_frame",
            "orc_add_feature_frame", "orc_update_feature_mesh", "orc_get_feature_mesh", "orc_mesh_num_triangles",
            "orc_get_mesh_topology", "orc_num_blocks",
            "orc_get_block_indices", "orc_get_tsdf_block", "orc_get_all_tsdf", "orc_get_feature_block",
            "orc_get_all_features", "orc_get_color_block", "orc_get_all_colors", "orc_last_view_blocks",
            "orc_last_counts", "orc_get_synthetic_depth", "orc_render_synthetic_depth", "orc_query_features",
            "orc_query_tsdf", "orc_upsample_features",
        ]:
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


def default_params(**overrides) -> OrcParams:
    p = OrcParams()
    lib().orc_default_params(C.byref(p))
    set_params(p, **overrides)
    return p


def set_params(p: OrcParams, **kw) -> OrcParams:
    for k, v in kw.items():
        if k in ("ws_min", "ws_max"):
            arr = getattr(p, k)
            for i in range(3):
                arr[i] = float(v[i])
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


def upsample_features(low_hwc: np.ndarray, hf: int, wf: int, cpad: int, fma_contraction: bool = False) -> np.ndarray:
    """[h,w,Cin] f32 (channels last) -> [Hf,Wf,cpad] f16 bit patterns (uint16): the C restatement of the feature up-sample,
    exact under either setting of ``fma_contraction`` (oracle/image_ops.upsample_features is the numpy one, uncontracted)."""
    low = np.ascontiguousarray(low_hwc, dtype=np.float32)
    h, w, cin = low.shape
    out = np.empty((hf, wf, cpad), dtype=np.uint16)
    lib().orc_upsample_features(_ptr(low), C.c_int(h), C.c_int(w), C.c_int(cin), _ptr(out), C.c_int(hf), C.c_int(wf), C.c_int(cpad),
                                C.c_int(1 if fma_contraction else 0))
    return out


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleMapper:
    """One voxel map (one ``mapper_id`` of the nvblox_torch Mapper)."""

    TSDF, COLOR, FEATURE = 0, 1, 2

    def __init__(self, params: OrcParams):
        self.params = params
        self.C = params.feature_channels
        self._h = C.c_void_p(lib().orc_create(C.byref(params)))

    def __del__(self):
        try:
            if self._h:
                lib().orc_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- integration ------------------------------------------------------------------
    def add_depth_frame(self, depth, T_W_C, K, mask=None):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        H, W = depth.shape
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        T = np.ascontiguousarray(T_W_C, dtype=np.float32)
        Kk = np.ascontiguousarray(K, dtype=np.float32)
        rc = lib().orc_add_depth_frame(self._h, _ptr(depth), _ptr(mask), H, W, _ptr(T), _ptr(Kk))
        assert rc == 0

    def add_color_frame(self, rgb, T_W_C, K, mask=None):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        H, W, _ = rgb.shape
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        T = np.ascontiguousarray(T_W_C, dtype=np.float32)
        Kk = np.ascontiguousarray(K, dtype=np.float32)
        rc = lib().orc_add_color_frame(self._h, _ptr(rgb), _ptr(mask), H, W, _ptr(T), _ptr(Kk))
        assert rc == 0

    def add_feature_frame(self, feat, T_W_C, K, mask=None):
        feat = np.ascontiguousarray(feat, dtype=np.float16)
        H, W, Cc = feat.shape
        mask = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        T = np.ascontiguousarray(T_W_C, dtype=np.float32)
        Kk = np.ascontiguousarray(K, dtype=np.float32)
        rc = lib().orc_add_feature_frame(self._h, _ptr(feat.view(np.uint16)), _ptr(mask), H, W, Cc, _ptr(T), _ptr(Kk))
        assert rc == 0, "feature channel count mismatch"

    def decay(self):
        lib().orc_decay(self._h)

    def clear(self):
        lib().orc_clear(self._h)

    # -- inspection -------------------------------------------------------------------
    def num_blocks(self, layer=0) -> int:
        return lib().orc_num_blocks(self._h, layer)

    def block_indices(self, layer=0) -> np.ndarray:
        n = self.num_blocks(layer)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().orc_get_block_indices(self._h, layer, _ptr(out))
        return out

    def all_tsdf(self) -> np.ndarray:
        """[n, 8, 8, 8, 2] (distance, weight), live order."""
        n = self.num_blocks(0)
        out = np.zeros((n, 8, 8, 8, 2), dtype=np.float32)
        if n:
            lib().orc_get_all_tsdf(self._h, _ptr(out))
        return out

    def all_features(self):
        """([n, 8, 8, 8, C] f16, [n, 8, 8, 8] f32 weights), live order."""
        n = self.num_blocks(2)
        f = np.zeros((n, 8, 8, 8, self.C), dtype=np.float16)
        w = np.zeros((n, 8, 8, 8), dtype=np.float32)
        if n:
            lib().orc_get_all_features(self._h, _ptr(f), _ptr(w))
        return f, w

    def all_colors(self):
        n = self.num_blocks(1)
        c = np.zeros((n, 8, 8, 8, 3), dtype=np.uint8)
        w = np.zeros((n, 8, 8, 8), dtype=np.float32)
        if n:
            lib().orc_get_all_colors(self._h, _ptr(c), _ptr(w))
        return c, w

    def last_view_blocks(self) -> np.ndarray:
        n = lib().orc_last_view_blocks(self._h, None)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().orc_last_view_blocks(self._h, _ptr(out))
        return out

    def last_counts(self):
        a, b = C.c_int(0), C.c_int(0)
        lib().orc_last_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def synthetic_depth(self) -> np.ndarray:
        ws, hs = C.c_int(0), C.c_int(0)
        lib().orc_get_synthetic_depth(self._h, None, C.byref(ws), C.byref(hs))
        out = np.zeros((hs.value, ws.value), dtype=np.float32)
        if out.size:
            lib().orc_get_synthetic_depth(self._h, _ptr(out), C.byref(ws), C.byref(hs))
        return out

    def render_synthetic_depth(self, H, W, T_W_C, K) -> np.ndarray:
        T = np.ascontiguousarray(T_W_C, dtype=np.float32)
        Kk = np.ascontiguousarray(K, dtype=np.float32)
        lib().orc_render_synthetic_depth(self._h, H, W, _ptr(T), _ptr(Kk))
        return self.synthetic_depth()

    def feature_mesh(self):
        """(vertices [V,3] f32, vertex_features [V,C] f16)."""
        V = lib().orc_update_feature_mesh(self._h)
        v = np.zeros((V, 3), dtype=np.float32)
        f = np.zeros((V, self.C), dtype=np.float16)
        if V:
            lib().orc_get_feature_mesh(self._h, _ptr(v), _ptr(f))
        return v, f

    def mesh_topology(self):
        """(triangles [T,3] int32 into the vertices of feature_mesh(), vertex colours [V,3] uint8) of the current map."""
        V = lib().orc_update_feature_mesh(self._h)
        T = lib().orc_mesh_num_triangles(self._h)
        t = np.zeros((T, 3), dtype=np.int32)
        c = np.zeros((V, 3), dtype=np.uint8)
        lib().orc_get_mesh_topology(self._h, _ptr(t), _ptr(c))
        return t, c

    def query_features(self, pts) -> np.ndarray:
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.zeros((pts.shape[0], self.C + 1), dtype=np.float32)
        if pts.shape[0]:
            lib().orc_query_features(self._h, _ptr(pts), pts.shape[0], _ptr(out))
        return out

    def query_tsdf(self, pts) -> np.ndarray:
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        out = np.zeros((pts.shape[0], 2), dtype=np.float32)
        if pts.shape[0]:
            lib().orc_query_tsdf(self._h, _ptr(pts), pts.shape[0], _ptr(out))
        return out


def num_threads() -> int:
    return lib().orc_num_threads()


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))
