"""nvblox_torch.mapper.Mapper on MI355X: ctypes front-end of libmmfusion.so.

Call contract taken from the reference's call sites:
  Mapper(voxel_sizes_m, integrator_types, mapper_parameters)   mapping/helpers/nvblox_mapping_helpers.py:72-76
  add_depth_frame(depth, T_W_C, K, mask, mapper_id)            :207-209
  add_color_frame(rgb, T_W_C, K, mask_frame=, mapper_id=)      :212-218
  add_feature_frame(feat, T_W_C, K_feat, mask, mapper_id)      :255-261
  decay() / clear()                                            mapping/isaaclab_nvblox_mapper.py:252-258
  update_feature_mesh / get_feature_mesh -> vertices(), vertex_features()   mapping/helpers/nvblox_output_helpers.py:49-52
  tsdf_layer_view / feature_layer_view / query_layer           visualization/visualizer.py:678-691, paper/utils/utils.py:101-121
All device tensors are borrowed by pointer on torch's current HIP stream; nothing is copied to the host
on the per-frame path.  Errors surface as RuntimeError (upstream raises from C++ asserts).
"""
import ctypes as C
from enum import Enum
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .. import _lib
from . import constants as _constants
from .mapper_params import MapperParams
from .projective_integrator_types import ProjectiveIntegratorType


class QueryType(Enum):
    TSDF = 0
    FEATURE = 2


def _host_f32(x, shape) -> np.ndarray:
    """Pose / intrinsics arrive as CPU torch tensors in the reference (``.cpu()`` at the call site)."""
    if isinstance(x, torch.Tensor):
        x = x.detach().to("cpu", torch.float32).numpy()
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    if a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _check_dev(t: torch.Tensor, name: str, dtype, ndim: int) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a tensor on the GPU")
    if t.dtype != dtype:
        raise ValueError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if t.ndim != ndim:
        raise ValueError(f"{name} must have {ndim} dimensions, got shape {tuple(t.shape)}")
    return t if t.is_contiguous() else t.contiguous()


def _mask_u8(mask: Optional[torch.Tensor], shape) -> Optional[torch.Tensor]:
    if mask is None:
        return None
    if not mask.is_cuda:
        raise ValueError("mask must be a tensor on the GPU")
    if tuple(mask.shape) != tuple(shape):
        raise ValueError(f"mask shape {tuple(mask.shape)} does not match the frame {tuple(shape)}")
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)  # 0/1 bytes: reinterpret, no copy
    if mask.dtype != torch.uint8:
        mask = mask.to(torch.uint8)
    return mask.contiguous()


def _write_ply(path: str, vertices, triangles, colors=None) -> None:
    """Binary little-endian PLY (what Open3D / MeshLab read): float32 xyz [+ uchar rgb], int32 triangle indices."""
    v = vertices.detach().to("cpu", torch.float32).numpy()
    t = triangles.detach().to("cpu", torch.int32).numpy()
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {v.shape[0]}", "property float x", "property float y",
              "property float z"]
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        header += ["property uchar red", "property uchar green", "property uchar blue"]
    header += [f"element face {t.shape[0]}", "property list uchar int vertex_indices", "end_header"]
    vert = np.zeros(v.shape[0], dtype=fields)
    vert["x"], vert["y"], vert["z"] = v[:, 0], v[:, 1], v[:, 2]
    if colors is not None:
        c = colors.detach().to("cpu", torch.uint8).numpy()
        vert["red"], vert["green"], vert["blue"] = c[:, 0], c[:, 1], c[:, 2]
    face = np.zeros(t.shape[0], dtype=[("n", "u1"), ("i", "<i4", (3,))])
    face["n"], face["i"] = 3, t
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(vert.tobytes())
        f.write(face.tobytes())


class FeatureMesh:
    """Result of Mapper.get_feature_mesh(): surface vertices with their feature vectors; triangles on demand."""

    def __init__(self, vertices: torch.Tensor, vertex_features: torch.Tensor, mapper: "Mapper" = None, mapper_id: int = 0):
        self._v = vertices
        self._f = vertex_features
        self._mapper, self._id = mapper, mapper_id
        self._t = None

    def vertices(self) -> torch.Tensor:
        """[V,3] float32 on the GPU, world frame."""
        return self._v

    def vertex_features(self) -> torch.Tensor:
        """[V,C] float16 on the GPU (zeros where no feature was observed)."""
        return self._f

    def vertex_appearances(self) -> torch.Tensor:
        return self._f

    def triangles(self) -> torch.Tensor:
        """[T,3] int32 indices into vertices() (marching-cubes connectivity, normals towards free space); extracted on
        first use -- the policy consumes vertices only (paper/utils/utils.py:84-92 is the consumer)."""
        if self._t is None:
            if self._mapper is None:
                raise RuntimeError("this FeatureMesh is not attached to a Mapper")
            tris, _, V = self._mapper._mesh_topology(self._id, want_colors=False)
            if V != self._v.shape[0]:
                raise RuntimeError("the map changed since this mesh was extracted; call get_feature_mesh() again")
            self._t = tris
        return self._t


class ColorMesh:
    """Result of Mapper.get_color_mesh(): vertices, per-vertex colours, triangles (visualization/visualizer.py:656-672)."""

    def __init__(self, vertices: torch.Tensor, triangles: torch.Tensor, colors_u8: torch.Tensor):
        self._v, self._t, self._c = vertices, triangles, colors_u8

    def vertices(self) -> torch.Tensor:
        return self._v

    def triangles(self) -> torch.Tensor:
        return self._t

    def vertex_colors(self) -> torch.Tensor:
        """[V,3] float32 in [0,1] (Open3D convention)."""
        return self._c.to(torch.float32) / 255.0

    def vertex_colors_u8(self) -> torch.Tensor:
        return self._c

    def vertex_appearances(self) -> torch.Tensor:
        return self.vertex_colors()

    def save(self, path: str) -> None:
        """Write a PLY file (the reference calls mesh.save(path) with a .ply name, visualizer.py:667-672)."""
        _write_ply(path, self._v, self._t, self._c)

    def to_open3d(self):
        import open3d as o3d  # not a dependency of this package

        mesh = o3d.geometry.TriangleMesh()
        mesh.vertices = o3d.utility.Vector3dVector(self._v.cpu().numpy().astype(np.float64))
        mesh.triangles = o3d.utility.Vector3iVector(self._t.cpu().numpy())
        mesh.vertex_colors = o3d.utility.Vector3dVector(self.vertex_colors().cpu().numpy().astype(np.float64))
        return mesh


class _LayerView:
    def __init__(self, mapper: "Mapper", mapper_id: int, layer: int):
        self._m, self._id, self._layer = mapper, mapper_id, layer

    def voxel_size(self) -> float:
        return self._m._voxel_sizes[self._id]

    def block_size(self) -> float:
        return 8.0 * self.voxel_size()

    def num_allocated_blocks(self) -> int:
        return self._m._num_blocks(self._id, self._layer)

    def get_all_block_indices(self) -> torch.Tensor:
        """[n,3] int32, allocation order."""
        return self._m._block_indices(self._id, self._layer)


class TsdfLayerView(_LayerView):
    def get_all_blocks(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """([n,8,8,8,2] float32 with [...,0]=distance, [...,1]=weight, [n,3] int32 indices)."""
        idx = self.get_all_block_indices()
        n = idx.shape[0]
        out = torch.empty((n, 8, 8, 8, 2), dtype=torch.float32, device=self._m.device)
        if n:
            L = _lib.lib()
            _lib.check(L.mmf_get_tsdf_blocks(self._m._h, self._id, _lib.dptr(out), n, self._m._stream()), "mmf_get_tsdf_blocks")
        return out, idx

    def get_block_at_index(self, index) -> Optional[torch.Tensor]:
        blocks, idx = self.get_all_blocks()
        key = torch.as_tensor(index, dtype=torch.int32, device=idx.device).view(1, 3)
        hit = torch.nonzero(torch.all(idx == key, dim=1))
        return blocks[hit[0, 0]] if hit.numel() else None

    def get_tsdfs_below_zero(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(tsdf_and_weight [N,2], voxel-centre points [N,3]) of observed voxels with distance < 0."""
        from .indexing import get_voxel_center_grids

        blocks, idx = self.get_all_blocks()
        centres = get_voxel_center_grids(idx, self.voxel_size(), device=blocks.device)
        sel = (blocks[..., 0] < 0) & (blocks[..., 1] > 0)
        return blocks[sel], centres[sel]


class FeatureLayerView(_LayerView):
    def get_all_blocks(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """([n,8,8,8,C+1] float16 with [...,-1]=weight, [n,3] int32 indices)."""
        feats, weights, idx = self.get_all_blocks_split()
        return torch.cat([feats, weights.unsqueeze(-1).to(torch.float16)], dim=-1), idx

    def get_all_blocks_split(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """([n,8,8,8,C] float16, [n,8,8,8] float32 weights, [n,3] int32 indices)."""
        idx = self.get_all_block_indices()
        n, Cc = idx.shape[0], self._m.feature_channels
        feats = torch.empty((n, 8, 8, 8, Cc), dtype=torch.float16, device=self._m.device)
        weights = torch.empty((n, 8, 8, 8), dtype=torch.float32, device=self._m.device)
        if n:
            L = _lib.lib()
            _lib.check(L.mmf_get_feature_blocks(self._m._h, self._id, _lib.dptr(feats), _lib.dptr(weights), n, self._m._stream()),
                       "mmf_get_feature_blocks")
        return feats, weights, idx

    def get_block_at_index(self, index) -> Optional[torch.Tensor]:
        blocks, idx = self.get_all_blocks()
        key = torch.as_tensor(index, dtype=torch.int32, device=idx.device).view(1, 3)
        hit = torch.nonzero(torch.all(idx == key, dim=1))
        return blocks[hit[0, 0]] if hit.numel() else None


class ColorLayerView(_LayerView):
    def get_all_blocks_split(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """([n,8,8,8,3] uint8, [n,8,8,8] float32 weights, [n,3] int32 indices)."""
        idx = self.get_all_block_indices()
        n = idx.shape[0]
        rgb = torch.empty((n, 8, 8, 8, 3), dtype=torch.uint8, device=self._m.device)
        weights = torch.empty((n, 8, 8, 8), dtype=torch.float32, device=self._m.device)
        if n:
            L = _lib.lib()
            _lib.check(L.mmf_get_color_blocks(self._m._h, self._id, _lib.dptr(rgb), _lib.dptr(weights), n, self._m._stream()),
                       "mmf_get_color_blocks")
        return rgb, weights, idx


class Mapper:
    def __init__(
        self,
        voxel_sizes_m: Union[float, Sequence[float]],
        integrator_types: Optional[List[ProjectiveIntegratorType]] = None,
        mapper_parameters: Optional[MapperParams] = None,
        device: Union[None, int, str, torch.device] = None,
        feature_channels: Optional[int] = None,
    ):
        if isinstance(voxel_sizes_m, (int, float)):
            voxel_sizes_m = [float(voxel_sizes_m)]
        self._voxel_sizes = [float(v) for v in voxel_sizes_m]
        n = len(self._voxel_sizes)
        if integrator_types is None:
            integrator_types = [ProjectiveIntegratorType.TSDF] * n
        if len(integrator_types) != n:
            raise ValueError("integrator_types must have one entry per voxel size")
        for t in integrator_types:
            if t != ProjectiveIntegratorType.TSDF:
                raise NotImplementedError("only ProjectiveIntegratorType.TSDF is supported")
        self._params = mapper_parameters if mapper_parameters is not None else MapperParams()
        self.feature_channels = int(feature_channels or _constants.constants.feature_array_num_elements())

        L = _lib.lib()  # raises if the HIP extension has not been built
        _lib.require_gpu()
        if device is None:
            dev_index = torch.cuda.current_device()
        else:
            d = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            dev_index = d.index if d.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        arr = (_lib.MmfParams * n)(*[self._params.to_c(v, self.feature_channels) for v in self._voxel_sizes])
        h = C.c_void_p()
        _lib.check(L.mmf_mapper_create(n, arr, dev_index, C.byref(h)), "mmf_mapper_create")
        self._h = h
        self._n = n
        self._fma_contraction = bool(arr[0].fma_contraction)
        self._mesh_V = {}
        self._held_rows = {}  # mapper_id -> (tensors of a frame whose appearance tail is deferred, their version counters)
        self._deferred_mode = {}  # mapper_id -> set_deferred_feature_rows state

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mmf_mapper_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def fma_contraction(self) -> bool:
        """The spec switch ``projective_integrator_fma_contraction`` THIS mapper was built with (the switch was read at construction:
        later edits of the parameter bag do not reach the native mapper).  Ops beside the mapper that must blend with the same
        arithmetic -- the materialised feature up-sampling, image_processing/feature_resize.py -- take it from here, not from the
        process environment."""
        return self._fma_contraction

    # -- helpers ------------------------------------------------------------------------
    def _stream(self) -> C.c_void_p:
        return _lib.stream_ptr(self.device)

    def _check_id(self, mapper_id: int) -> int:
        mapper_id = int(mapper_id)
        if not 0 <= mapper_id < self._n:
            raise ValueError(f"mapper_id {mapper_id} out of range [0, {self._n})")
        return mapper_id

    def _num_blocks(self, mapper_id: int, layer: int) -> int:
        n = C.c_int(0)
        _lib.check(_lib.lib().mmf_num_allocated_blocks(self._h, mapper_id, layer, self._stream(), C.byref(n)), "mmf_num_allocated_blocks")
        return n.value

    def _block_indices(self, mapper_id: int, layer: int) -> torch.Tensor:
        n = self._num_blocks(mapper_id, layer)
        out = torch.empty((n, 3), dtype=torch.int32, device=self.device)
        if n:
            _lib.check(_lib.lib().mmf_get_block_indices(self._h, mapper_id, layer, _lib.dptr(out), n, self._stream()), "mmf_get_block_indices")
        return out

    # -- nvblox_torch API ---------------------------------------------------------------
    def num_mappers(self) -> int:
        return self._n

    def add_depth_frame(self, depth_frame: torch.Tensor, t_w_c, intrinsics, mask_frame: Optional[torch.Tensor] = None,
                        mapper_id: int = 0) -> None:
        mapper_id = self._check_id(mapper_id)
        depth = _check_dev(depth_frame, "depth_frame", torch.float32, 2)
        H, W = depth.shape
        mask = _mask_u8(mask_frame, (H, W))
        T = _host_f32(t_w_c, (4, 4))
        K = _host_f32(intrinsics, (3, 3))
        _lib.check(_lib.lib().mmf_add_depth_frame(self._h, mapper_id, _lib.dptr(depth), _lib.dptr(mask), H, W, T.ctypes.data, K.ctypes.data,
                                                  self._stream()), "mmf_add_depth_frame")

    def add_color_frame(self, color_frame: torch.Tensor, t_w_c, intrinsics, mask_frame: Optional[torch.Tensor] = None,
                        mapper_id: int = 0) -> None:
        mapper_id = self._check_id(mapper_id)
        rgb = _check_dev(color_frame, "color_frame", torch.uint8, 3)
        H, W, ch = rgb.shape
        if ch != 3:
            raise ValueError("color_frame must be [H,W,3]")
        mask = _mask_u8(mask_frame, (H, W))
        T = _host_f32(t_w_c, (4, 4))
        K = _host_f32(intrinsics, (3, 3))
        _lib.check(_lib.lib().mmf_add_color_frame(self._h, mapper_id, _lib.dptr(rgb), _lib.dptr(mask), H, W, T.ctypes.data, K.ctypes.data,
                                                  self._stream()), "mmf_add_color_frame")

    def add_feature_frame(self, feature_frame: torch.Tensor, t_w_c, intrinsics, mask_frame: Optional[torch.Tensor] = None,
                          mapper_id: int = 0) -> None:
        mapper_id = self._check_id(mapper_id)
        feat = _check_dev(feature_frame, "feature_frame", torch.float16, 3)
        Hf, Wf, ch = feat.shape
        if ch != self.feature_channels:
            raise ValueError(
                f"feature_frame has {ch} channels but the mapper stores {self.feature_channels} "
                "(constants.feature_array_num_elements())")
        mask = _mask_u8(mask_frame, (Hf, Wf))
        T = _host_f32(t_w_c, (4, 4))
        K = _host_f32(intrinsics, (3, 3))
        _lib.check(_lib.lib().mmf_add_feature_frame(self._h, mapper_id, _lib.dptr(feat), _lib.dptr(mask), Hf, Wf, ch, T.ctypes.data,
                                                    K.ctypes.data, self._stream()), "mmf_add_feature_frame")

    def add_feature_frame_lowres(self, lowres_features: torch.Tensor, feature_size, t_w_c, intrinsics,
                                 mask_frame: Optional[torch.Tensor] = None, mapper_id: int = 0) -> None:
        """Extension: ``add_feature_frame`` fed with the backbone's low-res map [h,w,Cin] float32
        (channels last) instead of the up-sampled, padded f16 image.  The kernel evaluates
        ``f16(pad(bilinear(lowres -> feature_size)))`` at each tap itself (feature_extraction.py:188-191,198-210), so the
        result is bit-identical to ``upsample_features`` + ``add_feature_frame`` while the [Hf,Wf,C_pad] image
        (403 MB at 512x512x768) is never written or read.  ``intrinsics`` are those of the (virtual) feature image."""
        mapper_id = self._check_id(mapper_id)
        low = _check_dev(lowres_features, "lowres_features", torch.float32, 3)
        lh, lw, cin = low.shape
        Hf, Wf = int(feature_size[0]), int(feature_size[1])
        mask = _mask_u8(mask_frame, (Hf, Wf))
        T = _host_f32(t_w_c, (4, 4))
        K = _host_f32(intrinsics, (3, 3))
        _lib.check(_lib.lib().mmf_add_feature_frame_lowres(self._h, mapper_id, _lib.dptr(low), lh, lw, cin, _lib.dptr(mask), Hf, Wf,
                                                           T.ctypes.data, K.ctypes.data, self._stream()),
                   "mmf_add_feature_frame_lowres")

    def _frame_desc(self, depth_frame, color_frame, feature_frame, lowres_features, input_mask, t_w_c, intrinsics,
                    min_depth_m, input_mask_erosion_iterations, valid_depth_mask_erosion_iterations, border_percent, invert_input_mask):
        """Fill an ``mmf_frame``; returns (descriptor, objects to keep alive until the call returned, depth mask, feature mask)."""
        depth = _check_dev(depth_frame, "depth_frame", torch.float32, 2)
        rgb = _check_dev(color_frame, "color_frame", torch.uint8, 3)
        H, W = depth.shape
        if tuple(rgb.shape) != (H, W, 3):
            raise ValueError("color_frame must be [H,W,3] with the depth frame's H,W")
        f = _lib.MmfFrame()
        f.struct_size = C.sizeof(_lib.MmfFrame)
        if lowres_features is not None:
            low = _check_dev(lowres_features, "lowres_features", torch.float32, 3)
            f.lowres_features = low.data_ptr()
            f.lowres_h, f.lowres_w, f.lowres_channels = low.shape
            Hf, Wf, ch = H, W, self.feature_channels
            keep = low
        else:
            feat = _check_dev(feature_frame, "feature_frame", torch.float16, 3)
            Hf, Wf, ch = feat.shape
            if ch != self.feature_channels:
                raise ValueError(f"feature_frame has {ch} channels but the mapper stores {self.feature_channels}")
            f.features_f16 = feat.data_ptr()
            keep = feat
        mask = _mask_u8(input_mask, (H, W))
        T = _host_f32(t_w_c, (4, 4))
        K = _host_f32(intrinsics, (3, 3))
        dm = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        fm = torch.empty((Hf, Wf), dtype=torch.uint8, device=self.device)
        f.depth, f.rgb, f.input_mask = depth.data_ptr(), rgb.data_ptr(), mask.data_ptr()
        f.invert_input_mask = 1 if invert_input_mask else 0
        f.H, f.W, f.Hf, f.Wf, f.feature_channels = H, W, Hf, Wf, ch
        f.T_W_C, f.K = T.ctypes.data, K.ctypes.data
        f.min_depth_m = float(min_depth_m)
        f.input_mask_erosion_iterations = int(input_mask_erosion_iterations)
        f.valid_depth_mask_erosion_iterations = int(valid_depth_mask_erosion_iterations)
        f.border_percent = int(border_percent)
        f.depth_mask_out, f.feature_mask_out = dm.data_ptr(), fm.data_ptr()
        return f, (keep, depth, rgb, mask, T, K), dm, fm

    def _integrate_frame_desc(self, depth_frame, color_frame, feature_frame, lowres_features, input_mask, t_w_c, intrinsics,
                              min_depth_m, input_mask_erosion_iterations, valid_depth_mask_erosion_iterations, border_percent,
                              mapper_id, invert_input_mask):
        mapper_id = self._check_id(mapper_id)
        f, keep, dm, fm = self._frame_desc(depth_frame, color_frame, feature_frame, lowres_features, input_mask, t_w_c, intrinsics,
                                           min_depth_m, input_mask_erosion_iterations, valid_depth_mask_erosion_iterations,
                                           border_percent, invert_input_mask)
        previous = self._check_held_rows(mapper_id)
        try:
            _lib.check(_lib.lib().mmf_integrate_frame_desc(self._h, mapper_id, C.byref(f), self._stream()), "mmf_integrate_frame_desc")
        except Exception:
            # the call failed (bad argument, HIP error) but the PREVIOUS frame's tail may still be pending with pointers into
            # `previous`: keep those tensors alive until something runs it
            if previous is not None and _lib.lib().mmf_deferred_feature_rows_pending(self._h, mapper_id) == 1:
                self._held_rows[mapper_id] = previous
            raise
        del previous
        if _lib.lib().mmf_deferred_feature_rows_pending(self._h, mapper_id) == 1:
            # the native side still reads the feature image (or low-res map), the colour image and the two masks it has just written: keep them
            # allocated, and remember their version counters
            held = (keep[0], keep[2], dm, fm)
            self._held_rows[mapper_id] = (held, tuple(t._version for t in held))
        del keep
        return dm, fm

    def _check_held_rows(self, mapper_id: int):
        """Returns the tensors of the previous frame that were kept alive for the native side: the caller holds them until its own
        native call has returned (the kernels that read them are enqueued by that call; nothing allocated in between may land in
        their memory)."""
        held = self._held_rows.pop(mapper_id, None)
        if held is None or _lib.lib().mmf_deferred_feature_rows_pending(self._h, mapper_id) != 1:
            return held
        if tuple(t._version for t in held[0]) != held[1]:
            raise RuntimeError(
                "an image of the previous frame (features, colour or one of the returned masks) was modified in place while its "
                "appearance update was deferred (set_deferred_feature_rows): the map would differ from the undeferred sequence.  "
                "Hand every frame its own tensors, or call flush() before reusing the buffers.")
        return held

    def set_deferred_feature_rows(self, on: bool = True, mapper_id: int = -1) -> None:
        """Extension (``mmf_set_deferred_feature_rows``): consecutive ``integrate_frame`` calls are software-pipelined -- a
        frame's last launch (the feature-row update of the voxels that passed the gate) rides in the NEXT frame's sphere-trace
        launch, and its colour update + feature gating in the next frame's first launch; anything else that touches the mapper
        runs them first, so every result is bit-identical to the undeferred sequence.  While on, the feature image, the colour
        image and the two returned masks of a frame must not be modified in place before the next call on the mapper (this
        object keeps the tensors alive and checks their version counters -- which see torch's in-place operations only: a write
        through a raw pointer, e.g. by another native library or a DLPack consumer, goes unnoticed).  Off by default."""
        previous = []
        if not on:  # switching it off runs what is pending: the same in-place check as before any other consumer of the held images
            previous = [self._check_held_rows(i) for i in (list(self._held_rows) if int(mapper_id) < 0 else [int(mapper_id)])]
        _lib.check(_lib.lib().mmf_set_deferred_feature_rows(self._h, int(mapper_id), 1 if on else 0), "mmf_set_deferred_feature_rows")
        for i in (range(self._n) if int(mapper_id) < 0 else [int(mapper_id)]):
            self._deferred_mode[i] = bool(on)
        del previous

    def integrate_frame_sequence(self, frames, mapper_id: int = 0, decay_before_each: bool = True) -> list:
        """Extension: a recorded stream into one mapper -- ``frames`` yields dicts with the arguments of ``integrate_frame``
        (``depth_frame``, ``color_frame``, ``feature_frame`` or ``lowres_features``, ``input_mask``, ``t_w_c``, ``intrinsics``,
        ``min_depth_m``, ``input_mask_erosion_iterations``, ``valid_depth_mask_erosion_iterations``, ``border_percent``, optionally
        ``invert_input_mask``), each preceded by ``decay()`` as in the reference's control step.  Consecutive frames are
        software-pipelined (``set_deferred_feature_rows``) for the duration of the call and the last one is completed before it
        returns; the mode the mapper was in is restored.  Returns [(depth_mask, feature_mask), ...]."""
        mapper_id = self._check_id(mapper_id)
        was_on = bool(self._deferred_mode.get(mapper_id, False))
        self.set_deferred_feature_rows(True, mapper_id)
        out = []
        try:
            for fr in frames:
                if decay_before_each:
                    self.decay(mapper_id)
                out.append(self._integrate_frame_desc(
                    fr["depth_frame"], fr["color_frame"], None if fr.get("lowres_features") is not None else fr["feature_frame"],
                    fr.get("lowres_features"), fr["input_mask"], fr["t_w_c"], fr["intrinsics"], fr["min_depth_m"],
                    fr["input_mask_erosion_iterations"], fr["valid_depth_mask_erosion_iterations"], fr["border_percent"], mapper_id,
                    bool(fr.get("invert_input_mask", False))))
        finally:
            self.flush(mapper_id)
            if not was_on:
                self.set_deferred_feature_rows(False, mapper_id)
        return out

    def flush(self, mapper_id: int = -1) -> None:
        """Enqueue whatever is pending on the mapper (a deferred row update, a lazy ``decay()``)."""
        previous = [self._check_held_rows(i) for i in ([int(mapper_id)] if int(mapper_id) >= 0 else list(self._held_rows))]
        _lib.check(_lib.lib().mmf_flush(self._h, int(mapper_id), self._stream()), "mmf_flush")
        del previous

    def integrate_frame_multi(self, depth_frame: torch.Tensor, color_frame: torch.Tensor, feature_frame: torch.Tensor, t_w_c, intrinsics,
                              min_depth_m: float, border_percent: int, jobs, lowres_features: Optional[torch.Tensor] = None):
        """One camera frame into SEVERAL mappers with one native call (``mmf_integrate_frame_multi``): what the reference's
        ``nvblox_integrate(include_dynamic=True)`` does with two ``integrate_frame`` calls (nvblox_mapping_helpers.py:128-156).
        ``jobs``: one dict per mapper -- ``mapper_id``, ``input_mask`` [H,W] bool/u8, ``input_mask_erosion_iterations``,
        ``valid_depth_mask_erosion_iterations`` and optionally ``invert_input_mask``.  Pairs of jobs run as roles of the same five
        launches (bit-identical to the calls in sequence).  Returns [(depth_mask, feature_mask), ...] in job order."""
        n = len(jobs)
        descs = (_lib.MmfFrame * n)()
        ids = (C.c_int * n)()
        # the jobs share the camera frame: its checks, host copies of pose / intrinsics and descriptor fields are done once
        # (first job) and copied; per job only the mask, the erosions and the two output masks differ
        first = jobs[0]
        base, keep, dm0, fm0 = self._frame_desc(depth_frame, color_frame, None if lowres_features is not None else feature_frame,
                                                lowres_features, first["input_mask"], t_w_c, intrinsics, min_depth_m,
                                                first["input_mask_erosion_iterations"], first["valid_depth_mask_erosion_iterations"],
                                                border_percent, bool(first.get("invert_input_mask", False)))
        H, W, Hf, Wf = base.H, base.W, base.Hf, base.Wf
        out = [(dm0, fm0)]
        keep = [keep]
        if n > 1:
            dms = torch.empty((n - 1, H, W), dtype=torch.uint8, device=self.device)
            fms = torch.empty((n - 1, Hf, Wf), dtype=torch.uint8, device=self.device)
        for i, job in enumerate(jobs):
            ids[i] = self._check_id(job["mapper_id"])
            if i == 0:
                descs[0] = base
                continue
            C.memmove(C.byref(descs[i]), C.byref(base), C.sizeof(_lib.MmfFrame))
            f = descs[i]
            mask = _mask_u8(job["input_mask"], (H, W))
            f.input_mask = mask.data_ptr()
            f.invert_input_mask = 1 if job.get("invert_input_mask", False) else 0
            f.input_mask_erosion_iterations = int(job["input_mask_erosion_iterations"])
            f.valid_depth_mask_erosion_iterations = int(job["valid_depth_mask_erosion_iterations"])
            dm, fm = dms[i - 1], fms[i - 1]
            f.depth_mask_out, f.feature_mask_out = dm.data_ptr(), fm.data_ptr()
            keep.append(mask)
            out.append((dm, fm))
        previous = [self._check_held_rows(ids[i]) for i in range(n)]
        _lib.check(_lib.lib().mmf_integrate_frame_multi(self._h, n, ids, descs, self._stream()), "mmf_integrate_frame_multi")
        for i in range(n):  # mappers in deferred mode still read the frame's images and their own two masks
            if _lib.lib().mmf_deferred_feature_rows_pending(self._h, ids[i]) == 1:
                held = (keep[0][0], keep[0][2], out[i][0], out[i][1])
                self._held_rows[ids[i]] = (held, tuple(t._version for t in held))
        del keep, previous
        return out

    def integrate_frame_lowres(self, depth_frame: torch.Tensor, color_frame: torch.Tensor, lowres_features: torch.Tensor,
                               input_mask: torch.Tensor, t_w_c, intrinsics, min_depth_m: float,
                               input_mask_erosion_iterations: int, valid_depth_mask_erosion_iterations: int,
                               border_percent: int, mapper_id: int = 0, invert_input_mask: bool = False):
        """``integrate_frame`` with the low-res feature source of ``add_feature_frame_lowres`` (virtual feature image at
        the depth resolution).  Returns (depth_mask uint8 [H,W], feature_mask uint8 [H,W])."""
        return self._integrate_frame_desc(depth_frame, color_frame, None, lowres_features, input_mask, t_w_c, intrinsics,
                                          min_depth_m, input_mask_erosion_iterations, valid_depth_mask_erosion_iterations,
                                          border_percent, mapper_id, invert_input_mask)

    def integrate_frame(self, depth_frame: torch.Tensor, color_frame: torch.Tensor, feature_frame: torch.Tensor,
                        input_mask: torch.Tensor, t_w_c, intrinsics, min_depth_m: float, input_mask_erosion_iterations: int,
                        valid_depth_mask_erosion_iterations: int, border_percent: int, mapper_id: int = 0,
                        invert_input_mask: bool = False):
        """Extension: the reference's ``integrate_frame`` (mapping/helpers/nvblox_mapping_helpers.py:162-273) as one
        native call -- mask algebra + add_depth_frame + add_color_frame + add_feature_frame with identical results in
        five fused launches (a preceding ``decay()`` is folded into the first two).  Needs the feature image at the depth
        resolution.  ``invert_input_mask``: use ``~input_mask`` (the reference's static mask from the dynamic mask,
        :116-117) without materialising it.  Returns (depth_mask uint8 [H,W], feature_mask uint8 [Hf,Wf])."""
        return self._integrate_frame_desc(depth_frame, color_frame, feature_frame, None, input_mask, t_w_c, intrinsics,
                                          min_depth_m, input_mask_erosion_iterations, valid_depth_mask_erosion_iterations,
                                          border_percent, mapper_id, invert_input_mask)

    def save_map(self, path: str, mapper_id: int = 0) -> None:
        """``mapper.save_map(path, mapper_id)`` (nvblox_to_disk_helpers.py:88-93): checkpoint of one mapper -- every
        block of the TSDF, colour and feature layer in allocation order.  Own container, see io/map_file.py."""
        from ..io.map_file import write_map_file

        mapper_id = self._check_id(mapper_id)
        tsdf, tidx = self.tsdf_layer_view(mapper_id).get_all_blocks()
        rgb, cw, cidx = self.color_layer_view(mapper_id).get_all_blocks_split()
        feat, fw, fidx = self.feature_layer_view(mapper_id).get_all_blocks_split()
        arrays = {"tsdf_idx": tidx, "tsdf": tsdf, "color_idx": cidx, "color_rgb": rgb, "color_w": cw,
                  "feature_idx": fidx, "feature": feat, "feature_w": fw}
        meta = {"voxel_size_m": self._voxel_sizes[mapper_id], "feature_channels": self.feature_channels}
        write_map_file(path, meta, {k: v.cpu().numpy() for k, v in arrays.items()})

    def load_from_file(self, path: str, mapper_id: int = 0) -> None:
        """Replace the content of mapper `mapper_id` by a map saved with ``save_map``; fusion then continues exactly as it
        would have on the mapper that was saved (same block order, same values).  The mapper must have the voxel size and
        feature width of the saved one, and workspace bounds that contain it."""
        from ..io.map_file import read_map_file

        mapper_id = self._check_id(mapper_id)
        meta, a = read_map_file(path)
        if abs(float(meta["voxel_size_m"]) - self._voxel_sizes[mapper_id]) > 1e-9:
            raise ValueError(f"map was saved at voxel size {meta['voxel_size_m']}, this mapper uses {self._voxel_sizes[mapper_id]}")
        if int(meta["feature_channels"]) != self.feature_channels:
            raise ValueError(f"map was saved with {meta['feature_channels']} feature channels, this mapper stores {self.feature_channels}")
        L = _lib.lib()

        def dev(x, dtype):
            return torch.from_numpy(np.ascontiguousarray(x)).to(self.device, dtype)

        jobs = [(_lib.MMF_LAYER_TSDF, "tsdf_idx", dev(a["tsdf"], torch.float32), None),
                (_lib.MMF_LAYER_COLOR, "color_idx", dev(a["color_rgb"], torch.uint8), dev(a["color_w"], torch.float32)),
                (_lib.MMF_LAYER_FEATURE, "feature_idx", dev(a["feature"], torch.float16), dev(a["feature_w"], torch.float32))]
        for layer, key, payload, weights in jobs:
            idx = dev(a[key], torch.int32)
            n = int(idx.shape[0])
            if n == 0 and layer != _lib.MMF_LAYER_TSDF and not self._num_blocks(mapper_id, layer):
                continue  # never-used appearance layer stays unallocated
            _lib.check(L.mmf_import_blocks(self._h, mapper_id, layer, _lib.dptr(idx), _lib.dptr(payload), _lib.dptr(weights), n,
                                           self._stream()), "mmf_import_blocks")
        self._mesh_V.pop(mapper_id, None)

    def decay(self, mapper_id: int = -1) -> None:
        _lib.check(_lib.lib().mmf_decay(self._h, int(mapper_id), self._stream()), "mmf_decay")

    def clear(self, mapper_id: int = -1) -> None:
        _lib.check(_lib.lib().mmf_clear(self._h, int(mapper_id), self._stream()), "mmf_clear")

    def update_feature_mesh(self, mapper_id: int = 0) -> int:
        mapper_id = self._check_id(mapper_id)
        n = C.c_int(0)
        _lib.check(_lib.lib().mmf_update_feature_mesh(self._h, mapper_id, self._stream(), C.byref(n)), "mmf_update_feature_mesh")
        self._mesh_V[mapper_id] = n.value
        return n.value

    def get_feature_mesh(self, mapper_id: int = 0) -> FeatureMesh:
        """Mesh of the last update_feature_mesh (re-extracted if the map changed since)."""
        mapper_id = self._check_id(mapper_id)
        L = _lib.lib()
        V = self._mesh_V.get(mapper_id)
        for attempt in range(2):
            if V is None:
                V = self.update_feature_mesh(mapper_id)
            verts = torch.empty((V, 3), dtype=torch.float32, device=self.device)
            feats = torch.empty((V, self.feature_channels), dtype=torch.float16, device=self.device)
            rc = L.mmf_get_feature_mesh(self._h, mapper_id, _lib.dptr(verts), _lib.dptr(feats), self._stream())
            if rc == 4 and attempt == 0:  # MMF_ERR_BAD_STATE: map changed since the last update
                V = None
                continue
            _lib.check(rc, "mmf_get_feature_mesh")
            break
        return FeatureMesh(verts, feats, self, mapper_id)

    def model_inputs_prepare(self, mapper_id: int, aabb_min, aabb_max, used_channels: int, remove_zero_features: bool) -> int:
        """Extension (mmf_model_inputs_prepare): the row filters of the reference's ``get_vertices_and_features``
        (nvblox_output_helpers.py:49-74: mesh update, strict AABB, pad-channel strip, all-zero rows) in one launch; returns
        the number of rows that pass, in the order of the reference's filtered mesh.  ``aabb_*``: 3 host floats each."""
        mapper_id = self._check_id(mapper_id)
        lo = (C.c_float * 3)(*[float(x) for x in aabb_min])
        hi = (C.c_float * 3)(*[float(x) for x in aabb_max])
        n = C.c_int(0)
        _lib.check(_lib.lib().mmf_model_inputs_prepare(self._h, mapper_id, lo, hi, int(used_channels), int(bool(remove_zero_features)),
                                                       self._stream(), C.byref(n)), "mmf_model_inputs_prepare")
        self._mi_used = int(used_channels)
        return n.value

    def model_inputs_gather(self, mapper_id: int, rows: Optional[torch.Tensor], n_take: int, n_out: int,
                            features_dtype: Optional[torch.dtype] = torch.float32):
        """Extension (mmf_model_inputs_gather): rows of the last ``model_inputs_prepare`` -> (vertices [n_out,3] f32,
        features [n_out,used] ``features_dtype`` (None: vertices only), valid [n_out] bool).  ``rows``: int64 ranks among the
        kept rows on the device (None = the first ``n_take``); rows n_take .. n_out-1 are zero padding."""
        mapper_id = self._check_id(mapper_id)
        if rows is not None:
            rows = _check_dev(rows, "rows", torch.int64, 1)
            assert rows.shape[0] >= n_take
        verts = torch.empty((n_out, 3), dtype=torch.float32, device=self.device)
        feats = None
        if features_dtype is not None:
            assert features_dtype in (torch.float32, torch.float16)
            feats = torch.empty((n_out, self._mi_used), dtype=features_dtype, device=self.device)
        valid = torch.empty((n_out,), dtype=torch.bool, device=self.device)
        _lib.check(_lib.lib().mmf_model_inputs_gather(self._h, mapper_id, _lib.dptr(rows), int(n_take), int(n_out), _lib.dptr(verts),
                                                      _lib.dptr(feats), int(features_dtype == torch.float32), _lib.dptr(valid),
                                                      self._stream()), "mmf_model_inputs_gather")
        return verts, feats, valid

    def _mesh_topology(self, mapper_id: int, want_colors: bool):
        """(triangles [T,3] int32, colours [V,3] uint8 or None, V) of the current map."""
        L = _lib.lib()
        nv, nt = C.c_int(0), C.c_int(0)
        _lib.check(L.mmf_update_mesh_topology(self._h, mapper_id, self._stream(), C.byref(nv), C.byref(nt)), "mmf_update_mesh_topology")
        self._mesh_V[mapper_id] = nv.value
        tris = torch.empty((nt.value, 3), dtype=torch.int32, device=self.device)
        cols = torch.empty((nv.value, 3), dtype=torch.uint8, device=self.device) if want_colors else None
        _lib.check(L.mmf_get_mesh_topology(self._h, mapper_id, _lib.dptr(tris), _lib.dptr(cols), self._stream()), "mmf_get_mesh_topology")
        return tris, cols, nv.value

    def update_color_mesh(self, mapper_id: int = 0) -> None:
        """Kept for call compatibility (visualizer.py:657): extraction happens in get_color_mesh, from the current map."""
        self._check_id(mapper_id)

    def get_color_mesh(self, mapper_id: int = 0) -> Optional[ColorMesh]:
        """Surface mesh with per-vertex colours of the colour layer, or None when the map has no surface yet."""
        mapper_id = self._check_id(mapper_id)
        tris, cols, V = self._mesh_topology(mapper_id, want_colors=True)
        if V == 0:
            return None
        mesh = self.get_feature_mesh(mapper_id)
        return ColorMesh(mesh.vertices(), tris, cols)

    def tsdf_layer_view(self, mapper_id: int = 0) -> TsdfLayerView:
        return TsdfLayerView(self, self._check_id(mapper_id), _lib.MMF_LAYER_TSDF)

    def feature_layer_view(self, mapper_id: int = 0) -> FeatureLayerView:
        return FeatureLayerView(self, self._check_id(mapper_id), _lib.MMF_LAYER_FEATURE)

    def color_layer_view(self, mapper_id: int = 0) -> ColorLayerView:
        return ColorLayerView(self, self._check_id(mapper_id), _lib.MMF_LAYER_COLOR)

    def query_layer(self, query_type: QueryType, query: torch.Tensor, mapper_id: int = 0) -> torch.Tensor:
        """TSDF: [N,2] (distance, weight).  FEATURE: [N,C+1] float32 (features, weight last)."""
        mapper_id = self._check_id(mapper_id)
        pts = _check_dev(query, "query", torch.float32, 2)
        n = pts.shape[0]
        if query_type == QueryType.TSDF:
            layer, width = _lib.MMF_LAYER_TSDF, 2
        elif query_type == QueryType.FEATURE:
            layer, width = _lib.MMF_LAYER_FEATURE, self.feature_channels + 1
        else:
            raise ValueError(f"unsupported query type {query_type}")
        out = torch.empty((n, width), dtype=torch.float32, device=self.device)
        if n:
            _lib.check(_lib.lib().mmf_query_layer(self._h, mapper_id, layer, _lib.dptr(pts), n, _lib.dptr(out), self._stream()), "mmf_query_layer")
        return out

    # -- diagnostics / measurement (extensions) -------------------------------------------
    def last_view_blocks(self, mapper_id: int = 0) -> torch.Tensor:
        """Block indices the last add_depth_frame found in view, sorted (x,y,z): [n,3] int32."""
        n = C.c_int(0)
        _lib.check(_lib.lib().mmf_last_view_block_count(self._h, mapper_id, self._stream(), C.byref(n)), "mmf_last_view_block_count")
        out = torch.empty((n.value, 3), dtype=torch.int32, device=self.device)
        if n.value:
            _lib.check(_lib.lib().mmf_get_last_view_blocks(self._h, mapper_id, _lib.dptr(out), n.value, self._stream()), "mmf_get_last_view_blocks")
        return out

    def synthetic_depth(self, mapper_id: int = 0) -> torch.Tensor:
        hs, ws = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().mmf_get_synthetic_depth_dims(self._h, mapper_id, C.byref(hs), C.byref(ws)), "mmf_get_synthetic_depth_dims")
        out = torch.empty((hs.value, ws.value), dtype=torch.float32, device=self.device)
        if out.numel():
            _lib.check(_lib.lib().mmf_get_synthetic_depth(self._h, mapper_id, _lib.dptr(out), self._stream()), "mmf_get_synthetic_depth")
        return out

    def render_synthetic_depth(self, height: int, width: int, t_w_c, intrinsics, mapper_id: int = 0) -> torch.Tensor:
        T = _host_f32(t_w_c, (4, 4))
        K = _host_f32(intrinsics, (3, 3))
        _lib.check(_lib.lib().mmf_render_synthetic_depth(self._h, mapper_id, int(height), int(width), T.ctypes.data, K.ctypes.data, self._stream()),
                   "mmf_render_synthetic_depth")
        return self.synthetic_depth(mapper_id)

    def stats(self, mapper_id: int = 0) -> dict:
        buf = (C.c_int64 * _lib.MMF_NUM_STATS)()
        _lib.check(_lib.lib().mmf_get_stats(self._h, mapper_id, self._stream(), buf), "mmf_get_stats")
        names = ["depth_frames", "tsdf_blocks_updated", "tsdf_blocks_allocated", "color_frames", "color_blocks_updated",
                 "feature_frames", "feature_blocks_updated", "feature_blocks_allocated", "feature_voxels_updated"]
        return dict(zip(names, [int(x) for x in buf]))

    def debug_alloc_recoveries(self, mapper_id: int = 0) -> int:
        """New TSDF blocks integrated by the sweeper of k_alloc_tsdf (their waiter abandoned its wait): 0 in normal operation."""
        out = C.c_int64(0)
        _lib.check(_lib.lib().mmf_debug_alloc_recoveries(self._h, mapper_id, self._stream(), C.byref(out)), "mmf_debug_alloc_recoveries")
        return int(out.value)

    def hash_state(self, mapper_id: int = 0, layer: int = _lib.MMF_LAYER_TSDF) -> dict:
        """Diagnostics of a layer's block index: hash table size (0 = dense table of a bounded workspace), tombstones, rebuilds, live blocks."""
        buf = (C.c_int64 * 8)()
        _lib.check(_lib.lib().mmf_debug_hash_state(self._h, mapper_id, int(layer), self._stream(), buf), "mmf_debug_hash_state")
        out = dict(zip(["table_entries", "tombstones", "rebuilds", "live_blocks"], [int(x) for x in buf[:4]]))
        out["view_grid"] = [int(buf[4]), int(buf[5]), int(buf[6])]
        out["lazy_decays"] = int(buf[7])
        return out

    def count_tombstones(self, mapper_id: int = 0, layer: int = _lib.MMF_LAYER_TSDF) -> int:
        """Diagnostics: tombstone entries found by a scan of the layer's hash table (== ``hash_state()["tombstones"]``)."""
        out = C.c_int64(0)
        _lib.check(_lib.lib().mmf_debug_count_tombstones(self._h, mapper_id, int(layer), self._stream(), C.byref(out)), "mmf_debug_count_tombstones")
        return int(out.value)

    def reset_stats(self, mapper_id: int = 0) -> None:
        _lib.check(_lib.lib().mmf_reset_stats(self._h, mapper_id, self._stream()), "mmf_reset_stats")

    def profile_enable(self, on: bool = True, kernels=None, stride: int = 1) -> None:
        """Time kernel classes with HIP events on the launch stream (`kernels`: names of _lib.KERNEL_IDS, default all);
        `stride` = time every stride-th launch only."""
        _lib.check(_lib.lib().mmf_profile_set_stride(self._h, int(stride)), "mmf_profile_set_stride")
        mask = 0
        if on:
            for k in (kernels if kernels is not None else _lib.KERNEL_IDS):
                mask |= 1 << _lib.KERNEL_IDS[k]
        _lib.check(_lib.lib().mmf_profile_enable(self._h, mask), "mmf_profile_enable")

    def profile_reset(self) -> None:
        _lib.check(_lib.lib().mmf_profile_reset(self._h), "mmf_profile_reset")

    def profile(self) -> dict:
        """{kernel class: (total_ms, launches)} measured with HIP events on the launch stream."""
        out = {}
        for name, kid in _lib.KERNEL_IDS.items():
            ms, n = C.c_double(0), C.c_int64(0)
            _lib.check(_lib.lib().mmf_profile_get(self._h, kid, C.byref(ms), C.byref(n)), "mmf_profile_get")
            out[name] = (ms.value, n.value)
        return out


def integrate_frames_batch(frames) -> list:
    """Extension (``mmf_integrate_frame_batch``): N independent frames -- each for a different mapper, of the same or of
    different ``Mapper`` objects on one device, each with its own camera, images and masks -- as roles of ONE set of five
    launches (up to eight frames per set).  What several replicas of the fusion path on one GPU call instead of one
    ``integrate_frame`` each (data generation over several demos, several environments per GPU): bit-identical maps, the chip
    filled instead of half idle.  A ``decay()`` pending on a mapper is folded in as in the single call.

    ``frames``: one dict per frame with the arguments of ``Mapper.integrate_frame`` -- ``mapper`` (the Mapper), ``mapper_id``,
    ``depth_frame``, ``color_frame``, ``feature_frame`` (or ``lowres_features``), ``input_mask``, ``t_w_c``, ``intrinsics``,
    ``min_depth_m``, ``input_mask_erosion_iterations``, ``valid_depth_mask_erosion_iterations``, ``border_percent`` and
    optionally ``invert_input_mask``.  Returns [(depth_mask, feature_mask), ...] in frame order."""
    n = len(frames)
    if n == 0:
        return []
    descs = (_lib.MmfFrame * n)()
    ids = (C.c_int * n)()
    handles = (C.c_void_p * n)()
    keep, out = [], []
    for i, fr in enumerate(frames):
        m = fr["mapper"]
        if m.device != frames[0]["mapper"].device:
            raise ValueError("integrate_frames_batch: all mappers must live on one device")
        f, k, dm, fm = m._frame_desc(fr["depth_frame"], fr["color_frame"], None if fr.get("lowres_features") is not None else fr["feature_frame"],
                                     fr.get("lowres_features"), fr["input_mask"], fr["t_w_c"], fr["intrinsics"], fr["min_depth_m"],
                                     fr["input_mask_erosion_iterations"], fr["valid_depth_mask_erosion_iterations"], fr["border_percent"],
                                     bool(fr.get("invert_input_mask", False)))
        descs[i] = f
        ids[i] = m._check_id(fr.get("mapper_id", 0))
        handles[i] = m._h
        keep.append(k)
        out.append((dm, fm))
        keep.append(m._check_held_rows(ids[i]))  # (the previous frame's tensors: until the call below has enqueued their readers)
    _lib.check(_lib.lib().mmf_integrate_frame_batch(n, handles, ids, descs, frames[0]["mapper"]._stream()), "mmf_integrate_frame_batch")
    for i, fr in enumerate(frames):  # mappers in deferred mode (set_deferred_feature_rows) still read this frame's images
        m = fr["mapper"]
        if _lib.lib().mmf_deferred_feature_rows_pending(m._h, ids[i]) == 1:
            k = keep[2 * i]
            held = (k[0], k[2], out[i][0], out[i][1])
            m._held_rows[ids[i]] = (held, tuple(t._version for t in held))
    del keep
    return out
