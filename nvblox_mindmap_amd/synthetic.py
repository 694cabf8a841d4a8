"""Deterministic synthetic RGB-D + feature stream.

Used by the parity tests, ``bench.py`` and ``__graft_entry__.smoke()`` so that the HIP path
and the CPU oracle always see the same bytes.  numpy only (float64 ray casting, cast to f32).

Scene: ground plane z=0, a sphere, an axis-aligned box -- all inside the DRILL_IN_BOX
workspace AABB of the reference (mindmap/mapping/nvblox_mapper_constants.py:62-70).
Camera: pinhole, ROS optical frame (x right, y down, z forward), on a circle around the
workspace centre looking at it.  Depth is ``distance_to_image_plane`` (z-depth, the
convention of mindmap/isaaclab_utils/isaaclab_camera_handler.py:155) of the ray through the
pixel centre (col+0.5, row+0.5).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

# Reference workspace of the Drill-in-Box task (nvblox_mapper_constants.py:65-66).
DRILL_IN_BOX_AABB_MIN = np.array([-0.37, -0.75, -0.13], dtype=np.float32)
DRILL_IN_BOX_AABB_MAX = np.array([0.95, 0.75, 0.65], dtype=np.float32)

SPHERE_C = np.array([0.3, 0.0, 0.25])
SPHERE_R = 0.25
BOX_MIN = np.array([-0.15, 0.30, 0.0])
BOX_MAX = np.array([0.15, 0.60, 0.20])


@dataclass
class StreamConfig:
    width: int = 640
    height: int = 480
    fx: float = 525.0
    fy: float = 525.0
    cx: float = 319.5
    cy: float = 239.5
    num_poses: int = 200
    radius_m: float = 1.2
    height_m: float = 0.6
    invalid_holes: bool = True
    # "pixels": isolated invalid pixels where (u*73856093 ^ v*19349663) % 97 == 0;
    # "patches": 16x16 invalid patches on ~1/61 of the patch grid.  The reference erodes the valid-depth mask
    # by 20 pixels before feature fusion (nvblox_mapper_constants.py:70): 1 %-density pixel holes would
    # erase the whole feature mask, so streams that run the reference's mask algebra use "patches".
    hole_mode: str = "pixels"

    def intrinsics(self) -> np.ndarray:
        return np.array([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]], dtype=np.float32)


def look_at_pose(pos: np.ndarray, target: np.ndarray) -> np.ndarray:
    """4x4 float64 T_W_C, ROS optical frame, world up = +z."""
    fwd = target - pos
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    T = np.eye(4)
    T[:3, 0] = right
    T[:3, 1] = down
    T[:3, 2] = fwd
    T[:3, 3] = pos
    return T


def camera_pose(cfg: StreamConfig, index: int) -> np.ndarray:
    """T_W_C (float32 4x4) of pose ``index`` on the orbit (built in float64, then cast)."""
    centre = 0.5 * (DRILL_IN_BOX_AABB_MIN.astype(np.float64) + DRILL_IN_BOX_AABB_MAX.astype(np.float64))
    theta = 2.0 * np.pi * (index % cfg.num_poses) / cfg.num_poses
    pos = np.array([centre[0] + cfg.radius_m * np.cos(theta), centre[1] + cfg.radius_m * np.sin(theta), cfg.height_m])
    return look_at_pose(pos, centre).astype(np.float32)


def _ray_scene(o: np.ndarray, d: np.ndarray) -> np.ndarray:
    """Smallest positive ray parameter t (d not normalised) hitting the scene; inf if none."""
    t_best = np.full(d.shape[:-1], np.inf)
    # plane z = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -o[2] / d[..., 2]
    t = np.where((t > 1e-9) & np.isfinite(t), t, np.inf)
    t_best = np.minimum(t_best, t)
    # sphere
    oc = o - SPHERE_C
    a = np.sum(d * d, axis=-1)
    b = 2.0 * np.sum(d * oc, axis=-1)
    c = float(np.dot(oc, oc) - SPHERE_R**2)
    disc = b * b - 4 * a * c
    sq = np.sqrt(np.maximum(disc, 0.0))
    t = (-b - sq) / (2 * a)
    t = np.where((disc >= 0) & (t > 1e-9), t, np.inf)
    t_best = np.minimum(t_best, t)
    # box (slab test)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0 = (BOX_MIN - o) * inv
        t1 = (BOX_MAX - o) * inv
    tmin = np.max(np.minimum(t0, t1), axis=-1)
    tmax = np.min(np.maximum(t0, t1), axis=-1)
    t = np.where((tmax >= tmin) & (tmin > 1e-9), tmin, np.inf)
    t_best = np.minimum(t_best, t)
    return t_best


def render_depth(cfg: StreamConfig, T_W_C: np.ndarray) -> np.ndarray:
    """[H, W] float32 z-depth in metres; 0 where invalid."""
    T = T_W_C.astype(np.float64)
    cols = np.arange(cfg.width) + 0.5
    rows = np.arange(cfg.height) + 0.5
    u, v = np.meshgrid(cols, rows)
    dc = np.stack([(u - cfg.cx) / cfg.fx, (v - cfg.cy) / cfg.fy, np.ones_like(u)], axis=-1)
    dw = dc @ T[:3, :3].T
    t = _ray_scene(T[:3, 3], dw)  # dc.z == 1  =>  t is the z-depth
    depth = np.where(np.isfinite(t), t, 0.0).astype(np.float32)
    if cfg.invalid_holes:
        uu, vv = np.meshgrid(np.arange(cfg.width, dtype=np.int64), np.arange(cfg.height, dtype=np.int64))
        if cfg.hole_mode == "patches":
            holes = (((uu // 16) * 73856093) ^ ((vv // 16) * 19349663)) % 61 == 0
        else:
            holes = ((uu * 73856093) ^ (vv * 19349663)) % 97 == 0
        depth[holes] = 0.0
    return depth


def render_rgb(cfg: StreamConfig, index: int) -> np.ndarray:
    """[H, W, 3] uint8 smooth colour pattern, different per frame."""
    uu, vv = np.meshgrid(np.arange(cfg.width), np.arange(cfg.height))
    r = (uu * 255 // max(cfg.width - 1, 1)).astype(np.uint8)
    g = (vv * 255 // max(cfg.height - 1, 1)).astype(np.uint8)
    b = ((uu + vv + 7 * index) % 256).astype(np.uint8)
    return np.ascontiguousarray(np.stack([r, g, b], axis=-1))


def bilinear_resize_hwc(lowres: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Bilinear, align_corners=False (the F.interpolate call of
    mindmap/image_processing/feature_extraction.py:126-128), float32, HWC."""
    in_h, in_w, _ = lowres.shape

    def src(n_out, n_in):
        x = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) - np.float32(0.5)
        x = np.maximum(x, np.float32(0.0))
        i0 = np.minimum(np.floor(x).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        w1 = (x - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, w1

    y0, y1, wy = src(out_h, in_h)
    x0, x1, wx = src(out_w, in_w)
    lr = lowres.astype(np.float32)
    top = lr[y0][:, x0] * (1 - wx)[None, :, None] + lr[y0][:, x1] * wx[None, :, None]
    bot = lr[y1][:, x0] * (1 - wx)[None, :, None] + lr[y1][:, x1] * wx[None, :, None]
    return (top * (1 - wy)[:, None, None] + bot * wy[:, None, None]).astype(np.float32)


def render_features(cfg: StreamConfig, index: int, channels: int, lowres: int = 16,
                    out_hw: Optional[Tuple[int, int]] = None) -> np.ndarray:
    """[Hf, Wf, C] float16, HWC contiguous: seeded low-res map -> bilinear -> f16 (mimics A7)."""
    rng = np.random.Generator(np.random.PCG64(1000003 * (index + 1)))
    low = rng.standard_normal((lowres, lowres, channels), dtype=np.float32)
    h, w = out_hw if out_hw is not None else (cfg.height, cfg.width)
    return np.ascontiguousarray(bilinear_resize_hwc(low, h, w).astype(np.float16))


def frame(cfg: StreamConfig, index: int, channels: int = 0):
    """Returns dict(depth, T_W_C, K, rgb[, features])."""
    T = camera_pose(cfg, index)
    out = {
        "depth": render_depth(cfg, T),
        "T_W_C": T,
        "K": cfg.intrinsics(),
        "rgb": render_rgb(cfg, index),
    }
    if channels > 0:
        out["features"] = render_features(cfg, index, channels)
    return out
