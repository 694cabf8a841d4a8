"""What every leg of bench.py shares: the synthetic stream, one step of the hot path, the byte model of a frame, the launch tables
and the replayed counter summaries (profiles/).  bench.py re-exports these names (tools import them as ``bench.build_stream`` ...)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.image_processing.feature_resize import upsample_features  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper, integrate_frame  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3e12 achievable

CPU_THREAD_SWEEP = (8, 16, 32, 64)  # + all host threads; the best setting is the reported CPU baseline

def lowres_features(index: int, channels: int, lowres: int = 16) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(1000003 * (index + 1)))
    return rng.standard_normal((channels, lowres, lowres), dtype=np.float32)

def build_stream(cfg: S.StreamConfig, n_frames: int, channels: int, device):
    """Pre-generate the frames on the device: depth f32, rgb u8, features f16 HWC (via the HIP upsample
    kernel, the path's own K11 replacement), dynamic mask, pose, K."""
    frames = []
    stride = max(cfg.num_poses // n_frames, 1)
    for k in range(n_frames):
        idx = (k * stride) % cfg.num_poses
        T = S.camera_pose(cfg, idx)
        depth = S.render_depth(cfg, T)
        rgb = S.render_rgb(cfg, idx)
        low = torch.from_numpy(lowres_features(idx, channels)).to(device)
        feat = upsample_features(low, (cfg.height, cfg.width), channels)
        frames.append({
            "index": idx,
            "lowres": low.permute(1, 2, 0).contiguous(),  # [h,w,C] f32: the backbone output the image was made from
            "depth": torch.from_numpy(depth).to(device),
            "rgb": torch.from_numpy(rgb).to(device),
            "features": feat,
            "dynamic_mask": torch.zeros((cfg.height, cfg.width), dtype=torch.bool, device=device),
            "T_W_C": torch.from_numpy(T),
            "K": torch.from_numpy(cfg.intrinsics()),
        })
    torch.cuda.synchronize(device)
    return frames

def step(mapper, mcfg, fr):
    """decay + the STATIC-mapper half of nvblox_integrate (nvblox_mapping_helpers.py:116-141): static mask = ~dynamic mask
    (read inverted by the native call), depth + colour + feature integration."""
    mapper.decay()
    integrate_frame(mapper=mapper, nvblox_mapping_config=mcfg, depth_frame=fr["depth"], feature_frame=fr["features"],
                    intrinsics=fr["K"], camera_pose=fr["T_W_C"], rgb=fr["rgb"], input_mask=fr["dynamic_mask"],
                    input_mask_erosion_iterations=mcfg.static_mask_erosion_iterations,
                    valid_depth_mask_erosion_iterations=mcfg.valid_depth_mask_erosion_iterations,
                    mapper_id=MAPPER_TO_ID.STATIC, invert_input_mask=True)

def thread_settings():
    ncpu = os.cpu_count() or 1
    return sorted({t for t in CPU_THREAD_SWEEP if t < ncpu} | {ncpu})

def cpu_quota():
    """CPUs this process may actually use: the cgroup's CPU bandwidth quota (a container on a 256-thread host is often capped
    well below os.cpu_count()), else the affinity mask.  Every CPU-side figure of the line (cpu_baseline, the loader) is
    bounded by it."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    try:
        return float(len(os.sched_getaffinity(0)))
    except AttributeError:
        return float(os.cpu_count() or 1)

def flat_bytes_per_voxel(C: int) -> int:
    """Algorithmic bytes k_feature_flat moves per updated voxel: the voxel's f16 channel row read and written (2 x 2C), its
    four bilinear taps of the f16 feature image (4 x 2C) and its 20-byte survivor record."""
    return 2 * 2 * C + 4 * 2 * C + 20

def measure_d2d_copy(device, mib=1024, iters=10):
    """Device-to-device copy rate of the box (the measured counterpart of the 8 TB/s spec peak): bytes read +
    bytes written per second of a large torch copy."""
    src = torch.empty(mib * 1024 * 1024, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize(device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize(device)
    sec = a.elapsed_time(b) / iters * 1e-3
    del src, dst
    torch.cuda.empty_cache()
    return 2.0 * mib * 1024 * 1024 / sec / 1e9

def frame_byte_model(cfg, C, n_live, n_tsdf_upd, n_cand, n_surv, with_decay=True):
    """ALGORITHMIC bytes per launch of the fused frame = what THIS implementation's algorithm has to move between HBM and the
    chip, counted from the run's own device counters (DESIGN.md section 5 states the same formulas):
      k_front        depth f32 + input mask u8 read, masked depth f32 + two bit-row planes written
      k_alloc_tsdf   every LIVE TSDF block read (8 B/voxel: the pass flags appearance candidates on the voxels it holds) and,
                     with a decay pending, written back (else only the blocks the frame integrates); masked depth read once;
                     bit rows read, depth mask + feature mask u8 written; 32 B of list / stamp / summary words per live block
      k_sphere_alloc near-surface (candidate) blocks' TSDF voxels read once + the 1/4-resolution synthetic depth written
                     + the two candidate lists (16 B per candidate and layer)
      k_app_frame    candidate blocks: colour voxels (8 B) + feature weights (4 B) read and written; rgb u8x3 + two masks +
                     synthetic depth read; 20 B survivor record per surviving voxel
      k_feature_flat per surviving voxel: f16 row read + written (2 x 2C), four bilinear taps (4 x 2C), 20 B record
    """
    HW = cfg.height * cfg.width
    synth = (cfg.height // 4) * (cfg.width // 4) * 4
    return {
        "k_front": HW * (4 + 1 + 4) + 2 * HW / 8,
        "k_alloc_tsdf": n_live * 512 * 8 + (n_live if with_decay else n_tsdf_upd) * 512 * 8 + HW * 4 + 2 * HW / 8 + 2 * HW + 32 * n_live,
        "k_sphere_alloc": n_cand * 512 * 8 + synth + 2 * 16 * n_cand,
        "k_app_frame": n_cand * 512 * (8 + 4) * 2 + HW * (3 + 1 + 1) + synth + 20 * n_surv,
        "k_feature_flat": n_surv * flat_bytes_per_voxel(C),
    }

KERNEL_OF_CLASS = {"raycast": "k_front", "tsdf": "k_alloc_tsdf", "sphere": "k_sphere_alloc", "feature": "k_app_frame",
                   "feature_flat": "k_feature_flat"}

# What bounds each launch, with the counter evidence it rests on (tools/profile_sq.sh -> profiles/*_sq_summary.json: SQ counters of
# the same bench command, fractions of SQ_WAVE_CYCLES).  valu_issue: the achieved rate is VALU wave-instructions/s against the
# chip's issue peak (1 024 SIMDs x clock / 4 cycles per wave64 instruction); latency: most wave-cycles are parked in s_waitcnt on
# dependent loads at full occupancy; hbm: bytes/s against the HBM peak.
VALU_ISSUE_PEAK_PER_S = 1024 * 2.4e9 / 4.0

BOUND_OF_KERNEL = {"k_front": "valu_issue", "k_alloc_tsdf": "latency", "k_sphere_alloc": "valu_issue", "k_app_frame": "latency",
                   "k_feature_flat": "hbm", "k_front_app": "valu_issue", "k_sphere_alloc_flat": "valu_issue"}

# The launches of a frame: (profile class, kernel, the roles whose algorithmic bytes it moves).  Deferred mode (the headline:
# mmf_set_deferred_feature_rows): the colour update + feature gating and the row update of frame N are roles of launches 1 and 3
# of frame N + 1 -- three launches per frame in a stream.
LAUNCHES_EAGER = [("raycast", "k_front", ["k_front"]), ("tsdf", "k_alloc_tsdf", ["k_alloc_tsdf"]), ("sphere", "k_sphere_alloc", ["k_sphere_alloc"]),
                  ("feature", "k_app_frame", ["k_app_frame"]), ("feature_flat", "k_feature_flat", ["k_feature_flat"])]

LAUNCHES_DEFERRED = [("raycast", "k_front_app", ["k_front", "k_app_frame"]), ("tsdf", "k_alloc_tsdf", ["k_alloc_tsdf"]),
                     ("sphere", "k_sphere_alloc_flat", ["k_sphere_alloc", "k_feature_flat"])]

def counters_stamp(path):
    """The `__csrc_sha16__` a counter summary under profiles/ carries (the native sources it was collected on), or None (an older
    summary without a stamp)."""
    try:
        with open(path) as fh:
            return json.load(fh).get("__csrc_sha16__")
    except Exception:
        return None

def latest_sq_summary():
    """File name (under profiles/) of the SQ counter summary the bench replays: the one collected on the sources of THIS build when
    there is one (`__csrc_sha16__`), else the last by name -- which `counters_stale` then reports."""
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sq_summary.json"))
    if not files:
        return None
    try:
        from nvblox_mindmap_amd._lib import source_hash

        build = source_hash()
        for f in reversed(files):
            if counters_stamp(os.path.join(ROOT, "profiles", f)) == build:
                return f
    except Exception:
        pass
    return files[-1]


def sq_evidence():
    """Per-kernel SQ summary of the latest committed counter run (a replayed constant like roofline.traffic: labelled)."""
    name = latest_sq_summary()
    if not name:
        return {}, None
    files = [name]
    with open(os.path.join(ROOT, "profiles", files[-1])) as fh:
        raw = json.load(fh)
    out = {}
    for name, v in raw.items():
        if not isinstance(v, dict) or "SQ_WAVES" not in v:
            continue
        base = name.split("<")[0]
        if base in BOUND_OF_KERNEL and base not in out:
            out[base] = {k: v.get(k) for k in ("SQ_WAVES", "valu_per_wave", "salu_per_wave", "frac_parked", "frac_issuing", "frac_issue_stall")}
    return out, f"profiles/{files[-1]} ({raw.get('__source__', '')})"

def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/latest_pmc.json):
    2 x FETCH_SIZE (gfx950 correction of the guide) + WRITE_SIZE.  A replayed constant from the builder's profile run, not a
    live measurement -- the line says so (`traffic_source`)."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
            pmc = json.load(f)
        src = pmc.get("__source__", "profiles/latest_pmc.json (builder's rocprofv3 --pmc passes of this command)")
        out = {}
        for k, v in pmc.items():  # keys carry the template arguments (k_front<true>): fold them onto the kernel's base name
            if isinstance(v, dict) and "FETCH_SIZE_KB" in v:
                out.setdefault(k.split("<")[0], (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0)
        return out, src
    except Exception:
        return {}, None
