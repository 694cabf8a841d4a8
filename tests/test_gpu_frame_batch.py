"""N independent frames in ONE set of launches (``mmf_integrate_frame_batch`` / ``nvblox_torch.mapper.integrate_frames_batch``):
every map must be bit-identical to the one the single calls build, and to its own CPU oracle -- with different cameras, images,
masks, feature widths per Mapper object, pending decays, a frame that cannot take the merged path in the middle of a batch, and
more frames than one set of launches carries."""
import numpy as np
import pytest
import torch

from fusion_common import make_mapper, make_oracle, small_cfg
from nvblox_mindmap_amd import synthetic as S
from nvblox_mindmap_amd.nvblox_torch.mapper import integrate_frames_batch
from test_gpu_fusion_parity import compare_features, compare_tsdf, dev

pytestmark = pytest.mark.gpu


def same_maps(a, b, mapper_id=0):
    for layer in ("tsdf", "color", "feature"):
        va = getattr(a, layer + "_layer_view")(mapper_id)
        vb = getattr(b, layer + "_layer_view")(mapper_id)
        xa = va.get_all_blocks() if layer == "tsdf" else va.get_all_blocks_split()
        xb = vb.get_all_blocks() if layer == "tsdf" else vb.get_all_blocks_split()
        assert len(xa) == len(xb)
        for p, q in zip(xa, xb):
            assert p.shape == q.shape and torch.equal(p, q), layer


def frame_args(cfg, mapper, index, channels, k, k_in=3, k_depth=4, border=5, min_d=0.3):
    f = S.frame(cfg, index, channels)
    dyn = np.zeros(f["depth"].shape, dtype=bool)
    dyn[(5 * k) % 40 + 10: (5 * k) % 40 + 40, 20 + 3 * k: 70 + 3 * k] = True
    return dict(mapper=mapper, mapper_id=0, depth_frame=dev(f["depth"]), color_frame=dev(f["rgb"]), feature_frame=dev(f["features"]),
                input_mask=dev(dyn), t_w_c=torch.from_numpy(f["T_W_C"]), intrinsics=torch.from_numpy(f["K"]), min_depth_m=min_d,
                input_mask_erosion_iterations=k_in, valid_depth_mask_erosion_iterations=k_depth, border_percent=border,
                invert_input_mask=True), f, dyn


def single(args):
    a = dict(args)
    m = a.pop("mapper")
    mid = a.pop("mapper_id")
    return m.integrate_frame(a["depth_frame"], a["color_frame"], a["feature_frame"], a["input_mask"], a["t_w_c"], a["intrinsics"],
                             a["min_depth_m"], a["input_mask_erosion_iterations"], a["valid_depth_mask_erosion_iterations"],
                             a["border_percent"], mid, invert_input_mask=a["invert_input_mask"])


@pytest.mark.parametrize("n", [3, 5, 8, 11])
def test_batch_equals_single_calls_and_oracles(oracle_mod, n):
    from oracle import image_ops as IO

    base = small_cfg(2)  # 320x240
    cfg = S.StreamConfig(width=base.width, height=base.height, fx=base.fx, fy=base.fy, cx=base.cx, cy=base.cy, hole_mode="patches")
    C = 16
    batched = [make_mapper(C) for _ in range(n)]
    alone = [make_mapper(C) for _ in range(n)]
    orcs = [make_oracle(oracle_mod, C) for _ in range(min(n, 3))]  # (the oracle is slow: the first three streams)
    for step in range(4):
        entries_b, entries_a, frames = [], [], []
        for q in range(n):
            index = (17 * q + 9 * step) % 200  # every stream has its own camera path
            eb, f, dyn = frame_args(cfg, batched[q], index, C, q + step)
            ea = dict(eb, mapper=alone[q])
            entries_b.append(eb)
            entries_a.append(ea)
            frames.append((f, dyn))
        for m in batched + alone:
            m.decay()
        masks_b = integrate_frames_batch(entries_b)
        masks_a = [single(e) for e in entries_a]
        for (dmb, fmb), (dma, fma) in zip(masks_b, masks_a):
            assert torch.equal(dmb, dma) and torch.equal(fmb, fma)
        for q, orc in enumerate(orcs):
            f, dyn = frames[q]
            odm, ofm = IO.frame_masks(~dyn, f["depth"], 0.3, 3, 4, 5, cfg.height, cfg.width)
            orc.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
            orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
    for q in range(n):
        same_maps(batched[q], alone[q])
    for q, orc in enumerate(orcs):
        mx, exact = compare_tsdf(orc, batched[q])
        assert exact
        compare_features(orc, batched[q])


def test_batch_with_mixed_mappers(oracle_mod):
    """Different Mapper objects in one batch: feature widths 16 / 32, an UNBOUNDED mapper (not eligible for the merged path: it is
    integrated on its own, in order, between two sets of launches), both mappers of a two-mapper object, a mapper without a
    pending decay beside ones with it."""
    from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg

    base = small_cfg(4)
    cfg = S.StreamConfig(width=base.width, height=base.height, fx=base.fx, fy=base.fy, cx=base.cx, cy=base.cy, hole_mode="patches")

    def build():
        two = get_nvblox_mapper(NvbloxMappingCfg("DRILL_IN_BOX"), feature_channels=16)
        return [(make_mapper(16), 0, 16), (make_mapper(32), 0, 32), (make_mapper(16, workspace_bounds_type=0, voxel_size=0.02,
                                                                                  max_integration_distance_m=3.0), 0, 16),
                (two, 0, 16), (two, 1, 16), (make_mapper(16), 0, 16)]

    A, B = build(), build()
    for step in range(4):
        for side, use_batch in ((A, True), (B, False)):
            entries = []
            for q, (m, mid, C) in enumerate(side):
                e, _, _ = frame_args(cfg, m, (23 * q + 7 * step) % 200, C, q + step)
                e["mapper_id"] = mid
                entries.append(e)
                if q != 5 or step % 2 == 0:  # the last mapper decays every other step only
                    m.decay(mid)
            if use_batch:
                integrate_frames_batch(entries)
            else:
                for e in entries:
                    single(e)
    for (ma, mida, _), (mb, midb, _) in zip(A, B):
        same_maps(ma, mb, mida)
    with pytest.raises(RuntimeError, match="different mapper"):
        e, _, _ = frame_args(cfg, A[0][0], 0, 16, 0)
        integrate_frames_batch([e, dict(e)])
