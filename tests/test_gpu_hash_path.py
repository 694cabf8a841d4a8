"""The voxel-block HASH path (kUnbounded workspace: open-addressing table, CAS insertion, tombstones, amortised rebuild) at
BASELINE's full size -- 640x480 depth + 64-channel features, 1 cm voxels -- against the CPU oracle, bit for bit.  The reference's
tasks configure a bounding-box workspace (nvblox_mapping_helpers.py:53-59), where the block index is a dense table and the hash is
not kept at all; nvblox's own default (``Mapper(voxel_sizes_m=0.01)``, paper/teaser/convert_maps_usd.py:38) is unbounded."""
import numpy as np
import pytest
import torch

import fusion_common
from fusion_common import make_mapper, make_oracle
from nvblox_mindmap_amd import _lib
from nvblox_mindmap_amd import synthetic as S
from test_gpu_fusion_parity import compare_features, compare_tsdf, dev

pytestmark = pytest.mark.gpu


def fused_frame(gpu, orc, cfg, index, channels, k_in, k_depth, border, min_d=0.3):
    """decay + the reference's integrate_frame (mask algebra + depth + colour + features) on both sides."""
    from oracle import image_ops as IO

    f = S.frame(cfg, index, channels)
    static = np.ones(f["depth"].shape, dtype=bool)
    odm, ofm = IO.frame_masks(static, f["depth"], min_d, k_in, k_depth, border, cfg.height, cfg.width)
    orc.decay()
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
    orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
    orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
    gpu.decay()
    gpu.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), dev(static), torch.from_numpy(f["T_W_C"]),
                        torch.from_numpy(f["K"]), min_d, k_in, k_depth, border, 0)


def test_unbounded_full_size_fused_frames_match_oracle(oracle_mod):
    """Six fused frames of the benchmark stream in an UNBOUNDED workspace (5 m integration distance as the reference sets it,
    reference mask algebra: 17 / 20-pixel erosions, 5 % border)."""
    cfg = S.StreamConfig(hole_mode="patches")
    over = dict(workspace_bounds_type=0)
    gpu, orc = make_mapper(64, **over), make_oracle(oracle_mod, 64, **over)
    for i in (0, 7, 14, 21, 28, 35):
        fused_frame(gpu, orc, cfg, i, 64, 17, 20, 5)
    st = gpu.hash_state(0)
    assert st["table_entries"] >= 2 * st["live_blocks"] > 0, st
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)
    rgb, w, idx = gpu.color_layer_view(0).get_all_blocks_split()
    orgb, ow = orc.all_colors()
    assert np.array_equal(idx.cpu().numpy(), orc.block_indices(1))
    assert np.array_equal(w.cpu().numpy(), ow) and np.array_equal(rgb.cpu().numpy(), orgb)
    ov, of = orc.feature_mesh()
    mesh = gpu.get_feature_mesh(0)
    assert np.array_equal(mesh.vertices().cpu().numpy(), ov)
    assert np.array_equal(mesh.vertex_features().cpu().numpy().view(np.uint16), of.view(np.uint16))


def test_unbounded_full_size_decay_churn_rebuilds_the_table(oracle_mod):
    """Strong decay on a moving camera: every frame deallocates the blocks that left the view (tombstones in the table) and
    allocates new ones into reused slots, long enough for the tombstones to exceed a quarter of the table -- the amortised
    rebuild -- at least once.  Pool of 16 384 blocks -> 32 768 table entries.  TSDF + features, stand-alone and fused calls
    mixed (the stand-alone decay is the eager two-launch form, the fused frame folds it in)."""
    cfg = S.StreamConfig(hole_mode="patches")
    over = dict(workspace_bounds_type=0, tsdf_decay_factor=0.2, decayed_weight_threshold=1e-3, max_integration_distance_m=2.5,
                num_preallocated_blocks=16384)
    gpu, orc = make_mapper(64, **over), make_oracle(oracle_mod, 64, **over)
    rebuilds_seen = 0
    for k in range(24):
        index = (k * 23) % cfg.num_poses
        if k % 3 == 2:  # stand-alone calls: eager decay, three-kernel allocation
            f = S.frame(cfg, index, 64)
            orc.decay()
            gpu.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
            gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"])
            gpu.add_feature_frame(dev(f["features"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
        else:
            fused_frame(gpu, orc, cfg, index, 64, 3, 4, 5)
        st = gpu.hash_state(0)
        rebuilds_seen = max(rebuilds_seen, st["rebuilds"])
        assert st["live_blocks"] == orc.num_blocks(0), (k, st)
        # the counter behind the amortised rebuild is exact: insertion takes a reused tombstone off it (a scan of the table agrees)
        for layer in (_lib.MMF_LAYER_TSDF, _lib.MMF_LAYER_FEATURE):
            assert gpu.count_tombstones(0, layer) == gpu.hash_state(0, layer)["tombstones"], (k, layer, gpu.hash_state(0, layer))
        if k % 6 == 5:
            compare_tsdf(orc, gpu)
    assert rebuilds_seen >= 1, f"no tombstone rebuild in 24 frames: {gpu.hash_state(0)}"
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)
    stats = gpu.stats(0)
    assert stats["tsdf_blocks_allocated"] - gpu.hash_state(0)["live_blocks"] > 8192, "the churn must exceed a quarter of the table"


def test_lazy_decay_of_a_large_map_matches_the_eager_oracle(oracle_mod):
    """Fused frames into an unbounded map decay LAZILY: a decay multiplies the per-block summaries, a block's
    voxels catch up -- multiplication by multiplication -- when the block is next integrated, sampled by the sphere tracer, or read.
    A camera hopping around the orbit under a strong decay (blocks leave the view, stay behind for several decays, die or come
    back) against the oracle's eager decay, bit for bit: with a mesh read and a stand-alone call in mid-stream (both force the
    catch-up and the second drops the lazy state), and the laziness really engaged."""
    cfg = S.StreamConfig(hole_mode="patches")
    over = dict(workspace_bounds_type=0, tsdf_decay_factor=0.4, decayed_weight_threshold=1e-3, max_integration_distance_m=2.5,
                num_preallocated_blocks=32768)
    gpu, orc = make_mapper(16, **over), make_oracle(oracle_mod, 16, **over)
    for k in range(16):
        index = (k * 37) % cfg.num_poses
        if k == 9:  # a stand-alone call between fused frames: eager decay + three-kernel chain on caught-up voxels
            f = S.frame(cfg, index, 16)
            orc.decay()
            gpu.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
            gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
            continue
        fused_frame(gpu, orc, cfg, index, 16, 3, 4, 5)
        if k == 5:  # a reader in mid-stream sees current weights
            ov, _ = orc.feature_mesh()
            assert np.array_equal(gpu.get_feature_mesh(0).vertices().cpu().numpy(), ov)
    st = gpu.hash_state(0)
    # (mappers under MMF_SPEC_FLIPS take the stand-alone launches: eager decay -- the map must still be the oracle's)
    assert (st["lazy_decays"] >= 10 or bool(fusion_common.SPEC_FLIPS)) and st["live_blocks"] == orc.num_blocks(0), st
    assert gpu.stats(0)["tsdf_blocks_allocated"] > st["live_blocks"] + 500, "blocks must have died along the way"
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)
    ov, of = orc.feature_mesh()
    mesh = gpu.get_feature_mesh(0)
    assert np.array_equal(mesh.vertices().cpu().numpy(), ov)


@pytest.mark.parametrize("merged", [True, False])
def test_large_map_launch_arrangements_give_the_same_map(oracle_mod, monkeypatch, merged):
    """Round 5: a large map's frame hosts the light decay's list compaction in its first launch and the appearance allocation beside the
    sphere trace, the conditional hash rebuild follows a compaction only now and then, and a pipelined stream's tail rides in those launches.
    MMF_NO_BIG_MERGE=1 (read when the mapper is created) keeps the round-4 sequence -- both must give the oracle's map, pipelined or not."""
    if not merged:
        monkeypatch.setenv("MMF_NO_BIG_MERGE", "1")
    cfg = S.StreamConfig(hole_mode="patches")
    over = dict(workspace_bounds_type=0, tsdf_decay_factor=0.9)
    gpu, piped, orc = make_mapper(64, **over), make_mapper(64, **over), make_oracle(oracle_mod, 64, **over)
    piped.set_deferred_feature_rows(True)
    from oracle import image_ops as IO

    for i in (0, 9, 18, 27, 36):
        f = S.frame(cfg, i, 64)
        static = np.ones(f["depth"].shape, dtype=bool)
        odm, ofm = IO.frame_masks(static, f["depth"], 0.3, 17, 20, 5, cfg.height, cfg.width)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
        for m in (gpu, piped):
            m.decay()
            m.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), dev(static), torch.from_numpy(f["T_W_C"]),
                              torch.from_numpy(f["K"]), 0.3, 17, 20, 5, 0)
        pend = _lib.lib().mmf_deferred_feature_rows_pending(piped._h, 0)
        # (without the merged launches -- or with fma_contraction -- a large map's frame is complete when the call returns)
        assert pend == (1 if (merged and not fusion_common.NOT_DEFAULT_ROUTE) else 0)
    for m in (gpu, piped):
        _, exact = compare_tsdf(orc, m)
        assert exact
        compare_features(orc, m)
