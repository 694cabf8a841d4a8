"""nvblox pin kit, CPU side: the kit's mask algebra against the reference-generated golden masks, the kit + consumer
round trip on the CPU oracle, and -- when somebody has dumped real nvblox vectors into tests/golden/nvblox_*.npz with
tools/dump_nvblox_golden.py -- the oracle against them (skipped until then: the integrator oracle is parity-unpinned)."""
import json
import os

import numpy as np
import pytest

import nvblox_golden_common as NG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_kit_mask_algebra_matches_the_reference_generated_masks():
    """frame_masks of the kit == integrate_frame's masks as composed from the reference's own primitives (masks.npz: fm_*
    cases of tests/golden/make_golden.py, min distance 0.30) wherever the feature image has the depth image's size."""
    kit = NG.load_kit()
    g = np.load(os.path.join(GOLD, "masks.npz"))
    n = 0
    for name in ("same", "up", "down", "odd", "sq"):
        h, w, hf, wf, k_in, k_depth, pct = [int(v) for v in g[f"fm_{name}_params"]]
        if (hf, wf) != (h, w):
            continue
        unpack = lambda a: np.unpackbits(a)[: h * w].reshape(h, w).astype(bool)  # noqa: E731
        dm, fm = kit.frame_masks(unpack(g[f"fm_{name}_in"]), g[f"fm_{name}_depth"], 0.30, k_in, k_depth, pct)
        assert np.array_equal(dm, unpack(g[f"fm_{name}_depth_mask"])), name
        assert np.array_equal(fm, unpack(g[f"fm_{name}_feature_mask"])), name
        n += 1
    assert n == 2


def test_kit_mask_algebra_matches_the_numpy_oracle_on_random_masks():
    from oracle import image_ops as IMG

    kit = NG.load_kit()
    rng = np.random.default_rng(0)
    for H, W, k1, k2, border in ((48, 64, 3, 5, 5), (120, 160, 17, 20, 5), (37, 53, 0, 1, 0), (64, 64, 40, 2, 10)):
        m = rng.uniform(size=(H, W)) > 0.02
        depth = np.where(rng.uniform(size=(H, W)) > 0.01, rng.uniform(0.2, 2.0, size=(H, W)), 0.0).astype(np.float32)
        dm, fm = kit.frame_masks(m, depth, 0.3, k1, k2, border)
        dm2, fm2 = IMG.frame_masks(m, depth, 0.3, k1, k2, border, H, W)
        assert np.array_equal(dm, dm2) and np.array_equal(fm, fm2)


def test_kit_round_trip_on_the_cpu_oracle(tmp_path):
    """dump (kit.replay on the oracle adaptor) -> file -> consumer replay -> compare: identical.  Exercises the kit's whole
    code path and the file format without nvblox or a GPU; a flipped spec parameter must show up in the report."""
    kit = NG.load_kit()
    out = kit.replay(NG.oracle_backend(), "small", "patches", 4, True, True, device="cpu")
    path = tmp_path / "nvblox_small_patches.npz"
    np.savez_compressed(path, **out)
    gold = np.load(path, allow_pickle=False)
    meta = json.loads(str(gold["meta"]))
    assert meta["config"] == "small" and len(meta["frame_indices"]) == 4 and meta["spec_items"]
    assert len(gold["tsdf_indices"]) > 50 and int(gold["n_vertices"]) > 100 and len(gold["feature_indices"]) > 0
    same = NG.compare(gold, NG.replay_like(gold, NG.oracle_backend(), "cpu"))
    assert NG.passes_north_star(same), same
    assert NG.passes_reference_tolerances(same), same
    flipped = NG.compare(gold, NG.replay_like(gold, NG.oracle_backend(weighting_mode=0), "cpu"))
    assert not NG.passes_north_star(flipped), flipped


def test_pin_report_flips_the_spec_items_without_code_edits(tmp_path, capsys):
    """tests/pin_report.py on a file dumped from the oracle itself: "as specified" reproduces it, at least twenty spec items are
    parameters (flipped with zero code edits), and the flips are not no-ops -- all but a few change the replay."""
    import pin_report

    kit = NG.load_kit()
    out = kit.replay(NG.oracle_backend(), "small", "patches", 3, True, True, device="cpu")
    path = tmp_path / "nvblox_small_patches.npz"
    np.savez_compressed(path, **out)
    meta = json.loads(str(np.load(path, allow_pickle=False)["meta"]))
    items = [it for it in meta["spec_items"] if it.get("param") is not None]
    assert len(items) >= 20, [it["item"] for it in items]
    assert {"weighting_mode", "raycast_to_truncation", "decay_appearance_layers", "raycast_walk_from_camera",
            "appearance_blend_division", "fma_contraction", "block_index_by_division", "view_truncation_band_marking",
            "bilinear_four_weight_sum"} <= {it["param"] for it in items}
    wm = next(it for it in items if it["param"] == "weighting_mode")
    assert sorted(wm["flips"] + [wm["ours"]]) == [0, 1, 2, 3, 4, 5], "upstream's six weighting functions"
    results = pin_report.main([str(path)])
    capsys.readouterr()
    name0, r0 = results[0]
    assert name0 == "as specified" and NG.passes_north_star(r0)
    changed = {name for name, r in results[1:] if not NG.passes_north_star(r)}
    params_changed = {n.split("=")[0] for n in changed}
    # (three frames cannot saturate the maximum weights or reach the ray-length limit: those flips need a longer stream)
    assert len(params_changed) >= 9, sorted(params_changed)
    for mode in (0, 2, 3, 4, 5):
        assert f"weighting_mode={mode}" in changed, "every member of the weighting family must give a different map"
    # the two GPU-motivated arrangements of the spec are flips like any other: walking from the camera gives
    # the SAME map (that is the claim it was adopted on), dividing per channel stays within the north star's 1e-5
    by_name = dict(results)
    assert NG.passes_north_star(by_name["raycast_walk_from_camera=1"]) and by_name["raycast_walk_from_camera=1"]["tsdf_blocks_missing"] == 0
    assert by_name["raycast_walk_from_camera=1"]["tsdf_max_abs_distance_diff"] == 0.0
    assert by_name["appearance_blend_division=1"]["feature_max_abs_diff"] <= 2e-3  # <= 1 f16 ulp of values of magnitude ~2
    # the arithmetic mode (round 5): contracting the multiply-adds changes last bits -- same blocks, TSDF inside the north star's 1e-5,
    # features within one f16 ulp -- and NOT bit-identical: a dump from a contracting build is told apart from a non-contracting one
    # by this flip alone (before it existed no flip could have explained such a difference)
    fma = by_name["fma_contraction=1"]
    assert fma["tsdf_blocks_missing"] == fma["tsdf_blocks_extra"] == fma["feature_blocks_missing"] == fma["feature_blocks_extra"] == 0, fma
    assert 0.0 < fma["tsdf_max_abs_distance_diff"] <= 1e-5 and 0.0 < fma["feature_max_abs_diff"] <= 2e-3, fma
    assert NG.passes_reference_tolerances(fma), fma
    # round 6: three more recollection risks are flips -- the four-tap bilinear sum moves values at the last bits (same blocks), the
    # second marking pass only ever ADDS blocks, the division rule keeps this stream's maps (its differences live on block faces)
    w4, band, bdiv = by_name["bilinear_four_weight_sum=1"], by_name["view_truncation_band_marking=1"], by_name["block_index_by_division=1"]
    # (that it is not a no-op is asserted on whole layers in tests/test_gpu_fusion_parity.py; the six sampled blocks of three frames may agree)
    assert w4["tsdf_blocks_missing"] == w4["tsdf_blocks_extra"] == 0 and w4["tsdf_max_abs_distance_diff"] <= 1e-5, w4
    assert NG.passes_reference_tolerances(w4), w4
    assert band["tsdf_blocks_missing"] == 0 and band["tsdf_blocks_extra"] > 0 and band["tsdf_max_abs_distance_diff"] == 0.0, band
    assert NG.passes_reference_tolerances(band), band
    assert bdiv["tsdf_blocks_missing"] == bdiv["tsdf_blocks_extra"] == 0, bdiv


@pytest.mark.parametrize("path", NG.golden_files() or [None])
def test_oracle_matches_dumped_nvblox_vectors(path):
    if path is None:
        pytest.skip("no tests/golden/nvblox_*.npz dumped from upstream nvblox_torch yet (tools/dump_nvblox_golden.py): "
                    "the integrator oracle stays parity-unpinned")
    gold = np.load(path, allow_pickle=False)
    meta = json.loads(str(gold["meta"]))
    if meta["config"] != "small":
        pytest.skip("full-size streams are checked on the GPU (tests/test_gpu_nvblox_golden.py)")
    r = NG.compare(gold, NG.replay_like(gold, NG.oracle_backend(), "cpu"))
    assert NG.passes_north_star(r) or NG.passes_reference_tolerances(r), r
