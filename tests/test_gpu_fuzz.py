"""Randomised differential test of the fused frame (mmf_integrate_frame_desc / mmf_integrate_frame_multi) against the CPU
oracle: every seed draws an image size (incl. sizes that are not multiples of the 16-pixel mask groups or the 8-pixel
sphere-trace patches), a feature width, a voxel size, the three mask parameters, a decay factor, a camera path and a moving
dynamic region, then runs the reference's call sequence (nvblox_mapping_helpers.py:79-156) -- single mapper or static +
dynamic pair -- on the HIP path and the same frames through the oracle's add_depth/add_color/add_feature chain.

Bar: block index lists equal in ORDER, TSDF within 1e-5 abs, feature halves and weights bit-equal, colours bit-equal, masks
bit-equal.  Seeds are fixed: a failure reproduces.  ``MMF_FUZZ_SEEDS=first:count`` runs that seed range INSTEAD (a campaign:
``MMF_FUZZ_SEEDS=100:400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 4``; profiles/r05j_fuzz_campaign.txt)."""
import os

import numpy as np
import pytest
import torch

from nvblox_mindmap_amd import synthetic as S

from fusion_common import make_oracle

pytestmark = pytest.mark.gpu

def _seeds(default):
    spec = os.environ.get("MMF_FUZZ_SEEDS")
    if not spec:
        return range(default)
    first, count = (int(v) for v in spec.split(":"))
    return range(first, first + count)


SIZES = [(96, 72), (128, 96), (136, 104), (160, 120), (200, 152), (216, 168), (320, 240), (328, 248)]
CHANNELS = [8, 16, 24, 40, 64, 128]


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def draw(seed):
    r = np.random.default_rng(1000 + seed)
    w, h = SIZES[r.integers(len(SIZES))]
    k_depth = int(r.integers(0, 7))
    # 1 % isolated invalid pixels + an erosion of the valid-depth mask by more than a pixel or two leaves no feature mask at all
    hole_mode = str(r.choice(["patches", "pixels"])) if k_depth <= 1 else "patches"
    return dict(
        w=w, h=h, C=int(CHANNELS[r.integers(len(CHANNELS))]), voxel=float(r.choice([0.01, 0.0125, 0.02])),
        k_static=int(r.integers(0, 6)), k_dynamic=int(r.integers(0, 4)), k_depth=k_depth, border=int(r.choice([0, 5, 12])),
        decay=float(r.choice([0.98, 0.8, 0.4])), min_d=float(r.choice([0.1, 0.3, 0.37])), weight=float(r.choice([1.0, 0.5])),
        frames=[int(i) for i in r.integers(0, 200, size=int(r.integers(3, 7)))], pair=bool(r.integers(2)),
        rect=[int(v) for v in (r.integers(0, h // 2), r.integers(h // 2, h), r.integers(0, w // 2), r.integers(w // 2, w))],
        drift=[int(v) for v in r.integers(-5, 6, size=2)], hole_mode=hole_mode)


@pytest.mark.parametrize("seed", _seeds(24))
def test_random_configuration_matches_oracle(oracle_mod, seed, monkeypatch):
    import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers as H
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg
    from oracle import image_ops as IO

    p = draw(seed)
    print(p)
    f0 = 525.0 * p["w"] / 640.0
    cfg = S.StreamConfig(width=p["w"], height=p["h"], fx=f0, fy=f0, cx=p["w"] / 2 - 0.5, cy=p["h"] / 2 - 0.5, hole_mode=p["hole_mode"])
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX", voxel_size_m_override=p["voxel"], measurement_weight_override=p["weight"])
    mcfg.tsdf_decay_factor = p["decay"]
    mcfg.static_mask_erosion_iterations, mcfg.dynamic_mask_erosion_iterations = p["k_static"], p["k_dynamic"]
    mcfg.valid_depth_mask_erosion_iterations, mcfg.feature_mask_border_percent = p["k_depth"], p["border"]
    mcfg.min_integration_distance_m = p["min_d"]
    gpu = H.get_nvblox_mapper(mcfg, feature_channels=p["C"])
    over = dict(voxel_size=p["voxel"], tsdf_decay_factor=p["decay"], appearance_measurement_weight=p["weight"])
    orcs = [make_oracle(oracle_mod, p["C"], **over) for _ in range(2 if p["pair"] else 1)]
    monkeypatch.setattr(H, "PAIR_MAPPERS", True)

    class Extractor:
        def compute(self, rgb):
            return self.next.unsqueeze(0)

    ex = Extractor()
    for k, i in enumerate(p["frames"]):
        f = S.frame(cfg, i, p["C"])
        dyn = np.zeros(f["depth"].shape, dtype=bool)
        r0, r1, c0, c1 = p["rect"]
        dr, dc = p["drift"][0] * k, p["drift"][1] * k
        dyn[max(r0 + dr, 0): max(r1 + dr, 0), max(c0 + dc, 0): max(c1 + dc, 0)] = True
        ex.next = dev(f["features"])
        gpu.decay()
        out = H.nvblox_integrate(gpu, mcfg, ex, dev(f["depth"]), torch.from_numpy(f["K"]), torch.from_numpy(f["T_W_C"]), dev(f["rgb"]),
                                 dev(dyn), include_dynamic=p["pair"])
        jobs = [("STATIC", ~dyn, p["k_static"])] + ([("DYNAMIC", dyn, p["k_dynamic"])] if p["pair"] else [])
        for orc, (name, mask, k_in) in zip(orcs, jobs):
            odm, ofm = IO.frame_masks(mask, f["depth"], p["min_d"], k_in, p["k_depth"], p["border"], cfg.height, cfg.width)
            assert np.array_equal(out[name]["depth_mask"].cpu().numpy().astype(bool), odm), (name, k, "depth mask")
            assert np.array_equal(out[name]["feature_mask"].cpu().numpy().astype(bool), ofm), (name, k, "feature mask")
            orc.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
            orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
    total = 0
    for mid, orc in zip((MAPPER_TO_ID.STATIC, MAPPER_TO_ID.DYNAMIC), orcs):
        t, ti = gpu.tsdf_layer_view(mid).get_all_blocks()
        assert np.array_equal(ti.cpu().numpy(), orc.block_indices(0)), "TSDF block indices / order"
        if ti.shape[0]:
            assert np.abs(t.cpu().numpy() - orc.all_tsdf()).max() <= 1e-5
        fv, fw, fi = gpu.feature_layer_view(mid).get_all_blocks_split()
        of, ow = orc.all_features()
        assert np.array_equal(fi.cpu().numpy(), orc.block_indices(2)), "feature block indices / order"
        assert np.array_equal(fw.cpu().numpy(), ow), "feature weights"
        assert np.array_equal(fv.cpu().numpy().view(np.uint16), of.view(np.uint16)), "feature values"
        rgb, cw, ci = gpu.color_layer_view(mid).get_all_blocks_split()
        orgb, ocw = orc.all_colors()
        assert np.array_equal(ci.cpu().numpy(), orc.block_indices(1)), "colour block indices / order"
        assert np.array_equal(cw.cpu().numpy(), ocw) and np.array_equal(rgb.cpu().numpy(), orgb), "colours"
        total += int(ti.shape[0])
    assert total > 0, "the drawn stream integrated nothing"


def draw_parameters(seed):
    r = np.random.default_rng(5000 + seed)
    w, h = SIZES[r.integers(len(SIZES))]
    bounds = int(r.choice([0, 1, 2, 2]))  # kUnbounded (hash table, frustum view grid) | kHeightBounds | kBoundingBox (dense table)
    voxel = float(r.choice([0.02, 0.04])) if bounds != 2 else float(r.choice([0.01, 0.0125, 0.02]))
    over = dict(
        voxel_size=voxel, workspace_bounds_type=bounds, max_integration_distance_m=float(r.choice([2.0, 3.0, 5.0])),
        truncation_distance_vox=float(r.choice([2.0, 4.0, 6.0])), weighting_mode=int(r.integers(6)), max_weight=float(r.choice([5.0, 100.0])),
        st_subsampling=int(r.choice([2, 4, 4, 8])), raycast_subsampling=int(r.choice([1, 1, 2, 4])),
        tsdf_decay_factor=float(r.choice([0.98, 0.5, 0.1])), decayed_weight_threshold=float(r.choice([1e-3, 1e-2])),
        appearance_measurement_weight=float(r.choice([1.0, 0.25])), raycast_to_truncation=int(r.choice([1, 1, 0])),
        decay_appearance_layers=int(r.choice([0, 0, 1])), lin_interp_max_diff_vox=float(r.choice([2.0, 2.0, 0.0, 1e9])),
        mesh_min_weight=float(r.choice([1e-4, 1e-2])), appearance_max_weight=float(r.choice([5.0, 100.0])))
    return dict(w=w, h=h, C=int(CHANNELS[r.integers(len(CHANNELS))]), over=over, frames=[int(i) for i in r.integers(0, 200, size=int(r.integers(2, 6)))],
                use_mask=bool(r.integers(2)), color=bool(r.integers(2)), decay=bool(r.integers(2)))


@pytest.mark.parametrize("seed", _seeds(16))
def test_random_parameters_stand_alone_calls_match_oracle(oracle_mod, seed):
    """The nvblox_torch surface call by call (decay / add_depth_frame / add_color_frame / add_feature_frame, then the feature
    mesh and a rendered depth image) with integrator parameters off the reference's defaults: truncation distance, weighting
    mode, maximum weight, sphere-trace and raycast subsampling, workspace bounds type (hash table vs dense table), decay."""
    from fusion_common import frame_masks, make_mapper

    p = draw_parameters(seed)
    print(p)
    f0 = 525.0 * p["w"] / 640.0
    cfg = S.StreamConfig(width=p["w"], height=p["h"], fx=f0, fy=f0, cx=p["w"] / 2 - 0.5, cy=p["h"] / 2 - 0.5)
    orc, gpu = make_oracle(oracle_mod, p["C"], **p["over"]), make_mapper(p["C"], **p["over"])
    for k, i in enumerate(p["frames"]):
        f = S.frame(cfg, i, p["C"])
        mask = frame_masks(f["depth"], k) if p["use_mask"] else None
        T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
        dmask = None if mask is None else dev(mask)
        if p["decay"]:
            orc.decay()
            gpu.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], mask)
        gpu.add_depth_frame(dev(f["depth"]), T, K, dmask, 0)
        if p["color"]:
            orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], mask)
            gpu.add_color_frame(dev(f["rgb"]), T, K, mask_frame=dmask, mapper_id=0)
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], mask)
        gpu.add_feature_frame(dev(f["features"]), T, K, dmask, 0)
    t, ti = gpu.tsdf_layer_view(0).get_all_blocks()
    assert np.array_equal(ti.cpu().numpy(), orc.block_indices(0)), "TSDF block indices / order"
    assert ti.shape[0] > 0
    assert np.array_equal(t.cpu().numpy().view(np.uint32), orc.all_tsdf().view(np.uint32)), "TSDF values"
    fv, fw, fi = gpu.feature_layer_view(0).get_all_blocks_split()
    of, ow = orc.all_features()
    assert np.array_equal(fi.cpu().numpy(), orc.block_indices(2)) and np.array_equal(fw.cpu().numpy(), ow)
    assert np.array_equal(fv.cpu().numpy().view(np.uint16), of.view(np.uint16)), "feature values"
    if p["color"]:
        rgb, cw, ci = gpu.color_layer_view(0).get_all_blocks_split()
        orgb, ocw = orc.all_colors()
        assert np.array_equal(ci.cpu().numpy(), orc.block_indices(1)) and np.array_equal(cw.cpu().numpy(), ocw)
        assert np.array_equal(rgb.cpu().numpy(), orgb)
    f = S.frame(cfg, p["frames"][-1], 0)
    so = orc.render_synthetic_depth(cfg.height, cfg.width, f["T_W_C"], f["K"])
    sg = gpu.render_synthetic_depth(cfg.height, cfg.width, f["T_W_C"], f["K"]).cpu().numpy()
    assert np.array_equal(so.view(np.uint32), sg.view(np.uint32)), "sphere-traced depth"
    ov, ofeat = orc.feature_mesh()
    gpu.update_feature_mesh(0)
    mesh = gpu.get_feature_mesh(0)
    gv, gf = mesh.vertices().cpu().numpy(), mesh.vertex_features().cpu().numpy()
    assert gv.shape == ov.shape and np.array_equal(gv.view(np.uint32), ov.view(np.uint32)), "mesh vertices"
    assert np.array_equal(gf.view(np.uint16), ofeat.view(np.uint16)), "mesh vertex features"


@pytest.mark.parametrize("pipelined", [False, True])
@pytest.mark.parametrize("seed", _seeds(12))
def test_random_call_sequences_match_oracle(oracle_mod, seed, tmp_path, pipelined):
    """A random walk over the Mapper's entry points on a two-mapper Mapper -- decay, the fused frame, the same frame call by
    call, the fused frame from a low-res feature map, both mappers in one call, clear, mesh update, save + load into a NEW
    mapper that carries on, point queries, depth rendering -- against two oracle maps driven with the equivalent calls.  What
    this exercises is the STATE between calls: pending lazy decay, grid / hand-over tags, slot reuse after clear and
    deallocation, hints sized by earlier frames, a restored map continuing.  ``pipelined``: the same walk with the appearance
    half of every fused frame deferred to the next one (``set_deferred_feature_rows``): whatever call comes next either hosts it or
    completes it first."""
    import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers as H
    from nvblox_mindmap_amd.image_processing import upsample_features
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg
    from nvblox_mindmap_amd.nvblox_torch.mapper import QueryType
    from oracle import image_ops as IO

    r = np.random.default_rng(9000 + seed)
    W, Hh, C = 160, 120, 16
    cfg = S.StreamConfig(width=W, height=Hh, fx=525.0 / 4, fy=525.0 / 4, cx=W / 2 - 0.5, cy=Hh / 2 - 0.5, hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    mcfg.tsdf_decay_factor = float(r.choice([0.98, 0.6, 0.2]))
    k_in, k_depth, border, min_d = 2, 3, mcfg.feature_mask_border_percent, mcfg.min_integration_distance_m
    gpu = H.get_nvblox_mapper(mcfg, feature_channels=C)
    gpu.set_deferred_feature_rows(pipelined)
    orcs = [make_oracle(oracle_mod, C, tsdf_decay_factor=mcfg.tsdf_decay_factor) for _ in range(2)]

    def frame_inputs():
        i = int(r.integers(0, 200))
        f = S.frame(cfg, i, C)
        low = r.standard_normal((8, 8, C)).astype(np.float32)  # a backbone output; its up-sampled image replaces the stream's
        img = upsample_features(dev(low).permute(2, 0, 1).contiguous(), (Hh, W), C)
        f["low"], f["features"] = dev(low), img.cpu().numpy()
        m = np.ones((Hh, W), dtype=bool)
        y0, x0 = int(r.integers(0, Hh - 30)), int(r.integers(0, W - 40))
        m[y0: y0 + int(r.integers(5, 30)), x0: x0 + int(r.integers(5, 40))] = False
        return f, m

    def oracle_frame(orc, f, mask):
        odm, ofm = IO.frame_masks(mask, f["depth"], min_d, k_in, k_depth, border, Hh, W)
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
        return odm, ofm

    def check(mid):
        orc = orcs[mid]
        t, ti = gpu.tsdf_layer_view(mid).get_all_blocks()
        assert np.array_equal(ti.cpu().numpy(), orc.block_indices(0)), ("TSDF blocks", mid, log)
        if ti.shape[0]:
            assert np.array_equal(t.cpu().numpy().view(np.uint32), orc.all_tsdf().view(np.uint32)), ("TSDF values", mid, log)
        fv, fw, fi = gpu.feature_layer_view(mid).get_all_blocks_split()
        of, ow = orc.all_features()
        assert np.array_equal(fi.cpu().numpy(), orc.block_indices(2)) and np.array_equal(fw.cpu().numpy(), ow), ("feature blocks", mid, log)
        assert np.array_equal(fv.cpu().numpy().view(np.uint16), of.view(np.uint16)), ("feature values", mid, log)
        rgb, cw, ci = gpu.color_layer_view(mid).get_all_blocks_split()
        orgb, ocw = orc.all_colors()
        assert np.array_equal(ci.cpu().numpy(), orc.block_indices(1)) and np.array_equal(cw.cpu().numpy(), ocw), ("colour blocks", mid, log)
        assert np.array_equal(rgb.cpu().numpy(), orgb), ("colours", mid, log)

    log = []
    ops = ["decay", "fused", "calls", "lowres", "pair", "clear", "mesh", "reload", "query", "render"]
    weights = np.array([3, 4, 2, 3, 4, 0.6, 1.5, 1, 1, 1], dtype=np.float64)
    for step in range(28):
        op = str(r.choice(ops, p=weights / weights.sum())) if step > 1 else "fused"
        mid = int(r.integers(2))
        log.append((op, mid))
        if op == "decay":  # Mapper.decay() decays every mapper (nvblox_torch: mapper_id = -1)
            gpu.decay()
            orcs[0].decay(), orcs[1].decay()
        elif op in ("fused", "lowres"):
            f, m = frame_inputs()
            T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
            if op == "fused":
                dm, fm = gpu.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), dev(~m), T, K, min_d, k_in, k_depth, border,
                                             mid, invert_input_mask=True)
            else:
                dm, fm = gpu.integrate_frame_lowres(dev(f["depth"]), dev(f["rgb"]), f["low"], dev(m), T, K, min_d, k_in, k_depth, border, mid)
            odm, ofm = oracle_frame(orcs[mid], f, m)
            assert np.array_equal(dm.cpu().numpy().astype(bool), odm) and np.array_equal(fm.cpu().numpy().astype(bool), ofm), log
        elif op == "calls":
            f, m = frame_inputs()
            T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
            odm, ofm = oracle_frame(orcs[mid], f, m)
            gpu.add_depth_frame(dev(f["depth"]), T, K, dev(odm.astype(np.uint8)), mid)
            gpu.add_color_frame(dev(f["rgb"]), T, K, mask_frame=dev(odm.astype(np.uint8)), mapper_id=mid)
            gpu.add_feature_frame(dev(f["features"]), T, K, dev(ofm.astype(np.uint8)), mid)
        elif op == "pair":
            f, m = frame_inputs()
            T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
            jobs = [{"mapper_id": 0, "input_mask": dev(~m), "invert_input_mask": True, "input_mask_erosion_iterations": k_in,
                     "valid_depth_mask_erosion_iterations": k_depth},
                    {"mapper_id": 1, "input_mask": dev(~m), "invert_input_mask": False, "input_mask_erosion_iterations": k_in,
                     "valid_depth_mask_erosion_iterations": k_depth}]
            use_low = bool(r.integers(2))
            gpu.integrate_frame_multi(dev(f["depth"]), dev(f["rgb"]), None if use_low else dev(f["features"]), T, K, min_d, border, jobs,
                                      lowres_features=f["low"] if use_low else None)
            oracle_frame(orcs[0], f, m)
            oracle_frame(orcs[1], f, ~m)
        elif op == "clear":
            gpu.clear(mid)
            orcs[mid].clear()
        elif op == "mesh":
            ov, ofeat = orcs[mid].feature_mesh()
            gpu.update_feature_mesh(mid)
            mesh = gpu.get_feature_mesh(mid)
            gv = mesh.vertices().cpu().numpy()
            assert gv.shape == ov.shape and np.array_equal(gv.view(np.uint32), ov.view(np.uint32)), ("mesh", log)
            assert np.array_equal(mesh.vertex_features().cpu().numpy().view(np.uint16), ofeat.view(np.uint16)), ("mesh features", log)
        elif op == "reload":  # both mappers to disk, a NEW Mapper reads them and carries on
            paths = [str(tmp_path / f"s{seed}_{step}_{j}.nvblx") for j in range(2)]
            for j in range(2):
                gpu.save_map(paths[j], j)
            gpu = H.get_nvblox_mapper(mcfg, feature_channels=C)
            gpu.set_deferred_feature_rows(pipelined)
            for j in range(2):
                gpu.load_from_file(paths[j], j)
                os.remove(paths[j])  # (a campaign of thousands of seeds otherwise fills /tmp: pytest keeps tmp_path until the session ends)
        elif op == "query":
            pts = r.uniform(-0.4, 0.9, size=(257, 3)).astype(np.float32)
            assert np.array_equal(gpu.query_layer(QueryType.TSDF, dev(pts), mid).cpu().numpy(), orcs[mid].query_tsdf(pts)), log
            assert np.array_equal(gpu.query_layer(QueryType.FEATURE, dev(pts), mid).cpu().numpy(), orcs[mid].query_features(pts)), log
        elif op == "render":
            f = S.frame(cfg, int(r.integers(0, 200)), 0)
            so = orcs[mid].render_synthetic_depth(Hh, W, f["T_W_C"], f["K"])
            sg = gpu.render_synthetic_depth(Hh, W, f["T_W_C"], f["K"], mapper_id=mid).cpu().numpy()
            assert np.array_equal(so.view(np.uint32), sg.view(np.uint32)), log
        if step % 5 == 4:
            check(0), check(1)
    check(0), check(1)
    assert orcs[0].num_blocks(0) + orcs[1].num_blocks(0) > 0 or ("clear", 0) in log or ("clear", 1) in log


@pytest.mark.parametrize("seed", range(8))
def test_random_policy_shapes_fused_inference_matches_composite(seed):
    """The policy's inference kernels (whole-layer matrix-core kernels, split cross-attention, paired output stacks, step
    prologue / tail, HIP FPS) against the composite torch ops over drawn shapes: batch 1-3, 1-2 grippers, horizon 1-2, context
    lengths that are not multiples of the 16-token tiles, random vertex padding (one sample fully padded now and then), a few
    denoising steps.  Float-rounding agreement (different summation orders), identical graph replay."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig
    from nvblox_mindmap_amd.training import build_model, synthetic_batch
    from nvblox_mindmap_amd.training.trainer import unpack_batch

    r = np.random.default_rng(7000 + seed)
    B, G, L = int(r.integers(1, 4)), int(r.integers(1, 3)), int(r.integers(1, 3))
    n_vert = int(r.choice([45, 160, 333, 1000, 2048, 3072]))
    cfg = DiffuserActorConfig(data_type="mesh", feature_dim=int(r.choice([32, 64])), diffusion_timesteps=int(r.integers(3, 7)), ngrippers=G,
                              prediction_horizon=L, predict_head_yaw=bool(r.integers(2)), fps_subsampling_factor=int(r.choice([3, 5])))
    print(dict(B=B, G=G, L=L, n_vert=n_vert, steps=cfg.diffusion_timesteps, yaw=cfg.predict_head_yaw, fps=cfg.fps_subsampling_factor))
    torch.manual_seed(seed)
    model = build_model(cfg, device="cuda").eval()
    for p in model.parameters():  # AdaLN / output layers start at zero: perturb so that every path matters
        if p.requires_grad:
            p.data.add_(0.02 * torch.randn_like(p))
    s = unpack_batch(cfg, synthetic_batch(cfg, B, "cuda", num_vertices=n_vert, seed=seed))
    valid = torch.from_numpy(r.uniform(size=(B, n_vert)) > r.uniform(0.0, 0.6)).cuda()
    if B > 1 and r.integers(3) == 0:
        valid[int(r.integers(B))] = False
    valid[0, 0] = True

    def infer():
        torch.manual_seed(11)
        with torch.no_grad():
            return model(None, None, None, None, None, s["vertex_features"], s["vertices"], valid, None, s["gripper_history"], run_inference=True)[:2]

    ref, ref_yaw = infer()
    try:
        DiffuserActor.enable_fused_inference(True)
        fused, fused_yaw = infer()
        model.enable_graph_sampling(True)
        graph, graph_yaw = infer()
    finally:
        DiffuserActor.enable_fused_inference(False)
        model.enable_graph_sampling(False)
    assert torch.isfinite(ref).all() and torch.equal(fused, graph)
    assert torch.allclose(fused, ref, rtol=1e-3, atol=2e-4), float((fused - ref).abs().max())
    if cfg.predict_head_yaw:
        assert torch.equal(fused_yaw, graph_yaw) and torch.allclose(fused_yaw, ref_yaw, rtol=1e-3, atol=2e-4)
