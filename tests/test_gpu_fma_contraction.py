"""The 17th switchable spec item: ``fma_contraction`` (include/mmfusion.h mmf_params, oracle/mmf_oracle.c orc_params).

The library and the oracle are built with -ffp-contract=off; nvcc contracts by default, so CUDA nvblox's voxels almost certainly
hold fused multiply-adds (the pin kit could not attribute such a difference to anything before this item existed).  With the
switch on, both sides contract the same expressions the same way -- and must still agree BIT FOR BIT, on every route the frames of
such a mapper take: the fused frame of a bounded workspace (un-merged launches), the hash path with lazy decay, the stand-alone
calls, the low-res feature source (taps computed in the row kernel == the materialised image of mmf_upsample_features_spec == the
oracle's C restatement).  The whole parity / fuzz / soak suites run under the switch with MMF_FMA_CONTRACTION=1 (tests/fusion_common.py)."""
import numpy as np
import pytest
import torch

import fusion_common
from fusion_common import make_mapper, make_oracle, small_cfg
from nvblox_mindmap_amd import synthetic as S
from test_gpu_fusion_parity import _fused_vs_oracle, _lowres_map, compare_features, compare_tsdf, dev, run_both

pytestmark = pytest.mark.gpu
FMA = dict(fma_contraction=1)


def test_fused_frames_bounded_workspace(oracle_mod):
    cfg = small_cfg(4)
    gpu, orc = make_mapper(16, **FMA), make_oracle(oracle_mod, 16, **FMA)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 6, 40, 46, 90], 16)
    if fusion_common.FMA:
        return  # (MMF_FMA_CONTRACTION=1: "the default" is contracted too)
    # the flip is not a no-op: the default arithmetic gives another map (same blocks, values a few ulp apart)
    ref = make_oracle(oracle_mod, 16)
    _fused_vs_oracle(oracle_mod, make_mapper(16), ref, cfg, [0, 6, 40, 46, 90], 16)
    assert np.array_equal(ref.block_indices(0), orc.block_indices(0))
    a, b = ref.all_tsdf(), orc.all_tsdf()
    assert not np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # all but a handful of voxels stay inside the north star's 1e-5 across the flip (a last-bit change of a projection can move a
    # voxel's depth sample to the neighbouring pixel: those few differ by a real amount)
    d = np.abs(a - b)[..., 0]
    assert float((d > 1e-5).mean()) < 1e-3 and float(np.median(d[d > 0])) < 1e-6


def test_fused_frames_with_the_pipelined_mode_requested(oracle_mod):
    """set_deferred_feature_rows on a contracted mapper: its frames are complete when the call returns (nothing is deferred), the
    map is the oracle's."""
    cfg = small_cfg(4)
    gpu, orc = make_mapper(16, **FMA), make_oracle(oracle_mod, 16, **FMA)
    gpu.set_deferred_feature_rows(True)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 6, 40], 16)


def test_stand_alone_calls(oracle_mod):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 16, [0, 5, 30], **FMA)
    compare_tsdf(orc, gpu)
    compare_features(orc, gpu)


def test_hash_path_with_lazy_decay(oracle_mod):
    """Unbounded workspace at full size: k_alloc_big + k_tsdf_classify + k_tsdf_pass<LAZY, FMA>."""
    from test_gpu_hash_path import fused_frame

    cfg = S.StreamConfig(hole_mode="patches")
    over = dict(workspace_bounds_type=0, **FMA)
    gpu, orc = make_mapper(64, **over), make_oracle(oracle_mod, 64, **over)
    for i in (0, 7, 14, 21):
        fused_frame(gpu, orc, cfg, i, 64, 17, 20, 5)
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)


@pytest.mark.parametrize("fma", [False, True])
@pytest.mark.parametrize("cin,cpad,lh,lw", [(16, 16, 16, 16), (24, 32, 5, 7), (13, 16, 9, 4)])
def test_upsample_equals_the_c_restatement(oracle_mod, fma, cin, cpad, lh, lw):
    from nvblox_mindmap_amd.image_processing import upsample_features

    low = _lowres_map(3, cin, lh, lw)  # [lh, lw, cin]
    up = upsample_features(dev(np.ascontiguousarray(low.transpose(2, 0, 1))), (120, 160), cpad, fma_contraction=fma)
    want = oracle_mod.upsample_features(low, 120, 160, cpad, fma_contraction=fma)
    assert np.array_equal(up.cpu().numpy().view(np.uint16), want)
    other = oracle_mod.upsample_features(low, 120, 160, cpad, fma_contraction=not fma)
    assert not np.array_equal(want, other)  # (the contraction changes last bits of the image)
    d = np.abs(want.view(np.float16).astype(np.float32) - other.view(np.float16).astype(np.float32))
    assert float(d.max()) <= 2.0 ** -8  # one f16 ulp at magnitude < 4


@pytest.mark.parametrize("scale,cin,channels,lh,lw", [(4, 16, 16, 16, 16), (2, 64, 64, 16, 16)])
def test_lowres_source_is_upsample_plus_add(oracle_mod, scale, cin, channels, lh, lw):
    from nvblox_mindmap_amd.image_processing import upsample_features

    cfg = small_cfg(scale)
    fused, two_step, orc = make_mapper(channels, **FMA), make_mapper(channels, **FMA), make_oracle(oracle_mod, channels, **FMA)
    for i in [0, 4, 11]:
        f = S.frame(cfg, i, 0)
        low = _lowres_map(i, cin, lh, lw)
        T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
        up = upsample_features(dev(np.ascontiguousarray(low.transpose(2, 0, 1))), (cfg.height, cfg.width), channels, fma_contraction=True)
        mask = np.ones((cfg.height, cfg.width), dtype=np.uint8)
        for m in (fused, two_step):
            m.add_depth_frame(dev(f["depth"]), T, K, None, 0)
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], None)
        fused.add_feature_frame_lowres(dev(low), (cfg.height, cfg.width), T, K, dev(mask), 0)
        two_step.add_feature_frame(up, T, K, dev(mask), 0)
        orc.add_feature_frame(up.cpu().numpy(), f["T_W_C"], f["K"], mask)
    fa, wa, ia = fused.feature_layer_view(0).get_all_blocks_split()
    fb, wb, ib = two_step.feature_layer_view(0).get_all_blocks_split()
    assert torch.equal(ia, ib) and torch.equal(wa, wb) and torch.equal(fa.view(torch.int16), fb.view(torch.int16))
    assert int((wa > 0).sum()) > 500
    compare_features(orc, fused)


def test_both_arrangement_switches_together(oracle_mod):
    """fma_contraction with appearance_blend_division (k_color_integrate<DIV, FMA>, k_feature_integrate<*, DIV, FMA>)."""
    cfg = small_cfg(4)
    over = dict(appearance_blend_division=1, **FMA)
    _fused_vs_oracle(oracle_mod, make_mapper(16, **over), make_oracle(oracle_mod, 16, **over), cfg, [0, 6, 40], 16)
