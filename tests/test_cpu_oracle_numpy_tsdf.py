"""An INDEPENDENT restatement of the projective TSDF update -- numpy float32, vectorised over all voxels of all blocks, written from the
spec in the header of oracle/mmf_oracle.c, sharing no code with the C oracle -- must give the C oracle's voxels bit for bit, for a first
frame on an empty map and for a second frame blended into it (with a decay in between), masked and unmasked.

What this anchors: the C oracle is "the spec as written" (operation order, the depth-sampling rule with its nearest / bilinear switch, the
1/d^2 weight, clamping, the weight cap) and not an accident of its own indexing.  What it does not: that the spec is nvblox's (DESIGN.md
section 6: parity unpinned).  Blocks in view are taken from the oracle (their own anchor: tests/test_cpu_raycast_walk.py)."""
import numpy as np
import pytest

from fusion_common import FMA, REF_PARAMS, make_oracle, small_cfg
from nvblox_mindmap_amd import synthetic as S

# (the restatements below are of the spec's DEFAULT arithmetic: under MMF_FMA_CONTRACTION=1 every oracle the tests build contracts its
# multiply-adds and numpy, which cannot, has nothing to say about it)
pytestmark = pytest.mark.skipif(FMA, reason="numpy restates the uncontracted arithmetic; MMF_FMA_CONTRACTION=1 builds contracting oracles")

F = np.float32


def rigid_inverse(T):
    """T_C_L from T_L_C with the oracle's operation order (R^T; t' = -((r0 t0 + r1 t1) + r2 t2))."""
    R = T[:3, :3].astype(F)
    t = T[:3, 3].astype(F)
    Ri = R.T.copy()
    ti = np.array([-F(F(F(Ri[i, 0] * t[0]) + F(Ri[i, 1] * t[1])) + F(Ri[i, 2] * t[2])) for i in range(3)], dtype=F)
    return Ri, ti


def tsdf_update_numpy(idx, D, W, depth, mask, T_W_C, K, voxel, trunc_vox=4.0, max_w=5.0, max_dist=5.0, lin_vox=2.0):
    """D, W: [n, 512] float32 (updated in place); idx: [n, 3] block indices."""
    H, Wd = depth.shape
    v = F(voxel)
    bs = F(F(8.0) * v)
    trunc = F(F(trunc_vox) * v)
    md = F(F(lin_vox) * v)
    R, t = rigid_inverse(T_W_C)
    fx, fy, cx, cy = F(K[0, 0]), F(K[1, 1]), F(K[0, 2]), F(K[1, 2])
    lin = np.arange(512)
    vx, vy, vz = (lin >> 6).astype(F), ((lin >> 3) & 7).astype(F), (lin & 7).astype(F)
    b = idx.astype(F)
    c0 = b[:, 0:1] * bs + (vx + F(0.5))[None, :] * v
    c1 = b[:, 1:2] * bs + (vy + F(0.5))[None, :] * v
    c2 = b[:, 2:3] * bs + (vz + F(0.5))[None, :] * v
    p = [((R[i, 0] * c0 + R[i, 1] * c1) + R[i, 2] * c2) + t[i] for i in range(3)]
    ok = p[2] > F(1e-6)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iz = F(1.0) / p[2]
        u = fx * (p[0] * iz) + cx
        w = fy * (p[1] * iz) + cy
    ok &= ~((u < 0) | (w < 0) | (u > F(Wd)) | (w > F(H)))
    if max_dist > 0:
        ok &= ~(p[2] > F(max_dist))
    u = np.where(ok, u, F(0.0))
    w = np.where(ok, w, F(0.0))

    def tap(x, y):
        d = depth[y, x]
        valid = d > 0
        if mask is not None:
            valid &= mask[y, x] != 0
        return d, valid

    xn = np.minimum(np.floor(u).astype(np.int64), Wd - 1)
    yn = np.minimum(np.floor(w).astype(np.int64), H - 1)
    dn, vn = tap(xn, yn)
    uc, wc = u - F(0.5), w - F(0.5)
    fx0, fy0 = np.floor(uc), np.floor(wc)
    ix, iy = fx0.astype(np.int64), fy0.astype(np.int64)
    inside = ~((ix < 0) | (iy < 0) | (ix + 1 > Wd - 1) | (iy + 1 > H - 1))
    ixc, iyc = np.clip(ix, 0, Wd - 2), np.clip(iy, 0, H - 2)
    wx, wy = uc - fx0, wc - fy0
    a00, v00 = tap(ixc, iyc)
    a10, v10 = tap(ixc + 1, iyc)
    a01, v01 = tap(ixc, iyc + 1)
    a11, v11 = tap(ixc + 1, iyc + 1)
    lin_ok = inside & v00 & v10 & v01 & v11
    if md > 0:
        lin_ok &= ~((np.abs(a00 - dn) > md) | (np.abs(a10 - dn) > md) | (np.abs(a01 - dn) > md) | (np.abs(a11 - dn) > md))
    top = (F(1.0) - wx) * a00 + wx * a10
    bot = (F(1.0) - wx) * a01 + wx * a11
    d = np.where(lin_ok, (F(1.0) - wy) * top + wy * bot, dn).astype(F)
    ok &= vn
    sdf = d - p[2]
    ok &= ~(sdf < -trunc)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        wm = F(1.0) / (d * d)
        ok &= wm > 0
        Dn = (sdf * wm + D * W) / (wm + W)
    Dn = np.where(Dn > 0, np.minimum(trunc, Dn), np.maximum(-trunc, Dn)).astype(F)
    Wn = np.minimum(W + wm, F(max_w)).astype(F)
    D[ok] = Dn[ok]
    W[ok] = Wn[ok]
    return int(ok.sum())


@pytest.mark.parametrize("masked", [False, True])
def test_numpy_restatement_equals_the_c_oracle_bit_for_bit(oracle_mod, masked):
    cfg = small_cfg(4)
    orc = make_oracle(oracle_mod, 8)
    D = W = idx = None
    for k, i in enumerate((0, 7)):
        f = S.frame(cfg, i, 0)
        mask = None
        if masked:
            mask = np.ones(f["depth"].shape, dtype=np.uint8)
            mask[10 + k:40 + k, 20:70] = 0
        if k:
            orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], mask)
        new_idx = orc.block_indices(0)
        n = new_idx.shape[0]
        if idx is None:
            idx, D, W = new_idx, np.zeros((n, 512), F), np.zeros((n, 512), F)
        else:
            # decay: W <- W * factor on every voxel of every block; a block whose voxels are ALL below the threshold (e.g. one that was
            # in view but never received a measurement) is deallocated, the others keep their order; this frame's new blocks are appended
            W *= F(REF_PARAMS["tsdf_decay_factor"])
            keep = ~np.all(W < F(1e-3), axis=1)
            prev = {tuple(r): j for j, r in enumerate(idx.tolist()) if keep[j]}
            assert [tuple(r) for r in new_idx.tolist()][: len(prev)] == [tuple(r) for j, r in enumerate(idx.tolist()) if keep[j]]
            D2, W2 = np.zeros((n, 512), F), np.zeros((n, 512), F)
            for j2, r in enumerate(new_idx.tolist()[: len(prev)]):
                D2[j2], W2[j2] = D[prev[tuple(r)]], W[prev[tuple(r)]]
            idx, D, W = new_idx, D2, W2
        # the update touches the frame's blocks in view only; voxels of other blocks fail the projection or the depth test anyway?  No:
        # a live block outside this frame's view set is NOT integrated by the spec -- restrict to the oracle's view set
        view = orc.last_view_blocks()
        in_view = {tuple(r) for r in view.tolist()}
        sel = np.array([tuple(r) in in_view for r in idx.tolist()])
        Ds, Ws = D[sel], W[sel]
        n_upd = tsdf_update_numpy(idx[sel], Ds, Ws, f["depth"], mask, f["T_W_C"], f["K"], REF_PARAMS["voxel_size"],
                                  max_dist=REF_PARAMS["max_integration_distance_m"])
        D[sel], W[sel] = Ds, Ws
        assert n_upd > 20000
        got = orc.all_tsdf().reshape(n, 512, 2)
        assert np.array_equal(got[..., 0].view(np.uint32), D.view(np.uint32)), f"distances differ after frame {k}"
        assert np.array_equal(got[..., 1].view(np.uint32), W.view(np.uint32)), f"weights differ after frame {k}"


def _project(idx, T_W_C, K, voxel, H, Wd, max_dist):
    v = F(voxel)
    bs = F(F(8.0) * v)
    R, t = rigid_inverse(T_W_C)
    fx, fy, cx, cy = F(K[0, 0]), F(K[1, 1]), F(K[0, 2]), F(K[1, 2])
    lin = np.arange(512)
    vx, vy, vz = (lin >> 6).astype(F), ((lin >> 3) & 7).astype(F), (lin & 7).astype(F)
    b = idx.astype(F)
    c0 = b[:, 0:1] * bs + (vx + F(0.5))[None, :] * v
    c1 = b[:, 1:2] * bs + (vy + F(0.5))[None, :] * v
    c2 = b[:, 2:3] * bs + (vz + F(0.5))[None, :] * v
    p = [((R[i, 0] * c0 + R[i, 1] * c1) + R[i, 2] * c2) + t[i] for i in range(3)]
    ok = p[2] > F(1e-6)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iz = F(1.0) / p[2]
        u = fx * (p[0] * iz) + cx
        w = fy * (p[1] * iz) + cy
    ok &= ~((u < 0) | (w < 0) | (u > F(Wd)) | (w > F(H)))
    if max_dist > 0:
        ok &= ~(p[2] > F(max_dist))
    return np.where(ok, u, F(0)), np.where(ok, w, F(0)), p[2], ok


def _footprint(u, w, Wd, H):
    uc, wc = u - F(0.5), w - F(0.5)
    fx0, fy0 = np.floor(uc), np.floor(wc)
    ix, iy = fx0.astype(np.int64), fy0.astype(np.int64)
    inside = ~((ix < 0) | (iy < 0) | (ix + 1 > Wd - 1) | (iy + 1 > H - 1))
    return np.clip(ix, 0, Wd - 2), np.clip(iy, 0, H - 2), uc - fx0, wc - fy0, inside


def _bilin(a00, a10, a01, a11, wx, wy):
    top = (F(1.0) - wx) * a00 + wx * a10
    bot = (F(1.0) - wx) * a01 + wx * a11
    return ((F(1.0) - wy) * top + wy * bot).astype(F)


def test_numpy_restatement_of_the_feature_update(oracle_mod):
    """The appearance half restated the same way (candidate blocks, the per-voxel gate against the oracle's own sphere-traced depth image,
    the f16 blend): block list, weights and feature values of the C oracle bit for bit over two frames."""
    C = 8
    cfg = small_cfg(4)
    orc = make_oracle(oracle_mod, C)
    trunc = F(F(4.0) * F(REF_PARAMS["voxel_size"]))
    wm, max_w, sf = F(REF_PARAMS["appearance_measurement_weight"]), F(5.0), F(4.0)
    feat_idx, A, Wf = np.zeros((0, 3), np.int32), np.zeros((0, 512, C), np.float16), np.zeros((0, 512), F)
    for k, i in enumerate((0, 6)):
        f = S.frame(cfg, i, C)
        mask = np.ones(f["depth"].shape, dtype=np.uint8)
        mask[5:20, 100:140] = 0
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], None)
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], mask)
        H, Wd = f["depth"].shape
        tsdf_idx = orc.block_indices(0)
        tsdf = orc.all_tsdf().reshape(-1, 512, 2)
        synth = orc.synthetic_depth()
        Hs, Ws = synth.shape
        u, w, z, proj = _project(tsdf_idx, f["T_W_C"], f["K"], REF_PARAMS["voxel_size"], H, Wd, REF_PARAMS["max_integration_distance_m"])
        near = (tsdf[..., 1] > 0) & (np.abs(tsdf[..., 0]) < trunc)
        cand = np.any(near & proj, axis=1)
        cand_idx = tsdf_idx[cand]
        # the feature layer allocates the candidates in TSDF live order; blocks it already holds keep their place
        have = {tuple(r): j for j, r in enumerate(feat_idx.tolist())}
        new = [r for r in cand_idx.tolist() if tuple(r) not in have]
        feat_idx = np.concatenate([feat_idx, np.array(new, np.int32).reshape(-1, 3)])
        A = np.concatenate([A, np.zeros((len(new), 512, C), np.float16)])
        Wf = np.concatenate([Wf, np.zeros((len(new), 512), F)])
        assert np.array_equal(orc.block_indices(2), feat_idx), "feature block list / order"
        rows = np.array([{tuple(r): j for j, r in enumerate(feat_idx.tolist())}[tuple(r)] for r in cand_idx.tolist()])
        uu, ww, zz, ok = u[cand], w[cand], z[cand], proj[cand].copy()
        sx, sy, swx, swy, s_in = _footprint(uu / sf, ww / sf, Ws, Hs)
        ok &= s_in
        s00, s10, s01, s11 = synth[sy, sx], synth[sy, sx + 1], synth[sy + 1, sx], synth[sy + 1, sx + 1]
        ok &= (s00 > 0) & (s10 > 0) & (s01 > 0) & (s11 > 0)
        ok &= ~(np.abs(_bilin(s00, s10, s01, s11, swx, swy) - zz) > trunc)
        x0, y0, wx, wy, f_in = _footprint(uu, ww, Wd, H)
        ok &= f_in
        ok &= (mask[y0, x0] != 0) & (mask[y0, x0 + 1] != 0) & (mask[y0 + 1, x0] != 0) & (mask[y0 + 1, x0 + 1] != 0)
        feat = f["features"]  # [H, W, C] float16
        t00, t10 = feat[y0, x0].astype(F), feat[y0, x0 + 1].astype(F)
        t01, t11 = feat[y0 + 1, x0].astype(F), feat[y0 + 1, x0 + 1].astype(F)
        a = _bilin(t00, t10, t01, t11, wx[..., None], wy[..., None])
        Wv = Wf[rows]
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = F(1.0) / (Wv + wm)
        An = ((A[rows].astype(F) * Wv[..., None] + a * wm) * inv[..., None]).astype(np.float16)
        Ar, Wr = A[rows], Wf[rows]
        Ar[ok] = An[ok]
        Wr[ok] = np.minimum(Wv + wm, max_w)[ok]
        A[rows], Wf[rows] = Ar, Wr
        assert int(ok.sum()) > 3000
        of, ow = orc.all_features()
        assert np.array_equal(ow.reshape(-1, 512).view(np.uint32), Wf.view(np.uint32)), f"feature weights differ after frame {k}"
        assert np.array_equal(of.reshape(-1, 512, C).view(np.uint16), A.view(np.uint16)), f"feature values differ after frame {k}"


def test_numpy_restatement_of_the_sphere_traced_depth(oracle_mod):
    """The synthetic depth image of the appearance gate (sphere tracing through the TSDF, one ray per 4 x 4 pixels): all rays marched in
    lockstep in numpy float32 -- same samples, same steps, same termination rules as the spec -- against the C oracle's image, bit for bit."""
    cfg = small_cfg(2)
    orc = make_oracle(oracle_mod, 8)
    for i in (0, 5, 11):
        f = S.frame(cfg, i, 0)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], None)
    f = S.frame(cfg, 14, 0)
    H, Wd = f["depth"].shape
    want = orc.render_synthetic_depth(H, Wd, f["T_W_C"], f["K"])
    idx = orc.block_indices(0)
    tsdf = orc.all_tsdf().reshape(-1, 512, 2)
    lo = idx.min(axis=0)
    ext = idx.max(axis=0) - lo + 1
    table = -np.ones(tuple(ext), dtype=np.int64)
    table[tuple((idx - lo).T)] = np.arange(idx.shape[0])
    v = F(REF_PARAMS["voxel_size"])
    bs = F(F(8.0) * v)
    inv_bs, inv_v = F(F(1.0) / bs), F(F(1.0) / v)
    trunc, eps = F(F(4.0) * v), F(F(0.1) * v)
    sf, max_steps, max_len = 4, 100, F(15.0)
    Ws, Hs = Wd // sf, H // sf
    K, T = f["K"].astype(F), f["T_W_C"].astype(F)
    cs, rs = np.meshgrid(np.arange(Ws, dtype=F), np.arange(Hs, dtype=F))
    u, w = (cs + F(0.5)) * F(sf), (rs + F(0.5)) * F(sf)
    x, y = (u - K[0, 2]) / K[0, 0], (w - K[1, 2]) / K[1, 1]
    n = np.sqrt((x * x + y * y) + F(1.0)).astype(F)
    dC = [x / n, y / n, F(1.0) / n]
    R = T[:3, :3]
    dL = [(R[i, 0] * dC[0] + R[i, 1] * dC[1]) + R[i, 2] * dC[2] for i in range(3)]
    o = T[:3, 3]
    t = np.zeros((Hs, Ws), F)
    last_pos = np.zeros((Hs, Ws), bool)
    alive = np.ones((Hs, Ws), bool)
    success = np.zeros((Hs, Ws), bool)
    for step in range(max_steps):
        alive &= t < max_len
        if not alive.any():
            break
        p = [o[i] + t * dL[i] for i in range(3)]
        b = [np.floor(p[i] * inv_bs).astype(np.int64) for i in range(3)]
        q = [np.clip(np.floor((p[i] - b[i].astype(F) * bs) * inv_v).astype(np.int64), 0, 7) for i in range(3)]
        rel = [b[i] - lo[i] for i in range(3)]
        inside = np.all([(rel[i] >= 0) & (rel[i] < ext[i]) for i in range(3)], axis=0)
        row = np.where(inside, table[tuple(np.clip(rel[i], 0, ext[i] - 1) for i in range(3))], -1)
        lin = (q[0] * 8 + q[1]) * 8 + q[2]
        Wv = np.where(row >= 0, tsdf[np.maximum(row, 0), lin, 1], F(0))
        Dv = np.where(row >= 0, tsdf[np.maximum(row, 0), lin, 0], F(0))
        valid = (row >= 0) & (Wv > F(1e-4))
        fail = alive & ((~valid & last_pos) | (valid & (Dv < eps) & ~last_pos))
        hit = alive & valid & (Dv < eps) & last_pos
        t = np.where(hit, t + Dv, t).astype(F)
        success |= hit
        alive &= ~(fail | hit)
        stepv = np.where(valid, Dv, trunc).astype(F)
        t = np.where(alive, t + stepv, t).astype(F)
        last_pos |= alive & valid
    got = np.where(success, t * dC[2], F(-1.0)).astype(F)
    assert want.shape == got.shape and (want > 0).mean() > 0.3
    assert np.array_equal(want.view(np.uint32), got.view(np.uint32))


def test_numpy_restatement_of_the_colour_update(oracle_mod):
    """Same gate, the colour blend: A' = floor((A W + a w) / (W + w) + 0.5) per channel as uint8, W' = min(W + w, max) -- two frames."""
    cfg = small_cfg(4)
    orc = make_oracle(oracle_mod, 8)
    trunc = F(F(4.0) * F(REF_PARAMS["voxel_size"]))
    wm, max_w, sf = F(REF_PARAMS["appearance_measurement_weight"]), F(5.0), F(4.0)
    col_idx, A, Wc = np.zeros((0, 3), np.int32), np.zeros((0, 512, 3), np.uint8), np.zeros((0, 512), F)
    for k, i in enumerate((0, 6)):
        f = S.frame(cfg, i, 0)
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], None)
        orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], None)
        H, Wd = f["depth"].shape
        tsdf_idx = orc.block_indices(0)
        tsdf = orc.all_tsdf().reshape(-1, 512, 2)
        synth = orc.synthetic_depth()
        Hs, Ws = synth.shape
        u, w, z, proj = _project(tsdf_idx, f["T_W_C"], f["K"], REF_PARAMS["voxel_size"], H, Wd, REF_PARAMS["max_integration_distance_m"])
        cand = np.any((tsdf[..., 1] > 0) & (np.abs(tsdf[..., 0]) < trunc) & proj, axis=1)
        cand_idx = tsdf_idx[cand]
        have = {tuple(r) for r in col_idx.tolist()}
        new = [r for r in cand_idx.tolist() if tuple(r) not in have]
        col_idx = np.concatenate([col_idx, np.array(new, np.int32).reshape(-1, 3)])
        A = np.concatenate([A, np.zeros((len(new), 512, 3), np.uint8)])
        Wc = np.concatenate([Wc, np.zeros((len(new), 512), F)])
        assert np.array_equal(orc.block_indices(1), col_idx)
        where = {tuple(r): j for j, r in enumerate(col_idx.tolist())}
        rows = np.array([where[tuple(r)] for r in cand_idx.tolist()])
        uu, ww, zz, ok = u[cand], w[cand], z[cand], proj[cand].copy()
        sx, sy, swx, swy, s_in = _footprint(uu / sf, ww / sf, Ws, Hs)
        ok &= s_in
        s00, s10, s01, s11 = synth[sy, sx], synth[sy, sx + 1], synth[sy + 1, sx], synth[sy + 1, sx + 1]
        ok &= (s00 > 0) & (s10 > 0) & (s01 > 0) & (s11 > 0)
        ok &= ~(np.abs(_bilin(s00, s10, s01, s11, swx, swy) - zz) > trunc)
        x0, y0, wx, wy, f_in = _footprint(uu, ww, Wd, H)
        ok &= f_in
        rgb = f["rgb"].astype(F)
        a = _bilin(rgb[y0, x0], rgb[y0, x0 + 1], rgb[y0 + 1, x0], rgb[y0 + 1, x0 + 1], wx[..., None], wy[..., None])
        Wv = Wc[rows]
        inv = F(1.0) / (Wv + wm)
        An = np.floor((A[rows].astype(F) * Wv[..., None] + a * wm) * inv[..., None] + F(0.5)).astype(np.uint8)
        Ar, Wr = A[rows], Wc[rows]
        Ar[ok] = An[ok]
        Wr[ok] = np.minimum(Wv + wm, max_w)[ok]
        A[rows], Wc[rows] = Ar, Wr
        orgb, ow = orc.all_colors()
        assert np.array_equal(ow.reshape(-1, 512).view(np.uint32), Wc.view(np.uint32)) and np.array_equal(orgb.reshape(-1, 512, 3), A), k
    assert int((Wc > 0).sum()) > 3000
