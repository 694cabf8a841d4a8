#!/usr/bin/env python3
"""Generates tests/golden/*.npz by IMPORTING the reference's Python (works only in the authoring
container, where /root/reference exists; the fixtures -- inputs and expected outputs -- are committed,
this script is committed so they can be regenerated; nothing here ships to the GPU box at run time).

Reference functions exercised (all CPU-runnable):
  mindmap.image_processing.backprojection.{pose_to_homo, backproject_depth_to_pointcloud, get_camera_pointcloud}
  mindmap.image_processing.image_mask_operations.{erode_mask, get_border_mask, downscale_mask}
  mindmap.data_loading.vertex_sampling.sample_to_n_vertices
  mindmap.geometry.utils.quaternion_to_matrix
  torch.nn.functional.interpolate as called by feature_extraction.scale_image (:126-128)
`transforms3d` is absent from the image: an import stub provides quat2mat with transforms3d's published
formula (cross-checked against scipy below).  integrate_frame itself cannot be imported (nvblox_torch is
missing), so its mask algebra (nvblox_mapping_helpers.py:201-253) is composed here from the imported
reference primitives, line by line.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_transforms3d_stub():
    def quat2mat(q):
        w, x, y, z = [float(v) for v in q]
        Nq = w * w + x * x + y * y + z * z
        if Nq < np.finfo(np.float64).eps:
            return np.eye(3)
        s = 2.0 / Nq
        X, Y, Z = x * s, y * s, z * s
        wX, wY, wZ = w * X, w * Y, w * Z
        xX, xY, xZ = x * X, x * Y, x * Z
        yY, yZ, zZ = y * Y, y * Z, z * Z
        return np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])

    t3d = types.ModuleType("transforms3d")
    quats = types.ModuleType("transforms3d.quaternions")
    quats.quat2mat = quat2mat
    t3d.quaternions = quats
    sys.modules["transforms3d"] = t3d
    sys.modules["transforms3d.quaternions"] = quats
    # cross-check the stub against scipy (xyzw order there)
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        assert np.allclose(quat2mat(q), Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix(), atol=1e-12)


def main():
    assert os.path.isdir(REF), "the reference is only available in the authoring container"
    _install_transforms3d_stub()
    sys.path.insert(0, REF)
    from mindmap.data_loading.vertex_sampling import VertexSamplingMethod, sample_to_n_vertices
    from mindmap.image_processing.backprojection import get_camera_pointcloud, pose_to_homo
    from mindmap.image_processing.image_mask_operations import downscale_mask, erode_mask, get_border_mask

    torch.set_num_threads(1)
    rng = np.random.default_rng(1234)

    # ---------------- back-projection ----------------
    bp = {}
    for name, (B, H, W) in {"a": (2, 48, 64), "b": (1, 30, 41), "c": (3, 17, 20)}.items():
        depth = rng.uniform(0.2, 3.0, size=(B, H, W)).astype(np.float32)
        depth[:, 3, 5] = 0.0
        depth[0, 7, 7] = np.nan
        depth[0, 8, 9] = np.inf
        K = np.tile(np.array([[W * 0.9, 0.0, W / 2.0], [0.0, W * 0.92, H / 2.0], [0.0, 0.0, 1.0]], dtype=np.float32), (B, 1, 1))
        K[:, 0, 2] += rng.uniform(-1, 1, size=B).astype(np.float32)
        pos = rng.uniform(-1, 1, size=(B, 3)).astype(np.float32)
        quat = rng.standard_normal((B, 4)).astype(np.float32)
        quat /= np.linalg.norm(quat, axis=1, keepdims=True)
        out = get_camera_pointcloud(torch.from_numpy(K), torch.from_numpy(depth), torch.from_numpy(pos), torch.from_numpy(quat))
        homo = pose_to_homo(torch.from_numpy(np.concatenate([pos, quat], axis=1)))
        bp.update({f"{name}_depth": depth, f"{name}_K": K, f"{name}_pos": pos, f"{name}_quat": quat,
                   f"{name}_out": out.numpy(), f"{name}_homo": homo.numpy()})
    np.savez_compressed(os.path.join(HERE, "backprojection.npz"), **bp)

    # ---------------- masks ----------------
    mk = {}
    H, W = 48, 64
    rnd = rng.uniform(size=(H, W)) > 0.03
    struct = np.ones((H, W), dtype=bool)
    struct[10:14, 20:30] = False
    struct[0, 0] = False
    struct[H - 1, W // 2] = False
    struct[30, W - 1] = False
    mk["random"] = np.packbits(rnd)
    mk["struct"] = np.packbits(struct)
    mk["shape"] = np.array([H, W])
    for k in (1, 3, 10, 17, 20):
        for nm, m in (("random", rnd), ("struct", struct)):
            mk[f"erode_{nm}_{k}"] = np.packbits(erode_mask(torch.from_numpy(m), iterations=k).numpy())
    for (h, w, pct) in ((48, 64, 5), (512, 512, 5), (480, 640, 5), (10, 12, 5), (100, 30, 7)):
        bm, bh, bw = get_border_mask((h, w), pct, "cpu")
        mk[f"border_{h}_{w}_{pct}"] = np.packbits(bm.numpy())
        mk[f"border_{h}_{w}_{pct}_hw"] = np.array([bh, bw])
    dm_in = rng.uniform(size=(2, 1, 16, 24)) > 0.2
    mk["downscale_in"] = np.packbits(dm_in)
    mk["downscale_out_2"] = np.packbits(downscale_mask(torch.from_numpy(dm_in), 2).numpy())
    mk["downscale_out_4"] = np.packbits(downscale_mask(torch.from_numpy(dm_in), 4).numpy())

    # integrate_frame's mask algebra (nvblox_mapping_helpers.py:201-253), composed from reference primitives
    def ref_frame_masks(input_mask, depth, min_d, k_in, k_depth, border_pct, hf, wf):
        input_mask = torch.from_numpy(input_mask)
        depth = torch.from_numpy(depth)
        valid_depth_mask = depth > min_d
        depth_mask = torch.logical_and(input_mask, valid_depth_mask)
        input_mask_eroded = erode_mask(input_mask, iterations=k_in)
        valid_depth_mask_eroded = erode_mask(valid_depth_mask, iterations=k_depth)
        depth_mask_eroded = torch.logical_and(input_mask_eroded, valid_depth_mask_eroded)
        up = F.interpolate(depth_mask_eroded.unsqueeze(0).unsqueeze(0).to(torch.uint8), size=(hf, wf), mode="nearest").squeeze(0).squeeze(0).to(torch.bool)
        border_mask = get_border_mask((hf, wf), border_pct, "cpu")[0]
        return depth_mask.numpy(), torch.logical_and(border_mask, up).to(torch.uint8).numpy()

    cases = [("same", 48, 64, 48, 64, 2, 3, 5), ("up", 48, 64, 96, 128, 3, 2, 5), ("down", 48, 64, 24, 32, 1, 4, 5),
             ("odd", 48, 64, 50, 70, 2, 2, 5), ("sq", 64, 64, 64, 64, 17, 20, 5)]
    for nm, h, w, hf, wf, k_in, k_depth, pct in cases:
        im = rng.uniform(size=(h, w)) > 0.01
        im[5:9, 5:15] = False
        d = rng.uniform(0.1, 2.0, size=(h, w)).astype(np.float32)
        d[20:23, 30:40] = 0.0
        d[rng.uniform(size=(h, w)) > 0.995] = 0.05
        dm, fm = ref_frame_masks(im, d, 0.30, k_in, k_depth, pct, hf, wf)
        mk[f"fm_{nm}_in"] = np.packbits(im)
        mk[f"fm_{nm}_depth"] = d
        mk[f"fm_{nm}_params"] = np.array([h, w, hf, wf, k_in, k_depth, pct])
        mk[f"fm_{nm}_depth_mask"] = np.packbits(dm)
        mk[f"fm_{nm}_feature_mask"] = np.packbits(fm.astype(bool))
    np.savez_compressed(os.path.join(HERE, "masks.npz"), **mk)

    # ---------------- vertex sampling ----------------
    vs = {}
    for nm, (V, N) in {"down": (300, 64), "equal": (64, 64), "pad": (20, 64)}.items():
        verts = rng.standard_normal((V, 3)).astype(np.float32)
        feats = rng.standard_normal((V, 5)).astype(np.float32)
        vs[f"{nm}_verts"], vs[f"{nm}_feats"], vs[f"{nm}_N"] = verts, feats, np.array(N)
        for method in (VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, VertexSamplingMethod.RANDOM_WITH_REPLACEMENT,
                       VertexSamplingMethod.LOWEST, VertexSamplingMethod.NONE):
            v, f, m = sample_to_n_vertices(torch.from_numpy(verts), torch.from_numpy(feats), N, method, seed=7)
            vs[f"{nm}_{method.value}_v"], vs[f"{nm}_{method.value}_f"], vs[f"{nm}_{method.value}_m"] = v.numpy(), f.numpy(), m.numpy()
    np.savez_compressed(os.path.join(HERE, "vertex_sampling.npz"), **vs)

    # ---------------- feature upsample (scale_image + rearrange + zero pad + f16 cast) ----------------
    up = {}
    for nm, (c, h, w, hf, wf, cpad) in {"a": (5, 4, 4, 12, 12, 8), "b": (12, 16, 16, 40, 56, 16), "c": (3, 8, 8, 8, 8, 8),
                                        "d": (24, 6, 9, 15, 20, 24)}.items():
        low = rng.standard_normal((1, c, h, w)).astype(np.float32)
        t = F.interpolate(torch.from_numpy(low), size=(hf, wf), mode="bilinear", align_corners=False)
        t = t.permute(0, 2, 3, 1)
        t = torch.cat((t, torch.zeros(1, hf, wf, cpad - c)), dim=3)
        up[f"{nm}_low"] = low[0]
        up[f"{nm}_out"] = t[0].contiguous().to(torch.float16).numpy()
        up[f"{nm}_out_f32"] = t[0].contiguous().numpy()
    np.savez_compressed(os.path.join(HERE, "feature_upsample.npz"), **up)

    # ---------------- quaternion -> matrix (mindmap.geometry.utils, wxyz) ----------------
    from mindmap.geometry.utils import quaternion_to_matrix

    q = rng.standard_normal((16, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    np.savez_compressed(os.path.join(HERE, "quaternion.npz"), q=q, R=quaternion_to_matrix(torch.from_numpy(q)).numpy())

    # ---------------- policy math (normalisation, rotations, position codes, loss) ----------------
    from mindmap.diffuser_actor.position_encodings import RotaryPositionEncoding, RotaryPositionEncoding3D, SinusoidalPosEmb
    from mindmap.geometry.utils import (compute_rotation_matrix_from_ortho6d, get_ortho6d_from_rotation_matrix,
                                        matrix_to_quaternion)
    from mindmap.model_utils.loss import LossWeights, compute_loss
    from mindmap.model_utils.normalization import normalize_pos, normalize_trajectory, unnormalize_trajectory

    pm = {}
    wb = torch.tensor([[-0.37, -0.75, -0.13], [0.95, 0.75, 0.65]])
    pm["wb"] = wb.numpy()
    traj = torch.from_numpy(rng.uniform(-0.5, 0.9, size=(3, 2, 2, 7)).astype(np.float32))
    traj[..., 3:] = torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal((3, 2, 2, 4)).astype(np.float32)), dim=-1)
    pm["traj"] = traj.numpy()
    n9 = normalize_trajectory(traj.clone(), wb, "6D_from_query", "wxyz")
    pm["traj_norm"] = n9.numpy()
    n10 = torch.cat([n9, torch.from_numpy(rng.standard_normal((3, 2, 2, 1)).astype(np.float32))], dim=-1)
    pm["traj_norm10"] = n10.numpy()
    pm["traj_unnorm"] = unnormalize_trajectory(n10.clone(), wb, "6D_from_query", "wxyz").numpy()
    pts = torch.from_numpy(rng.uniform(-1, 1.2, size=(5, 7, 3)).astype(np.float32))
    pn, pv = normalize_pos(pts, wb)
    pm["pts"], pm["pts_norm"], pm["pts_valid"] = pts.numpy(), pn.numpy(), pv.numpy()
    d6 = torch.from_numpy(rng.standard_normal((9, 6)).astype(np.float32))
    R = compute_rotation_matrix_from_ortho6d(d6)
    pm["d6"], pm["d6_R"] = d6.numpy(), R.numpy()
    pm["R_d6"] = get_ortho6d_from_rotation_matrix(R).numpy()
    pm["R_quat"] = matrix_to_quaternion(R).numpy()
    xyz = torch.from_numpy(rng.uniform(-1, 1, size=(2, 11, 3)).astype(np.float32))
    pe = RotaryPositionEncoding3D(120)(xyz)
    pm["rot_xyz"], pm["rot_code"] = xyz.numpy(), pe.numpy()
    x = torch.from_numpy(rng.standard_normal((2, 11, 120)).astype(np.float32))
    pm["rot_x"] = x.numpy()
    pm["rot_applied"] = RotaryPositionEncoding.embed_rotary(x, pe[..., 0], pe[..., 1]).numpy()
    t = torch.tensor([0.0, 1.0, 17.0, 99.0])
    pm["sin_t"], pm["sin_emb"] = t.numpy(), SinusoidalPosEmb(120)(t).numpy()
    pred = torch.from_numpy(rng.standard_normal((4, 1, 2, 10)).astype(np.float32))
    tgt = torch.from_numpy(rng.standard_normal((4, 1, 2, 9)).astype(np.float32))
    go = (torch.from_numpy(rng.uniform(size=(4, 1, 2, 1))) > 0.5).float()
    hy_p = torch.from_numpy(rng.standard_normal((4, 1, 1)).astype(np.float32))
    hy_g = torch.from_numpy(rng.uniform(-1, 1, size=(4, 1, 1)).astype(np.float32))
    L = compute_loss(pred, hy_p, tgt, go, hy_g, LossWeights(), True, rotation_form="6D")
    pm["loss_pred"], pm["loss_tgt"], pm["loss_open"], pm["loss_hyp"], pm["loss_hyg"] = pred.numpy(), tgt.numpy(), go.numpy(), hy_p.numpy(), hy_g.numpy()
    pm["loss_out"] = np.array([float(v) for v in L])
    np.savez_compressed(os.path.join(HERE, "policy_math.npz"), **pm)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
