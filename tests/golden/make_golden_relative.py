#!/usr/bin/env python3
"""Generates tests/golden/relative_conversions.npz by IMPORTING the reference's mindmap.model_utils.relative_conversions
(:15-133) in the authoring container (CPU).  Inputs + the reference's outputs only; no reference code is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def main():
    assert os.path.isdir("/root/reference"), "the reference is only available in the authoring container"
    from mindmap.model_utils import relative_conversions as RC

    rng = np.random.default_rng(77)

    def poses(*shape):  # [..., 8]: position, unit quaternion (both signs of the real part), openness
        p = rng.uniform(-1, 1, size=shape + (8,)).astype(np.float32)
        q = rng.standard_normal(shape + (4,)).astype(np.float32)
        p[..., 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
        p[..., 7] = rng.integers(0, 2, size=shape)
        return p

    out = {}
    for tag, (B, nhist, ngrip, L) in {"arm": (3, 3, 1, 1), "humanoid": (4, 3, 2, 2)}.items():
        hist, traj = poses(B, nhist, ngrip), poses(B, L, ngrip)
        cur = RC.get_current_pose_from_gripper_history(torch.from_numpy(hist))
        rel = RC.to_relative_trajectory(torch.from_numpy(traj), cur)
        out.update({f"{tag}_history": hist, f"{tag}_trajectory": traj, f"{tag}_current": cur.numpy(),
                    f"{tag}_history_rel": RC.to_relative_gripper_history(torch.from_numpy(hist), cur).numpy(),
                    f"{tag}_trajectory_rel": rel.numpy(), f"{tag}_trajectory_back": RC.to_absolute_trajectory(rel, cur).numpy()})
    pcd = rng.uniform(-2, 2, size=(3, 2, 3, 5, 7)).astype(np.float32)
    pose2d = poses(3)
    out.update(pcd=pcd, pcd_pose=pose2d, pcd_rel=RC.to_relative_pcd(torch.from_numpy(pcd), torch.from_numpy(pose2d)).numpy())
    a, b = poses(6)[..., 3:7], poses(6)[..., 3:7]
    out.update(qa=a, qb=b, q_ab=RC.quaternion_multiply(torch.from_numpy(a), torch.from_numpy(b)).numpy(),
               q_inv=RC.quaternion_invert(torch.from_numpy(a)).numpy())
    np.savez_compressed(os.path.join(HERE, "relative_conversions.npz"), **out)
    print("wrote relative_conversions.npz:", sorted(out))


if __name__ == "__main__":
    main()
