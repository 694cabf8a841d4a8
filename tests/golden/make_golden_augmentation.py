#!/usr/bin/env python3
"""Generates tests/golden/augmentation.npz by IMPORTING the reference's mindmap.data_loading.sample_transformer (:76-300) in the
authoring container (CPU): seeds, inputs and the reference's outputs."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def main():
    assert os.path.isdir("/root/reference"), "the reference is only available in the authoring container"
    # the module's import chain pulls in typed-argument-parser (absent here) for an argument class none of the functions
    # exercised below touch: an empty base class lets the import through (the functions are torch + pytorch3d-style math only)
    tap = type(sys)("tap")
    tap.Tap = type("Tap", (), {})
    sys.modules.setdefault("tap", tap)
    from mindmap.data_loading import sample_transformer as ST

    rng = np.random.default_rng(5)
    poses = rng.uniform(-1, 1, size=(3, 2, 8)).astype(np.float32)
    q = rng.standard_normal((3, 2, 4)).astype(np.float32)
    poses[..., 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    poses[..., 7] = rng.integers(0, 2, size=(3, 2))
    flat_poses = poses.reshape(6, 8).copy()
    vertices = rng.uniform(-1, 1, size=(50, 3)).astype(np.float32)
    t_range, rpy_range = ([-0.1, -0.2, -0.05], [0.1, 0.2, 0.05]), ([-10.0, -5.0, -180.0], [10.0, 5.0, 180.0])
    out = dict(poses=poses, flat_poses=flat_poses, vertices=vertices, t_lo=np.array(t_range[0]), t_hi=np.array(t_range[1]),
               rpy_lo=np.array(rpy_range[0]), rpy_hi=np.array(rpy_range[1]))
    random.seed(11)
    t, quat = ST.random_transform_uniform(t_range, rpy_range)
    out.update(uniform_t=t.numpy(), uniform_q=quat.numpy())
    random.seed(12)
    aug = ST.GeometryAugmentor(t_range, rpy_range)
    out["aug_poses"] = aug(torch.from_numpy(poses)).numpy()
    out["aug_vertices"] = aug({"vertices": torch.from_numpy(vertices)})["vertices"].numpy()
    aug.reset()
    out["aug_vertices_after_reset"] = aug(torch.from_numpy(vertices)).numpy()
    torch.manual_seed(21)
    t, quat = ST.random_transform_gaussian(0.01, 2.0, 7)
    out.update(gauss_t=t.numpy(), gauss_q=quat.numpy())
    torch.manual_seed(22)
    noiser = ST.GeometryNoiser(0.02, 3.0)
    out["noisy_vertices"] = noiser({"vertices": torch.from_numpy(vertices)})["vertices"].numpy()
    out["noisy_flat_poses"] = noiser(torch.from_numpy(flat_poses)).numpy()
    np.savez_compressed(os.path.join(HERE, "augmentation.npz"), **out)
    print("wrote augmentation.npz:", sorted(out))


if __name__ == "__main__":
    main()
