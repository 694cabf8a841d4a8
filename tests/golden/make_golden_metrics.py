#!/usr/bin/env python3
"""Generates tests/golden/metrics.npz by IMPORTING the reference's mindmap.model_utils.loss.compute_metrics (:83-139) in the
authoring container (CPU).  Inputs + the reference's outputs only."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


def main():
    assert os.path.isdir("/root/reference"), "the reference is only available in the authoring container"
    from mindmap.model_utils.loss import compute_metrics

    rng = np.random.default_rng(123)
    out = {}
    for tag, (B, L, G) in {"arm": (5, 1, 1), "humanoid": (4, 2, 2)}.items():
        def poses():
            p = rng.uniform(-1, 1, size=(B, L, G, 8)).astype(np.float32)
            q = rng.standard_normal((B, L, G, 4)).astype(np.float32)
            p[..., 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
            p[..., 7] = rng.uniform(0, 1, size=(B, L, G))
            return p

        pred, gt = poses(), poses()
        pred[0] = gt[0]  # an exact hit: the small-angle branch of the rotation error
        pred[1, ..., 3:7] = -gt[1, ..., 3:7]  # the same rotation with the other sign
        yaw_p, yaw_g = rng.uniform(-3, 3, size=(B, L, 1)).astype(np.float32), rng.uniform(-3, 3, size=(B, L, 1)).astype(np.float32)
        m = compute_metrics(torch.from_numpy(pred), torch.from_numpy(yaw_p), torch.from_numpy(gt), torch.from_numpy(yaw_g),
                            predict_head_yaw=True, rotation_form="quaternion")
        out.update({f"{tag}_pred": pred, f"{tag}_gt": gt, f"{tag}_yaw_pred": yaw_p, f"{tag}_yaw_gt": yaw_g})
        out.update({f"{tag}_m_{k}": np.asarray(v.numpy()) for k, v in m.items()})
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("wrote metrics.npz:", sorted(k for k in out if "_m_" in k))


if __name__ == "__main__":
    main()
