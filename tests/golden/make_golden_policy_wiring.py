#!/usr/bin/env python3
"""Generates tests/golden/policy_wiring.npz by IMPORTING the reference's top-level model, mindmap/diffuser_actor/diffuser_actor.py
(DiffuserActor :29) with its Encoder (diffuser_actor/encoder.py:32), and running its TRAINING forward (:518-690) and its
INFERENCE forward (compute_trajectory :411-516, conditional_sample :337-409) on the CPU, data type MESH (authoring container
only; the fixture -- parameter names / shapes, seeds, checksums and the reference's OUTPUTS -- is committed, the reference's
code is not).

What this pins: the WIRING above the already-pinned pieces (tests/golden/policy_head.npz: DiffusionHead, attention stacks) --
gripper-history split and closedness, normalisation of history / vertices / targets, Encoder.encode_feature_pointcloud,
encode_gripper_history, run_fps (masking, gather, sequence-first layout), the conditioning of the head, compute_loss, the
reverse-diffusion loop's bookkeeping (what is fed back, what is concatenated), unnormalize_trajectory and the head-yaw clamp.

What it does NOT pin (third-party modules absent here; import-only stand-ins, see below):
  dgl.geometry.farthest_point_sampler  -> tests/fps_restatement.py (the published definition, numpy)
  diffusers DDPMScheduler              -> nvblox_mindmap_amd.diffuser_actor.scheduler.DDPMScheduler behind an adaptor with diffusers'
                                          constructor keywords, `.config.num_train_timesteps`, `.timesteps`, `.add_noise`, `.step`
so the numerics of those two stay restatements of their published algorithms.  clip / torchvision / wandb are never reached.
The reference moves modules to "cuda" in constructors (encoder.py:24); `.to("cuda")` is redirected to the CPU for the run.
The Gaussian noise of the sampling loop is fed from a list (the reference draws it with torch.randn inside conditional_sample and
inside diffusers' step; this repository's loop takes it as one tensor): same numbers on both sides.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")

import golden_seeded as GS  # noqa: E402

CFG = dict(embedding_dim=120, nhist=3, ngrippers=2, prediction_horizon=1, diffusion_timesteps=6, fps_subsampling_factor=5,
           feature_dim=3, n_vertices=40, batch=2)
WORKSPACE = np.array([[-0.37, -0.75, -0.13], [0.95, 0.75, 0.65]], dtype=np.float32)


def stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def model_inputs(seed):
    """World-frame inputs of DiffuserActor.forward for data type MESH (shapes of diffuser_actor.py:532-547)."""
    rng = np.random.default_rng(seed)
    B, N, C, nh, G, L = CFG["batch"], CFG["n_vertices"], CFG["feature_dim"], CFG["nhist"], CFG["ngrippers"], CFG["prediction_horizon"]
    f32 = np.float32

    def poses(n):
        q = rng.standard_normal((B, n, G, 4))
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        p = rng.uniform(WORKSPACE[0] + 0.05, WORKSPACE[1] - 0.05, size=(B, n, G, 3))
        o = (rng.uniform(size=(B, n, G, 1)) > 0.5).astype(np.float64)
        return np.concatenate([p, q, o], axis=-1).astype(f32)

    x = {"vertices": rng.uniform(WORKSPACE[0], WORKSPACE[1], size=(B, N, 3)).astype(f32),
         "vertex_features": rng.standard_normal((B, N, C)).astype(f32),
         "vertices_valid_mask": rng.uniform(size=(B, N)) > 0.2,
         "gripper_history": poses(nh), "gt_gripper_pred": poses(L),
         "gt_head_yaw": rng.uniform(-3.0, 3.0, size=(B, L, 1)).astype(f32)}
    T = CFG["diffusion_timesteps"]
    x["sampling_noise"] = rng.standard_normal((1 + T, B, L, G, 9)).astype(f32)
    return x


def main():
    assert os.path.isdir("/root/reference"), "the reference is only available in the authoring container"
    import nvblox_mindmap_amd

    nvblox_mindmap_amd.install_as_nvblox_torch()
    from fps_restatement import farthest_point_sampling_numpy
    from nvblox_mindmap_amd.diffuser_actor.scheduler import DDPMScheduler as OurScheduler

    stub("clip", load=None)
    stub("clip.model", ModifiedResNet=object)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.ops = stub("torchvision.ops", FeaturePyramidNetwork=object)
    stub("wandb")

    def fps(x, npoints, start_idx=0):  # dgl.geometry.farthest_point_sampler(pos [B,N,C], npoints, start_idx) -> [B,npoints] int64
        return torch.from_numpy(farthest_point_sampling_numpy(x.detach().cpu().numpy(), int(npoints), int(start_idx)))

    dgl = stub("dgl")
    dgl.geometry = stub("dgl.geometry", farthest_point_sampler=fps)

    feed = []  # Gaussian noise handed to the reference's sampling loop, in the order it asks for it

    class DiffusersAdaptor:
        """diffusers.schedulers.scheduling_ddpm.DDPMScheduler's surface as the reference uses it (diffuser_actor.py:147-156,
        349-397,644-662), on this repository's restated scheduler."""

        def __init__(self, num_train_timesteps, beta_schedule, prediction_type):
            assert prediction_type == "epsilon"
            self.inner = OurScheduler(num_train_timesteps, beta_schedule)
            self.config = types.SimpleNamespace(num_train_timesteps=num_train_timesteps)

        def set_timesteps(self, n):
            self.inner.set_timesteps(n)
            self.timesteps = self.inner.timesteps

        def add_noise(self, x, noise, t):
            return self.inner.add_noise(x, noise, t)

        def step(self, model_output, t, sample):
            k = self.inner.timesteps.tolist().index(int(t))
            noise = feed[1 + k][..., :3] if model_output.shape[-1] == 3 else feed[1 + k][..., 3:9]
            return types.SimpleNamespace(prev_sample=self.inner.step(model_output, int(t), sample, noise=noise))

    stub("diffusers")
    stub("diffusers.schedulers")
    stub("diffusers.schedulers.scheduling_ddpm", DDPMScheduler=DiffusersAdaptor)

    orig_tensor_to, orig_module_to = torch.Tensor.to, torch.nn.Module.to

    def cpu_only(args, kwargs):
        if kwargs.get("device") == "cuda":
            kwargs = dict(kwargs, device="cpu")
        return tuple("cpu" if (isinstance(a, str) and a == "cuda") else a for a in args), kwargs

    def tensor_to(self, *a, **k):
        a, k = cpu_only(a, k)
        return orig_tensor_to(self, *a, **k)

    def module_to(self, *a, **k):
        a, k = cpu_only(a, k)
        return orig_module_to(self, *a, **k)

    torch.Tensor.to, torch.nn.Module.to = tensor_to, module_to
    out = {}
    try:
        from mindmap.data_loading.data_types import DataType
        from mindmap.diffuser_actor.diffuser_actor import DiffuserActor
        from mindmap.image_processing.feature_extraction import FeatureExtractorType

        ref = DiffuserActor(feature_type=FeatureExtractorType.RGB, image_size=(256, 256), embedding_dim=CFG["embedding_dim"],
                            use_instruction=False, fps_subsampling_factor=CFG["fps_subsampling_factor"],
                            workspace_bounds=torch.from_numpy(WORKSPACE), rotation_parametrization="6D_from_query",
                            quaternion_format="wxyz", diffusion_timesteps=CFG["diffusion_timesteps"], nhist=CFG["nhist"],
                            ngrippers=CFG["ngrippers"], prediction_horizon=CFG["prediction_horizon"], relative=False,
                            predict_head_yaw=True, data_type=DataType.MESH, use_fps=True, encode_openness=True,
                            use_shared_feature_encoder=False, add_external_cam=False)
        ref.vis = None
        spec = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
        state = GS.seeded_state(spec, 4242)
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        ref.eval()
        out["spec"] = np.array(GS.spec_to_json(spec))
        out["state_checksum"] = np.array(GS.state_checksum(state))
        x = model_inputs(77)
        t = {k: torch.from_numpy(v) for k, v in x.items()}

        def call(run_inference):
            return ref(t["gt_gripper_pred"].clone(), t["gt_head_yaw"].clone(), None, None, None, t["vertex_features"].clone(),
                       t["vertices"].clone(), t["vertices_valid_mask"].clone(), None, t["gripper_history"].clone(),
                       run_inference=run_inference)

        # training forward: torch.randn(gt.shape) then torch.randint(0, T, (B,)) on the CPU default generator (:644-652)
        torch.manual_seed(1234)
        with torch.no_grad():
            losses, fixed, _ = call(False)
        out["train_losses"] = np.array([float(v) for v in losses], dtype=np.float64)
        for k in ("context_feats", "context", "adaln_gripper_feats", "fps_feats", "fps_pos"):
            out["enc_" + k] = fixed[k].numpy()
        out["enc_fps_mask"] = fixed["fps_mask"].numpy()
        # inference forward: the loop's noise comes from the feed
        feed[:] = [t["sampling_noise"][i] for i in range(t["sampling_noise"].shape[0])]
        real_randn = torch.randn
        torch.randn = lambda *a, **k: feed[0].clone()  # conditional_sample's only direct draw: x_T (:357-359)
        try:
            with torch.no_grad():
                traj, head_yaw, inf_losses, _, _ = call(True)
        finally:
            torch.randn = real_randn
        out["infer_trajectory"] = traj.numpy()
        out["infer_head_yaw"] = head_yaw.numpy()
        out["infer_losses"] = np.array([float(v) for v in inf_losses], dtype=np.float64)
    finally:
        torch.Tensor.to, torch.nn.Module.to = orig_tensor_to, orig_module_to
    path = os.path.join(HERE, "policy_wiring.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items()}, os.path.getsize(path))
    print("train losses", out["train_losses"], "inference losses", out["infer_losses"])


if __name__ == "__main__":
    main()
