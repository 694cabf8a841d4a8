"""GPU tests of the policy side: HIP farthest-point sampling against the plain-torch restatement, one training step of
the full input pipeline (HIP back-projection inside unpack_batch) on cuda:0."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(4, 3072, 120, 614), (2, 500, 7, 100), (1, 8192, 16, 33), (3, 64, 1024, 64)])
def test_fps_kernel_matches_reference(shape):
    from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling, farthest_point_sampling_reference

    B, N, C, n = shape
    torch.manual_seed(N)
    x = torch.randn(B, N, C, device="cuda")
    x[:, N // 3: N // 3 + N // 10] = 0  # masked-out tokens are zeroed by the encoder: many identical points
    got = farthest_point_sampling(x, n, 0)
    ref = farthest_point_sampling_reference(x, n, 0)
    assert got.dtype == torch.int64 and got.shape == (B, n)
    assert torch.equal(got, ref)
    assert torch.equal(farthest_point_sampling(x, min(n, 5), 3)[:, 0], torch.full((B,), 3, device="cuda"))


def test_fps_rejects_bad_arguments():
    from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling

    with pytest.raises(RuntimeError):
        farthest_point_sampling(torch.zeros(1, 10, 4, device="cuda"), 11, 0)
    with pytest.raises(RuntimeError):
        farthest_point_sampling(torch.zeros(1, 10, 4), 2, 0)


def test_training_step_on_gpu_with_images():
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import build_model, build_optimizer, synthetic_batch, train_one_step

    torch.manual_seed(0)
    cfg = DiffuserActorConfig(data_type="rgbd_and_mesh", image_size=(128, 128), feature_dim=768)
    model = build_model(cfg, device="cuda")
    opt = build_optimizer(model)
    before = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad]).clone()
    losses = [train_one_step(cfg, model, opt, synthetic_batch(cfg, 2, "cuda", num_vertices=256, seed=i)) for i in range(2)]
    assert all(torch.isfinite(l[0]) for l in losses)
    after = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad])
    assert not torch.equal(before, after)
    model.eval()
    from nvblox_mindmap_amd.training.trainer import unpack_batch

    s = unpack_batch(cfg, synthetic_batch(cfg, 1, "cuda", num_vertices=256, seed=9))
    assert s["pcds"].shape == (1, 1, 3, 128, 128) and s["pcds"].is_cuda
    cfg.diffusion_timesteps = 100
    traj, yaw, _, _, _ = model(None, None, s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"],
                               s["vertices_valid_mask"], None, s["gripper_history"], run_inference=True)
    assert traj.shape == (1, 1, 2, 8) and torch.isfinite(traj).all()
