"""CPU tests of the policy side: math pinned against golden vectors generated from the
reference's importable modules, the DDPM scheduler's invariants, attention masking, the model's plumbing
(BASELINE configs[0]) and the data-parallel path on gloo with world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pm():
    return np.load(f"{GOLD}/policy_math.npz")


def test_normalisation_and_rotations_match_reference(pm):
    from nvblox_mindmap_amd.diffuser_actor import rotations as R

    wb = torch.from_numpy(pm["wb"])
    n9 = R.normalize_trajectory(torch.from_numpy(pm["traj"]), wb, "wxyz")
    assert np.allclose(n9.numpy(), pm["traj_norm"], atol=1e-6)
    un = R.unnormalize_trajectory(torch.from_numpy(pm["traj_norm10"]), wb, "wxyz").numpy()
    ref = pm["traj_unnorm"]
    assert np.allclose(un[..., :3], ref[..., :3], atol=1e-5) and np.allclose(un[..., 7:], ref[..., 7:], atol=1e-6)
    q, qr = un[..., 3:7], ref[..., 3:7]  # quaternions are equal up to sign
    assert np.allclose(np.abs((q * qr).sum(-1)), 1.0, atol=1e-5)
    pn, pv = R.normalize_pos(torch.from_numpy(pm["pts"]), wb)
    assert np.allclose(pn.numpy(), pm["pts_norm"], atol=1e-6) and np.array_equal(pv.numpy(), pm["pts_valid"])
    assert np.allclose(R.unnormalize_pos(pn, wb).numpy(), pm["pts"], atol=1e-5)
    Rm = R.ortho6d_to_matrix(torch.from_numpy(pm["d6"]))
    assert np.allclose(Rm.numpy(), pm["d6_R"], atol=1e-5)
    assert np.allclose(R.matrix_to_ortho6d(Rm).numpy(), pm["R_d6"], atol=1e-5)
    qq = R.matrix_to_quat_wxyz(torch.from_numpy(pm["d6_R"])).numpy()
    assert np.allclose(np.abs((qq * pm["R_quat"]).sum(-1)), 1.0, atol=1e-5)
    # xyzw input format
    t = torch.from_numpy(pm["traj"])
    t_xyzw = torch.cat([t[..., :3], t[..., (4, 5, 6, 3)]], dim=-1)
    assert np.allclose(R.normalize_trajectory(t_xyzw, wb, "xyzw").numpy(), pm["traj_norm"], atol=1e-6)


def test_position_codes_match_reference(pm):
    from nvblox_mindmap_amd.diffuser_actor.layers import apply_rotary, rotary3d, sinusoidal_embedding

    cos, sin = rotary3d(torch.from_numpy(pm["rot_xyz"]), 120)
    assert np.allclose(cos.numpy(), pm["rot_code"][..., 0], atol=1e-6) and np.allclose(sin.numpy(), pm["rot_code"][..., 1], atol=1e-6)
    assert np.allclose(apply_rotary(torch.from_numpy(pm["rot_x"]), cos, sin).numpy(), pm["rot_applied"], atol=1e-5)
    assert np.allclose(sinusoidal_embedding(torch.from_numpy(pm["sin_t"]), 120).numpy(), pm["sin_emb"], atol=1e-5)


def test_loss_matches_reference(pm):
    from nvblox_mindmap_amd.diffuser_actor.loss import LossWeights, compute_loss

    out = compute_loss(torch.from_numpy(pm["loss_pred"]), torch.from_numpy(pm["loss_hyp"]), torch.from_numpy(pm["loss_tgt"]),
                       torch.from_numpy(pm["loss_open"]), torch.from_numpy(pm["loss_hyg"]), LossWeights(), True)
    assert np.allclose([float(v) for v in out], pm["loss_out"], rtol=1e-6)


def test_ddpm_scheduler():
    from nvblox_mindmap_amd.diffuser_actor.scheduler import DDPMScheduler

    for sched in ("scaled_linear", "squaredcos_cap_v2"):
        s = DDPMScheduler(100, sched)
        assert s.betas.shape == (100,) and bool((s.betas > 0).all()) and bool((s.betas < 1).all())
        assert bool((s.alphas_cumprod[1:] < s.alphas_cumprod[:-1]).all())
        assert s.timesteps.tolist() == list(range(99, -1, -1))
    s = DDPMScheduler(100, "scaled_linear")
    assert abs(float(s.betas[0]) - 1e-4) < 1e-9 and abs(float(s.betas[-1]) - 0.02) < 1e-8
    c = DDPMScheduler(100, "squaredcos_cap_v2")
    import math
    ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    assert abs(float(c.betas[10]) - (1 - ab(11 / 100) / ab(10 / 100))) < 1e-7 and float(c.betas[-1]) == pytest.approx(0.999)
    # forward process: x_t = sqrt(abar) x0 + sqrt(1 - abar) eps, per-sample timestep
    x0, eps = torch.randn(4, 1, 2, 3) * 0.3, torch.randn(4, 1, 2, 3)
    t = torch.tensor([0, 10, 50, 99])
    xt = s.add_noise(x0, eps, t)
    for i in range(4):
        a = s.alphas_cumprod[t[i]]
        assert torch.allclose(xt[i], a.sqrt() * x0[i] + (1 - a).sqrt() * eps[i], atol=1e-6)
    # reverse step with the TRUE noise: the posterior mean moves towards x0; at t = 0 it returns x0 exactly (no noise)
    assert torch.allclose(s.step(eps[0:1], 0, s.add_noise(x0[0:1], eps[0:1], torch.tensor([0]))), x0[0:1].clamp(-1, 1), atol=1e-5)
    g = torch.Generator().manual_seed(0)
    xt50 = s.add_noise(x0, eps, torch.full((4,), 50))
    prev = s.step(eps, 50, xt50, generator=g)
    a50, a49 = float(s.alphas_cumprod[50]), float(s.alphas_cumprod[49])
    beta = 1 - a50 / a49
    mean = (a49 ** 0.5 * beta / (1 - a50)) * x0.clamp(-1, 1) + ((a50 / a49) ** 0.5 * (1 - a49) / (1 - a50)) * xt50
    var = (1 - a49) / (1 - a50) * beta
    assert float(((prev - mean) / var ** 0.5).std()) == pytest.approx(1.0, abs=0.35)
    # strided inference grid
    s.set_timesteps(10)
    assert s.timesteps.tolist() == list(range(90, -1, -10))


def test_attention_ignores_masked_keys():
    """Key-padding mask invariance (what the reference's tests/test_attention_masking.py:29-120 checks)."""
    from nvblox_mindmap_amd.diffuser_actor.layers import AttentionStack, rotary3d

    torch.manual_seed(0)
    D = 60
    stack = AttentionStack(D, 4, 2, use_adaln=True).eval()
    for m in stack.modules():  # AdaLN starts as identity: give it non-trivial weights
        if hasattr(m, "proj") and m.proj.out_features == 2 * D:
            torch.nn.init.normal_(m.proj.weight, std=0.1)
    q, mem = torch.randn(2, 3, D), torch.randn(2, 10, D)
    qpos, mpos = torch.rand(2, 3, 3), torch.rand(2, 10, 3)
    cond = torch.randn(2, D)
    pad = torch.zeros(2, 10, dtype=torch.bool)
    pad[:, 6:] = True
    out1, w1 = stack(q, mem, cond, rotary3d(qpos, D), rotary3d(mpos, D), key_padding_mask=pad, need_weights=True)
    mem2 = mem.clone()
    mem2[:, 6:] = torch.randn(2, 4, D) * 100
    out2, _ = stack(q, mem2, cond, rotary3d(qpos, D), rotary3d(mpos, D), key_padding_mask=pad)
    assert torch.allclose(out1, out2, atol=1e-5)  # also: explicit-softmax path == SDPA path
    assert torch.all(w1[..., 6:] == 0) and torch.allclose(w1.sum(-1), torch.ones_like(w1.sum(-1)), atol=1e-5)
    out3, _ = stack(q, mem2, cond, rotary3d(qpos, D), rotary3d(mpos, D))
    assert not torch.allclose(out1, out3, atol=1e-3)


def test_fps_reference_semantics():
    from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling_cpu

    x = torch.tensor([[[0.0, 0], [1, 0], [10, 0], [10.5, 0], [-3, 0], [0, 0], [0, 0]]])
    idx = farthest_point_sampling_cpu(x, 4, 0)[0].tolist()
    assert idx[0] == 0 and idx[1] == 3 and idx[2] == 4 and idx[3] == 1  # farthest from {0}, then from {0, 3}, ...
    torch.manual_seed(1)
    y = torch.randn(3, 200, 16)
    i2 = farthest_point_sampling_cpu(y, 40, 0)
    assert i2.shape == (3, 40) and all(len(set(r.tolist())) == 40 for r in i2)
    z = torch.zeros(1, 5, 3)  # all ties: first index every time
    assert farthest_point_sampling_cpu(z, 3, 0)[0].tolist() == [0, 0, 0]


def _tiny_cfg(**kw):
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig

    base = dict(data_type="mesh", feature_dim=24, embedding_dim=60, num_attn_heads=4, diffusion_timesteps=10)
    base.update(kw)
    return DiffuserActorConfig(**base)


def _tiny_batch(cfg, B, seed):
    from nvblox_mindmap_amd.training.trainer import synthetic_batch

    return synthetic_batch(cfg, B, "cpu", num_vertices=96, seed=seed)


def test_model_single_forward_cpu_plumbing():
    """BASELINE configs[0]: one forward of the policy on a cached-sample-shaped input, PyTorch CPU."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import get_workspace_bounds
    from nvblox_mindmap_amd.training.trainer import unpack_batch

    torch.manual_seed(0)
    cfg = _tiny_cfg()
    model = DiffuserActor(cfg, get_workspace_bounds("DRILL_IN_BOX"))
    s = unpack_batch(cfg, _tiny_batch(cfg, 2, 0))
    s["vertices_valid_mask"][1, :] = False  # a sample with an empty context must not produce NaN
    losses, enc, _ = model(s["gt_gripper_pred"], s["gt_head_yaw"], None, None, None, s["vertex_features"], s["vertices"],
                           s["vertices_valid_mask"], None, s["gripper_history"])
    assert all(torch.isfinite(x) for x in losses)
    assert enc["context_feats"].shape == (2, 96, 60) and enc["fps_feats"].shape == (2, 96 // 5, 60)
    assert enc["history_feats"].shape == (2, cfg.num_history * cfg.ngrippers, 60)
    losses[0].backward()
    unused = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    assert unused and all(("vl_" in n or "instruction" in n or "goal_embed" in n or "traj_lang_" in n) for n in unused), unused
    model.eval()
    traj, yaw, l2, _, _ = model(s["gt_gripper_pred"], s["gt_head_yaw"], None, None, None, s["vertex_features"], s["vertices"],
                                s["vertices_valid_mask"], None, s["gripper_history"], run_inference=True)
    assert traj.shape == (2, 1, 2, 8) and yaw.shape == (2, 1, 1) and torch.isfinite(traj).all()
    assert torch.allclose(traj[..., 3:7].norm(dim=-1), torch.ones(2, 1, 2), atol=1e-4)  # unit quaternions
    assert bool(((traj[..., 7] >= 0) & (traj[..., 7] <= 1)).all())  # openness probability
    assert l2 is not None and torch.isfinite(l2[0])


def test_relative_conversions_match_reference():
    """model_utils/relative_conversions.py:15-133, vectors from the imported reference (tests/golden/make_golden_relative.py);
    the reference's own test is the round trip (tests/test_relative_conversions.py:33)."""
    from nvblox_mindmap_amd.diffuser_actor import relative_conversions as RC

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "relative_conversions.npz"))
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    assert torch.equal(RC.quaternion_invert(t("qa")), t("q_inv"))
    assert np.allclose(RC.quaternion_multiply(t("qa"), t("qb")).numpy(), g["q_ab"], atol=1e-6)
    assert bool((RC.quaternion_multiply(t("qa"), t("qb"))[..., 0] >= 0).all())
    for tag in ("arm", "humanoid"):
        cur = RC.get_current_pose_from_gripper_history(t(f"{tag}_history"))
        assert torch.equal(cur, t(f"{tag}_current"))
        hist = t(f"{tag}_history")
        keep = hist.clone()
        assert np.allclose(RC.to_relative_gripper_history(hist, cur).numpy(), g[f"{tag}_history_rel"], atol=1e-6)
        assert torch.equal(hist, keep), "the caller's history must not be modified"
        rel = RC.to_relative_trajectory(t(f"{tag}_trajectory"), cur)
        assert np.allclose(rel.numpy(), g[f"{tag}_trajectory_rel"], atol=1e-6)
        back = RC.to_absolute_trajectory(rel, cur)
        assert np.allclose(back.numpy(), g[f"{tag}_trajectory_back"], atol=1e-6)
        # the round trip: positions / state exact to float error, rotation equal up to the quaternion's sign
        tr = g[f"{tag}_trajectory"]
        assert np.allclose(back[..., :3].numpy(), tr[..., :3], atol=1e-6) and np.array_equal(back[..., 7].numpy(), tr[..., 7])
        assert np.allclose(np.abs((back[..., 3:7].numpy() * tr[..., 3:7]).sum(-1)), 1.0, atol=1e-5)
    assert np.allclose(RC.to_relative_pcd(t("pcd"), t("pcd_pose")).numpy(), g["pcd_rel"], atol=1e-6)
    with pytest.raises(RuntimeError):  # per-gripper poses have no single origin (the reference's view() raises as well)
        RC.to_relative_pcd(t("pcd"), t("humanoid_current")[:3])


def test_relative_action_model_equals_the_absolute_model_on_shifted_inputs():
    """cfg.relative_action (reference ``relative``, diffuser_actor.py:554-566, :509): the relative model on world inputs ==
    the absolute model fed the hand-converted history / target, its prediction mapped back by the current pose."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor
    from nvblox_mindmap_amd.diffuser_actor import relative_conversions as RC
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import get_workspace_bounds
    from nvblox_mindmap_amd.training.trainer import unpack_batch

    torch.manual_seed(0)
    cfg = _tiny_cfg()
    absolute = DiffuserActor(cfg, get_workspace_bounds("DRILL_IN_BOX")).eval()
    import copy
    import dataclasses

    relative = copy.deepcopy(absolute)
    relative.cfg = dataclasses.replace(cfg, relative_action=True)
    s = unpack_batch(cfg, _tiny_batch(cfg, 2, 0))
    cur = RC.get_current_pose_from_gripper_history(s["gripper_history"][..., :7])
    hist_rel = torch.cat([RC.to_relative_gripper_history(s["gripper_history"][..., :7], cur), s["gripper_history"][..., 7:]], dim=-1)
    gt_rel = RC.to_relative_trajectory(s["gt_gripper_pred"], cur)
    args = (None, None, None, s["vertex_features"], s["vertices"], s["vertices_valid_mask"], None)
    torch.manual_seed(3)
    la = absolute(gt_rel, s["gt_head_yaw"], *args, hist_rel)[0]
    torch.manual_seed(3)
    lr = relative(s["gt_gripper_pred"], s["gt_head_yaw"], *args, s["gripper_history"])[0]
    assert all(torch.allclose(a, b, atol=1e-6) for a, b in zip(la, lr))
    torch.manual_seed(4)
    ta, ya = absolute(None, None, *args, hist_rel, run_inference=True)[:2]
    torch.manual_seed(4)
    tr, yr = relative(None, None, *args, s["gripper_history"], run_inference=True)[:2]
    assert torch.allclose(RC.to_absolute_trajectory(ta, cur), tr, atol=1e-6) and torch.allclose(ya, yr, atol=1e-6)
    assert not torch.allclose(ta, tr, atol=1e-3)


def test_model_image_branch_shapes_cpu():
    from nvblox_mindmap_amd.diffuser_actor.model import DiffuserActorConfig, Encoder

    torch.manual_seed(0)
    cfg = DiffuserActorConfig(data_type="rgbd_and_mesh", image_size=(64, 64), feature_dim=768, embedding_dim=60, num_attn_heads=4)
    enc = Encoder(cfg).eval()
    rgb, pcd = torch.rand(2, 1, 3, 64, 64), torch.rand(2, 1, 3, 64, 64) * 2 - 1
    valid = torch.ones(2, 1, 64, 64, dtype=torch.bool)
    valid[0, 0, :16, :16] = False
    valid[0, 0, 20, 40] = False
    tokens, pos, m = enc.encode_images(rgb, pcd, valid)
    assert tokens.shape == (2, 16, 60) and pos.shape == (2, 16, 3) and m.shape == (2, 16)
    assert not m[0, 0] and not m[0, 1 * 4 + 2] and int(m[0].sum()) == 14 and bool(m[1].all())
    assert all(not p.requires_grad for p in enc.backbone.parameters())


def test_distributed_sampler_union_equals_single_process():
    from nvblox_mindmap_amd.training.sampler import DistributedWeightedSampler

    w = torch.rand(103) + 0.1
    single = DistributedWeightedSampler(w, 103, replacement=True, seed=5, num_replicas=1, rank=0)
    single.set_epoch(3)
    ref = single.global_indices().tolist()
    parts = []
    for r in range(4):
        s = DistributedWeightedSampler(w, 103, replacement=True, seed=5, num_replicas=4, rank=r)
        s.set_epoch(3)
        parts.append(list(iter(s)))
        assert len(parts[-1]) == len(s) == 26
    merged = [parts[i % 4][i // 4] for i in range(104)]
    assert merged[:103] == ref and merged[103] == ref[0]
    s.set_epoch(4)
    assert list(iter(s)) != parts[-1]
    uni = DistributedWeightedSampler(torch.ones(50), 50, replacement=False, seed=1, num_replicas=2, rank=0)
    uni2 = DistributedWeightedSampler(torch.ones(50), 50, replacement=False, seed=1, num_replicas=2, rank=1)
    assert sorted(list(iter(uni)) + list(iter(uni2))) == list(range(50))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from nvblox_mindmap_amd.training import (ProcessGroup, all_gather_objects, build_optimizer, get_rank, get_world_size,
                                             max_over_ranks, train_one_step, wrap_ddp)
    from nvblox_mindmap_amd.training.trainer import build_model

    with ProcessGroup(backend="gloo"):
        assert get_world_size() == world and get_rank() == rank
        torch.manual_seed(0)  # same initial weights on every rank
        cfg = _tiny_cfg()
        model = build_model(cfg, device="cpu")
        ddp = wrap_ddp(model, "cpu")
        opt = build_optimizer(ddp, lr=1e-3)
        torch.manual_seed(100 + rank)  # different noise / timesteps per rank
        losses = []
        for it in range(2):
            losses.append(float(train_one_step(cfg, ddp, opt, _tiny_batch(cfg, 2, seed=10 * rank + it))[0]))
        vec = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad])
        gathered = all_gather_objects({"rank": rank, "checksum": float(vec.double().sum()), "loss": losses})
        tmax = max_over_ranks(1.0 + rank)
        if rank == 0:
            out.put((gathered, tmax))


def test_ddp_two_ranks_gloo():
    """world_size-2 data-parallel step on CPU (gloo): gradients are all-reduced, so both ranks hold identical weights
    after the optimizer step although they saw different data; the timing reduction is a MAX over ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g["rank"] for g in gathered] == [0, 1]
    assert gathered[0]["checksum"] == gathered[1]["checksum"]
    assert gathered[0]["loss"] != gathered[1]["loss"]
    assert tmax == 2.0


def _flat_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from nvblox_mindmap_amd.training import (GraphedTrainStep, ProcessGroup, all_gather_objects, build_optimizer, train_one_step,
                                             wrap_ddp)
    from nvblox_mindmap_amd.training.trainer import build_lr_scheduler, build_model

    with ProcessGroup(backend="gloo"):
        cfg = _tiny_cfg()
        steps, lr = 4, 1e-3
        batches = [_tiny_batch(cfg, 2, seed=10 * rank + it) for it in range(steps)]
        # (a) the reference-shaped step: DDP(find_unused_parameters) + AdamW over the individual parameters + LinearLR
        torch.manual_seed(0)
        ref = build_model(cfg, device="cpu")
        ddp = wrap_ddp(ref, "cpu")
        opt = build_optimizer(ddp, lr=lr)
        sched = build_lr_scheduler(opt, train_iters=8)
        torch.manual_seed(100 + rank)
        ref_losses = [[float(x) for x in train_one_step(cfg, ddp, opt, b, scheduler=sched)] for b in batches]
        # (b) flat buffers + one explicit all-reduce + AdamW over the two flat segments (the eager form of the captured step)
        torch.manual_seed(0)
        model = build_model(cfg, device="cpu")
        flat = GraphedTrainStep(cfg, model, batches[0], lr=lr, use_graphs=False)
        flat.linear_lr(train_iters=8)
        torch.manual_seed(100 + rank)
        flat_losses = []
        for b in batches:
            flat_losses.append([float(x) for x in flat.step(b)])
            flat.scheduler_step()
        names = dict(model.named_parameters())
        diff = max(float((p - names[n]).abs().max()) for n, p in ref.named_parameters())
        equal = all(torch.equal(p, names[n]) for n, p in ref.named_parameters())
        # (c) a step that ONE rank takes alone while the group exists (bench.py's rank-0-only file-fed leg): data_parallel=False
        # issues no collective -- a collective here would wait for rank 1 forever
        alone_world = None
        if rank == 0:
            torch.manual_seed(0)
            alone = GraphedTrainStep(cfg, build_model(cfg, device="cpu"), batches[0], lr=lr, use_graphs=False, data_parallel=False)
            alone.step(batches[0])
            alone_world = (alone.world, alone.observed_world())
        unused_ref = sorted(n for n, p in ref.named_parameters() if p.requires_grad and p.grad is None)
        vec = torch.cat([p.detach().flatten() for p in model.parameters() if p.requires_grad])
        gathered = all_gather_objects({"diff": diff, "equal": equal, "ref_losses": ref_losses, "flat_losses": flat_losses,
                                       "unused": (unused_ref, sorted(flat.unused_names)), "checksum": float(vec.double().sum()),
                                       "lr": (opt.param_groups[0]["lr"], flat.lr), "world": flat.observed_world(),
                                       "payload": flat.flat_grad.numel(), "steps": flat.steps_done, "alone": alone_world})
        if rank == 0:
            out.put(gathered)


def test_flat_allreduce_step_matches_ddp_two_ranks_gloo():
    """training.GraphedTrainStep (flat gradient buffer, ONE explicit all-reduce, AdamW over two flat segments, unused parameters
    found by a probe) against the reference-shaped DDP step on 2 gloo ranks that see different data: the same losses, the same
    weights after 4 steps with a LinearLR ramp, the same set of parameters left without a gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = g
    for r in (a, b):
        assert r["unused"][0] == r["unused"][1] and len(r["unused"][0]) > 0  # the instruction branch, on both paths
        assert r["world"] == 2 and r["steps"] == 4 and r["payload"] > 0
        assert r["lr"][0] == r["lr"][1] < 1e-3
        # losses: 5 values x 4 steps (total, position, rotation, gripper, head yaw)
        assert np.allclose(np.array(r["ref_losses"], dtype=np.float64), np.array(r["flat_losses"], dtype=np.float64), rtol=1e-6, atol=1e-7)
        assert r["diff"] <= 1e-7, r["diff"]  # elementwise-identical arithmetic (bit-equal on this build: see "equal")
    assert a["checksum"] == b["checksum"]  # the explicit all-reduce kept the ranks in step
    assert tuple(a["alone"]) == (1, 1) and b["alone"] is None
    assert a["flat_losses"] != b["flat_losses"]  # (they saw different data)
    assert a["equal"] and b["equal"], (a["diff"], b["diff"])


def test_compute_metrics_matches_reference():
    """model_utils/loss.py:83-139 (vectors from the imported reference, tests/golden/make_golden_metrics.py): an exact hit
    exercises the small-angle branch, a sign-flipped quaternion the double cover."""
    from nvblox_mindmap_amd.diffuser_actor.loss import compute_metrics

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz"))
    for tag in ("arm", "humanoid"):
        t = lambda k: torch.from_numpy(g[f"{tag}_{k}"])  # noqa: E731
        m = compute_metrics(t("pred"), t("yaw_pred"), t("gt"), t("yaw_gt"), predict_head_yaw=True)
        names = [k[len(tag) + 3:] for k in g.files if k.startswith(f"{tag}_m_")]
        assert set(names) == set(m.keys()) and len(names) == 13
        for k in names:
            assert np.allclose(m[k].numpy(), g[f"{tag}_m_{k}"], rtol=1e-5, atol=1e-5), (tag, k, m[k], g[f"{tag}_m_{k}"])
        assert "head_yaw_error_deg" not in compute_metrics(t("pred"), None, t("gt"), None, predict_head_yaw=False)


class _MemoryDataset(torch.utils.data.Dataset):
    """Per-sample dicts cut out of synthetic batches (default collate stacks them back)."""

    def __init__(self, cfg, n, seed):
        from nvblox_mindmap_amd.training import synthetic_batch

        b = synthetic_batch(cfg, n, "cpu", num_vertices=48, seed=seed)
        self.items = [{k: v[i] for k, v in b.items() if torch.is_tensor(v)} for i in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def _loop_worker(rank, world, port, out, ckpt_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import dataclasses

    from torch.utils.data import DataLoader

    from nvblox_mindmap_amd.training import (DistributedWeightedSampler, ProcessGroup, build_optimizer, load_train_checkpoint, run_training,
                                             wrap_ddp)
    from nvblox_mindmap_amd.training.trainer import build_model

    with ProcessGroup(backend="gloo"):
        cfg = dataclasses.replace(_tiny_cfg(), diffusion_timesteps=3)
        train_set, val_set = _MemoryDataset(cfg, 12, 1), _MemoryDataset(cfg, 8, 2)
        ts = DistributedWeightedSampler(torch.ones(len(train_set)), len(train_set), seed=3)
        vs = DistributedWeightedSampler(torch.ones(len(val_set)), len(val_set), seed=4)
        train_loader, val_loader = DataLoader(train_set, batch_size=2, sampler=ts), DataLoader(val_set, batch_size=2, sampler=vs)
        assert len(train_loader) == 3  # 12 samples / 2 ranks / batch 2: the loop crosses epoch boundaries

        def fresh():
            torch.manual_seed(0)
            model = build_model(cfg, device="cpu")
            ddp = wrap_ddp(model, "cpu")
            return model, ddp, build_optimizer(ddp, lr=1e-3)

        evals = []
        model, ddp, opt = fresh()
        torch.manual_seed(50 + rank)
        done, best = run_training(cfg, ddp, opt, train_loader, val_loader, train_iters=4, val_freq=2, train_sampler=ts, validation_sampler=vs,
                                  checkpoint_dir=ckpt_dir, num_batches_per_test_eval=1, num_batches_per_train_eval=1,
                                  on_eval=lambda step, split, v: evals.append((step, split, v)))
        assert done == 4 and ddp.training
        # resume from what rank 0 wrote (every rank reads it after the loop's barrier) and carry on to iteration 6
        model2, ddp2, opt2 = fresh()
        start, best2 = load_train_checkpoint(os.path.join(ckpt_dir, "last.pth"), ddp2, opt2, initial_learning_rate=1e-3)
        same = all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), model2.state_dict().values()))
        torch.manual_seed(70 + rank)
        done2, best3 = run_training(cfg, ddp2, opt2, train_loader, val_loader, train_iters=6, val_freq=2, train_sampler=ts, validation_sampler=vs,
                                    start_iter=start, best_loss=best2, checkpoint_dir=ckpt_dir, num_batches_per_test_eval=-1)
        vec = torch.cat([p.detach().flatten() for p in model2.parameters() if p.requires_grad])
        from nvblox_mindmap_amd.training import all_gather_objects

        gathered = all_gather_objects({"evals": evals, "best": (best, best2, best3), "start": start, "done2": done2, "same": same,
                                       "checksum": float(vec.double().sum()), "lr": opt2.param_groups[0]["lr"]})
        if rank == 0:
            out.put(gathered)


def test_training_loop_two_ranks_gloo(tmp_path):
    """training.run_training on 2 ranks (gloo): iteration-based loop over epoch boundaries, evaluation in inference mode with the
    per-rank means all-gathered and averaged (run_training.py:371-375), rank 0 writes last.pth / best.pth, every rank resumes
    from it and carries on."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loop_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = g
    assert [(s, split) for s, split, _ in a["evals"]] == [(1, "train-val"), (1, "val"), (3, "train-val"), (3, "val")]
    for (_, _, va), (_, _, vb) in zip(a["evals"], b["evals"]):
        assert va == vb and np.isfinite(list(va.values())).all()  # the averaged values are the same on every rank
        assert {"mean_total_loss", "mean_pos_loss", "mean_distance_m", "mean_rot_error_deg", "mean_bias_z", "mean_head_yaw_error_deg"} <= set(va)
    assert a["best"][0] is not None and b["best"][0] is None  # rank 0 (the writer) tracks the best loss ...
    assert a["best"][1] == b["best"][1] == min(v["mean_total_loss"] for s, split, v in a["evals"] if split == "val")  # ... the file carries it
    assert a["start"] == b["start"] == 4 and a["done2"] == b["done2"] == 6 and a["same"] and b["same"]
    assert a["checksum"] == b["checksum"]  # DDP kept the ranks in step after the resume
    assert abs(a["lr"] - 7.5e-4) < 1e-12  # the ramp (1.0 -> 0.5 over 75 % of 6 iterations) restarted at the resume, 2 steps in
    assert os.path.exists(tmp_path / "best.pth") and os.path.exists(tmp_path / "last.pth")


def _graphed_loop_worker(rank, world, port, out, ckpt_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import dataclasses

    from torch.utils.data import DataLoader

    from nvblox_mindmap_amd.training import (DistributedWeightedSampler, GraphedTrainStep, ProcessGroup, all_gather_objects, build_optimizer,
                                             load_train_checkpoint, run_training, wrap_ddp)
    from nvblox_mindmap_amd.training.trainer import build_model

    with ProcessGroup(backend="gloo"):
        cfg = dataclasses.replace(_tiny_cfg(), diffusion_timesteps=3)
        train_set, val_set = _MemoryDataset(cfg, 12, 1), _MemoryDataset(cfg, 8, 2)

        def loaders():
            ts = DistributedWeightedSampler(torch.ones(len(train_set)), len(train_set), seed=3)
            vs = DistributedWeightedSampler(torch.ones(len(val_set)), len(val_set), seed=4)
            return DataLoader(train_set, batch_size=2, sampler=ts), DataLoader(val_set, batch_size=2, sampler=vs), ts, vs

        def run(graphed: bool, directory):
            torch.manual_seed(0)
            model = build_model(cfg, device="cpu")
            tl, vl, ts, vs = loaders()
            evals = []
            if graphed:
                step = GraphedTrainStep(cfg, model, next(iter(tl)), lr=1e-3, use_graphs=False)
                wrapped, opt = model, step
            else:
                step, wrapped = None, wrap_ddp(model, "cpu")
                opt = build_optimizer(wrapped, lr=1e-3)
            torch.manual_seed(50 + rank)
            done, best = run_training(cfg, wrapped, opt, tl, vl, train_iters=5, val_freq=2, train_sampler=ts, validation_sampler=vs,
                                      checkpoint_dir=directory, num_batches_per_test_eval=1, graphed=step,
                                      on_eval=lambda s_, split, v: evals.append((s_, v["mean_total_loss"])))
            return model, opt, evals, done

        ref, _, ref_evals, _ = run(False, os.path.join(ckpt_dir, "ddp"))
        got, step, got_evals, done = run(True, os.path.join(ckpt_dir, "flat"))
        same = all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), got.state_dict().values()))
        # resume the captured-step run from its own checkpoint (written by rank 0 after iteration 4 = the second evaluation)
        torch.manual_seed(0)
        model2 = build_model(cfg, device="cpu")
        tl, vl, ts, vs = loaders()
        step2 = GraphedTrainStep(cfg, model2, next(iter(tl)), lr=1e-3, use_graphs=False)
        start, best = load_train_checkpoint(os.path.join(ckpt_dir, "flat", "last.pth"), model2, step2, initial_learning_rate=1e-3)
        gathered = all_gather_objects({"same": same, "evals": (ref_evals, got_evals), "done": done, "start": start, "lr": step2.lr,
                                       "moments": float(sum(v["exp_avg"].abs().sum() for v in step2.optimizer.state.values()))})
        if rank == 0:
            out.put(gathered)


def test_training_loop_on_the_flat_step_equals_the_ddp_loop(tmp_path):
    """training.run_training(graphed=GraphedTrainStep) on 2 gloo ranks: the iteration loop with its read-ahead, the LinearLR ramp, the
    evaluations and rank 0's checkpoints gives the weights and the evaluation values of the reference-shaped loop (DDP + AdamW over
    the individual parameters), bit for bit; its checkpoint resumes (weights, Adam moments, iteration)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graphed_loop_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in g:
        assert r["same"], "weights after 5 iterations differ between the flat step and DDP"
        assert r["evals"][0] == r["evals"][1] and [s for s, _ in r["evals"][0]] == [1, 3]
        assert r["done"] == 5 and r["start"] == 4 and r["lr"] == 1e-3 and r["moments"] > 0.0


def test_training_checkpoint_resume(tmp_path):
    """save_checkpoint / load_train_checkpoint (model_utils/checkpoint.py:30-52,117-136): resuming reproduces the run that
    never stopped, bit for bit; best.pth only moves when the validation loss improves."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import build_model, build_optimizer, load_train_checkpoint, save_checkpoint

    from nvblox_mindmap_amd.training import synthetic_batch, train_one_step

    cfg = DiffuserActorConfig(data_type="mesh", feature_dim=24)

    def step(model, opt, seed):
        torch.manual_seed(seed)  # noise / timestep draws inside the model
        return float(train_one_step(cfg, model, opt, synthetic_batch(cfg, 2, "cpu", num_vertices=64, seed=seed))[0])

    torch.manual_seed(0)
    a = build_model(cfg, device="cpu")
    oa = build_optimizer(a)
    for s in range(2):
        step(a, oa, s)
    best = save_checkpoint(str(tmp_path), a, oa, step_id=1, new_loss=0.5, best_loss=None)
    assert best == 0.5 and (tmp_path / "best.pth").exists() and (tmp_path / "last.pth").exists()
    torch.manual_seed(123)
    b = build_model(cfg, device="cpu")  # different initial weights
    ob = build_optimizer(b)
    start, best_b = load_train_checkpoint(str(tmp_path / "last.pth"), b, ob, initial_learning_rate=1e-4)
    assert start == 2 and best_b == 0.5
    la, lb = step(a, oa, 7), step(b, ob, 7)
    assert la == lb
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    mtime = (tmp_path / "best.pth").stat().st_mtime_ns
    assert save_checkpoint(str(tmp_path), a, oa, step_id=2, new_loss=0.9, best_loss=0.5) == 0.5
    assert (tmp_path / "best.pth").stat().st_mtime_ns == mtime  # worse loss: best.pth untouched
    # a DDP-wrapped save ("module." names) loads into a bare model
    wrapped = {"weight": {"module." + k: v for k, v in a.state_dict().items()}, "iter": 5}
    torch.save(wrapped, tmp_path / "ddp.pth")
    assert load_train_checkpoint(str(tmp_path / "ddp.pth"), b, ob)[0] == 5


def test_checkpoint_of_the_other_kind_of_step_is_refused_by_name(tmp_path):
    """Round-4 advisor finding: both loops write last.pth / best.pth; a flat-buffer step's optimizer state loaded into a per-parameter
    AdamW failed with KeyError 'param_groups' (and the other way round with a layout error) at resume time.  Now: a clear message."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import GraphedTrainStep, build_model, build_optimizer, load_train_checkpoint, save_checkpoint, synthetic_batch

    cfg = DiffuserActorConfig(data_type="mesh", feature_dim=16, embedding_dim=48, num_attn_heads=4, diffusion_timesteps=4)
    batch = synthetic_batch(cfg, 2, "cpu", num_vertices=64, seed=0)
    torch.manual_seed(0)
    m = build_model(cfg, device="cpu")
    g = GraphedTrainStep(cfg, m, batch, lr=1e-3, use_graphs=False, data_parallel=False)
    g.step(batch)
    save_checkpoint(str(tmp_path / "flat"), m, g, 0, 1.0, None)
    assert torch.load(str(tmp_path / "flat" / "last.pth"), weights_only=True)["optimizer"]["format"] == "flat-v1"
    m2 = build_model(cfg, device="cpu")
    with pytest.raises(ValueError, match="GraphedTrainStep"):
        load_train_checkpoint(str(tmp_path / "flat" / "last.pth"), m2, build_optimizer(m2))
    # the same kind resumes; the eager flat step takes only the moments and step counts from the file
    g2 = GraphedTrainStep(cfg, m2, batch, lr=1e-3, use_graphs=False, data_parallel=False)
    assert load_train_checkpoint(str(tmp_path / "flat" / "last.pth"), m2, g2, initial_learning_rate=1e-3)[0] == 1
    torch.manual_seed(3)
    la = g.step(batch).clone()
    torch.manual_seed(3)
    lb = g2.step(batch).clone()
    assert torch.equal(la, lb) and all(torch.equal(p, q) for p, q in zip(m.parameters(), m2.parameters()))
    opt = build_optimizer(m)
    save_checkpoint(str(tmp_path / "plain"), m, opt, 0, 1.0, None)
    with pytest.raises(ValueError, match="per-parameter AdamW"):
        load_train_checkpoint(str(tmp_path / "plain" / "last.pth"), m2, g2)
