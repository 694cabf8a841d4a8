"""Child process of tests/test_gpu_rccl_single_rank.py (a fresh process: it owns the GPU from its first call, the process group
from start to end; a hang in RCCL costs the parent's timeout, not the suite).

What runs here over ``backend="nccl"`` (RCCL) with ONE rank, forced-collective mode (training/distributed.py):
  * ``GraphedTrainStep``: the weight broadcast, the 1 / world scale + all-reduce of the flat gradient between the two captured HIP
    graphs, ``observed_world()`` -- against the plain captured step (no process group) from the same seed: bit for bit;
  * ``wrap_ddp`` (DistributedDataParallel, find_unused_parameters=True, mindmap/run_training.py:608-613) + ``train_one_step`` against
    the bare model's eager step;
  * ``max_over_ranks`` on a device tensor, ``barrier``, ``all_gather_objects``.
Prints ONE JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ROC_AQL_QUEUE_SIZE", "65536")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import (GraphedTrainStep, ProcessGroup, all_gather_objects, barrier, build_model, build_optimizer,
                                             max_over_ranks, synthetic_batch, train_one_step, wrap_ddp)
    from nvblox_mindmap_amd.training.distributed import collectives_active, force_collectives

    full = len(sys.argv) > 1 and sys.argv[1] == "full"
    if full:  # the benchmark's shapes (batch 32, 512 x 512, 2 048 vertices): the 10.9 MB payload
        cfg, B, V = DiffuserActorConfig(), 32, 2048
    else:
        cfg, B, V = DiffuserActorConfig(data_type="rgbd_and_mesh", image_size=(128, 128), feature_dim=768), 2, 256
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    batches = [synthetic_batch(cfg, B, dev, num_vertices=V, seed=i) for i in range(3)]
    n_steps = 4

    def graphed_run(**kw):
        torch.manual_seed(0)
        model = build_model(cfg, device=dev)
        g = GraphedTrainStep(cfg, model, batches[0], lr=1e-3, **kw)
        g.initial = g.flat_param.clone()
        torch.manual_seed(1)
        losses = [g.step(batches[i % 3], batches[(i + 1) % 3]).clone() for i in range(n_steps)]
        torch.cuda.synchronize(dev)
        return g, torch.stack(losses), g.flat_param.clone()

    def eager_run(wrap):
        torch.manual_seed(0)
        model = build_model(cfg, device=dev)
        m = wrap_ddp(model, dev) if wrap else model
        opt = build_optimizer(m, lr=1e-3)
        torch.manual_seed(1)
        losses = [torch.stack(list(train_one_step(cfg, m, opt, batches[i % 3]))).clone() for i in range(2)]
        torch.cuda.synchronize(dev)
        return m, torch.stack(losses), torch.cat([p.detach().reshape(-1) for p in model.parameters() if p.requires_grad])

    assert not collectives_active()
    g0, l0, w0 = graphed_run()
    assert not g0.collective and g0.observed_world() == 1
    _, el0, ew0 = eager_run(False)

    out = {}
    t0 = time.perf_counter()
    with ProcessGroup(backend="nccl", force=True):
        out["init_s"] = time.perf_counter() - t0
        assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
        assert force_collectives() and collectives_active()
        out["backend"] = dist.get_backend()
        g1, l1, w1 = graphed_run()
        assert g1.collective and g1.world == 1 and g1.graph_fb is not None and g1.graph_opt is not None
        out["observed_world"] = g1.observed_world()
        out["graphed_losses_equal"] = bool(torch.equal(l0, l1))
        out["graphed_weights_equal"] = bool(torch.equal(w0, w1))
        out["graphed_max_weight_diff"] = float((w0 - w1).abs().max())
        out["weights_moved_by"] = float((w1 - g1.initial).abs().max())  # (the steps did train: not a comparison of two no-ops)
        # the all-reduce's own duration, HIP events on the issuing stream
        g1.time_allreduce = True
        for i in range(6):
            g1.step(batches[i % 3], batches[(i + 1) % 3])
        ms = g1.collect_allreduce_ms()
        out["allreduce_ms"] = {"mean": sum(ms) / len(ms), "min": min(ms), "max": max(ms), "payload_MB": g1.flat_grad.numel() * 4 / 1e6}
        # the benchmark's timing reductions
        barrier()
        out["max_over_ranks_device"] = max_over_ranks(1.25, dev)
        out["all_gather_objects"] = all_gather_objects({"rank": dist.get_rank()})
        # the reference-shaped wrapper over the same one-rank group
        m, el1, ew1 = eager_run(True)
        out["ddp_wrapped"] = type(m).__name__
        out["ddp_losses_equal"] = bool(torch.equal(el0, el1))
        out["ddp_max_loss_diff"] = float((el0 - el1).abs().max())
        out["ddp_max_weight_diff"] = float((ew0 - ew1).abs().max())
        barrier()
    assert not dist.is_initialized() and not collectives_active()
    out["shapes"] = "bench" if full else "small"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
