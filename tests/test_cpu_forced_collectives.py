"""Forced-collective mode on the CPU (gloo, ONE rank): the training path issues its collectives over a one-rank group and the step is
unchanged bit for bit; outside the mode (and after the group is gone) nothing is issued.  The RCCL twin of this test is
tests/test_gpu_rccl_single_rank.py.  Reference: mindmap/model_utils/multi_gpu.py:21-34, mindmap/run_training.py:608-613."""
import os

import torch
import torch.distributed as dist

from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
from nvblox_mindmap_amd.training import (GraphedTrainStep, ProcessGroup, all_gather_objects, barrier, build_model, max_over_ranks,
                                         synthetic_batch, wrap_ddp)
from nvblox_mindmap_amd.training.distributed import collectives_active, force_collectives


def _run(cfg, batches, **kw):
    torch.manual_seed(0)
    model = build_model(cfg, device="cpu")
    g = GraphedTrainStep(cfg, model, batches[0], lr=1e-3, **kw)
    torch.manual_seed(1)
    losses = torch.stack([g.step(batches[i % 2]).clone() for i in range(3)])
    return g, losses, g.flat_param.clone()


def test_one_rank_group_with_forced_collectives_changes_nothing(monkeypatch):
    monkeypatch.delenv("MMF_FORCE_COLLECTIVES", raising=False)
    cfg = DiffuserActorConfig(data_type="mesh", feature_dim=64)
    batches = [synthetic_batch(cfg, 2, "cpu", num_vertices=64, seed=i) for i in range(2)]
    launcher = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")
    for k in launcher:
        monkeypatch.delenv(k, raising=False)
    g0, l0, w0 = _run(cfg, batches)
    assert not g0.collective and not collectives_active() and not force_collectives()
    with ProcessGroup(backend="gloo", force=True):
        assert dist.is_initialized() and dist.get_world_size() == 1 and collectives_active()
        g1, l1, w1 = _run(cfg, batches)
        assert g1.collective and g1.world == 1 and g1.observed_world() == 1
        assert torch.equal(l0, l1) and torch.equal(w0, w1)
        # a step that was told to stand alone issues nothing, forced mode or not
        g2, l2, w2 = _run(cfg, batches, data_parallel=False)
        assert not g2.collective and torch.equal(l0, l2)
        assert max_over_ranks(2.5) == 2.5 and all_gather_objects("x") == ["x"]
        barrier()
        assert type(wrap_ddp(build_model(cfg, device="cpu"), "cpu")).__name__ == "DistributedDataParallel"
    assert not dist.is_initialized() and not collectives_active() and not force_collectives()
    assert not any(k in os.environ for k in launcher)  # the one-rank rendezvous leaves no launcher variables behind
    assert wrap_ddp(g0.model, "cpu") is g0.model


def test_force_without_a_group_is_an_error():
    cfg = DiffuserActorConfig(data_type="mesh", feature_dim=64)
    batch = synthetic_batch(cfg, 2, "cpu", num_vertices=64, seed=0)
    torch.manual_seed(0)
    model = build_model(cfg, device="cpu")
    try:
        GraphedTrainStep(cfg, model, batch, force_collectives=True)
    except RuntimeError as e:
        assert "process group" in str(e)
    else:
        raise AssertionError("expected a RuntimeError")
