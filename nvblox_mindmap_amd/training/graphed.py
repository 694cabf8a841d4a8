"""The policy's training step with the host taken off the critical path (config 5 of BASELINE.json; the data-parallel half of
the north star).

``train_one_step`` + ``wrap_ddp`` (trainer.py) is the reference-shaped step (mindmap/run_training.py:155-217,608-613): eager
PyTorch, DistributedDataParallel with ``find_unused_parameters=True``.  On MI355X that step is HOST-bound: ~6 000 kernel launches
under the interpreter lock, DDP's per-step graph traversal for unused parameters, ~150 gradient hooks.  This module is the same
arithmetic arranged for the machine:

  * every trainable parameter that receives a gradient lives in ONE flat float32 buffer (``flat_param``), its gradient in a
    second one of the same layout (``flat_grad``, 10.9 MB for the reference model); ``p.data`` / ``p.grad`` are views.  The
    parameters the reference model never uses at ``use_instruction = 0`` (instruction encoder, vision-language attention, goal
    embedding: why the reference needs ``find_unused_parameters``) are found ONCE by a probe backward and stay out of the buffers
    and of the optimizer -- exactly what DDP + AdamW do with a ``None`` gradient: nothing;
  * forward + backward is ONE captured HIP graph (static shapes: batch 32, 2 048 vertices); the frozen backbone of the NEXT batch
    is a second graph replayed beside it on another stream (it is frozen: its output does not depend on this step's update).  Both
    are single-stream graphs without memory nodes: measured on this runtime, a graph that forks streams or holds a captured
    hipMallocAsync is walked node by node from the HOST by hipGraphLaunch (tools/microbench/graph_block_probe.py);
  * the data-parallel exchange is ONE explicit ``all_reduce`` of ``flat_grad`` over RCCL (pre-scaled by 1 / world like DDP's
    bucket) between the two graphs -- no bucketing, no hooks;
  * AdamW (the reference's two groups: no weight decay for names containing "bias" / "LayerNorm.*", run_training.py:140-153) is
    ``torch.optim.AdamW`` over the two flat segments, captured as a second graph (``capturable=True``: step count and learning
    rate live on the device).

On a CPU (the gloo tests) or with ``use_graphs=False`` the same buffers, the same explicit all-reduce and the same optimizer run
eagerly: tests/test_cpu_policy.py checks that path against ``wrap_ddp`` + ``build_optimizer`` at world size 2 (same weights after
N steps), tests/test_gpu_policy.py the captured graphs against the eager step on one GPU.
"""
import time
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..diffuser_actor import DiffuserActorConfig
from .distributed import collectives_active, get_world_size
from .trainer import unpack_batch

AQL_QUEUE_PACKETS = 65536


def configure_runtime_for_graphs() -> None:
    """Call BEFORE the process touches the GPU (the HIP runtime reads its flags when it initialises).  The captured step is a
    graph of ~2 400 kernel nodes, several AQL packets each: with the runtime's default ring of 16 384 packets, launching step
    t + 1 while step t is still running blocks the host inside hipGraphLaunch until the ring has room (measured on MI355X: 26 ms of
    a 73 ms step; 0.7 ms with ROC_AQL_QUEUE_SIZE = 65536, tools/microbench/graph_block_probe.py and DESIGN.md section 7)."""
    import os

    os.environ.setdefault("ROC_AQL_QUEUE_SIZE", str(AQL_QUEUE_PACKETS))


TUNED_GEMMS_FILE = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "tunableop_gfx950.csv")


def enable_tuned_gemms(path: str = TUNED_GEMMS_FILE) -> bool:
    """Let PyTorch's TunableOp pick, per GEMM shape, the hipBLASLt / rocBLAS solution recorded in ``path`` (tuned once on an
    MI355X with tools/tune_train_gemms.sh; no tuning at run time).  The trainable half of the policy is full of tall-skinny float32
    GEMMs (98 304 x 120 -> 240 key / value projections, 65 536 x 768 -> 120 vertex embeddings) for which the libraries' default
    heuristic picks kernels 5 - 7x slower than their best (0.60 -> 0.08 ms for the projection; 73.0 -> 68.1 ms per step).  The
    file's validators (PyTorch, HIP, hipBLASLt, rocBLAS versions, gfx950) must match the running stack: otherwise it is ignored
    with torch's warning and the library defaults stay.  Returns whether TunableOp is on with entries loaded."""
    import os

    if not (torch.cuda.is_available() and os.path.exists(path)):
        return False
    import torch.cuda.tunable as tunable

    if os.environ.get("PYTORCH_TUNABLEOP_TUNING", "0") == "1":  # a tuning run (tools/tune_train_gemms.sh): the environment rules
        return tunable.is_enabled()
    tunable.enable(True)
    tunable.tuning_enable(False)
    tunable.record_untuned_enable(False)
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(False)  # (read-only use: no results file written at exit, whatever the rank count)
    try:
        ok = bool(tunable.read_file(path))
    except Exception:
        ok = False
    return ok and len(tunable.get_results()) > 0


_NO_DECAY = ("bias", "LayerNorm.weight", "LayerNorm.bias")  # run_training.py:140-153


def _forward_losses(cfg, model, s, backbone_feats=None):
    losses, _, _ = model(s["gt_gripper_pred"], s["gt_head_yaw"], s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"],
                         s["vertices"], s["vertices_valid_mask"], None, s["gripper_history"], backbone_feats=backbone_feats)
    return losses


class GraphedTrainStep:
    """``step(batch[, next_batch]) -> losses`` (total, position, rotation, gripper, head yaw; a float32 [5] tensor that the next
    step overwrites -- clone it to keep it).

    ``model`` is the BARE policy (not DDP-wrapped) on its device, every rank starting from the same weights.  ``example_batch``: a
    loader-shaped batch (``synthetic_batch`` / the dataset's collate output) of the shapes every later batch has.
    ``overlap_backbone``: evaluate the frozen image backbone of ``next_batch`` beside the trainable pass of ``batch`` (graph
    branch / side stream); the caller then hands batch t + 1 to step t, and the SAME dict object to step t + 1.
    ``data_parallel=False``: this rank steps ALONE although a process group exists (a measurement or an evaluation that only one
    rank runs): no collective is issued -- with the default every rank of the group must call ``step`` the same number of times.
    ``force_collectives``: issue the collectives (broadcast, gradient scale + all-reduce, ``observed_world``) although the group has ONE
    rank -- None follows ``training.distributed.force_collectives()`` (MMF_FORCE_COLLECTIVES=1).  The step's results do not change
    (a sum over one rank, a scale by 1.0); it is how RCCL (``backend="nccl"``) is exercised on a one-GPU box
    (tests/test_gpu_rccl_single_rank.py)."""

    def __init__(self, cfg: DiffuserActorConfig, model: nn.Module, example_batch: Dict[str, torch.Tensor], lr: float = 1e-4,
                 weight_decay: float = 5e-4, use_graphs: Optional[bool] = None, overlap_backbone: bool = True,
                 unpack: Optional[Callable] = None, process_group=None, tuned_gemms: bool = True, data_parallel: bool = True,
                 force_collectives: Optional[bool] = None):
        self.cfg, self.model = cfg, model
        self.unpack = unpack or unpack_batch
        self.group = process_group
        if not data_parallel:
            self.world = 1
            self.collective = False
        else:
            self.world = get_world_size() if process_group is None else dist.get_world_size(process_group)
            have_group = dist.is_available() and dist.is_initialized()
            forced = collectives_active() if force_collectives is None else (bool(force_collectives) and have_group)
            if force_collectives and not have_group:
                raise RuntimeError("force_collectives=True needs an initialised process group (training.ProcessGroup(force=True))")
            self.collective = self.world > 1 or forced  # the exchange steps are issued (always when there is someone to exchange with)
        p0 = next(model.parameters())
        self.device = p0.device
        self.use_graphs = (self.device.type == "cuda") if use_graphs is None else bool(use_graphs)
        if self.use_graphs and self.device.type != "cuda":
            raise ValueError("HIP graphs need the model on a GPU")
        enc = getattr(model, "encoder", None)
        self.has_backbone = bool(enc is not None and getattr(enc, "uses_images", False) and getattr(enc, "backbone", None) is not None)
        self.overlap = bool(overlap_backbone and self.has_backbone and self.device.type == "cuda")
        self.static = {k: v.to(self.device).clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        self.steps_done = 0
        self.host_enqueue_s = 0.0       # host wall time spent inside step() (enqueue only: step() never synchronises)
        self.host_cpu_s = 0.0           # CPU time of the calling thread inside step()
        self._ev = None                 # (start, end) HIP events around the last all-reduce when timing is on
        self.time_allreduce = False
        self.allreduce_ms: List[float] = []
        self._primed = None             # the batch dict whose backbone features are in ``self.feats`` (held: ids are not reused)

        self.tuned_gemms = bool(tuned_gemms and self.device.type == "cuda" and enable_tuned_gemms())
        model.train()
        self._find_used_parameters()
        self._flatten(lr, weight_decay)
        if self.collective:
            # what DistributedDataParallel does in its constructor: every replica starts from rank 0's weights (a per-rank seed or a
            # checkpoint loaded on rank 0 only would otherwise diverge silently -- the all-reduced gradients applied to different weights)
            dist.broadcast(self.flat_param, src=0 if process_group is None else dist.get_global_rank(process_group, 0), group=process_group)
        self.losses = torch.zeros(5, dtype=torch.float32, device=self.device)
        self.feats = self.next_feats = self.next_rgbs = None
        if self.overlap:
            self.next_rgbs = self.static["rgbs"].clone()
            # (Stream priorities were tried and dropped: the trainable chain on a high-priority queue slows the backbone's GEMMs so much
            # that a step takes 100 ms instead of 60; the backbone on the high-priority queue changes nothing.  DESIGN.md section 7.)
            self._side = torch.cuda.Stream(device=self.device)
        self.graph_fb = self.graph_opt = self.graph_bb = None
        if self.use_graphs:
            import os
            import warnings

            if int(os.environ.get("ROC_AQL_QUEUE_SIZE", "16384")) < AQL_QUEUE_PACKETS:
                warnings.warn("ROC_AQL_QUEUE_SIZE is below 65536: hipGraphLaunch of the captured step will block the host while the previous "
                              "step runs (call training.configure_runtime_for_graphs() before the first GPU call, or export it)")
            self._capture()

    # -- layout ---------------------------------------------------------------------------------------------------------------
    def _find_used_parameters(self) -> None:
        """One probe forward + backward: the trainable parameters that receive a gradient.  The set is a property of the model's
        configuration (static control flow), not of the data.  The caller's RNG streams are left untouched."""
        devs = [self.device] if self.device.type == "cuda" else []
        with torch.random.fork_rng(devices=devs):
            self.model.zero_grad(set_to_none=True)
            s = self.unpack(self.cfg, self.static)
            _forward_losses(self.cfg, self.model, s)[0].backward()
        named = [(n, p) for n, p in self.model.named_parameters() if p.requires_grad]
        self.unused_names = [n for n, p in named if p.grad is None]
        used = [(n, p) for n, p in named if p.grad is not None]
        self.model.zero_grad(set_to_none=True)
        # the reference's two AdamW groups, no-decay first (trainer.build_optimizer keeps the same rule and order)
        nd = [(n, p) for n, p in used if any(k in n for k in _NO_DECAY)]
        de = [(n, p) for n, p in used if not any(k in n for k in _NO_DECAY)]
        self.used = nd + de
        self.n_no_decay = sum(p.numel() for _, p in nd)
        self.n_total = sum(p.numel() for _, p in self.used)

    def _flatten(self, lr: float, weight_decay: float) -> None:
        dev = self.device
        self.flat_param = torch.empty(self.n_total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        off = 0
        self.layout = []
        self.grad_views = []
        with torch.no_grad():
            for name, p in self.used:
                n = p.numel()
                assert p.dtype == torch.float32, name
                self.flat_param[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                self.grad_views.append(p.grad)
                self.layout.append((name, off, n))
                off += n
        a = self.n_no_decay
        self.seg_params = [nn.Parameter(self.flat_param[:a], requires_grad=False), nn.Parameter(self.flat_param[a:], requires_grad=False)]
        self.seg_params[0].grad, self.seg_params[1].grad = self.flat_grad[:a], self.flat_grad[a:]
        groups = [{"params": [self.seg_params[0]], "weight_decay": 0.0}, {"params": [self.seg_params[1]], "weight_decay": weight_decay}]
        groups = [g for g in groups if g["params"][0].numel() > 0]
        if self.use_graphs:
            self.lr_tensor = torch.tensor(float(lr), dtype=torch.float32, device=dev)
            self.optimizer = torch.optim.AdamW(groups, lr=self.lr_tensor, capturable=True, foreach=True)
            for g in self.optimizer.param_groups:
                g["lr"] = self.lr_tensor  # ONE device scalar for both groups: set_lr() is one fill
        else:
            self.lr_tensor = None
            self.optimizer = torch.optim.AdamW(groups, lr=float(lr))
        self.lr = float(lr)

    # -- the pieces of a step (captured or eager) --------------------------------------------------------------------------------
    def _backbone_next(self) -> None:
        """self.next_feats <- frozen backbone of self.next_rgbs (its own graph, replayed on the side stream: two single-stream
        graphs side by side instead of one forked graph -- hipGraphLaunch walks a forked graph from the host, blocking it)."""
        out = self.model.encoder.backbone_features(self.next_rgbs)
        if self.next_feats is None:
            self.next_feats = out
        else:
            self.next_feats.copy_(out)

    def _forward_backward(self) -> None:
        """flat_grad <- d loss / d parameters of the batch in ``self.static`` (scaled by 1 / world), ``self.losses`` <- the
        losses.  With ``overlap`` the image tokens come from ``self.feats`` (the backbone's output for this batch)."""
        # autograd HANDS OVER a parameter's gradient when the parameter has none yet (no kernel); into existing .grad views it would
        # accumulate -- one small add_ per parameter tensor (177 launches) plus the buffer's memset.  So: backward into fresh
        # tensors, then ONE multi-tensor copy into the flat buffer's views (every view is overwritten: no zeroing; the same values
        # as 0 + g).
        for _, p in self.used:
            p.grad = None
        s = self.unpack(self.cfg, self.static)
        feats = self.feats if self.overlap else None
        losses = _forward_losses(self.cfg, self.model, s, backbone_feats=feats)
        losses[0].backward()
        with torch.no_grad():
            got = [(v, p.grad) for v, (_, p) in zip(self.grad_views, self.used)]
            torch._foreach_copy_([v for v, g in got if g is not None], [g for _, g in got if g is not None])
            for v, g in got:
                if g is None:
                    v.zero_()  # (a parameter the probe saw a gradient for: not expected)
            for v, (_, p) in zip(self.grad_views, self.used):
                p.grad = v
        if self.collective:
            self.flat_grad.mul_(1.0 / self.world)  # DDP scales the bucket before its all-reduce (sum)
        with torch.no_grad():
            zero = losses[0].detach().new_zeros(())
            self.losses.copy_(torch.stack([zero if x is None else x.detach().to(torch.float32) for x in losses]))

    def _optimizer_step(self) -> None:
        self.optimizer.step()

    def _capture(self) -> None:
        dev = self.device
        if self.overlap:
            self.feats = self.model.encoder.backbone_features(self.static["rgbs"]).clone()
        # Warm-up on a side stream (lazy initialisation of the GEMM library, the split-weight caches, autograd's buffers), as
        # torch's whole-network capture recipe asks; forward + backward only: no weight changes before the first real step.
        warm = torch.cuda.Stream(device=dev)
        warm.wait_stream(torch.cuda.current_stream(dev))
        with torch.random.fork_rng(devices=[dev]):
            with torch.cuda.stream(warm):
                for _ in range(2):
                    self._forward_backward()
                # optimizer state is created by its first step(): take that step with a zero gradient and a zero learning rate (no
                # parameter moves, the moments stay zero), then rewind the step counters
                self.flat_grad.zero_()
                self.lr_tensor.fill_(0.0)
                self.optimizer.step()
                for st in self.optimizer.state.values():
                    st["step"].zero_()
                self.lr_tensor.fill_(self.lr)
            torch.cuda.current_stream(dev).wait_stream(warm)
            torch.cuda.synchronize(dev)
            # capture_error_mode="thread_local": a loader's pin-memory / prefetch thread may allocate pinned memory or copy while this
            # thread captures -- in the default "global" mode such a call from ANY thread invalidates the capture (seen as an
            # intermittent hipErrorStreamCaptureInvalidated when the step is built while a DataLoader is already running)
            mode = dict(capture_error_mode="thread_local")
            if self.overlap:
                self.graph_bb = torch.cuda.CUDAGraph()  # (its own memory pool: it runs BESIDE the other two graphs)
                with torch.cuda.graph(self.graph_bb, **mode):
                    self._backbone_next()
            self.graph_fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_fb, **mode):
                self._forward_backward()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph_fb.pool(), **mode):
                self._optimizer_step()
        torch.cuda.synchronize(dev)

    # -- public ------------------------------------------------------------------------------------------------------------------
    def set_lr(self, lr: float) -> None:
        """New learning rate for both groups (what a LinearLR step does): one device fill, no synchronisation."""
        self.lr = float(lr)
        if self.lr_tensor is not None:
            self.lr_tensor.fill_(self.lr)
        else:
            for g in self.optimizer.param_groups:
                g["lr"] = self.lr

    def linear_lr(self, train_iters: int = 100000, convergence_percentage: float = 0.75, end_factor: float = 0.5):
        """The reference's LinearLR ramp (run_training.py:603-611) for this step: a torch LinearLR over a host-side stand-in
        optimizer whose rate ``scheduler_step()`` copies to the device (the ramp's arithmetic stays torch's, in Python floats)."""
        self._host_param = nn.Parameter(torch.zeros(()))
        self._host_opt = torch.optim.SGD([self._host_param], lr=self.lr)
        self._sched = torch.optim.lr_scheduler.LinearLR(self._host_opt, start_factor=1.0, end_factor=end_factor,
                                                        total_iters=int(train_iters * convergence_percentage))
        return self._sched

    def scheduler_step(self) -> None:
        self._host_opt.step()  # (keeps torch's "optimizer.step() before lr_scheduler.step()" order check quiet; the parameter has no gradient)
        self._sched.step()
        self.set_lr(self._host_opt.param_groups[0]["lr"])

    def _frozen_linears(self):
        if getattr(self, "_frozen_cache", None) is None:  # (walked once: step() checks these layers' version counters every call)
            enc = getattr(self.model, "encoder", None)
            bb = getattr(enc, "backbone", None) if self.has_backbone else None
            self._frozen_cache = [m for m in bb.modules() if isinstance(m, nn.Linear)] if bb is not None else []
        return self._frozen_cache

    def refresh_frozen_weights(self) -> int:
        """The frozen backbone's weights changed after capture (a checkpoint loaded into the model with load_state_dict: the documented
        resume flow builds the step first): the captured backbone graph multiplies by fp16 SPLIT COPIES of its Linear weights whose
        addresses it holds (diffuser_actor/split_linear.py) -- they are recomputed in place, the graph stays valid.  Called by
        ``step()`` whenever a copy is stale (a version check of ~50 layers: microseconds); returns the number of layers refreshed."""
        from ..diffuser_actor import split_linear as SL

        n = 0
        for lin in self._frozen_linears():
            if SL.stale(lin):
                if not SL.refresh_in_place(lin):
                    if self.use_graphs:
                        raise RuntimeError("a frozen backbone weight changed after graph capture and its split copy cannot be refreshed in "
                                           "place (values beyond fp16's range?): build a new GraphedTrainStep")
                    continue
                n += 1
        if n:
            self._primed = None  # features computed with the old weights
        return n

    def _load(self, batch: Dict[str, torch.Tensor]) -> None:
        for k, dst in self.static.items():
            if k == "rgbs" and self.overlap:
                continue  # the trainable pass reads the backbone's OUTPUT for this batch (self.feats)
            src = batch[k]
            if src.shape != dst.shape:
                raise ValueError(f"batch[{k!r}] has shape {tuple(src.shape)}, the captured step {tuple(dst.shape)}")
            dst.copy_(src, non_blocking=True)

    def prime(self, batch: Dict[str, torch.Tensor]) -> None:
        """Backbone features of ``batch`` into the step's buffer (the first batch of a stream, or one that was not announced)."""
        with torch.no_grad():
            f = self.model.encoder.backbone_features(batch["rgbs"].to(self.device))
            if self.feats is None:
                self.feats = f.clone()
            else:
                self.feats.copy_(f)
        self._primed = batch

    def step(self, batch: Dict[str, torch.Tensor], next_batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        t0, c0 = time.perf_counter(), time.thread_time()
        if self.has_backbone:
            self.refresh_frozen_weights()
        announce = self.overlap and next_batch is not None
        if self.overlap:
            if self._primed is not batch:
                self.prime(batch)
            main = torch.cuda.current_stream(self.device)
            if announce:  # the next batch's frozen backbone, beside everything below
                self.next_rgbs.copy_(next_batch["rgbs"], non_blocking=True)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    if self.graph_bb is not None:
                        self.graph_bb.replay()
                    else:
                        with torch.no_grad():
                            self._backbone_next()
        self._load(batch)
        if self.graph_fb is not None:
            self.graph_fb.replay()
        else:
            self._forward_backward()
        if self.collective:
            if self.time_allreduce and self.device.type == "cuda":
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                dist.all_reduce(self.flat_grad, group=self.group)
                b.record()
                self._pending_events = getattr(self, "_pending_events", []) + [(a, b)]
            else:
                dist.all_reduce(self.flat_grad, group=self.group)
        if self.graph_opt is not None:
            self.graph_opt.replay()
        else:
            self._optimizer_step()
        if self.overlap:
            if announce:
                main.wait_stream(self._side)
                self.feats.copy_(self.next_feats)  # 100 MB at batch 32 (~50 us), after both sides are done with their buffers
            self._primed = next_batch  # ``self.feats`` now holds the features of ``next_batch`` (None: stale, re-primed next time)
        self.steps_done += 1
        self.host_enqueue_s += time.perf_counter() - t0
        self.host_cpu_s += time.thread_time() - c0
        return self.losses

    def collect_allreduce_ms(self) -> List[float]:
        """Durations of the timed all-reduces so far (synchronises)."""
        ev = getattr(self, "_pending_events", [])
        if ev:
            torch.cuda.synchronize(self.device)
            self.allreduce_ms += [a.elapsed_time(b) for a, b in ev]
            self._pending_events = []
        return self.allreduce_ms

    def observed_world(self) -> int:
        """The number of ranks a sum over the process group actually reaches (an all-reduce of ones on this rank's device)."""
        if not self.collective:
            return 1
        one = torch.ones(1, dtype=torch.float32, device=self.device)
        dist.all_reduce(one, group=self.group)
        return int(round(float(one.item())))

    # -- checkpoints -----------------------------------------------------------------------------------------------------------------
    def state_dict(self) -> dict:
        return {"format": "flat-v1", "optimizer": self.optimizer.state_dict(), "layout": list(self.layout), "n_no_decay": self.n_no_decay,
                "steps_done": self.steps_done, "lr": self.lr}

    def load_state_dict(self, state: dict) -> None:
        if [tuple(x) for x in state["layout"]] != [tuple(x) for x in self.layout]:
            raise ValueError("the checkpoint's flat parameter layout is not this model's")
        lr = self.lr
        saved = state["optimizer"]["state"]
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        if self.graph_opt is None:
            # eager step: only the moments and step counts are taken -- the checkpoint's param_groups may be a captured step's
            # (capturable=True, device-side step / learning rate), which an eager AdamW on this device would refuse
            for i, p in enumerate(params):
                if i in saved:
                    st = self.optimizer.state[p]
                    for k, v in saved[i].items():
                        v = torch.as_tensor(v)
                        st[k] = v.detach().to("cpu", torch.float32).clone() if k == "step" else v.detach().to(p.device, p.dtype).clone()
        else:
            # the captured optimizer graph reads and writes THESE state tensors: copy into them (load_state_dict would replace them)
            for i, p in enumerate(params):
                if i in saved:
                    for k, v in saved[i].items():
                        self.optimizer.state[p][k].copy_(torch.as_tensor(v))
        self.steps_done = int(state.get("steps_done", 0))
        self.set_lr(state.get("lr", lr))
