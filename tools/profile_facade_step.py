#!/usr/bin/env python3
"""Where does the host time of ONE closed-loop fusion step go?  (0.66 ms through the facade vs 0.11 ms of
kernels.)  facade.decay() + facade.update_reconstruction_from_sample(sample, "pov") at the reference's shape (512 x 512, 768 feature
channels, backbone output handed over), `--steps` times:
  * wall time per step with a synchronise after every step (what bench.py's closed_loop.breakdown_ms.fusion measures) and without
    (host enqueue only);
  * cProfile of the same loop: top functions by cumulative and by own time -> stdout / --out.
Reference: mindmap/mapping/isaaclab_nvblox_mapper.py:96-165, mapping/helpers/nvblox_input_helpers.py:18-82."""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def closed_loop(args):
    """Phases of the fusion step inside the closed loop (cold host caches, a GPU that has just drained a 1 400-node graph)."""
    import nvblox_mindmap_amd.mapping.isaaclab_nvblox_mapper as F
    from benchlib.fusion_legs import build_facade
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig
    from nvblox_mindmap_amd.image_processing.backprojection import get_camera_pointcloud
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID
    from nvblox_mindmap_amd.training import build_model, synthetic_batch

    dev = torch.device("cuda", 0)
    cfg, C, frames, samples, ex, facade = build_facade(args.shape, dev, 4)
    if args.pipelining >= 0:
        facade.set_frame_pipelining(bool(args.pipelining))
    pcfg = DiffuserActorConfig()
    torch.manual_seed(0)
    model = build_model(pcfg, device=dev).eval()
    DiffuserActor.enable_fused_inference(True)
    model.enable_graph_sampling(True)
    hist = synthetic_batch(pcfg, 1, dev, seed=3)["gripper_history"]
    acc = {}

    def timed(name, fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
            return r
        return w

    F.frame_inputs_from_sample = timed("frame_inputs_from_sample", F.frame_inputs_from_sample)
    F.nvblox_integrate = timed("nvblox_integrate", F.nvblox_integrate)
    n = max(args.steps // 10, 8)

    probe = torch.zeros(64, device=dev)

    def control_step(i, between="policy"):
        with_policy = between in ("policy", "policy_poll", "policy_probe", "policy_nograph") or between.startswith("policy_spin")
        if between.startswith("policy_spin"):  # a busy host for N us after the inference, outside the fusion timers: is the cost transient?
            t_end = time.perf_counter() + float(between[len("policy_spin"):]) * 1e-6
            while time.perf_counter() < t_end:
                pass
        if between == "policy_probe":  # ONE trivial launch + wait before the fusion step: does the FIRST launch after the inference pay?
            p0 = time.perf_counter()
            probe.add_(1.0)
            p1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            p2 = time.perf_counter()
            acc["probe_launch_host"] = acc.get("probe_launch_host", 0.0) + p1 - p0
            acc["probe_wait"] = acc.get("probe_wait", 0.0) + p2 - p1
        fr, smp = frames[i % 4], samples[i % 4]
        ex.next, ex.low = fr["features"], fr["lowres"]
        t0 = time.perf_counter()
        facade.decay()
        t1 = time.perf_counter()
        facade.update_reconstruction_from_sample(smp, "pov")
        t2 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        acc["decay"] = acc.get("decay", 0.0) + t1 - t0
        acc["update_reconstruction_from_sample"] = acc.get("update_reconstruction_from_sample", 0.0) + t2 - t1
        acc["final_synchronize"] = acc.get("final_synchronize", 0.0) + t3 - t2
        acc["fusion_total"] = acc.get("fusion_total", 0.0) + t3 - t0
        if between == "sleep":  # what a blocked host thread looks like without any GPU work
            time.sleep(0.024)
        elif between == "spin":  # 24 ms of busy host, idle GPU
            t_end = time.perf_counter() + 0.024
            while time.perf_counter() < t_end:
                pass
        if not with_policy:
            return
        inp = facade.get_nvblox_model_inputs(MAPPER_TO_ID.STATIC, remove_zero_features=True)
        pcd = get_camera_pointcloud(smp["intrinsics"][0], smp["depths"][0], smp["camera_poses"][0, :, :3], smp["camera_poses"][0, :, 3:])
        with torch.no_grad():
            model(None, None, smp["rgbs"], pcd[:, None], (smp["depths"] > 0), inp["vertex_features"], inp["vertices"],
                  inp["vertices_valid_mask"], None, hist, run_inference=True)
        if between == "policy_poll":  # wait for the inference by polling an event: the host thread never blocks in the driver
            ev = torch.cuda.Event()
            ev.record()
            while not ev.query():
                pass
        else:
            torch.cuda.synchronize(dev)

    lines = []
    def graph_mode(on):
        model.enable_graph_sampling(on)

    for label, with_policy in (("closed loop (policy inference between fusion steps; torch.cuda.synchronize)", "policy"),
                               ("closed loop, the inference WITHOUT the captured HIP graph (same kernels launched one by one)", "policy_nograph"),
                               ("closed loop, the inference awaited by polling an event", "policy_poll"),
                               ("closed loop, one trivial launch + wait before each fusion step", "policy_probe"),
                               ("closed loop, 200 us of busy host between the inference and the fusion step", "policy_spin200"),
                               ("closed loop, 1 ms of busy host between the inference and the fusion step", "policy_spin1000"),
                               ("closed loop, 5 ms of busy host between the inference and the fusion step", "policy_spin5000"),
                               ("24 ms time.sleep between fusion steps (no GPU work)", "sleep"),
                               ("24 ms busy host loop between fusion steps (no GPU work)", "spin"),
                               ("fusion steps back to back", "none"),
                               ("closed loop again (the first mode, after everything else has run: is the first figure a warm-up effect?)", "policy")):
        graph_mode(with_policy != "policy_nograph")
        for i in range(3):
            control_step(i, with_policy)
        import gc

        gc.collect()
        gc.freeze()
        acc.clear()
        for i in range(n):
            control_step(3 + i, with_policy)
        lines.append(f"---- {label}: {n} steps, ms per step ----")
        for k, v in acc.items():
            lines.append(f"  {k:<40}{v / n * 1e3:9.4f}")
    DiffuserActor.enable_fused_inference(False)
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--shape", default="ref")
    ap.add_argument("--out", default=None)
    ap.add_argument("--pipelining", type=int, default=-1, help="-1: the facade's default; 0 / 1: set_frame_pipelining")
    ap.add_argument("--closed-loop", action="store_true", help="the fusion step as bench.py's closed-loop leg runs it: a policy inference "
                                                                "(24 ms of other work) between two fusion steps; phases timed separately")
    args = ap.parse_args()
    if args.closed_loop:
        return closed_loop(args)
    from benchlib.fusion_legs import build_facade

    dev = torch.device("cuda", 0)
    cfg, C, frames, samples, ex, facade = build_facade(args.shape, dev, 8)
    if args.pipelining >= 0:
        facade.set_frame_pipelining(bool(args.pipelining))

    def step(i):
        fr, smp = frames[i % 8], samples[i % 8]
        ex.next, ex.low = fr["features"], fr["lowres"]
        facade.decay()
        facade.update_reconstruction_from_sample(smp, "pov")

    for i in range(16):
        step(i)
    torch.cuda.synchronize(dev)
    import gc

    gc.collect()
    gc.freeze()
    lines = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
        torch.cuda.synchronize(dev)
    synced = (time.perf_counter() - t0) / args.steps * 1e3
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    enq = (time.perf_counter() - t0) / args.steps * 1e3
    torch.cuda.synchronize(dev)
    total = (time.perf_counter() - t0) / args.steps * 1e3
    lines.append(f"shape {args.shape}: per step, synchronised after each: {synced:.4f} ms; host enqueue only: {enq:.4f} ms; stream of {args.steps} steps: {total:.4f} ms")
    pr = cProfile.Profile()
    pr.enable()
    for i in range(args.steps):
        step(i)
        torch.cuda.synchronize(dev)
    pr.disable()
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
        lines.append(f"---- cProfile, {args.steps} steps, sorted by {key} ----")
        lines.append(s.getvalue())
    text = "\n".join(lines)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
