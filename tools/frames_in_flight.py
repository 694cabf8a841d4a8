"""N independent fusion streams on ONE MI355X: N Mapper handles, each driven by its own host thread on its own HIP stream
(the native calls release the GIL), the same decay + fused-frame step as bench.py's headline.  Aggregate frames/s.

    python3 tools/frames_in_flight.py [--n 1 2 4 8] [--steps 300]
"""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg  # noqa: E402


def run(n, frames, mcfg, channels, steps, warmup, device):
    mappers = [get_nvblox_mapper(mcfg, feature_channels=channels) for _ in range(n)]
    streams = [torch.cuda.Stream(device) for _ in range(n)]
    start = threading.Barrier(n + 1)
    done = threading.Barrier(n + 1)

    def worker(k):
        torch.cuda.set_device(device)
        with torch.cuda.stream(streams[k]):
            for i in range(warmup):
                bench.step(mappers[k], mcfg, frames[(i + 7 * k) % len(frames)])
            streams[k].synchronize()
            start.wait()
            for i in range(steps):
                bench.step(mappers[k], mcfg, frames[(warmup + i + 7 * k) % len(frames)])
            streams[k].synchronize()
            done.wait()

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(n)]
    for t in ts:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    done.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    return n * steps / dt, mappers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--switch-interval", type=float, default=2e-5)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    sys.setswitchinterval(args.switch_interval)  # the workers hand the GIL over at every native call; 5 ms (the default) convoys them
    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    frames = bench.build_stream(cfg, 100, 64, device)
    out = {}
    for n in args.n:
        fps, mappers = run(n, frames, mcfg, 64, args.steps, args.warmup, device)
        out[str(n)] = fps
        del mappers
    print(json.dumps({"frames_in_flight_aggregate_fps": out}))


if __name__ == "__main__":
    main()
