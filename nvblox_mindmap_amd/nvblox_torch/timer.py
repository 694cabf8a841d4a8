"""nvblox_torch.timer: global named accumulating timers with '/'-hierarchical names.

Import sites in the reference: mindmap/run_training.py:23,470-491,765; data_loading/batching.py:12;
data_loading/dataset.py:19; diffuser_actor/diffuser_actor.py:3; mapping/helpers/nvblox_mapping_helpers.py:21.
Usage kept: ``with Timer(name): ...`` and ``t = Timer(name); ...; t.stop()``.  Host wall-clock, not
GPU-synchronising (same as upstream).
"""
import threading
import time
from typing import Dict, List

_lock = threading.Lock()
_timers: Dict[str, List[float]] = {}  # name -> [count, total_s, last_s, min_s, max_s]


class Timer:
    def __init__(self, name: str, start: bool = True):
        self.name = name
        self._t0 = None
        if start:
            self.start()

    def start(self) -> None:
        self._t0 = time.perf_counter()

    def stop(self) -> float:
        if self._t0 is None:
            return 0.0
        dt = time.perf_counter() - self._t0
        self._t0 = None
        with _lock:
            rec = _timers.setdefault(self.name, [0, 0.0, 0.0, float("inf"), 0.0])
            rec[0] += 1
            rec[1] += dt
            rec[2] = dt
            rec[3] = min(rec[3], dt)
            rec[4] = max(rec[4], dt)
        return dt

    def __enter__(self):
        if self._t0 is None:
            self.start()
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


def get_last_time(name: str) -> float:
    with _lock:
        return _timers[name][2] if name in _timers else 0.0


def get_mean_time(name: str) -> float:
    with _lock:
        rec = _timers.get(name)
        return rec[1] / rec[0] if rec and rec[0] else 0.0


def get_total_time(name: str) -> float:
    with _lock:
        return _timers[name][1] if name in _timers else 0.0


def get_num_calls(name: str) -> int:
    with _lock:
        return int(_timers[name][0]) if name in _timers else 0


def timer_status_string() -> str:
    with _lock:
        names = sorted(_timers)
        lines = ["Timings [s]", f"{'name':<48}{'calls':>8}{'total':>12}{'mean':>12}{'min':>12}{'max':>12}"]
        for n in names:
            c, tot, _last, mn, mx = _timers[n]
            lines.append(f"{n:<48}{int(c):>8}{tot:>12.6f}{(tot / c if c else 0):>12.6f}{mn:>12.6f}{mx:>12.6f}")
    return "\n".join(lines)


def print_timers() -> None:
    print(timer_status_string())


def reset_timers() -> None:
    with _lock:
        _timers.clear()
