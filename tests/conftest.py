import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as O

    O.build()
    O.lib()
    # a frame's work is a few thousand blocks: beyond ~16 threads the OpenMP fan-out costs more than it buys (bench.py's thread
    # sweep: 39 frames/s on 16 threads, 1.6 on the GPU box's 256), and that box gives the container a 16-CPU quota anyway
    O.set_num_threads(min(16, os.cpu_count() or 1))
    return O
