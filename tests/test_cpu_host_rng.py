"""mmf_host_randperm_prefix (the draw of sample_to_n_vertices, mindmap/data_loading/vertex_sampling.py:143-145) against
torch itself: same indices as ``torch.randperm(n)[:k]`` on the CPU default generator and the same generator state afterwards."""
import random

import pytest
import torch

from nvblox_mindmap_amd.data_loading.vertex_sampling import randperm_prefix


def _cases():
    rnd = random.Random(7)
    fixed = [(0, 0), (1, 0), (1, 1), (2, 1), (2, 2), (5, 5), (623, 10), (624, 624), (625, 3), (1000, 1000), (3000, 2048),
             (26000, 2048), (32768, 2048), (32769, 2048), (43000, 2048), (100000, 2048)]
    drawn = []
    for _ in range(40):
        n = rnd.randint(1, 60000)
        drawn.append((n, rnd.randint(0, min(n, 4096))))
    return fixed + drawn


@pytest.mark.parametrize("case", list(enumerate(_cases())))
def test_randperm_prefix_matches_torch(case):
    seed, (n, k) = case
    torch.manual_seed(seed)
    torch.rand(seed * 37 % 1300)  # any phase of the engine's 624-word buffer
    start = torch.get_rng_state()
    want = torch.randperm(n)[:k]
    state_want = torch.get_rng_state()
    after_want = torch.randint(0, 2**31, (7,))
    torch.set_rng_state(start)
    got = randperm_prefix(n, k)
    assert got.dtype == torch.int64 and torch.equal(got, want)
    assert torch.equal(torch.get_rng_state(), state_want), "the generator must end where torch.randperm leaves it"
    assert torch.equal(torch.randint(0, 2**31, (7,)), after_want)
