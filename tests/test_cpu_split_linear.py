"""diffuser_actor/split_linear.py bookkeeping on the CPU (the arithmetic is GPU-tested in tests/test_gpu_policy.py): which layers have a
split copy, when a copy is stale, what can be refreshed in place."""
import torch
import torch.nn as nn

from nvblox_mindmap_amd.diffuser_actor import split_linear as SL


def test_a_layer_without_a_split_copy_is_never_stale():
    """Round-5 advisor finding: an unsplittable Linear (K % 8 != 0) caches None; after load_state_dict ``stale()`` said True,
    ``refresh_in_place`` False, and GraphedTrainStep.refresh_frozen_weights raised 'build a new GraphedTrainStep' although the captured
    graph multiplies by the float32 weight itself."""
    torch.manual_seed(0)
    odd = nn.Linear(70, 16)       # 70 % 8 != 0: no split copy
    ok = nn.Linear(128, 16)
    assert SL._w3(odd) is None and SL._w3(ok) is not None
    assert not SL.stale(odd) and not SL.stale(ok)
    with torch.no_grad():
        odd.weight.mul_(2.0)
        ok.weight.mul_(2.0)
    assert not SL.stale(odd)      # nothing to refresh: no copy exists
    assert SL.stale(ok)
    kept = SL._w3.__globals__["_W3_CACHE"][id(ok.weight)][2]
    assert SL.refresh_in_place(ok) and not SL.stale(ok)
    assert SL._W3_CACHE[id(ok.weight)][2] is kept                     # same storage: a captured graph's address stays valid
    assert torch.equal(kept, SL._split_weight(ok))
    assert not SL.refresh_in_place(odd)                               # (no copy: nothing it could do -- and nobody asks, see stale)


def test_out_of_range_weights_lose_their_copy_loudly():
    """A copy that existed and can no longer be recomputed (a value beyond fp16's range) IS reported: stale, not refreshable."""
    torch.manual_seed(0)
    lin = nn.Linear(128, 8)
    assert SL._w3(lin) is not None
    with torch.no_grad():
        lin.weight[0, 0] = 1.0e6
    assert SL.stale(lin) and not SL.refresh_in_place(lin)
