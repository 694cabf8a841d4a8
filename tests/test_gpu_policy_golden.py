"""The fused (matrix-core) inference path of the diffusion head against golden outputs of the REFERENCE DiffusionHead
(tests/golden/policy_head.npz; mindmap/diffuser_actor/diffusion_head.py:161,254), weights through the reference -> local
converter.  The CPU counterpart (composite torch ops) is tests/test_cpu_policy_golden.py."""
import numpy as np
import pytest
import torch

from test_cpu_policy_golden import GOLD, enc_of, head_case
from nvblox_mindmap_amd.diffuser_actor import layers as Ly
from nvblox_mindmap_amd.diffuser_actor import reference_weights as RW
from nvblox_mindmap_amd.diffuser_actor.layers import sinusoidal_embedding
from nvblox_mindmap_amd.diffuser_actor.model import DiffusionHead

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


def build(gold, name):
    cfg, ref_state, x = head_case(gold, name)
    head = DiffusionHead(cfg).eval()
    head.load_state_dict(RW.convert_head_state_dict(ref_state), strict=True)
    return cfg, head.cuda(), {k: v.cuda() for k, v in x.items()}


@pytest.mark.parametrize("name", ["head", "policy", "arm"])
def test_composite_head_on_gpu_matches_reference(gold, name):
    cfg, head, x = build(gold, name)
    with torch.no_grad():
        pred, head_yaw, _ = head(x["trajectory"], x["timestep"], enc_of(x))
    assert np.abs(pred.cpu().numpy() - gold[f"{name}_pred"]).max() <= TOL
    if cfg.predict_head_yaw:
        assert np.abs(head_yaw.cpu().numpy() - gold[f"{name}_head_yaw"]).max() <= TOL


@pytest.mark.parametrize("name", ["head", "policy"])
def test_fused_block_kernels_match_reference(gold, name):
    """FUSED_INFERENCE on, per-sample timesteps: the whole-layer kernels driven by the generic forward (no step prologue)."""
    cfg, head, x = build(gold, name)
    Ly.FUSED_INFERENCE = True
    try:
        with torch.no_grad():
            pred, head_yaw, _ = head(x["trajectory"], x["timestep"], enc_of(x))
    finally:
        Ly.FUSED_INFERENCE = False
    assert np.abs(pred.cpu().numpy() - gold[f"{name}_pred"]).max() <= TOL, float(np.abs(pred.cpu().numpy() - gold[f"{name}_pred"]).max())
    assert np.abs(head_yaw.cpu().numpy() - gold[f"{name}_head_yaw"]).max() <= TOL


def test_fused_mfma_step_matches_reference(gold):
    """The closed-loop shape (batch 1, 3 072 context tokens, 614 sub-sampled): ONE denoising-step evaluation on the
    end-to-end inference kernels (_forward_fused: step prologue, matrix-core layer kernels, split cross-attention, one-launch
    output heads) reproduces the reference head's output."""
    cfg, head, x = build(gold, "policy")
    Ly.FUSED_INFERENCE = True
    try:
        with torch.no_grad():
            P = head.prepare_context(enc_of(x))
            assert P.get("seq") is not None and P.get("ctx_pad16") is not None, "matrix-core path not selected"
            time_emb = head.time_mlp(sinusoidal_embedding(x["timestep"], cfg.embedding_dim))
            assert head.can_denoise_fused(P, x["trajectory"])
            pred, head_yaw, _ = head(x["trajectory"], None, enc_of(x), prepared=P, time_emb=time_emb)
    finally:
        Ly.FUSED_INFERENCE = False
    err = float(np.abs(pred.cpu().numpy() - gold["policy_pred"]).max())
    assert err <= TOL, err
    assert np.abs(head_yaw.cpu().numpy() - gold["policy_head_yaw"]).max() <= TOL
