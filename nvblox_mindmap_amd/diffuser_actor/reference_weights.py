"""State-dict converter between the reference's DiffuserActor and this package's (both directions).

The two models compute the same function with differently organised modules, so a reference checkpoint
(``best.pth`` / ``last.pth`` written by mindmap/model_utils/checkpoint.py:30-52, loaded by
``load_inference_checkpoint`` :103 and closed_loop/policies/nvblox_diffuser_actor_policy.py:240-245) has to be
renamed -- and, for the attention projections, split -- before ``load_state_dict``:

  reference (mindmap/diffuser_actor/...)                               here (nvblox_mindmap_amd/diffuser_actor/...)
  -------------------------------------------------------------------  ---------------------------------------------
  MultiheadCustomAttention.in_proj_weight [3D,D] / in_proj_bias [3D]   RelativeAttention.q_proj [D,D] + kv_proj [2D,D]
    (multihead_custom_attention.py:62,80; rows = q | k | v, :310)         (rows 0:D | D:3D)
  RelativeCrossAttentionLayer.multihead_attn / .norm / .adaln          AttentionBlock.attn / .norm / .adaln
    (layers.py:354-385)
  AdaLN.modulation[1] (layers.py:309-313; [0] is the SiLU)             AdaLN.proj
  FeedforwardLayer.linear1 / linear2 / norm / adaln (layers.py:327)    FeedForwardBlock.fc1 / fc2 / norm / adaln
  FFWRelative{Cross,Self}AttentionModule.attn_layers / ffw_layers      AttentionStack.attn / ffw
  ParallelAttentionLayer.cross_12 / norm_12 / ffn_12.{0,3} / norm_122  AttentionBlock (attn, norm) + FeedForwardBlock
    (layers.py:57-91; vision-language + trajectory-language attention)
  nn.Sequential indices that count Dropout / ReLU slots                nn.Sequential without the Dropout slots
    (diffusion_head.py:43-56,104-153)
  Encoder.image_feature_encoder.linear / reconstruction_encoder.linear Encoder.image_embed / mesh_embed
  Encoder.curr_open_close_encoder / gripper_history_embed /            Encoder.open_close / history_embed /
    gripper_context_head / goal_gripper_embed (encoder.py:92-126)        gripper_context / goal_embed

The reference's frozen RADIO / DINO extractor is not an nn.Module attribute (feature_extraction.py:132-160), so its
weights are not in a reference checkpoint; this package's stand-in backbone (``encoder.backbone.*``) is likewise left
alone by the converter.  A DistributedDataParallel wrapper prefixes every key with ``module.``; both directions accept
and drop it.
"""
import re
from typing import Dict, List, Tuple

import torch

# (reference prefix, local prefix) of the sub-modules; longest match wins.
_STACKS = [
    ("encoder.gripper_context_head", "encoder.gripper_context"),
    ("prediction_head.cross_attn", "prediction_head.cross_attn"),
    ("prediction_head.self_attn", "prediction_head.self_attn"),
    ("prediction_head.rotation_self_attn", "prediction_head.rotation_attn"),
    ("prediction_head.position_self_attn", "prediction_head.position_attn"),
]
_PLAIN = [
    ("encoder.image_feature_encoder.linear", "encoder.image_embed"),
    ("encoder.reconstruction_encoder.linear", "encoder.mesh_embed"),
    ("encoder.curr_open_close_encoder", "encoder.open_close"),
    ("encoder.gripper_history_embed", "encoder.history_embed"),
    ("encoder.goal_gripper_embed", "encoder.goal_embed"),
    ("encoder.instruction_encoder", "encoder.instruction_encoder"),
    ("prediction_head.traj_encoder.0", "prediction_head.traj_encoder"),
    ("prediction_head.time_emb.1", "prediction_head.time_mlp.0"),
    ("prediction_head.time_emb.4", "prediction_head.time_mlp.2"),
    ("prediction_head.gripper_history_emb.0", "prediction_head.history_mlp.0"),
    ("prediction_head.gripper_history_emb.3", "prediction_head.history_mlp.2"),
    ("prediction_head.rotation_proj.0", "prediction_head.rotation_proj"),
    ("prediction_head.position_proj.0", "prediction_head.position_proj"),
    ("prediction_head.rotation_predictor.0", "prediction_head.rotation_out.0"),
    ("prediction_head.rotation_predictor.3", "prediction_head.rotation_out.2"),
    ("prediction_head.position_predictor.0", "prediction_head.position_out.0"),
    ("prediction_head.position_predictor.3", "prediction_head.position_out.2"),
    ("prediction_head.openess_predictor.0", "prediction_head.openness_out.0"),
    ("prediction_head.openess_predictor.3", "prediction_head.openness_out.2"),
    ("prediction_head.head_yaw_predictor.0", "prediction_head.head_yaw_out.0"),
    ("prediction_head.head_yaw_predictor.3", "prediction_head.head_yaw_out.2"),
]
# ParallelAttention holders: reference "<ref>.layers.<i>.<part>" -> local "<attn>.<i>..." / "<ffn>.<i>..."
_PARALLEL = [
    ("encoder.vl_attention.0", "encoder.vl_attention", "encoder.vl_ffn"),
    ("prediction_head.traj_lang_attention.0", "prediction_head.traj_lang_attention", "prediction_head.traj_lang_ffn"),
]
_STACK_PART = [  # inside attn_layers.<i> / ffw_layers.<i>
    ("norm", "norm"), ("adaln.modulation.1", "adaln.proj"), ("linear1", "fc1"), ("linear2", "fc2"),
    ("multihead_attn.out_proj", "attn.out_proj"),
]
_PARALLEL_PART = [  # (reference part, local module kind, local part)
    ("cross_12.out_proj", "attn", "attn.out_proj"), ("norm_12", "attn", "norm"),
    ("ffn_12.0", "ffn", "fc1"), ("ffn_12.3", "ffn", "fc2"), ("norm_122", "ffn", "norm"),
]


def strip_ddp_prefix(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Drop the ``module.`` prefix of a DistributedDataParallel state dict (no-op otherwise)."""
    if all(k.startswith("module.") for k in state):
        return {k[len("module."):]: v for k, v in state.items()}
    return dict(state)


def _split_in_proj(prefix: str, leaf: str, tensor: torch.Tensor) -> List[Tuple[str, torch.Tensor]]:
    """in_proj_weight / in_proj_bias (rows q | k | v) -> q_proj + kv_proj."""
    D = tensor.shape[0] // 3
    assert tensor.shape[0] == 3 * D, f"{prefix}.in_proj_{leaf}: first dim {tensor.shape[0]} is not 3 x embed_dim"
    return [(f"{prefix}.q_proj.{leaf}", tensor[:D].clone()), (f"{prefix}.kv_proj.{leaf}", tensor[D:].clone())]


def _convert_key(key: str, value: torch.Tensor) -> List[Tuple[str, torch.Tensor]]:
    for ref, loc in _STACKS:
        m = re.fullmatch(re.escape(ref) + r"\.(attn_layers|ffw_layers)\.(\d+)\.(.+)\.(weight|bias)", key)
        m2 = re.fullmatch(re.escape(ref) + r"\.attn_layers\.(\d+)\.multihead_attn\.in_proj_(weight|bias)", key)
        if m2:
            return _split_in_proj(f"{loc}.attn.{m2.group(1)}.attn", m2.group(2), value)
        if m:
            kind = "attn" if m.group(1) == "attn_layers" else "ffw"
            for rp, lp in _STACK_PART:
                if m.group(3) == rp:
                    return [(f"{loc}.{kind}.{m.group(2)}.{lp}.{m.group(4)}", value)]
            raise KeyError(f"unknown parameter of an attention stack in the reference state dict: {key}")
    for ref, attn, ffn in _PARALLEL:
        m2 = re.fullmatch(re.escape(ref) + r"\.layers\.(\d+)\.cross_12\.in_proj_(weight|bias)", key)
        if m2:
            return _split_in_proj(f"{attn}.{m2.group(1)}.attn", m2.group(2), value)
        m = re.fullmatch(re.escape(ref) + r"\.layers\.(\d+)\.(.+)\.(weight|bias)", key)
        if m:
            for rp, kind, lp in _PARALLEL_PART:
                if m.group(2) == rp:
                    return [(f"{attn if kind == 'attn' else ffn}.{m.group(1)}.{lp}.{m.group(3)}", value)]
            raise KeyError(f"unknown parameter of a ParallelAttention layer in the reference state dict: {key}")
    for ref, loc in _PLAIN:
        if key.startswith(ref + "."):
            return [(loc + key[len(ref):], value)]
    raise KeyError(f"no counterpart for reference parameter {key!r}")


def convert_reference_state_dict(reference_state: Dict[str, torch.Tensor], ignore_prefixes=("encoder.feature_extractor.",)):
    """Reference DiffuserActor state dict -> state dict of nvblox_mindmap_amd.diffuser_actor.DiffuserActor.
    Keys under ``ignore_prefixes`` (a trainable CLIP FPN extractor: not supported here) are skipped and returned as the
    second value."""
    out, skipped = {}, []
    for key, value in strip_ddp_prefix(reference_state).items():
        if any(key.startswith(p) for p in ignore_prefixes):
            skipped.append(key)
            continue
        for k, v in _convert_key(key, value):
            assert k not in out, k
            out[k] = v
    return out, skipped


def convert_head_state_dict(reference_head_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The same for a bare reference ``DiffusionHead`` state dict (keys without the ``prediction_head.`` prefix)."""
    full, _ = convert_reference_state_dict({"prediction_head." + k: v for k, v in strip_ddp_prefix(reference_head_state).items()})
    return {k[len("prediction_head."):]: v for k, v in full.items()}


def to_reference_state_dict(local_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The inverse: this package's DiffuserActor state dict -> reference key names / layouts (``encoder.backbone.*``, the
    stand-in for the reference's non-Module feature extractor, is dropped)."""
    local = {k: v for k, v in strip_ddp_prefix(local_state).items() if not k.startswith("encoder.backbone.")}
    out = {}

    def join_in_proj(loc_attn: str, ref_attn: str):
        for leaf in ("weight", "bias"):
            q, kv = local.pop(f"{loc_attn}.q_proj.{leaf}"), local.pop(f"{loc_attn}.kv_proj.{leaf}")
            out[f"{ref_attn}.in_proj_{leaf}"] = torch.cat([q, kv], dim=0)

    for ref, loc in _STACKS:
        layers = sorted({int(m.group(1)) for k in local for m in [re.match(re.escape(loc) + r"\.attn\.(\d+)\.", k)] if m})
        for i in layers:
            join_in_proj(f"{loc}.attn.{i}.attn", f"{ref}.attn_layers.{i}.multihead_attn")
            for kind, rkind in (("attn", "attn_layers"), ("ffw", "ffw_layers")):
                for rp, lp in _STACK_PART:
                    for leaf in ("weight", "bias"):
                        k = f"{loc}.{kind}.{i}.{lp}.{leaf}"
                        if k in local:
                            out[f"{ref}.{rkind}.{i}.{rp}.{leaf}"] = local.pop(k)
    for ref, attn, ffn in _PARALLEL:
        layers = sorted({int(m.group(1)) for k in local for m in [re.match(re.escape(attn) + r"\.(\d+)\.", k)] if m})
        for i in layers:
            join_in_proj(f"{attn}.{i}.attn", f"{ref}.layers.{i}.cross_12")
            for rp, kind, lp in _PARALLEL_PART:
                for leaf in ("weight", "bias"):
                    k = f"{attn if kind == 'attn' else ffn}.{i}.{lp}.{leaf}"
                    if k in local:
                        out[f"{ref}.layers.{i}.{rp}.{leaf}"] = local.pop(k)
    for ref, loc in _PLAIN:
        for k in [k for k in local if k.startswith(loc + ".")]:
            out[ref + k[len(loc):]] = local.pop(k)
    assert not local, f"parameters without a reference counterpart: {sorted(local)[:5]}"
    return out


def load_reference_state_dict(model, reference_state: Dict[str, torch.Tensor]) -> None:
    """Load a reference state dict into this package's DiffuserActor.  Every converted key must exist with the same shape;
    the only parameters allowed to be absent from the reference are the stand-in backbone's."""
    converted, _ = convert_reference_state_dict(reference_state)
    wrapped = all(k.startswith("module.") for k in model.state_dict())
    if wrapped:
        converted = {"module." + k: v for k, v in converted.items()}
    result = model.load_state_dict(converted, strict=False)
    assert not result.unexpected_keys, f"converted keys unknown to the model: {result.unexpected_keys[:5]}"
    missing = [k for k in result.missing_keys if ".backbone." not in k]
    assert not missing, f"model parameters absent from the reference state dict: {missing[:5]}"


def is_reference_state_dict(state: Dict[str, torch.Tensor]) -> bool:
    return any(".multihead_attn.in_proj_weight" in k or ".cross_12.in_proj_weight" in k for k in state)
