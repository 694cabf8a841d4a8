"""3-D diffuser-actor policy on PyTorch-ROCm: the consumer of the fused map.

Same input-tensor API as mindmap/diffuser_actor/diffuser_actor.py:518-531 (``vertex_features (B,N,C)``,
``vertices (B,N,3)``, ``vertices_valid_mask (B,N)``, ``rgb_obs``, ``pcd_obs``, ``pcd_valid_mask``,
``gripper_history``), same architecture (layer counts, widths, AdaLN conditioning, 3-D rotary attention, two DDPM
schedules), written from scratch: batch-first tensors, ``scaled_dot_product_attention``, an own DDPM scheduler
(the reference uses `diffusers`, absent here) and a HIP farthest-point sampler (the reference uses a dgl CUDA op,
which has no ROCm build).
"""
from .model import DiffuserActor, DiffuserActorConfig  # noqa: F401
