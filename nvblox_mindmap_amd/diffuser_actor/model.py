"""DiffuserActor: encoder + diffusion head + two DDPM schedules.

Counterpart of mindmap/diffuser_actor/{diffuser_actor,encoder,diffusion_head}.py with the same forward() input
tensors (diffuser_actor.py:518-531) and the same architecture: context tokens = image-patch tokens (frozen backbone ->
Linear) and/or map-vertex tokens (Linear), 3-D rotary attention, gripper-history queries attending to the context
(3 layers), farthest-point subsampling of the context in feature space, a diffusion head with AdaLN conditioning on
(denoising step + gripper history): 2 cross-attention layers over the full context, 4 shared + 2 + 2 head-specific
self-attention layers over [trajectory tokens ; subsampled context], predictors for position noise (3), 6-D rotation
noise (6), gripper openness (1) and head yaw (1).
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nvblox_torch.timer import Timer
from .backbone import VitBackbone
from .fps import farthest_point_sampling, farthest_point_sampling_cpu
from . import layers as layers_mod
from .layers import AttentionBlock, AttentionStack, FeedForwardBlock, rotary3d, sinusoidal_embedding
from .loss import LossWeights, compute_loss
from .relative_conversions import (get_current_pose_from_gripper_history, to_absolute_trajectory, to_relative_gripper_history,
                                   to_relative_pcd, to_relative_trajectory)
from .rotations import normalize_pointcloud, normalize_pos, normalize_trajectory, unnormalize_trajectory
from .scheduler import DDPMScheduler


@dataclass
class DiffuserActorConfig:
    """Defaults = mindmap/cli/args.py:57-93 (ModelArgs)."""

    data_type: str = "rgbd_and_mesh"       # "rgbd" | "mesh" | "rgbd_and_mesh"
    image_size: Tuple[int, int] = (512, 512)
    feature_dim: int = 768                  # channels of image features and of the map's vertex features
    embedding_dim: int = 120
    num_attn_heads: int = 8
    num_history: int = 3
    ngrippers: int = 2                      # humanoid (Drill-in-Box): 2, arm: 1
    prediction_horizon: int = 1
    fps_subsampling_factor: int = 5
    use_fps: bool = True
    diffusion_timesteps: int = 100
    encode_openness: bool = True
    use_shared_feature_encoder: bool = False
    predict_head_yaw: bool = True
    use_instruction: bool = False
    quaternion_format: str = "wxyz"
    rotation_parametrization: str = "6D_from_query"  # reference default (cli/args.py); see rotations.unnormalize_trajectory
    add_external_cam: bool = False
    relative_action: bool = False           # poses relative to the newest gripper pose (the reference's ``relative``, diffuser_actor.py:46)
    dropout: float = 0.0
    backbone: str = "vit_b16"               # random-init stand-in for the frozen RADIO v2.5-B ("none": rgb tokens are given)
    # matmuls of the FROZEN image backbone: "float32"; "float16x3": every Linear as one fp16 GEMM of operands split into two fp16
    # values each (22-bit mantissas, f32 accumulation: f32 accuracy at ~2x the speed, split_linear.py); "float16": fp16 autocast
    # (a 10-bit mantissa, what the reference's TF32 setting keeps, feature_extraction.py:322)
    backbone_matmul_dtype: str = "float16x3"
    loss_weights: LossWeights = field(default_factory=LossWeights)


class Encoder(nn.Module):
    def __init__(self, cfg: DiffuserActorConfig):
        super().__init__()
        self.cfg = cfg
        D, nq = cfg.embedding_dim, cfg.num_history * cfg.ngrippers
        self.uses_images = cfg.data_type in ("rgbd", "rgbd_and_mesh")
        self.uses_mesh = cfg.data_type in ("mesh", "rgbd_and_mesh")
        if self.uses_images:
            self.backbone = VitBackbone(dim=cfg.feature_dim) if cfg.backbone == "vit_b16" else None
            if self.backbone is not None:
                for p in self.backbone.parameters():
                    p.requires_grad = False
            self.image_embed = nn.Linear(cfg.feature_dim, D)
        if self.uses_mesh and not cfg.use_shared_feature_encoder:
            self.mesh_embed = nn.Linear(cfg.feature_dim, D)
        if cfg.encode_openness:
            # binary closedness per (history step, gripper) -> one learnable query per state (encoder.py:108-114)
            self.open_close = nn.Linear(nq, nq * D)
        else:
            self.history_embed = nn.Embedding(nq, D)
        self.gripper_context = AttentionStack(D, cfg.num_attn_heads, 3, cfg.dropout, use_adaln=False)
        # present in the reference, unused when use_instruction = 0 (why DDP needs find_unused_parameters=True)
        self.goal_embed = nn.Embedding(1, D)
        self.instruction_encoder = nn.Linear(512, D)
        self.vl_attention = nn.ModuleList([AttentionBlock(D, cfg.num_attn_heads, cfg.dropout) for _ in range(2)])
        self.vl_ffn = nn.ModuleList([FeedForwardBlock(D, 4 * D, cfg.dropout) for _ in range(2)])

    def train(self, mode: bool = True):
        super().train(mode)
        if getattr(self, "backbone", None) is not None:
            self.backbone.eval()  # frozen
        return self

    # -- context tokens ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def backbone_features(self, rgb):
        """Frozen image backbone: rgb (B,ncam,3,H,W) in [0,1] -> (B*ncam, C, h, w) float32.  No parameter of it is trained,
        so a trainer may evaluate it for the NEXT batch on a second stream while the trainable part of the current batch
        runs (training.trainer.BackbonePrefetcher)."""
        # The reference runs the frozen backbone under AllowMatMulTf32 (image_processing/feature_extraction.py:322): 10-bit
        # mantissa inputs, fp32 accumulation.  gfx950 has no TF32 MFMA; float16 inputs carry the same mantissa ("float16").
        # Default: f32 accuracy from split fp16 operands ("float16x3"), ~2x the speed of the f32 GEMMs.
        fp16 = self.cfg.backbone_matmul_dtype == "float16" and rgb.is_cuda
        if hasattr(self.backbone, "split_gemm"):
            from .split_linear import supported

            self.backbone.split_gemm = self.cfg.backbone_matmul_dtype == "float16x3" and rgb.is_cuda and supported()
        with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
            feats = self.backbone(rgb.flatten(0, 1))
        return feats.float()

    def encode_images(self, rgb, pcd, valid_mask, backbone_feats=None):
        """rgb (B,ncam,3,H,W) in [0,1], pcd (B,ncam,3,H,W) normalised points, valid_mask (B,ncam,H,W) ->
        tokens (B,ncam*h*w,D), positions (B,ncam*h*w,3), mask (B,ncam*h*w).  ``backbone_feats``: the output of
        ``backbone_features(rgb)`` when it has been computed ahead of time."""
        B, ncam = pcd.shape[:2]
        feats = self.backbone_features(rgb) if backbone_feats is None else backbone_feats
        h, w = feats.shape[-2:]
        tokens = self.image_embed(feats.flatten(2).transpose(1, 2)).reshape(B, ncam * h * w, -1)
        pos = F.interpolate(pcd.flatten(0, 1), (h, w), mode="bilinear", align_corners=False)
        pos = pos.flatten(2).transpose(1, 2).reshape(B, ncam * h * w, 3)
        f = valid_mask.shape[-1] // w  # AND-pooling: a token is valid iff all its pixels are (image_mask_operations.py:71)
        m = valid_mask.reshape(B, ncam, h, f, w, f).all(dim=-1).all(dim=-2).reshape(B, ncam * h * w)
        return tokens, pos, m

    def encode_vertices(self, vertex_features, vertices):
        embed = self.image_embed if self.cfg.use_shared_feature_encoder else self.mesh_embed
        assert vertex_features.shape[-1] == embed.in_features, (
            f"vertex features have {vertex_features.shape[-1]} channels, the model expects {embed.in_features}")
        return embed(vertex_features.to(torch.float32)), vertices

    # -- gripper history queries --------------------------------------------------------------------------------------
    def encode_gripper_history(self, gripper_history, context_feats, context_pos, closedness):
        """gripper_history (B,nhist,ngrip,9), closedness (B,nhist,ngrip,1) -> (B,nhist*ngrip,D)."""
        B = gripper_history.shape[0]
        D = self.cfg.embedding_dim
        if self.cfg.encode_openness:
            q = self.open_close(closedness.flatten(1)).reshape(B, -1, D)
        else:
            q = self.history_embed.weight[None].expand(B, -1, -1)
        q_rot = rotary3d(gripper_history[..., :3].flatten(1, 2), D)
        out, _ = self.gripper_context(q, context_feats, None, q_rot, rotary3d(context_pos, D))
        return out

    def attend_instruction(self, context_feats, instruction):
        instr = self.instruction_encoder(instruction)
        for attn, ffn in zip(self.vl_attention, self.vl_ffn):
            context_feats, _ = attn(context_feats, instr)
            context_feats = ffn(context_feats)
        return context_feats, instr

    # -- farthest point subsampling -------------------------------------------------------------------------------------
    def run_fps(self, context_feats, context_pos, context_mask):
        """Subsample the context to N / fps_subsampling_factor tokens by FPS in feature space (invalid tokens zeroed
        first, so at most one of them is picked before every valid one; encoder.py:338-419)."""
        B, N, D = context_feats.shape
        masked = context_feats * context_mask[..., None]
        n_keep = max(N // self.cfg.fps_subsampling_factor, 1)
        fps = farthest_point_sampling if masked.is_cuda else farthest_point_sampling_cpu
        idx = fps(masked, n_keep, 0)
        feats = torch.gather(masked, 1, idx[..., None].expand(-1, -1, D))
        pos = torch.gather(context_pos, 1, idx[..., None].expand(-1, -1, 3))
        return feats, pos, (feats != 0).any(dim=-1)


class DiffusionHead(nn.Module):
    def __init__(self, cfg: DiffuserActorConfig):
        super().__init__()
        self.cfg = cfg
        D, H, p = cfg.embedding_dim, cfg.num_attn_heads, cfg.dropout
        nq = cfg.num_history * cfg.ngrippers
        self.traj_encoder = nn.Linear(9, D)
        self.time_mlp = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, D))
        self.history_mlp = nn.Sequential(nn.Linear(D * nq, D), nn.ReLU(), nn.Linear(D, D))
        # trajectory tokens attend to the language tokens (use_instruction only; diffusion_head.py:61-78,204-214).  The
        # feed-forward half exists in the reference's state dict but is never applied there (apply_ffn=False): kept so that
        # checkpoints convert both ways (reference_weights.py)
        self.traj_lang_attention = nn.ModuleList([AttentionBlock(D, H, p)])
        self.traj_lang_ffn = nn.ModuleList([FeedForwardBlock(D, 4 * D, p)])
        self.cross_attn = AttentionStack(D, H, 2, p, use_adaln=True)
        self.self_attn = AttentionStack(D, H, 4, p, use_adaln=True, self_attention=True)
        self.rotation_attn = AttentionStack(D, H, 2, p, use_adaln=True, self_attention=True)
        self.position_attn = AttentionStack(D, H, 2, p, use_adaln=True, self_attention=True)
        self.rotation_proj, self.position_proj = nn.Linear(D, D), nn.Linear(D, D)
        self.rotation_out = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 6))
        self.position_out = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 3))
        self.openness_out = nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, 1))
        self.head_yaw_out = nn.Sequential(nn.Linear(D * cfg.ngrippers, D), nn.ReLU(), nn.Linear(D, 1)) if cfg.predict_head_yaw else None
        self.drop = nn.Dropout(p)
        self._side_stream = None  # second stream of the fused inference path (rotation stack || position stack)

    def prepare_context(self, enc):
        """Everything the head needs from the encoder outputs that does not depend on the denoising step: masks made safe
        against fully masked samples (branch-free: no host synchronisation, so the sampling loop can be captured in a HIP
        graph), rotary embeddings of the context / sub-sampled context positions, the history conditioning."""
        D = self.cfg.embedding_dim
        ctx_mask, fps_mask = enc["context_mask"], enc["fps_mask"]
        ctx_feats, fps_feats = enc["context_feats"], enc["fps_feats"]
        # a sample whose context is fully masked would give NaN attention rows: attend to (zeroed) everything instead
        empty, empty_fps = ~ctx_mask.any(dim=-1), ~fps_mask.any(dim=-1)
        ctx_mask = ctx_mask | empty[:, None]
        fps_mask = fps_mask | empty_fps[:, None]
        ctx_feats = ctx_feats * (~empty)[:, None, None]      # x * 1.0 is exact: samples with context are untouched
        fps_feats = fps_feats * (~empty_fps)[:, None, None]
        P = {"ctx_feats": ctx_feats, "fps_feats": fps_feats, "ctx_pad": ~ctx_mask, "fps_pad": ~fps_mask,
             "ctx_rot": rotary3d(enc["context_pos"], D), "fps_rot": rotary3d(enc["fps_pos"], D),
             "history": self.history_mlp(enc["history_feats"].flatten(1)), "cross_kv": None}
        P["adaln"] = None
        if layers_mod._fused(ctx_feats):  # keys / values of the (step-invariant) context, once per inference instead of per step
            P["cross_kv"] = [blk.attn.project_kv(ctx_feats, P["ctx_rot"]) for blk in self.cross_attn.attn]
            P["adaln"] = layers_mod.AdaLNBatch([mod for mod in self.modules() if isinstance(mod, layers_mod.AdaLN)])
            from .fused_ops import MFMA_DIMS

            if (D, self.cfg.num_attn_heads) == MFMA_DIMS:
                from .fused_ops import pad_mask16

                P["cross_kv"] = [blk.attn.project_kv_heads(ctx_feats, P["ctx_rot"]) for blk in self.cross_attn.attn]
                P["ctx_pad16"] = pad_mask16(P["ctx_pad"])
                from .fused_ops import CrossHandover

                P["cross_handover"] = CrossHandover(ctx_feats.shape[0], self.cfg.num_attn_heads, ctx_feats.device)  # zeroed per inference
            if D in layers_mod._block_dims():
                # step-invariant parts of the step's sequence-wide tensors: the sub-sampled context rows of the token
                # sequence, of its rotary tables and of its padding mask are filled once; a step rewrites the trajectory rows
                B, nt = fps_feats.shape[0], self.cfg.prediction_horizon * self.cfg.ngrippers
                dev = fps_feats.device
                head = torch.zeros((B, nt, D), device=dev)
                P["seq"] = torch.cat([head, fps_feats], dim=1)
                P["seq_cos"] = torch.cat([head, P["fps_rot"][0].expand(B, -1, D)], dim=1)
                P["seq_sin"] = torch.cat([head, P["fps_rot"][1].expand(B, -1, D)], dim=1)
                P["seq_pad"] = torch.cat([torch.zeros((B, nt), dtype=torch.bool, device=dev), P["fps_pad"]], dim=1)
                if P.get("cross_handover") is not None:  # (matrix-core path) hand-over buffers of the one-launch self-attention layers
                    from .fused_ops import SelfHandover

                    P["self_handover"] = SelfHandover(B, P["seq"].shape[1], D, dev)
                if P.get("ctx_pad16") is not None:
                    from .fused_ops import pad_mask16

                    P["seq_pad16"] = pad_mask16(P["seq_pad"])
                    P["seq_pad16x2"] = torch.cat([P["seq_pad16"], P["seq_pad16"]], dim=0)  # rotation | position stacks as one batch
                P["pos_table"] = sinusoidal_embedding(torch.arange(nt, device=dev), D)
                third = D // 3
                P["rot_freq"] = torch.exp(torch.arange(0, third, 2, device=dev, dtype=torch.float32) * (-math.log(10000.0) / third))
        return P

    def time_embeddings(self, timesteps, device):
        """time_mlp(sinusoidal(t)) for a list of timesteps in one batched pass ([T, D]): the time part of the conditioning
        vector depends on the step only, not on the inputs."""
        timesteps = list(timesteps)
        n = len(timesteps)
        stride = timesteps[0] - timesteps[1] if n > 1 else 1
        assert all(timesteps[i] == timesteps[0] - i * stride for i in range(n)), "evenly strided, descending timesteps"
        # built on the device (no host-to-device copy: this runs inside HIP-graph capture)
        ts = timesteps[0] - torch.arange(n, device=device, dtype=torch.long) * stride
        return self.time_mlp(sinusoidal_embedding(ts, self.cfg.embedding_dim))

    def forward(self, trajectory, timestep, enc, need_weights: bool = False, prepared=None, time_emb=None):
        """trajectory (B,L,ngrip,9) noisy sample, timestep (B,), enc = Encoder outputs (`prepared` = prepare_context(enc),
        computed here when not given).  Returns (pred (B,L,ngrip,10), head_yaw (B,L,1) or None, cross-attention weights or None)."""
        cfg, D = self.cfg, self.cfg.embedding_dim
        B, L, G, _ = trajectory.shape
        nt = L * G
        P = prepared if prepared is not None else self.prepare_context(enc)
        if (not need_weights and time_emb is not None and P.get("seq") is not None and layers_mod._fused(trajectory)
                and nt == P["pos_table"].shape[0] and G <= 4 and not cfg.use_instruction):
            return self._forward_fused(trajectory, P, time_emb)
        tokens = self.drop(self.traj_encoder(trajectory)).flatten(1, 2)
        token_pos = sinusoidal_embedding(torch.arange(nt, device=tokens.device), D)[None]
        if cfg.use_instruction:
            lang = self.traj_lang_attention[0]
            out, _ = lang.attn(tokens + token_pos, enc["instr_feats"])
            tokens = lang.norm(tokens + lang.drop(out))
        tokens = tokens + token_pos
        cond = (self.time_mlp(sinusoidal_embedding(timestep, D)) if time_emb is None else time_emb) + P["history"]
        cond_act = F.silu(cond)  # every AdaLN block applies the same activation to the same conditioning vector
        if P.get("adaln") is not None and not need_weights:
            cond_act = P["adaln"].compute(cond_act)  # ... and all their projections are one GEMM
        ctx_feats, fps_feats, fps_rot = P["ctx_feats"], P["fps_feats"], P["fps_rot"]

        traj_rot = rotary3d(trajectory[..., :3].flatten(1, 2), D)
        tokens, weights = self.cross_attn(tokens, ctx_feats, cond, traj_rot, P["ctx_rot"], key_padding_mask=P["ctx_pad"],
                                          need_weights=need_weights, cond_act=cond_act, kv_caches=None if need_weights else P["cross_kv"])
        seq = torch.cat([tokens, fps_feats], dim=1)
        seq_rot = (torch.cat([traj_rot[0], fps_rot[0]], dim=1), torch.cat([traj_rot[1], fps_rot[1]], dim=1))
        pad = torch.cat([torch.zeros((B, nt), dtype=torch.bool, device=seq.device), P["fps_pad"]], dim=1)
        seq, _ = self.self_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=cond_act)
        if layers_mod._fused(seq):
            # the two output stacks are independent: fork the rotation stack onto a second stream (parallel branches of the
            # captured HIP graph; concurrent small kernels in eager mode), join before the projections
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=seq.device)
            main, side = torch.cuda.current_stream(seq.device), self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                rot_seq, _ = self.rotation_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=cond_act)
            pos_seq, _ = self.position_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=cond_act)
            main.wait_stream(side)
            rot_seq.record_stream(main)
        else:
            rot_seq, _ = self.rotation_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=cond_act)
            pos_seq, _ = self.position_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=cond_act)
        rot_feat = self.drop(self.rotation_proj(rot_seq[:, :nt]))
        pos_feat = self.drop(self.position_proj(pos_seq[:, :nt]))
        pred = torch.cat([self.position_out(pos_feat), self.rotation_out(rot_feat), self.openness_out(pos_feat)], dim=-1)
        head_yaw = self.head_yaw_out(pos_feat.reshape(B, L, G * D)) if self.head_yaw_out is not None else None
        if weights is not None:
            weights = weights.mean(dim=1)  # average over heads
        return pred.reshape(B, L, G, 10), head_yaw, weights


    def _stacks_fused(self, tokens, P, ada):
        """Cross-attention, self-attention and the two output stacks of a step on the whole-layer kernels over the
        preassembled sequence buffers; `ada` holds this step's AdaLN projections.  Returns (rotation seq, position seq)."""
        nt = tokens.shape[1]
        cond = P["history"]  # only its presence matters below: every AdaLN projection comes from `ada`
        traj_rot = (P["seq_cos"][:, :nt], P["seq_sin"][:, :nt])
        seq, seq_rot, pad, pad16 = P["seq"], (P["seq_cos"], P["seq_sin"]), P["seq_pad"], P.get("seq_pad16")
        head_rows = seq[:, :nt]  # batch 1: a contiguous view -- the cross-attention stack's last launch writes it directly
        tokens, _ = self.cross_attn(tokens, P["ctx_feats"], cond, traj_rot, P["ctx_rot"], key_padding_mask=P["ctx_pad"], cond_act=ada,
                                    kv_caches=P["cross_kv"], key_padding_mask16=P.get("ctx_pad16"),
                                    out_last=head_rows if head_rows.is_contiguous() else None, handover=P.get("cross_handover"))
        if tokens.data_ptr() != head_rows.data_ptr():
            head_rows.copy_(tokens)
        seq, _ = self.self_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=ada, key_padding_mask16=pad16,
                                handover=P.get("self_handover"))
        # the two output stacks are independent and of identical shape
        from . import fused_ops as FO

        ra, pa = self.rotation_attn, self.position_attn
        if (P.get("seq_pad16x2") is not None and (seq.shape[-1], ra.attn[0].attn.heads) == FO.MFMA_DIMS and len(ra.attn) == len(pa.attn)
                and ra.ffw[0].fc1.out_features == seq.shape[-1]):
            # ... every layer's three launches serve both (stack-major activations, the attention kernel sees a batch of 2 B)
            rot_seq, pos_seq = FO.paired_self_attention_stacks(ra, pa, seq, lambda adaln: None if adaln is None else ada.lookup(adaln),
                                                               seq_rot, P["seq_pad16x2"], ra.attn[0].attn.heads)
            return rot_seq.contiguous(), pos_seq.contiguous()
        # otherwise: fork the rotation stack onto a second stream (parallel branches of the captured HIP graph; concurrent small
        # kernels in eager mode), join before the output heads
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=seq.device)
        main, side = torch.cuda.current_stream(seq.device), self._side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            rot_seq, _ = self.rotation_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=ada, key_padding_mask16=pad16)
        pos_seq, _ = self.position_attn(seq, None, cond, seq_rot, key_padding_mask=pad, cond_act=ada, key_padding_mask16=pad16)
        main.wait_stream(side)
        rot_seq.record_stream(main)
        return rot_seq, pos_seq

    def _forward_fused(self, trajectory, P, time_emb):
        """One pass on the inference kernels end to end (DiffuserActor.enable_fused_inference): one prologue launch
        (trajectory tokens, conditioning, every AdaLN projection, trajectory rotary codes), the attention stacks, one launch
        for the projections and output MLPs."""
        from . import fused_ops as FO

        B, L, G, _ = trajectory.shape
        ada = P["adaln"]
        tokens, ada.all = FO.step_prologue(trajectory, self.traj_encoder, P["pos_table"], time_emb[0], P["history"], P["rot_freq"],
                                           ada.weight_t, ada.bias, P["seq_cos"], P["seq_sin"])
        rot_seq, pos_seq = self._stacks_fused(tokens, P, ada)
        pred, head_yaw = FO.head_outputs(self, rot_seq, pos_seq, B, L, G)
        return pred, head_yaw, None

    def can_denoise_fused(self, P, traj) -> bool:
        if self.cfg.use_instruction and layers_mod._fused(traj) and not getattr(self, "_warned_instruction", False):
            self._warned_instruction = True
            warnings.warn("use_instruction: the matrix-core inference kernels have no trajectory-language attention; the denoising "
                          "loop runs on the composite ops (about 10x slower)")
        return (P.get("seq") is not None and layers_mod._fused(traj) and traj.shape[1] * traj.shape[2] == P["pos_table"].shape[0]
                and traj.shape[2] <= 4 and not self.cfg.use_instruction)

    def denoise_fused(self, noise, P, time_table, coefs):
        """The whole reverse-diffusion loop on the inference kernels.  noise [1 + T, B, L, G, 9] (noise[0] = x_T), time_table
        [T, D] (time embeddings of the steps), coefs[k] = (position, rotation) DDPMScheduler.step_coefficients of step k.
        The AdaLN projections of ALL steps come from one GEMM (their input depends on the step index only); a step is
        [attention stacks] + ONE launch that produces the step's outputs, x_{t-1}, and the next step's tokens."""
        from . import fused_ops as FO

        traj = noise[0]
        ada = P["adaln"]
        T = time_table.shape[0]
        ada_steps = F.linear(F.silu(time_table[:, None, :] + P["history"][None]), ada.weight, ada.bias)  # [T, B, NA]
        tokens, _ = FO.step_prologue(traj, self.traj_encoder, P["pos_table"], None, None, P["rot_freq"], None, None, P["seq_cos"],
                                     P["seq_sin"])
        pred = head_yaw = None
        for k in range(T):
            ada.all = ada_steps[k]
            rot_seq, pos_seq = self._stacks_fused(tokens, P, ada)
            pred, head_yaw, traj, tokens = FO.step_tail(self, rot_seq, pos_seq, traj, noise[1 + k], coefs[k][0], coefs[k][1],
                                                        P["pos_table"], P["rot_freq"], P["seq_cos"], P["seq_sin"], last=(k == T - 1))
        # an in-launch hand-over that timed out (a peer workgroup was not running: fused_ops.Handover) must not pass for a result
        for key in ("cross_handover", "self_handover"):
            ho = P.get(key)
            if ho is not None:
                traj = torch.where(ho.words[-1] != 0, torch.full_like(traj, float("nan")), traj)
        return traj, pred, head_yaw


class DiffuserActor(nn.Module):
    def __init__(self, cfg: DiffuserActorConfig, workspace_bounds: torch.Tensor):
        super().__init__()
        self.cfg = cfg
        self.register_buffer("workspace_bounds", workspace_bounds.to(torch.float32).clone(), persistent=False)
        self.encoder = Encoder(cfg)
        self.prediction_head = DiffusionHead(cfg)
        self.position_noise_scheduler = DDPMScheduler(cfg.diffusion_timesteps, "scaled_linear")
        self.rotation_noise_scheduler = DDPMScheduler(cfg.diffusion_timesteps, "squaredcos_cap_v2")
        self._graph_sampler = None       # see enable_graph_sampling()
        self._inference_timesteps = []

    # -- shared encoding --------------------------------------------------------------------------------------------------
    def encode_inputs(self, rgb_obs, pcd_obs, pcd_valid_mask, vertex_features, vertices, vertices_valid_mask, instruction,
                      gripper_history, closedness, backbone_feats=None):
        enc = self.encoder
        feats, pos, mask = [], [], []
        if enc.uses_images:
            f, p, m = enc.encode_images(rgb_obs, pcd_obs, pcd_valid_mask, backbone_feats)
            feats.append(f), pos.append(p), mask.append(m)
        if enc.uses_mesh:
            assert vertices.ndim == 3 and vertices_valid_mask.ndim == 2 and vertices.shape[1] == vertices_valid_mask.shape[1]
            f, p = enc.encode_vertices(vertex_features, vertices)
            feats.append(f), pos.append(p), mask.append(vertices_valid_mask)
        context_feats, context_pos, context_mask = torch.cat(feats, 1), torch.cat(pos, 1), torch.cat(mask, 1)
        instr = None
        if self.cfg.use_instruction:
            context_feats, instr = enc.attend_instruction(context_feats, instruction)
        history_feats = enc.encode_gripper_history(gripper_history, context_feats, context_pos, closedness)
        if self.cfg.use_fps:
            with Timer("diffuser_actor/encode_inputs/fps"):
                fps_feats, fps_pos, fps_mask = enc.run_fps(context_feats, context_pos, context_mask)
        else:
            fps_feats, fps_pos, fps_mask = context_feats, context_pos, context_mask
        return {"context_feats": context_feats, "context_pos": context_pos, "context_mask": context_mask, "instr_feats": instr,
                "history_feats": history_feats, "fps_feats": fps_feats, "fps_pos": fps_pos, "fps_mask": fps_mask}

    # -- inference: full reverse diffusion ------------------------------------------------------------------------------
    @torch.no_grad()
    def _denoise(self, enc, noise):
        """The reverse diffusion loop as a pure function of tensors: noise[0] is x_T, noise[1 + k] feeds the variance term
        of step k.  No host synchronisation, no RNG: eager and HIP-graph replays give identical results."""
        cfg = self.cfg
        traj = noise[0]
        batch, device = traj.shape[0], traj.device
        pred = head_yaw = None
        prepared = self.prediction_head.prepare_context(enc)  # step-invariant part of the head, once per inference
        fused = layers_mod._fused(traj)
        time_table = self.prediction_head.time_embeddings(self._inference_timesteps, device) if fused else None
        head = self.prediction_head
        if fused and head.can_denoise_fused(prepared, traj):
            coefs = [(self.position_noise_scheduler.step_coefficients(t), self.rotation_noise_scheduler.step_coefficients(t))
                     for t in self._inference_timesteps]
            traj, pred, head_yaw = head.denoise_fused(noise, prepared, time_table, coefs)
            return torch.cat([traj, pred[..., 9:]], dim=-1), head_yaw
        for k, t in enumerate(self._inference_timesteps):
            if fused:
                from .fused_ops import ddpm_step

                pred, head_yaw, _ = self.prediction_head(traj, None, enc, prepared=prepared, time_emb=time_table[k:k + 1].expand(batch, -1))
                traj = ddpm_step(traj, pred, noise[1 + k], self.position_noise_scheduler.step_coefficients(t),
                                 self.rotation_noise_scheduler.step_coefficients(t))
                continue
            ts = torch.full((batch,), t, dtype=torch.long, device=device)
            pred, head_yaw, _ = self.prediction_head(traj, ts, enc, prepared=prepared)
            pos = self.position_noise_scheduler.step(pred[..., :3], t, traj[..., :3], noise=noise[1 + k][..., :3])
            rot = self.rotation_noise_scheduler.step(pred[..., 3:9], t, traj[..., 3:9], noise=noise[1 + k][..., 3:9])
            traj = torch.cat([pos, rot], dim=-1)
        return torch.cat([traj, pred[..., 9:]], dim=-1), head_yaw  # openness / head yaw are not diffused

    def sample_trajectory(self, enc, batch: int, device, generator=None):
        """DDPM sampling of the trajectory (diffuser_actor.py:conditional_sample).  All Gaussian noise of the loop is drawn
        up front in ONE call ([1 + T, B, L, G, 9]); with ``enable_graph_sampling()`` the loop itself is one HIP-graph launch."""
        cfg = self.cfg
        self.position_noise_scheduler.set_timesteps(cfg.diffusion_timesteps)
        self.rotation_noise_scheduler.set_timesteps(cfg.diffusion_timesteps)
        self._inference_timesteps = self.position_noise_scheduler.timesteps.tolist()
        shape = (1 + len(self._inference_timesteps), batch, cfg.prediction_horizon, cfg.ngrippers, 9)
        noise = torch.randn(shape, device=device, generator=generator)
        if self._graph_sampler is not None and torch.device(device).type == "cuda" and not torch.is_grad_enabled():
            return self._graph_sampler.run(enc, noise)
        return self._denoise(enc, noise)

    @staticmethod
    def enable_fused_inference(on: bool = True) -> None:
        """Use the fused HIP ops (rotary, AdaLN, small attention; cached context keys/values) in blocks that run without
        autograd on the GPU.  Process-wide switch; training (autograd on) always takes the composite torch ops."""
        layers_mod.FUSED_INFERENCE = bool(on)

    def enable_graph_sampling(self, on: bool = True) -> None:
        """Replay the denoising loop (T steps x ~150 small kernels, launch-bound at batch 1) as a captured HIP graph.
        Used under torch.no_grad() / inference_mode on the GPU; captured per distinct input shape, weights are read at
        replay time (loading new weights into the same parameters needs no re-capture)."""
        from .graph_sampler import GraphSampler

        self._graph_sampler = GraphSampler(self) if on else None

    # -- forward --------------------------------------------------------------------------------------------------------
    def forward(self, gt_gripper_pred, gt_head_yaw, rgb_obs, pcd_obs, pcd_valid_mask, vertex_features, vertices,
                vertices_valid_mask, instruction, gripper_history, run_inference: bool = False, backbone_feats=None):
        """Arguments as in the reference (diffuser_actor.py:518-531):
          gt_gripper_pred (B,L,ngrip,8) xyz + quaternion + openness (or None at inference), gt_head_yaw (B,L,1),
          rgb_obs (B,ncam,3,H,W) in [0,1], pcd_obs (B,ncam,3,H,W) world points, pcd_valid_mask (B,ncam,H,W),
          vertex_features (B,N,C), vertices (B,N,3) world, vertices_valid_mask (B,N), instruction (B,T,512) or None,
          gripper_history (B,nhist,ngrip,8).
        Training: returns ((total, pos, rot, gripper, head_yaw) losses, encoded inputs, None).
        Inference: returns (trajectory (B,L,ngrip,8), head_yaw, losses or None, encoded inputs, None)."""
        cfg, wb = self.cfg, self.workspace_bounds
        closedness = gripper_history[..., 7:8]
        history = gripper_history[..., :7]
        current_pose = None
        if cfg.relative_action:
            # reference :554-566 -- the point cloud and the history are translated to the newest gripper pose, the target
            # trajectory is translated and rotated; the map's vertices stay in the world frame there, so they do here
            current_pose = get_current_pose_from_gripper_history(history)
            if pcd_obs is not None:
                pcd_obs = to_relative_pcd(pcd_obs, current_pose)
            history = to_relative_gripper_history(history, current_pose)
            if gt_gripper_pred is not None:
                gt_gripper_pred = to_relative_trajectory(gt_gripper_pred, current_pose)
        history = normalize_trajectory(history, wb, cfg.quaternion_format)
        if pcd_obs is not None:
            pcd_obs, inside = normalize_pointcloud(pcd_obs, wb)
            pcd_valid_mask = pcd_valid_mask & inside
        if vertices is not None:
            vertices, _ = normalize_pos(vertices, wb)
        gt, gt_open = None, None
        if gt_gripper_pred is not None:
            assert gt_gripper_pred.shape[-1] == 8
            gt_open = gt_gripper_pred[..., 7:]
            gt = normalize_trajectory(gt_gripper_pred[..., :7], wb, cfg.quaternion_format)
        with Timer("diffuser_actor/encode_inputs"):
            enc = self.encode_inputs(rgb_obs, pcd_obs, pcd_valid_mask, vertex_features, vertices, vertices_valid_mask,
                                     instruction, history, closedness, backbone_feats)
        B, dev = history.shape[0], history.device
        if run_inference:
            traj, head_yaw = self.sample_trajectory(enc, B, dev)
            losses = None
            if gt is not None:
                losses = compute_loss(traj, head_yaw, gt, gt_open, gt_head_yaw, cfg.loss_weights, cfg.predict_head_yaw)
            traj = unnormalize_trajectory(traj, wb, cfg.quaternion_format, cfg.rotation_parametrization)
            if cfg.relative_action:
                traj = to_absolute_trajectory(traj, current_pose)  # reference :509-510
            if head_yaw is not None:
                head_yaw = head_yaw.clamp(-torch.pi, torch.pi - 1e-6)
            return traj, head_yaw, losses, enc, None
        noise = torch.randn(gt.shape, device=dev)
        timesteps = torch.randint(0, cfg.diffusion_timesteps, (B,), device=dev)
        noisy = torch.cat([self.position_noise_scheduler.add_noise(gt[..., :3], noise[..., :3], timesteps),
                           self.rotation_noise_scheduler.add_noise(gt[..., 3:9], noise[..., 3:9], timesteps)], dim=-1)
        with Timer("diffuser_actor/policy_forward_pass"):
            pred, head_yaw, _ = self.prediction_head(noisy, timesteps, enc)
        losses = compute_loss(pred, head_yaw, noise, gt_open, gt_head_yaw, cfg.loss_weights, cfg.predict_head_yaw)
        return losses, enc, None
