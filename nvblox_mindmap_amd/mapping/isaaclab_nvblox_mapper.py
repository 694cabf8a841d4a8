"""The object the policy and the data generator drive (mirror of mindmap/mapping/isaaclab_nvblox_mapper.py:35-258): owns the
two-mapper Mapper, the feature extractor and the task's mapping configuration, and exposes

    update_reconstruction_from_sample(sample, camera_name)     :96-122   loader sample -> one fused frame per mapper
    update_reconstruction_from_camera(camera_handler)          :74-94    live camera (duck-typed, the simulator is out of scope)
    get_nvblox_model_inputs(mapper_id, remove_zero_features)   :207-250  map -> {vertices, vertex_features, vertices_valid_mask}
    save_nvblox_map_to_disk(frame_index, root_directory)       :166-205
    decay() / clear()                                          :252-258

as called by closed_loop/policies/nvblox_diffuser_actor_policy.py:77-83,206-211 and run_isaaclab_datagen.py.  Differences
from the reference, none of them in behaviour: the constructor takes the handful of arguments it reads from the reference's
``args`` bag as keywords (``args`` with the same attribute names is accepted too), the feature extractor is passed in (the
reference builds RADIO / DINO / CLIP extractors from a registry that needs network access -- the DNN is out of scope, its
``compute`` / ``num_excess_features`` contract is not), and the visualiser-only point-cloud image is produced on first access.
"""
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from ..data_loading.vertex_sampling import VertexSamplingMethod
from ..image_processing.feature_extraction import _held_output
from ..image_processing.feature_resize import upsample_features
from .helpers.nvblox_input_helpers import frame_inputs_from_sample, get_nvblox_inputs_from_camera_handler
from .helpers.nvblox_mapping_helpers import get_nvblox_mapper, nvblox_integrate
from .helpers.nvblox_output_helpers import get_vertices_and_features
from .helpers.nvblox_to_disk_helpers import save_feature_mesh_to_disk, save_serialized_nvblox_map_to_disk
from .nvblox_mapper_constants import CAMERA_NAME_TO_ID, MAPPER_TO_ID, NvbloxMappingCfg


def includes_mesh(mapping_data_type) -> bool:
    """data_loading/data_types.py: MESH and RGBD_AND_MESH carry the feature mesh."""
    name = getattr(mapping_data_type, "name", str(mapping_data_type)).lower()
    return name in ("mesh", "rgbd_and_mesh")


class BackboneFeatureExtractor:
    """FeatureExtractor contract of the reference (image_processing/feature_extraction.py:132-216) around any image backbone
    ``(B,3,H,W) in [0,1] -> (B,C,h,w)``: ``compute(rgb [1,H,W,3] u8) -> [1,Hf,Wf,C_pad]`` with the bilinear resize to
    ``desired_output_size``, channels-last layout, zero padding to the mapper's channel count and the float16 cast done by ONE
    HIP kernel (``mmf_upsample_features``: the reference's :188-210 chain + nvblox_mapping_helpers.py:256)."""

    def __init__(self, backbone: torch.nn.Module, desired_output_size: Tuple[int, int], pad_to_channels: int, input_size=None):
        self.backbone = backbone.eval()
        self.desired_output_size = tuple(desired_output_size)
        self.pad_to_channels = int(pad_to_channels)
        self.input_size = input_size
        self._channels = None
        # the spec switch of the mapper this extractor feeds (None: the process default, MMF_FMA_CONTRACTION); set by
        # follow_mapper_arithmetic() -- the materialised image must blend with the arithmetic the low-res path uses inside the mapper
        self.fma_contraction = None

    def _backbone_output(self, rgb: torch.Tensor, hold: bool = False) -> torch.Tensor:
        assert rgb.ndim == 4 and rgb.shape[0] == 1 and rgb.shape[-1] == 3

        def run():
            x = rgb.permute(0, 3, 1, 2).to(torch.float32) / 255.0
            if self.input_size is not None and tuple(x.shape[-2:]) != tuple(self.input_size):
                x = F.interpolate(x, self.input_size, mode="bilinear", align_corners=False)
            low = self.backbone(x)
            self._channels = int(low.shape[1])
            return low[0]

        # compute_lowres then compute on one image object: one backbone run (feature_extraction._held_output: explicit, one-shot)
        return _held_output(self, rgb, hold, run)

    def release_lowres(self) -> None:
        self._held = None

    @torch.no_grad()
    def compute(self, rgb: torch.Tensor) -> torch.Tensor:
        return upsample_features(self._backbone_output(rgb), self.desired_output_size, self.pad_to_channels, self.fma_contraction).unsqueeze(0)

    @torch.no_grad()
    def compute_lowres(self, rgb: torch.Tensor):
        """Extension read by ``nvblox_integrate``: (the backbone's own output as [h, w, C] float32, the size ``compute`` would
        resize it to).  The native integration samples the low-res map itself -- same result as integrating ``compute(rgb)``,
        bit for bit, without the [Hf, Wf, C_pad] float16 image ever existing."""
        low = self._backbone_output(rgb, hold=True)
        if low.shape[0] % 8 != 0:  # the fused path moves 8 channels at a time
            return None, self.desired_output_size
        return low.permute(1, 2, 0).to(torch.float32).contiguous(), self.desired_output_size

    def num_excess_features(self) -> int:
        return 0 if self._channels is None else self.pad_to_channels - self._channels


class IsaacLabNvbloxMapper:
    def __init__(self, mapping_data_type="rgbd_and_mesh", args=None, device: str = "cuda", *, feature_extractor=None,
                 task: Optional[str] = None, include_dynamic: Optional[bool] = None, num_vertices_to_sample: Optional[int] = None,
                 vertex_sampling_method: Optional[VertexSamplingMethod] = None, save_serialized_nvblox_map_to_disk: Optional[bool] = None,
                 feature_channels: Optional[int] = None, frame_pipelining: Optional[bool] = None) -> None:
        def pick(value, name, default):
            return value if value is not None else getattr(args, name, default)

        self.mapping_data_type = mapping_data_type
        self.include_dynamic = bool(pick(include_dynamic, "include_dynamic", False))
        self.num_vertices_to_sample = pick(num_vertices_to_sample, "num_vertices_to_sample", 2048)
        self.vertex_sampling_method = pick(vertex_sampling_method, "vertex_sampling_method", VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
        self.save_serialized_nvblox_map_to_disk = bool(pick(save_serialized_nvblox_map_to_disk, "save_serialized_nvblox_map_to_disk", False))
        self.device = device
        self.mapping_config = NvbloxMappingCfg(pick(task, "task", "DRILL_IN_BOX"))
        self.mapper = get_nvblox_mapper(self.mapping_config, feature_channels=feature_channels)
        name = getattr(mapping_data_type, "name", str(mapping_data_type)).lower()
        if name == "mesh" and self.include_dynamic:
            raise ValueError("Dynamics are not supported for mesh generation yet.")
        if feature_extractor is None:
            raise ValueError("pass feature_extractor=: an object with compute(rgb=[1,H,W,3] u8) -> [1,Hf,Wf,C_pad] and "
                             "num_excess_features() (e.g. BackboneFeatureExtractor around the image backbone)")
        self.feature_extractor = feature_extractor
        # the last nvblox_integration_images per camera (the visualiser reads them)
        self.last_nvblox_integration_images: Dict[str, Dict] = {}
        # Consecutive frames are software-pipelined BY DEFAULT here (not in the plain Mapper): every tensor the deferred half of a
        # frame reads is made per frame by this object or its helpers (the uint8 image of frame_inputs_from_sample / the camera
        # handler's per-frame copy, the extractor's output, the masks the native call writes), and every reader of the map completes
        # the pending frame first.  The closed loop's step() integrates several cameras / frames between two map reads
        # (mindmap/closed_loop/policies/nvblox_diffuser_actor_policy.py:77-83,206-211): those overlap.  Results are bit-identical.
        # ``frame_pipelining=False`` (or MMF_FACADE_PIPELINING=0) gives the frame-at-a-time schedule back.
        if frame_pipelining is None:
            import os

            frame_pipelining = os.environ.get("MMF_FACADE_PIPELINING", "1") != "0"
        self.frame_pipelining = bool(frame_pipelining)
        self.set_frame_pipelining(self.frame_pipelining)

    # -- map update ---------------------------------------------------------------------------------------------------------
    def update_reconstruction_from_camera(self, camera_handler) -> None:
        inputs = get_nvblox_inputs_from_camera_handler(camera_handler, self.mapping_config.dynamic_class_labels)
        self._update_reconstruction(*inputs, camera_handler.camera_name)

    def update_reconstruction_from_sample(self, sample: Dict[str, torch.Tensor], camera_name: str) -> None:
        num_cams = sample["depths"].shape[1]
        camera_index = 0 if num_cams == 1 else CAMERA_NAME_TO_ID[camera_name]
        # (get_nvblox_inputs_from_sample with one native call and one synchronisation; the point cloud on first use)
        self._update_reconstruction(*frame_inputs_from_sample(sample, camera_index), camera_name)

    def _update_reconstruction(self, depth_frame, intrinsics, camera_pose, rgb, dynamic_mask, pointcloud, camera_name: str) -> None:
        images = nvblox_integrate(mapper=self.mapper, nvblox_mapping_config=self.mapping_config, feature_extractor=self.feature_extractor,
                                  depth_frame=depth_frame, intrinsics=intrinsics, camera_pose=camera_pose, rgb=rgb,
                                  dynamic_mask=dynamic_mask, include_dynamic=self.include_dynamic)
        size = self.mapping_config.upscaled_feature_image_size

        def pcd_image():  # visualization/utils.py:17-23 -- only the visualiser asks for it
            pcd = pointcloud() if callable(pointcloud) else pointcloud
            chw = pcd if pcd.shape[0] == 3 else pcd.permute(2, 0, 1)
            return F.interpolate(chw.unsqueeze(0), size, mode="bilinear").permute(0, 2, 3, 1)

        for mapper_name in [m.name for m in MAPPER_TO_ID]:
            if mapper_name in images and hasattr(images[mapper_name], "_lazy"):
                images[mapper_name]._lazy["pcd"] = pcd_image
        self.last_nvblox_integration_images[camera_name] = images

    # -- outputs ------------------------------------------------------------------------------------------------------------
    def save_nvblox_map_to_disk(self, frame_index: int, root_directory: str):
        features = vertices = None
        if includes_mesh(self.mapping_data_type):
            vertices, features = save_feature_mesh_to_disk(self.mapper, self.mapping_config, self.feature_extractor.num_excess_features(),
                                                           frame_index, root_directory, self.include_dynamic)
        if self.save_serialized_nvblox_map_to_disk:
            save_serialized_nvblox_map_to_disk(self.mapper, root_directory, frame_index, self.include_dynamic)
        return vertices, features

    def get_nvblox_model_inputs(self, mapper_id: int, remove_zero_features: bool) -> Dict[str, torch.Tensor]:
        """{"vertices" [1,N,3] f32, "vertex_features" [1,N,C] f32, "vertices_valid_mask" [1,N] bool} on ``self.device``."""
        if not includes_mesh(self.mapping_data_type):
            raise NotImplementedError(f"Invalid data type: {self.mapping_data_type}")
        samples = {}
        samples["vertices"], samples["vertex_features"], samples["vertices_valid_mask"] = get_vertices_and_features(
            self.mapper, mapper_id, self.mapping_config, remove_zero_features, self.feature_extractor.num_excess_features(),
            sample_vertices=True, number_of_vertices_to_sample=self.num_vertices_to_sample,
            vertex_sampling_method=self.vertex_sampling_method, features_dtype=torch.float32)
        # (the reference's .to(torch.float32).to(self.device) of both tensors, :243-246: the gather writes float32 on the device)
        samples["vertex_features"] = samples["vertex_features"].to(self.device)
        samples["vertices"] = samples["vertices"].to(self.device)
        return samples

    def set_frame_pipelining(self, on: bool = True) -> None:
        """Extension, ON by default (see the constructor): consecutive updates are software-pipelined in the native library
        (``Mapper.set_deferred_feature_rows``) -- the appearance half of a frame runs beside the geometry half of the
        next one.  It pays wherever the map is not read after every frame (several cameras or frames per control step, dataset
        generation over a recorded demo, replay).  Results are bit-identical; whatever reads the map (``get_nvblox_model_inputs``,
        ``save_nvblox_map_to_disk``, layer views) completes the last frame first.  Every update makes its own image tensors, which is
        all the mode asks for (with ``include_dynamic`` both mappers' frames share them); a feature extractor that REUSES one output
        buffer across frames with torch in-place operations is reported by the mapper (RuntimeError), not silently mis-fused."""
        self.frame_pipelining = bool(on)
        self.mapper.set_deferred_feature_rows(bool(on))

    def clear(self):
        self.mapper.clear()

    def decay(self):
        self.mapper.decay()


SampleNvbloxMapper = IsaacLabNvbloxMapper  # the simulator-free name
