"""Per-frame inputs of the mapper (mirror of mindmap/mapping/helpers/nvblox_input_helpers.py:18-124).

Same function names, argument meaning, returned tuple and assertions as the reference.  Two differences:
  * the reference hard-codes 512x512 in its shape assertions (:55,74); here the image size is whatever the sample holds (the
    layout ``[1, num_cams, ...]`` is asserted all the same), so the 640x480 benchmark stream goes through the same door;
  * the back-projection is the HIP kernel (``mmf_backproject_depth``) instead of batched matmuls; the single pose is
    converted on the host like in the reference (quat2mat in float64, backprojection.py:34-36) and the 4x4 is RETURNED ON THE
    HOST: every consumer of it takes it there (``integrate_frame`` calls ``camera_pose.cpu()``, nvblox_mapping_helpers.py:208),
    so the reference's device round trip (host -> device -> ``.cpu()``) and its synchronisation are saved.
"""
from typing import Dict, List, Tuple

import torch

from ...geometry.transforms import _pose_to_homo_host
from ...image_processing.backprojection import _backproject_chw


def get_nvblox_inputs_from_sample(sample: Dict[str, torch.Tensor], camera_index: int
                                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """One camera of a loader sample, formatted for nvblox_integrate (:18-82).

    sample: ``depths`` (1,ncam,H,W) f32 metres; ``intrinsics`` (1,ncam,3,3) f32; ``camera_poses`` (1,ncam,7) f32
    ``[x,y,z,qw,qx,qy,qz]``; ``rgbs`` (1,ncam,3,H,W) f32 in [0,1]; ``segmentation_masks`` (1,ncam,H,W) bool.
    Returns (depth_frame (H,W) f32, intrinsics (3,3) f32, camera_pose (4,4) f32 ON THE HOST, rgb (H,W,3) u8 -- ``(rgb * 255)`` TRUNCATED to
    uint8 as the reference does (:69), dynamic_mask (H,W) bool, pointcloud (3,H,W) f32 world frame)."""
    num_cams = sample["depths"].shape[1]
    assert camera_index < num_cams
    H, W = sample["depths"].shape[-2:]

    assert sample["depths"].shape == torch.Size([1, num_cams, H, W])
    assert sample["depths"].dtype == torch.float32
    depth_frame = sample["depths"].squeeze(0)[camera_index, ...]

    assert sample["intrinsics"].shape == torch.Size([1, num_cams, 3, 3]), f"intrinsics shape is {sample['intrinsics'].shape}"
    assert sample["intrinsics"].dtype == torch.float32
    intrinsics = sample["intrinsics"].squeeze(0)[camera_index, ...]

    assert sample["camera_poses"].shape == torch.Size([1, num_cams, 7]), f"camera_poses shape is {sample['camera_poses'].shape}"
    assert sample["camera_poses"].dtype == torch.float32
    camera_pose = sample["camera_poses"].squeeze(0)[camera_index, ...]
    camera_pose_homo = torch.from_numpy(_pose_to_homo_host(camera_pose.detach().to("cpu").numpy()))  # (4,4) float32, host

    assert sample["rgbs"].shape == torch.Size([1, num_cams, 3, H, W]), f"rgbs shape is {sample['rgbs'].shape}"
    assert sample["rgbs"].dtype == torch.float32
    lo_hi = torch.stack(torch.aminmax(sample["rgbs"])).tolist()  # the reference's two range assertions (:68) with ONE sync
    assert lo_hi[0] >= 0 and lo_hi[1] <= 1
    rgb = (sample["rgbs"].squeeze(0)[camera_index, ...].permute(1, 2, 0) * 255).to(torch.uint8)

    assert sample["segmentation_masks"].shape == torch.Size([1, num_cams, H, W])
    assert sample["segmentation_masks"].dtype == torch.bool
    dynamic_mask = sample["segmentation_masks"].squeeze(0)[camera_index]

    # get_camera_pointcloud(intrinsics, depth, position, orientation) (:76-81) with the transform already in hand
    pointcloud = _backproject_chw(depth_frame.unsqueeze(0), intrinsics.unsqueeze(0),
                                  camera_pose_homo.to(depth_frame.device, non_blocking=True).unsqueeze(0)).squeeze(0)
    return (depth_frame, intrinsics, camera_pose_homo, rgb, dynamic_mask, pointcloud)


def get_nvblox_inputs_from_camera_handler(camera_handler, dynamic_class_labels: List[str]
                                          ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """The same tuple from a live camera (:85-124).  ``camera_handler`` is the simulator's object in the reference
    (isaaclab_utils/isaaclab_camera_handler.py, out of scope here); anything with its six getters works:
    get_dynamic_segmentation(labels), get_depth(), get_intrinsics(), get_pose_as_homo(), get_rgb(), get_pcd()."""
    dynamic_mask = camera_handler.get_dynamic_segmentation(dynamic_class_labels).squeeze().to("cuda")
    assert dynamic_mask.dtype == torch.bool
    depth_frame = camera_handler.get_depth().to("cuda")
    assert depth_frame.dtype == torch.float32
    intrinsics = camera_handler.get_intrinsics().to("cuda")
    assert intrinsics.dtype == torch.float32
    camera_pose = camera_handler.get_pose_as_homo().to(torch.float32).to("cuda")
    rgb = camera_handler.get_rgb().to(torch.uint8).to("cuda")
    pointcloud = camera_handler.get_pcd()
    assert pointcloud.dtype == torch.float32
    return (depth_frame, intrinsics, camera_pose, rgb, dynamic_mask, pointcloud)
