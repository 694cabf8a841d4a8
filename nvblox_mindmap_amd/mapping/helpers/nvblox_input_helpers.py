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

import numpy as np
import torch

from ... import _lib
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


class _SampleScratch:
    """Per-device kernel scratch of ``frame_inputs_from_sample`` (calls are serialised inside the library)."""

    _by_device = {}

    def __init__(self, device):
        n = _lib.lib().mmf_sample_inputs_scratch_floats()
        self.scratch = torch.empty(n, dtype=torch.float32, device=device)
        self.scratch_ptr = _lib.dptr(self.scratch)
        self.call = _lib.lib().mmf_sample_frame_inputs_host

    @classmethod
    def of(cls, device):
        key = (device.type, device.index)
        sc = cls._by_device.get(key)
        if sc is None:
            sc = cls._by_device[key] = cls(device)
        return sc


def frame_inputs_from_sample(sample: Dict[str, torch.Tensor], camera_index: int):
    """``get_nvblox_inputs_from_sample`` for a caller that integrates the frame at once (the facade): same checks, same
    values, but ONE native call (rgb -> uint8 HWC, the range check's operands, pose and intrinsics gathered into one record)
    and ONE device->host copy instead of three synchronising reads (pose, range check, intrinsics); the intrinsics come back
    ON THE HOST like the pose (the mapper takes both there), and the world-frame point cloud -- read by the visualiser only --
    as a function that computes it on first use.
    Returns (depth_frame, intrinsics (3,3) host, camera_pose (4,4) host, rgb (H,W,3) u8, dynamic_mask, pointcloud_fn)."""
    num_cams = sample["depths"].shape[1]
    assert camera_index < num_cams
    H, W = sample["depths"].shape[-2:]
    assert sample["depths"].shape == torch.Size([1, num_cams, H, W])
    assert sample["depths"].dtype == torch.float32
    assert sample["intrinsics"].shape == torch.Size([1, num_cams, 3, 3]), f"intrinsics shape is {sample['intrinsics'].shape}"
    assert sample["intrinsics"].dtype == torch.float32
    assert sample["camera_poses"].shape == torch.Size([1, num_cams, 7]), f"camera_poses shape is {sample['camera_poses'].shape}"
    assert sample["camera_poses"].dtype == torch.float32
    assert sample["rgbs"].shape == torch.Size([1, num_cams, 3, H, W]), f"rgbs shape is {sample['rgbs'].shape}"
    assert sample["rgbs"].dtype == torch.float32
    assert sample["segmentation_masks"].shape == torch.Size([1, num_cams, H, W])
    assert sample["segmentation_masks"].dtype == torch.bool
    depth_frame = sample["depths"][0, camera_index]
    dynamic_mask = sample["segmentation_masks"][0, camera_index]
    rgb_chw = sample["rgbs"][0, camera_index]
    pose7 = sample["camera_poses"][0, camera_index]
    k_dev = sample["intrinsics"][0, camera_index]
    if not (rgb_chw.is_cuda and rgb_chw.is_contiguous() and pose7.is_contiguous() and k_dev.is_contiguous()):
        d, K, T, rgb, dyn, pcd = get_nvblox_inputs_from_sample(sample, camera_index)
        return d, K.detach().to("cpu"), T, rgb, dyn, (lambda: pcd)
    dev = rgb_chw.device
    sc = _SampleScratch.of(dev)
    rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    # the record arrives on the host with the call: the kernel stores it into coherent pinned memory the library polls -- no copy
    # engine, no stream synchronisation (the step's host path: DESIGN.md sections 4.5 and 8)
    # (the record is this call's own array: the native call serialises concurrent callers -- and with them the shared device scratch --
    # but hands each its result in the buffer it passed)
    rec = np.empty(20, dtype=np.float32)
    _lib.check(sc.call(rgb_chw.data_ptr(), H, W, pose7.data_ptr(), k_dev.data_ptr(), rgb.data_ptr(), sc.scratch_ptr, rec.ctypes.data,
                       _lib.stream_ptr(dev)), "mmf_sample_frame_inputs_host")
    # the reference's two range assertions (:68); a NaN fails them like it fails `min() >= 0`
    assert rec[2] == 0.0 and rec[0] >= 0 and rec[1] <= 1
    if num_cams > 1:  # the reference checks the range over every camera of the sample
        lo_hi = torch.stack(torch.aminmax(sample["rgbs"])).tolist()
        assert lo_hi[0] >= 0 and lo_hi[1] <= 1
    camera_pose_homo = torch.from_numpy(_pose_to_homo_host(rec[4:11]))
    intrinsics = torch.from_numpy(rec[11:20].reshape(3, 3))

    def pointcloud():  # get_camera_pointcloud(intrinsics, depth, position, orientation) (:76-81), on first use
        return _backproject_chw(depth_frame.unsqueeze(0), k_dev.unsqueeze(0), camera_pose_homo.to(dev).unsqueeze(0)).squeeze(0)

    return depth_frame, intrinsics, camera_pose_homo, rgb, dynamic_mask, pointcloud


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
    raw_rgb = camera_handler.get_rgb()
    rgb = raw_rgb.to(torch.uint8).to("cuda")
    if rgb is raw_rgb or rgb.data_ptr() == raw_rgb.data_ptr():
        # a uint8 image already on the GPU comes back as the handler's OWN tensor: the facade pipelines consecutive frames (the
        # appearance half of this frame reads the image while the next frame is being set up), so the frame gets a copy of its own
        # -- the reference's handler clones per frame anyway (isaaclab_utils/isaaclab_camera_handler.py:130); 0.8 MB, microseconds
        rgb = rgb.clone()
    pointcloud = camera_handler.get_pcd()
    assert pointcloud.dtype == torch.float32
    return (depth_frame, intrinsics, camera_pose, rgb, dynamic_mask, pointcloud)
