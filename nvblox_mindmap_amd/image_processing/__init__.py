from .backprojection import backproject_depth_to_pointcloud, get_camera_pointcloud  # noqa: F401
from .feature_resize import upsample_features  # noqa: F401
from .image_mask_operations import (  # noqa: F401
    depth_mask, downscale_mask, erode_mask, feature_mask, frame_masks, get_border_mask)
