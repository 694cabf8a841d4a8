from .transforms import pose_to_homo, quaternion_wxyz_to_matrix  # noqa: F401
