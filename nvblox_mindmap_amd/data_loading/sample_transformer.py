"""Per-item transforms of the data loader (mirror of mindmap/data_loading/sample_transformer.py:28-73).  They are plain
tensor ops and run wherever the sample lives -- on the GPU when the loader hands device tensors (section 8(f) N4)."""
import torch

from ..mapping.nvblox_mapper_constants import DEPTH_SCALE_FACTOR


class SampleTransformer:
    def reset(self):
        pass

    def __call__(self, sample: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class RgbTransformer(SampleTransformer):
    """[H,W,3] in [0,255] -> [3,H,W] float32 in [0,1] (image_processing/image_conversions.py:13-38)."""

    def __call__(self, image):
        assert image.dim() == 3 and image.shape[-1] == 3
        return (image / 255.0).permute(2, 0, 1).type(torch.float32)


class DepthTransformer(SampleTransformer):
    """u16 millimetres -> float32 metres (:62-73)."""

    def __call__(self, image):
        return (image / DEPTH_SCALE_FACTOR).to(torch.float32)


# ---- geometry augmentation (sample_transformer.py:76-300; off in the reference's default arguments, cli/args.py:85-86) -----------
def _quat_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), dim=-1)


def _euler_xyz_to_quaternion(angles_rad: torch.Tensor) -> torch.Tensor:
    """(..., 3) roll / pitch / yaw -> unit quaternions, real part first and non-negative, of R = Rx(roll) Ry(pitch) Rz(yaw)
    (pytorch3d's euler_angles_to_matrix(..., "XYZ") followed by matrix_to_quaternion)."""
    h = angles_rad * 0.5
    c, s = torch.cos(h), torch.sin(h)
    zero = torch.zeros_like(c[..., 0])
    qx = torch.stack((c[..., 0], s[..., 0], zero, zero), dim=-1)
    qy = torch.stack((c[..., 1], zero, s[..., 1], zero), dim=-1)
    qz = torch.stack((c[..., 2], zero, zero, s[..., 2]), dim=-1)
    q = _quat_raw_multiply(_quat_raw_multiply(qx, qy), qz)
    return torch.where(q[..., :1] < 0, -q, q)


def random_transform_uniform(random_translation_range_m, random_rpy_range_deg, rng=None):
    """One rigid transform, translation and roll / pitch / yaw uniform in the given (lower, upper) bounds; six draws from
    Python's ``random`` (or ``rng``, a ``random.Random`` of the caller's own), translation first, like the reference (:188-219).
    Returns (translation [3], quaternion [4])."""
    import random as _random

    random = rng if rng is not None else _random

    t = torch.tensor([random.uniform(random_translation_range_m[0][i], random_translation_range_m[1][i]) for i in range(3)])
    rpy = torch.tensor([random.uniform(random_rpy_range_deg[0][i], random_rpy_range_deg[1][i]) for i in range(3)])
    return t, _euler_xyz_to_quaternion(torch.deg2rad(rpy))


def random_transform_gaussian(pos_stddev_m: float, rot_stddev_deg: float, num_transforms: int, generator=None):
    """``num_transforms`` independent small transforms: zero-mean Gaussian translation and roll / pitch / yaw (two
    ``torch.normal`` draws of shape [N, 3], translation first, :222-245).  Returns ([N, 3], [N, 4])."""
    shape = (num_transforms, 3)
    t = torch.normal(mean=torch.zeros(shape), std=torch.full(shape, pos_stddev_m), generator=generator)
    rpy = torch.normal(mean=torch.zeros(shape), std=torch.full(shape, torch.deg2rad(torch.tensor(rot_stddev_deg))), generator=generator)
    return t, _euler_xyz_to_quaternion(rpy)


def apply_random_transform_to_sample(sample: torch.Tensor, random_translation: torch.Tensor, random_rotation: torch.Tensor) -> torch.Tensor:
    """T_BA applied to points (..., 3) or poses (..., 8 = position, quaternion real-first, gripper state): B_t = R_BA A_t + B_t_BA,
    R_BW = R_BA R_AW with the product's real part made non-negative (:248-297)."""
    from ..diffuser_actor.relative_conversions import quaternion_invert, quaternion_multiply

    assert sample.shape[-1] in (3, 8)
    pos = sample[..., :3]
    as_quat = torch.cat((pos.new_zeros(pos.shape[:-1] + (1,)), pos), dim=-1)
    rotated = _quat_raw_multiply(_quat_raw_multiply(random_rotation, as_quat), quaternion_invert(random_rotation))[..., 1:]
    moved = rotated + random_translation
    if sample.shape[-1] == 8:
        moved = torch.cat((moved, quaternion_multiply(random_rotation, sample[..., 3:7]), sample[..., 7:]), dim=-1)
    assert moved.shape == sample.shape
    return moved.to(sample.dtype)


def _geometry(sample):
    return sample["vertices"] if isinstance(sample, dict) else sample


def _with_geometry(sample, tensor):
    if isinstance(sample, dict):
        sample["vertices"] = tensor
        return sample
    return tensor


class GeometryAugmentor(SampleTransformer):
    """One random rigid transform per sample for everything geometric in it (mesh vertices, gripper history, target poses): the
    SAME object is registered for all those items and ``reset()`` draws the transform for the next sample (:76-114)."""

    def __init__(self, random_translation_range_m, random_rpy_range_deg):
        self._t_range, self._rpy_range = random_translation_range_m, random_rpy_range_deg
        self._transform = None
        self.reset()

    def reset(self, rng=None):
        if self._t_range is not None and self._rpy_range is not None:
            self._transform = random_transform_uniform(self._t_range, self._rpy_range, rng)

    def __call__(self, sample):
        return _with_geometry(sample, apply_random_transform_to_sample(_geometry(sample), *self._transform))


class GeometryNoiser(SampleTransformer):
    """Independent Gaussian pose noise: one small transform per leading index of the sample -- per vertex of a mesh, per entry of
    a pose list (:117-149).  (A [nhist, ngrippers, 8] history gets one transform per history entry, shared by its grippers: the
    reference's broadcast of [N, 4] against [N, G, 4] only lines up that way.)"""

    def __init__(self, pos_stddev_m: float, rot_stddev_deg: float):
        self._pos, self._rot = pos_stddev_m, rot_stddev_deg

    def __call__(self, sample, generator=None):
        x = _geometry(sample)
        t, q = random_transform_gaussian(self._pos, self._rot, x.shape[0], generator)
        for _ in range(x.dim() - 2):
            t, q = t.unsqueeze(1), q.unsqueeze(1)
        return _with_geometry(sample, apply_random_transform_to_sample(x, t.to(x.device), q.to(x.device)))


class VertexSampler(SampleTransformer):
    """{"vertices", "features"} -> exactly ``desired_num_vertices`` rows + "vertices_valid_mask" (:152-185)."""

    def __init__(self, desired_num_vertices, method, seed=None):
        from .vertex_sampling import VertexSamplingMethod

        assert isinstance(method, VertexSamplingMethod), "Require vertex_sampling_method when using mesh."
        if method != VertexSamplingMethod.NONE:
            assert desired_num_vertices is not None and desired_num_vertices > 0, "Require num_vertices_to_sample > 0 when using mesh."
        self.desired_num_vertices, self.method, self.seed = desired_num_vertices, method, seed

    def __call__(self, sample):
        from .vertex_sampling import sample_to_n_vertices

        sample["vertices"], sample["features"], sample["vertices_valid_mask"] = sample_to_n_vertices(
            sample["vertices"], sample["features"], self.desired_num_vertices, self.method, self.seed)
        return sample
