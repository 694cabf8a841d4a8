"""nvblox_mindmap_amd -- MI355X-native spatial-memory hot path of NVlabs/nvblox_mindmap.

RGB-D + deep-feature frame -> TSDF / feature voxel-block map -> (vertices, vertex_features) tensors
for diffuser_actor, plus the depth back-projection and mask algebra feeding it.  The integrator is
hand-written HIP for gfx950 behind the C ABI of ``include/mmfusion.h``; this package is the Python
host side that mirrors the reference's ``nvblox_torch`` / ``mindmap.mapping`` /
``mindmap.image_processing`` call surface.  There is no CPU fallback.
"""
import sys

__version__ = "0.1.0"


def install_as_nvblox_torch() -> None:
    """Make ``import nvblox_torch`` (as written in the reference) resolve to this package's drop-in."""
    from . import nvblox_torch as nt

    sys.modules.setdefault("nvblox_torch", nt)
    for sub in ("mapper", "mapper_params", "projective_integrator_types", "constants", "timer", "indexing", "layer"):
        sys.modules.setdefault(f"nvblox_torch.{sub}", getattr(nt, sub))
