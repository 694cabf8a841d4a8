"""nvblox_torch.visualization (import sites: mindmap/paper/utils/utils.py:17, paper/teaser/utils/utils.py).

``get_voxel_mesh(centers, voxel_size, colors=)`` is used to draw one cube per surface voxel
(paper/utils/utils.py:134-137).  Upstream returns an open3d TriangleMesh; open3d is a viewer dependency that is not part
of this build, so the cubes are built as plain tensors (on the device of ``centers``) and converted on request.
"""
from typing import Optional

import torch

# Corner c of the unit cube has offsets (c&1, c>>1&1, c>>2&1) - 0.5; two outward-facing triangles per face.
_CORNERS = [[(c >> a & 1) - 0.5 for a in range(3)] for c in range(8)]
_TRIANGLES = [
    [0, 2, 3], [0, 3, 1],  # -z
    [4, 5, 7], [4, 7, 6],  # +z
    [0, 1, 5], [0, 5, 4],  # -y
    [2, 6, 7], [2, 7, 3],  # +y
    [0, 4, 6], [0, 6, 2],  # -x
    [1, 3, 7], [1, 7, 5],  # +x
]


class VoxelMesh:
    """Cubes as tensors: ``vertices`` [8n,3] f32, ``triangles`` [12n,3] int32, ``vertex_colors`` [8n,3] f32 in [0,1] or None."""

    def __init__(self, vertices, triangles, vertex_colors):
        self.vertices, self.triangles, self.vertex_colors = vertices, triangles, vertex_colors

    def to_open3d(self):
        import open3d as o3d  # viewer-side dependency, deliberately not imported at module level

        mesh = o3d.geometry.TriangleMesh()
        mesh.vertices = o3d.utility.Vector3dVector(self.vertices.cpu().numpy().astype("float64"))
        mesh.triangles = o3d.utility.Vector3iVector(self.triangles.cpu().numpy())
        if self.vertex_colors is not None:
            mesh.vertex_colors = o3d.utility.Vector3dVector(self.vertex_colors.cpu().numpy().astype("float64"))
        return mesh


def get_voxel_mesh(centers: torch.Tensor, voxel_size_m: float, colors: Optional[torch.Tensor] = None) -> VoxelMesh:
    """One axis-aligned cube of edge ``voxel_size_m`` per centre.  ``colors``: [n,3], uint8 0..255 or float 0..1."""
    assert centers.ndim == 2 and centers.shape[1] == 3
    n, dev = centers.shape[0], centers.device
    corners = torch.tensor(_CORNERS, dtype=torch.float32, device=dev) * float(voxel_size_m)
    vertices = (centers.to(torch.float32)[:, None, :] + corners[None]).reshape(n * 8, 3)
    tri = torch.tensor(_TRIANGLES, dtype=torch.int32, device=dev)
    triangles = (tri[None] + 8 * torch.arange(n, dtype=torch.int32, device=dev)[:, None, None]).reshape(n * 12, 3)
    vertex_colors = None
    if colors is not None:
        assert colors.shape == (n, 3)
        c = colors.to(dev)
        c = c.to(torch.float32) / 255.0 if c.dtype == torch.uint8 else c.to(torch.float32)
        vertex_colors = c[:, None, :].expand(n, 8, 3).reshape(n * 8, 3)
    return VoxelMesh(vertices, triangles, vertex_colors)
