"""kaolin.ops.spc: the five structure calls of tools/prepare_data/generate_voxel.py (:149-150, 175-176, 185)."""
from neuralrecon_w_amd.spc import generate_points, scan_octrees, to_dense, unbatched_points_to_octree  # noqa: F401

from . import points  # noqa: F401
