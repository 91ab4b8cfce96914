"""kaolin.ops.spc.points: `quantize_points` (generate_voxel.py:149)."""
from neuralrecon_w_amd.spc import morton_to_points, points_to_morton, quantize_points  # noqa: F401
