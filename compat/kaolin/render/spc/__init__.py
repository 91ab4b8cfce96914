"""kaolin.render.spc: `unbatched_raytrace` (generate_voxel.py:358-368) on the HIP ray / voxel kernel."""
from neuralrecon_w_amd.spc import unbatched_raytrace  # noqa: F401
