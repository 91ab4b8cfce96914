"""`kaolin`-named boundary module for ROCm machines: exactly the six kaolin calls the reference makes
(tools/prepare_data/generate_voxel.py:149-150, 175-176, 185, 358-368), served by neuralrecon_w_amd.spc (torch ops on the
device + the HIP ray / voxel kernel `ncw_ray_voxel_trace`).  Put this directory's parent (`compat/`) on PYTHONPATH where the
CUDA-only kaolin package cannot be installed: the reference's own gen_octree / convert_to_dense / octree_to_spc /
get_near_far and NeuconWSystem.surface_selection / octree_update (lightning_modules/neuconw_system.py:186-312) then run
unedited.  Not a re-implementation of kaolin: anything else raises AttributeError.  Tensor formats: neuralrecon_w_amd/spc.py."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from . import ops, render  # noqa: E402,F401

__version__ = "0.0+neuralrecon_w_amd.compat"
