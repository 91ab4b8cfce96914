"""Golden vector for the octree refresh, produced by RUNNING the reference's own
`NeuconWSystem.surface_selection` (lightning_modules/neuconw_system.py:186-264) on CPU in the build container.

kaolin and CUDA are absent, so three seams are stubbed -- and only those: `convert_to_dense` returns the dense
occupancy grid we feed in (kaolin SPC -> dense conversion), `Tensor.cuda()` is the identity, and `renderer.sdf`
is an analytic float32 SDF (distance to a sphere of radius 0.5) so that the fixture pins the reference's INDEX
and COORDINATE arithmetic (nonzero order, up-sampling kernel, float32 `ind * voxel + origin`, thresholding)
independently of any network.   Run:  python tests/golden/make_golden_octree.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402


def main():
    ref_import.load()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    try:
        import importlib

        sysmod = importlib.import_module("lightning_modules.neuconw_system")
    finally:
        sys.path.remove(ref_import.REFERENCE_ROOT)
    level, train_level, threshold = 4, 6, 0.03
    G = 1 << level
    g = torch.Generator().manual_seed(7)
    c = (torch.stack(torch.meshgrid(*[torch.arange(G)] * 3, indexing="ij"), -1).float() + 0.5) * (2.0 / G) - 1.0
    dense = (((c.norm(dim=-1) - 0.5).abs() < 0.12) | (torch.rand(G, G, G, generator=g) < 0.01)).float()
    octree_origin = torch.tensor([0.02, 0.01, -0.03])
    octree_scale = 1.25
    origin, radius = torch.tensor([0.05, -0.02, 0.01]), 1.1
    sdf = lambda pts: pts.reshape(-1, 3).norm(dim=-1, keepdim=True) - 0.5  # noqa: E731
    renderer = types.SimpleNamespace(
        origin=origin, radius=radius, sdf=sdf,
        octree_data={"octree": None, "scene_origin": octree_origin, "scale": octree_scale, "level": level})
    fake_self = types.SimpleNamespace(renderer=renderer, hparams=types.SimpleNamespace(num_gpus=1))
    sysmod.convert_to_dense = lambda octree, lvl: dense
    sysmod.get_world_size = lambda: 1
    sysmod.get_rank = lambda: 0
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        pts, tvs = sysmod.NeuconWSystem.surface_selection(fake_self, train_level, threshold, device="cpu", chunk=4096)
    finally:
        torch.Tensor.cuda = orig_cuda
    out = os.path.join(HERE, "octree_refresh.npz")
    np.savez_compressed(out, dense=dense.numpy().astype(np.uint8), octree_origin=octree_origin.numpy(),
                        octree_scale=np.float64(octree_scale), level=level, train_level=train_level,
                        threshold=np.float64(threshold), origin=origin.numpy(), radius=np.float64(radius),
                        sparse_pc_sfm=np.asarray(pts), train_voxel_size=np.float64(tvs))
    print("wrote", out, np.asarray(pts).shape, np.asarray(pts).dtype, tvs, os.path.getsize(out))


if __name__ == "__main__":
    main()
