"""The oracle's octree-refresh restatement (oracle/neuconw_oracle.py: surface_selection, gen_octree_dense)
against the golden vector produced by the reference's own NeuconWSystem.surface_selection
(tests/golden/make_golden_octree.py).  Index/coordinate arithmetic: bit-exact."""
import os

import numpy as np
import torch

from tests._util import GOLDEN


def _golden():
    return np.load(os.path.join(GOLDEN, "octree_refresh.npz"))


def test_surface_selection_bit_exact_vs_reference():
    from oracle import neuconw_oracle as O

    g = _golden()
    dense = torch.from_numpy(np.asarray(g["dense"])).float()
    sdf = lambda x: x.reshape(-1, 3).norm(dim=-1, keepdim=True) - 0.5  # noqa: E731  (same analytic SDF as the fixture)
    pts, tvs, sdf_all, xyz = O.surface_selection(
        dense, torch.from_numpy(np.asarray(g["octree_origin"])), float(g["octree_scale"]), int(g["level"]),
        torch.from_numpy(np.asarray(g["origin"])), float(g["radius"]), int(g["train_level"]), float(g["threshold"]), sdf)
    ref = torch.from_numpy(np.asarray(g["sparse_pc_sfm"]))
    assert tvs == float(g["train_voxel_size"])
    assert pts.dtype == ref.dtype == torch.float32 and pts.shape == ref.shape
    assert torch.equal(pts, ref)  # same points, same order, same bits


def test_gen_octree_dense_properties():
    from oracle import neuconw_oracle as O

    g = _golden()
    ref = torch.from_numpy(np.asarray(g["sparse_pc_sfm"]))
    scale, origin = float(g["octree_scale"]), np.asarray(g["octree_origin"]).astype(np.float64)
    dense, level = O.gen_octree_dense(ref, float(g["train_voxel_size"]), origin, scale)
    assert level == int(g["train_level"])
    # every selected corner point lies in (or, by float32 rounding of the corner, one voxel below) its sub-voxel;
    # points on the cube's lower faces are dropped by the strict (-1,1) filter
    pn = (ref.double() - torch.from_numpy(origin)) / scale
    inside = ((pn > -1) & (pn < 1)).all(-1)
    assert 0 < int(dense.sum()) <= int(inside.sum())
    res = 2 ** level
    q = torch.round(res * (pn[inside] + 1) / 2).long().clamp(0, res - 1)
    near = torch.zeros_like(dense)
    for dx in (0, -1):
        for dy in (0, -1):
            for dz in (0, -1):
                qq = (q + torch.tensor([dx, dy, dz])).clamp(0, res - 1)
                near[qq[:, 0], qq[:, 1], qq[:, 2]] = True
    assert bool((dense & ~near).sum() == 0)
    # empty input -> empty grid
    d0, _ = O.gen_octree_dense(torch.zeros(0, 3), float(g["train_voxel_size"]), origin, scale)
    assert int(d0.sum()) == 0
