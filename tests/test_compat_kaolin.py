"""The kaolin-named boundary module (compat/kaolin over neuralrecon_w_amd.spc).

CPU, every box: the SPC tensor formats (octree bytes, pyramid, point hierarchy, dense conversion, bit masks) against a
brute-force construction, and round trips.  CPU, build container only (needs /root/reference): the reference's own
gen_octree_from_sfm / convert_to_dense / octree_to_spc and NeuconWSystem.surface_selection / octree_update run UNEDITED
over it in a fresh interpreter (tests/_compat_kaolin_worker.py) and reproduce the golden vectors captured from the reference.
GPU: tests/test_gpu_voxel.py (`unbatched_raytrace`)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import neuconw_oracle as O
from oracle import ref_import
from tests._util import GOLDEN, compat_kaolin

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spc():
    return compat_kaolin()


def _brute_octree(q, level):
    """Reference construction, node by node: python sets per level, children sorted by (x, y, z) bit triple."""
    cells = {tuple(int(v) for v in p) for p in q.tolist()}
    levels = [cells]
    for _ in range(level):
        cells = {(x >> 1, y >> 1, z >> 1) for x, y, z in cells}
        levels.append(cells)
    levels = levels[::-1]  # root first
    morton = lambda p, l: sum((((p[0] >> i) & 1) << (3 * i + 2)) | (((p[1] >> i) & 1) << (3 * i + 1)) | (((p[2] >> i) & 1) << (3 * i))  # noqa: E731
                              for i in range(l + 1))
    octree, hierarchy = [], []
    for l in range(level + 1):
        nodes = sorted(levels[l], key=lambda p: morton(p, l))
        hierarchy += nodes
        if l < level:
            for x, y, z in nodes:
                b = 0
                for c in range(8):
                    if (2 * x + ((c >> 2) & 1), 2 * y + ((c >> 1) & 1), 2 * z + (c & 1)) in levels[l + 1]:
                        b |= 1 << c
                octree.append(b)
    return np.array(octree, np.uint8), np.array(hierarchy, np.int16), [len(s) for s in levels]


@pytest.mark.parametrize("level,n", [(3, 40), (5, 300), (6, 1)])
def test_spc_formats_against_brute_force(level, n):
    spc, _ = _spc()
    g = torch.Generator().manual_seed(level)
    x = torch.rand(n, 3, generator=g) * 2.2 - 1.1  # some outside the cube: clamped by quantize_points
    q = spc.points.quantize_points(x, level)
    assert q.dtype == torch.int16 and int(q.min()) >= 0 and int(q.max()) <= 2 ** level - 1
    assert torch.equal(q, torch.floor(torch.clamp(2 ** level * (x + 1) / 2, 0, 2 ** level - 1)).short())
    q = torch.cat([q, q[: n // 3]])  # duplicates collapse
    octree = spc.unbatched_points_to_octree(q, level)
    bo, bh, counts = _brute_octree(q, level)
    assert octree.dtype == torch.uint8 and np.array_equal(octree.numpy(), bo)
    max_level, pyramid, exsum = spc.scan_octrees(octree, torch.tensor([len(octree)], dtype=torch.int32))
    assert max_level == level and pyramid.shape == (1, 2, level + 2) and pyramid.dtype == torch.int32
    assert pyramid[0, 0].tolist() == counts + [0]
    assert pyramid[0, 1].tolist() == [0] + np.cumsum(counts).tolist()
    assert exsum.shape[0] == len(octree) + 1 and int(exsum[-1]) == sum(counts) - 1
    pts = spc.generate_points(octree, pyramid, exsum)
    assert pts.dtype == torch.int16 and np.array_equal(pts.numpy(), bh)
    feat = torch.arange(counts[-1], dtype=torch.float32).reshape(-1, 1) + 1
    dense = spc.to_dense(pts, pyramid, feat, level)
    assert dense.shape == (1, 1, 2 ** level, 2 ** level, 2 ** level)
    leaf = pts[int(pyramid[0, 1, level]):].long()
    assert torch.equal(dense[0, 0, leaf[:, 0], leaf[:, 1], leaf[:, 2]], feat[:, 0]) and int((dense > 0).sum()) == counts[-1]
    # Morton codes <-> points
    assert torch.equal(spc.points.morton_to_points(spc.points.points_to_morton(leaf)), leaf.short())


def test_octree_data_presents_both_formats():
    """voxel.OctreeData: bit masks <-> the reference's `octree` / `spc_data` members, derived lazily in both directions."""
    from neuralrecon_w_amd import spc, voxel

    g = torch.Generator().manual_seed(3)
    level = 5
    q = torch.unique(torch.randint(0, 32, (500, 3), generator=g), dim=0)
    occ, brick = spc.occupancy_bits(q, level)
    od = voxel.OctreeData({"occ": occ, "brick": brick, "level": level, "scale": 1.0, "scene_origin": torch.zeros(3)})
    assert torch.equal(voxel.voxels_from_occupancy(od), q)  # lexicographic, like torch.unique(dim=0)
    dense = voxel.dense_from_occupancy(od)
    assert int(dense.sum()) == q.shape[0] and bool(dense[q[:, 0], q[:, 1], q[:, 2]].all())
    assert "octree" not in od
    octree, sd = od["octree"], od["spc_data"]
    assert np.array_equal(octree.numpy(), _brute_octree(q, level)[0])
    ref = {"octree": octree, "level": level, "scale": 1.0, "scene_origin": torch.zeros(3), "spc_data": sd}  # the reference's layout
    for d in (ref, {k: v for k, v in ref.items() if k != "spc_data"}):
        voxel.ensure_occupancy(d)
        assert torch.equal(d["occ"], occ) and torch.equal(d["brick"], brick)
    # brick mask: every occupied voxel's 8^3 brick is set
    Gb = 4
    b = q >> 3
    lin = (b[:, 0] * Gb + b[:, 1]) * Gb + b[:, 2]
    assert bool((((brick[lin >> 5].long() >> (lin & 31)) & 1) == 1).all())


def test_raytrace_refuses_cpu_tensors():
    from neuralrecon_w_amd.lib import NeuconwHipError

    spc, spc_render = _spc()
    octree = spc.unbatched_points_to_octree(torch.tensor([[1, 2, 3]], dtype=torch.int16), 3)
    _, pyramid, exsum = spc.scan_octrees(octree, torch.tensor([len(octree)], dtype=torch.int32))
    pts = spc.generate_points(octree, pyramid, exsum)
    with pytest.raises(NeuconwHipError):
        spc_render.unbatched_raytrace(octree, pts, pyramid[0], exsum, torch.zeros(2, 3), torch.ones(2, 3), 3, return_depth=True)


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present")
def test_reference_octree_code_runs_unedited_over_compat(tmp_path):
    out = str(tmp_path / "w.npz")
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_compat_kaolin_worker.py"), out], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    w = np.load(out)
    # 1. the coarse octree from the COLMAP scene: same cube, level and voxel set as the golden captured from the reference
    g = np.load(os.path.join(GOLDEN, "sfm_octree.npz"))
    assert int(w["sfm_level"]) == int(g["level"]) and np.array_equal(w["sfm_origin"], g["scene_origin"]) and float(w["sfm_scale"]) == float(g["scale"])
    o = O.gen_octree_from_sfm(os.path.join(GOLDEN, "sfm_scene"), int(g["min_track_length"]), float(g["voxel_size"]))
    assert np.array_equal(w["sfm_dense"] > 0, o["dense"] > 0) and 0 < (w["sfm_dense"] > 0).sum() < w["sfm_dense"].size
    lvl = int(w["sfm_level"])
    leaf = w["sfm_points"][w["sfm_pyramid"][1, lvl]:w["sfm_pyramid"][1, lvl + 1]]
    assert leaf.shape[0] == int((o["dense"] > 0).sum())
    # 2. surface_selection: bit for bit the golden produced with the reference's convert_to_dense stubbed out
    z = np.load(os.path.join(GOLDEN, "octree_refresh.npz"))
    assert np.array_equal(w["sel_pts"], z["sparse_pc_sfm"]) and float(w["sel_voxel"]) == float(z["train_voxel_size"])
    # 3. octree_update's dictionary -> the bit masks the kernels read == the product's own quantisation of the same points
    from neuralrecon_w_amd import spc, voxel

    q, level = voxel.quantise_points(torch.from_numpy(w["sel_pts"]), float(w["fine_voxel"]), w["fine_origin"], float(w["fine_scale"]))
    assert level == int(w["fine_level"])
    occ, brick = spc.occupancy_bits(q, level)
    assert np.array_equal(w["fine_occ"], occ.numpy()) and np.array_equal(w["fine_brick"], brick.numpy())
    assert int(w["fine_pyramid"][0, level]) == torch.unique(q, dim=0).shape[0]
