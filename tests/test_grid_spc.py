"""CPU: the sparse evaluation grid of the mesh extraction (mesh.gen_grid_spc) against the golden produced by RUNNING the
reference's own `gen_grid_spc` (tools/extract_mesh.py:60-102; tests/golden/make_golden_gridspc.py), bit for bit."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_gen_grid_spc_bit_exact_on_cpu():
    from neuralrecon_w_amd import mesh, voxel

    z = np.load(os.path.join(GOLDEN, "grid_spc.npz"))
    zo = np.load(os.path.join(GOLDEN, "sfm_octree.npz"))
    dense = torch.from_numpy(z["dense"]).bool()
    bits = dense.reshape(-1, 32).to(torch.int64)
    occ = (bits << torch.arange(32).view(1, 32)).sum(1)
    occ = torch.where(occ >= 2 ** 31, occ - 2 ** 32, occ).to(torch.int32)  # the int32 bit mask ncw_voxel_build writes
    od = {"occ": occ, "level": int(z["level"]), "scale": float(zo["scale"]), "scene_origin": torch.from_numpy(zo["scene_origin"])}
    assert torch.equal(voxel.dense_from_occupancy(od), dense)
    sd = mesh.gen_grid_spc(od, int(z["eval_level"]))
    assert sd["dim"] == int(z["dim"]) and np.float64(sd["voxel_size"]) == z["voxel_size"][()]
    assert np.array_equal(np.asarray(sd["vol_origin"]), z["vol_origin"])
    assert str(sd["sparse_vol"].dtype) == str(z["sparse_vol_dtype"])
    assert np.array_equal(sd["sparse_vol"].numpy(), z["sparse_vol"])
    # the volume / mask construction of utils/visualization.py:91-110 runs on the CPU as well
    vol, mask, ind = mesh.sparse_volume(sd, torch.arange(sd["sparse_vol"].shape[0]).float())
    assert vol.shape == (sd["dim"],) * 3 and int(mask.sum()) > 0
    assert torch.equal(vol[ind[:, 0], ind[:, 1], ind[:, 2]], torch.arange(ind.shape[0]).float())
