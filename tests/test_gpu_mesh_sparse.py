"""GPU: the sparse mode of the mesh extraction (tools/extract_mesh.py:60-102 `gen_grid_spc`,
utils/visualization.py:47-61,91-110): the evaluation grid against the golden produced by the reference's own function
(tests/golden/make_golden_gridspc.py), the volume / mask construction against their definition, the sparse mesh against
marching cubes over the fully evaluated lattice under the same mask (bit-identical), and scripts/extract_mesh.py end to end."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


def _golden_octree():
    from neuralrecon_w_amd import voxel

    z = np.load(os.path.join(GOLDEN, "grid_spc.npz"))
    zo = np.load(os.path.join(GOLDEN, "sfm_octree.npz"))
    dense = torch.from_numpy(z["dense"]).bool().cuda()
    od = voxel.occupancy_from_dense(dense, zo["scene_origin"], float(zo["scale"]))
    od["scene_origin"] = torch.from_numpy(zo["scene_origin"]).cuda()  # float64, as voxel.octree_from_sfm stores it
    assert torch.equal(voxel.dense_from_occupancy(od), dense)
    return od, z, zo


def test_gen_grid_spc_matches_reference():
    from neuralrecon_w_amd import mesh

    od, z, zo = _golden_octree()
    sd = mesh.gen_grid_spc(od, int(z["eval_level"]))
    assert sd["dim"] == int(z["dim"])
    assert np.float64(sd["voxel_size"]) == z["voxel_size"][()]
    assert np.array_equal(np.asarray(sd["vol_origin"]), z["vol_origin"])
    got = sd["sparse_vol"].cpu().numpy()
    assert str(sd["sparse_vol"].dtype) == str(z["sparse_vol_dtype"]) and got.shape == z["sparse_vol"].shape
    assert np.array_equal(got, z["sparse_vol"])  # bit-identical, order included


def test_sparse_volume_and_mask_definition():
    from neuralrecon_w_amd import mesh

    od, z, zo = _golden_octree()
    sd = mesh.gen_grid_spc(od, int(z["eval_level"]) - 1)
    K, dim = sd["sparse_vol"].shape[0], sd["dim"]
    vals = torch.randn(K, device="cuda")
    vol, mask, ind = mesh.sparse_volume(sd, vals)
    ind = ind.cpu().numpy()
    # the index of every sparse point is its up-sampled voxel index; values land there, 1 elsewhere
    up = 2 ** (int(z["eval_level"]) - 1 - int(z["level"]))
    occ = np.kron(z["dense"], np.ones((up, up, up), dtype=np.uint8)).astype(bool)
    assert occ.sum() == K and occ[ind[:, 0], ind[:, 1], ind[:, 2]].all()
    v = vol.cpu().numpy()
    assert np.array_equal(v[ind[:, 0], ind[:, 1], ind[:, 2]], vals.cpu().numpy()) and (v[~occ] == 1.0).all()
    # mask[p] <=> p and its 7 lower neighbours (periodic, like torch.roll) were all evaluated
    want = np.ones_like(occ)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                want &= np.roll(occ, (dx, dy, dz), axis=(0, 1, 2))
    assert np.array_equal(mask.cpu().numpy(), want) and 0 < want.sum() < occ.sum()


def test_sparse_mesh_equals_masked_dense_lattice():
    """The sparse path evaluates the SDF only inside the occupied voxels; its mesh must be exactly the marching-cubes surface
    of the FULLY evaluated lattice restricted to the cubes whose 8 corners are sparse points."""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import mesh, voxel
    from tests._build import build_system

    emb, neuconw, nerf, rdr = build_system(seed=2, prec=nw.PREC_F32)  # geometric init: a sphere of radius ~0.5
    G = 16
    ax = (torch.arange(G, device="cuda").float() + 0.5) * (2.0 / G) - 1.0
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    r = torch.sqrt(X * X + Y * Y + Z * Z)
    shell = (r > 0.3) & (r < 0.72)  # coarse voxels around most of the surface (the mesh is open where it leaves them)
    scene_origin, scene_radius = [0.3, -0.2, 0.1], 1.7
    od = voxel.occupancy_from_dense(shell, scene_origin, scene_radius)  # octree cube = the training cube in SfM units
    od["scene_origin"] = torch.tensor(scene_origin, dtype=torch.float64, device="cuda")
    sd = mesh.gen_grid_spc(od, eval_level=6)  # 64^3 lattice, 4^3 sub-voxels per coarse voxel
    out = mesh.extract_mesh(rdr, 0, scene_radius, scene_origin, sparse_data=sd, with_color=True, embedding_a=emb.weight[3].detach())
    V, F = out["vertices_training"], out["faces"]
    assert V.shape[0] > 2000 and F.shape[0] > 4000 and out["colors"].shape == (V.shape[0], 3)
    s = rdr.sdf(V).reshape(-1)
    assert float(s.abs().max()) < 1e-2
    assert torch.allclose(out["vertices"], V * scene_radius + torch.tensor(scene_origin).cuda(), atol=1e-6)
    # the same lattice, every point evaluated
    dim = sd["dim"]
    idx = torch.stack(torch.meshgrid(*[torch.arange(dim, device="cuda")] * 3, indexing="ij"), -1).reshape(-1, 3)
    xyz_sfm = ((idx * float(sd["voxel_size"])).double() + torch.from_numpy(sd["vol_origin"]).cuda()).float()
    xyz = (xyz_sfm - torch.tensor(scene_origin).cuda()) / scene_radius
    full = rdr.sdf(xyz.reshape(-1, 1, 3)).reshape(dim, dim, dim)
    _, mask, _ = mesh.sparse_volume(dict(sd, sparse_vol=sd["sparse_vol"].float()), torch.zeros(sd["sparse_vol"].shape[0], device="cuda"))
    v2, f2 = mesh.isosurface(full, 0.0, mask)
    vol_origin = (torch.from_numpy(sd["vol_origin"]).float().cuda() - torch.tensor(scene_origin).cuda()) / scene_radius
    v2t = v2 * (float(sd["voxel_size"]) / scene_radius) + vol_origin
    assert v2t.shape == V.shape and torch.equal(v2t, V) and torch.equal(f2, F)


def test_extract_mesh_script(tmp_path):
    """scripts/extract_mesh.py on the golden COLMAP scene: checkpoint -> sparse (octree) and dense meshes -> PLY files."""
    import yaml

    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import config as C
    from neuralrecon_w_amd import trainer

    scene = os.path.join(GOLDEN, "sfm_scene")
    cfg_path = str(tmp_path / "exp.yaml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump({"DATASET": {"ROOT_DIR": scene},
                        "NEUCONW": {"SDF_CONFIG": {"d_hidden": 64, "n_layers": 4, "skip_in": "(2,)", "d_out": 65},
                                    "COLOR_CONFIG": {"d_hidden": 64, "d_feature": 64, "n_layers": 2, "head_channels": 32},
                                    "N_VOCAB": 1200, "N_SAMPLES": 8, "N_IMPORTANCE": 16}}, f)
    cfg = C.load_config(cfg_path)
    torch.manual_seed(5)
    emb, neuconw, nerf, rdr, sc = C.build_system(cfg, torch.device("cuda"))
    ckpt = str(tmp_path / "exp" / "last.ckpt")
    os.makedirs(os.path.dirname(ckpt))
    trainer.save_checkpoint(ckpt, emb, neuconw, nerf)
    env = dict(os.environ, PYTHONPATH=ROOT)
    base = [sys.executable, os.path.join(ROOT, "scripts", "extract_mesh.py"), "--cfg_path", cfg_path, "--ckpt_path", ckpt,
            "--out_dir", str(tmp_path / "mesh")]
    r = subprocess.run(base + ["--mesh_size", "64", "--vertex_color"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    p = tmp_path / "mesh" / "extracted_mesh_res_64_radius_1.0_colored.ply"
    head = open(p, "rb").read(400)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex ") and b"property uchar red" in head
    nv = int(head.split(b"element vertex ")[1].split(b"\n")[0])
    assert nv > 1000
    zo = np.load(os.path.join(GOLDEN, "sfm_octree.npz"))
    lvl = int(zo["level"]) + 2
    r = subprocess.run(base + ["--eval_level", str(lvl)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "sparse points" in r.stdout and os.path.exists(tmp_path / "mesh" / ("extracted_mesh_level_%d.ply" % lvl))
