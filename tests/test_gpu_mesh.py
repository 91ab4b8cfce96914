"""GPU: iso-surface extraction -- ncw_mt_count / ncw_mt_emit through mesh.isosurface against the
CPU restatement (same triangles, bit-identical vertices), geometry properties at a larger size, the colour pass
and the PLY writer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _field(D, seed):
    ax = np.linspace(-1, 1, D, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    g = np.random.default_rng(seed)
    f = np.sqrt((X - 0.05) ** 2 + (Y + 0.1) ** 2 + Z ** 2) - 0.55 + 0.08 * np.sin(5 * X) * np.cos(4 * Y)
    return (f + 0.01 * g.standard_normal(f.shape)).astype(np.float32)


@pytest.mark.parametrize("D,use_mask", [(9, False), (12, True)])
def test_matches_restatement(D, use_mask):
    from neuralrecon_w_amd import mesh
    from oracle.mesh_oracle import marching_tetrahedra

    f = _field(D, D)
    f[3, 3, 3] = 0.0  # a value exactly on the level
    mask = None
    if use_mask:
        mask = np.random.default_rng(1).random(f.shape) < 0.7
    tris, verts = marching_tetrahedra(f, 0.0, mask)
    v, fc = mesh.isosurface(torch.from_numpy(f).cuda(), 0.0, None if mask is None else torch.from_numpy(mask).cuda())
    v, fc = v.cpu().numpy(), fc.cpu().numpy()
    # same triangles (as unordered vertex triples with bit-identical coordinates); triangles with coincident
    # corners (a grid value exactly on the level collapses edges) are dropped on both sides, and the winding is
    # checked by the orientation / volume properties below, not per sliver triangle
    def canon(tri):
        return tuple(sorted(tri)) if len(set(tri)) == 3 else None

    ref, got = {}, {}
    for t in tris:
        k = canon(tuple(tuple(float(c) for c in verts[q]) for q in t))
        if k is not None:
            ref[k] = ref.get(k, 0) + 1
    for a, b, c in fc:
        k = canon(tuple(tuple(float(x) for x in v[j]) for j in (a, b, c)))
        if k is not None:
            got[k] = got.get(k, 0) + 1
    assert len(got) > 50 and got == ref


def test_sphere_properties_and_ply(tmp_path):
    from neuralrecon_w_amd import mesh

    D, r = 96, 0.62
    ax = torch.linspace(-1, 1, D, device="cuda")
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    sdf = torch.sqrt(X ** 2 + Y ** 2 + Z ** 2) - r
    v, f = mesh.isosurface(sdf)
    assert int(f.min()) == 0 and int(f.max()) == v.shape[0] - 1
    P = (v.double() * (2.0 / (D - 1)) - 1.0)
    assert float((P.norm(dim=-1) - r).abs().max()) < 2e-3
    a, b, c = P[f[:, 0]], P[f[:, 1]], P[f[:, 2]]
    vol = float((a * torch.cross(b, c, dim=-1)).sum() / 6.0)
    assert abs(vol - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 2e-3  # positive: outward winding
    e = torch.cat([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = e[:, 0] * v.shape[0] + e[:, 1]
    assert key.unique().numel() == key.numel()  # every directed edge once ...
    rev = e[:, 1] * v.shape[0] + e[:, 0]
    assert torch.equal(key.sort().values, rev.sort().values)  # ... and its reverse once: closed, oriented
    assert mesh.isosurface(torch.ones(5, 5, 5, device="cuda"))[1].shape[0] == 0  # empty
    path = os.path.join(tmp_path, "m.ply")
    cols = torch.randint(0, 255, (v.shape[0], 3), dtype=torch.uint8)
    mesh.write_ply(path, v, f, cols)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex %d" % v.shape[0] in head and b"element face %d" % f.shape[0] in head
    assert len(body) == v.shape[0] * 15 + f.shape[0] * 13
    first = np.frombuffer(body[:12], "<f4")
    assert np.allclose(first, v[0].cpu().numpy())


def test_extract_mesh_with_colors():
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import mesh
    from tests._build import build_system

    emb, neuconw, nerf, rdr = build_system(seed=2, prec=nw.PREC_F32)  # geometric init: a sphere of radius ~0.5
    out = mesh.extract_mesh(rdr, dim=48, scene_radius=2.0, scene_origin=[1.0, 2.0, 3.0], with_color=True,
                            embedding_a=emb.weight[3].detach())
    V, F = out["vertices"], out["faces"]
    assert V.shape[0] > 500 and F.shape[0] > 1000 and out["colors"].shape == (V.shape[0], 3)
    # the iso-surface is the network's zero level set: sdf at the (training-space) vertices ~ 0
    s = rdr.sdf(out["vertices_training"]).reshape(-1)
    assert float(s.abs().max()) < 1.5e-2  # a third of a voxel (2/47): linear interpolation of a curved field
    assert torch.allclose(V, out["vertices_training"] * 2.0 + torch.tensor([1.0, 2.0, 3.0]).cuda())
