"""CPU: geometry-level properties of the marching-tetrahedra restatement (oracle/mesh_oracle.py) that the GPU
extractor is pinned to.  (skimage is absent: the reference's own triangulation cannot be compared.)"""
import numpy as np


def _sphere(D, r=0.6, c=(0.03, -0.02, 0.05)):
    ax = np.linspace(-1, 1, D, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r).astype(np.float32), ax


def test_sphere_is_watertight_outward_and_has_the_right_volume():
    from oracle.mesh_oracle import marching_tetrahedra

    D = 14
    sdf, ax = _sphere(D)
    tris, verts = marching_tetrahedra(sdf)
    assert len(tris) > 200
    edges = {}
    for a, b, c in tris:
        for e in ((a, b), (b, c), (c, a)):
            edges[e] = edges.get(e, 0) + 1
    # closed + consistently oriented: every directed edge once, and its reverse once
    assert all(n == 1 for n in edges.values())
    assert all((e[1], e[0]) in edges for e in edges)
    h = ax[1] - ax[0]
    P = {k: np.array(v, dtype=np.float64) * h - 1.0 for k, v in verts.items()}
    vol = sum(np.dot(P[a], np.cross(P[b], P[c])) for a, b, c in tris) / 6.0
    assert abs(vol - 4.0 / 3.0 * np.pi * 0.6 ** 3) / (4.0 / 3.0 * np.pi * 0.6 ** 3) < 0.05 and vol > 0  # outward
    r = np.array([np.linalg.norm(p - np.array([0.03, -0.02, 0.05])) for p in P.values()])
    assert np.abs(r - 0.6).max() < 0.02  # linear zero crossings of a distance field lie on the sphere


def test_mask_and_empty():
    from oracle.mesh_oracle import marching_tetrahedra

    sdf, _ = _sphere(8)
    assert marching_tetrahedra(np.ones((4, 4, 4), np.float32))[0] == []
    mask = np.zeros(sdf.shape, bool)
    assert marching_tetrahedra(sdf, mask=mask)[0] == []
    mask[1:, 1:, 1:] = True  # every cube enabled
    assert len(marching_tetrahedra(sdf, mask=mask)[0]) == len(marching_tetrahedra(sdf)[0])
