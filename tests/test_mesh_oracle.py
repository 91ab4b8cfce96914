"""CPU: geometry-level properties of the marching-cubes restatement (oracle/mesh_oracle.py) that the GPU extractor is
pinned to, and the product's GENERATED case tables (neuralrecon_w_amd/mc_tables.py) against that restatement on all 256
configurations.  (skimage is absent: the reference's own triangulation of the ambiguous cases cannot be compared; its
vertex set -- one linear zero crossing per sign-changing grid edge -- is what `edge_vertices` restates.)"""
import numpy as np


def _sphere(D, r=0.6, c=(0.03, -0.02, 0.05)):
    ax = np.linspace(-1, 1, D, dtype=np.float32)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return (np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r).astype(np.float32), ax


def test_sphere_is_watertight_outward_and_has_the_right_volume():
    from oracle.mesh_oracle import marching_cubes

    D = 14
    sdf, ax = _sphere(D)
    tris, verts = marching_cubes(sdf)
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
    from oracle.mesh_oracle import marching_cubes

    sdf, _ = _sphere(8)
    assert marching_cubes(np.ones((4, 4, 4), np.float32))[0] == []
    mask = np.zeros(sdf.shape, bool)
    assert marching_cubes(sdf, mask=mask)[0] == []
    mask[1:, 1:, 1:] = True  # every cube enabled
    assert len(marching_cubes(sdf, mask=mask)[0]) == len(marching_cubes(sdf)[0])


def test_generated_case_tables_match_the_restatement_on_all_256_configurations():
    """Every configuration: the table's triangles == the polygons traced from corner values (as sets of oriented
    triangles over cut edges); 820 triangles in total (the classic table's count), at most 5 per cube; every cut edge of a
    configuration is used, every directed polygon edge inside a cube appears once."""
    from neuralrecon_w_amd import mc_tables
    from oracle.mesh_oracle import _cube_polygons

    tri, ntri, edges = mc_tables.tables()
    assert tri.shape == (256, 16) and int(ntri.sum()) == 820 and int(ntri.max()) == 5 and ntri[0] == ntri[255] == 0
    E = [tuple(e) for e in edges.tolist()]
    assert len(E) == 12 and all(bin(a ^ b).count("1") == 1 and a < b for a, b in E)

    def rot(t):  # canonical rotation of an oriented triangle
        i = t.index(min(t))
        return t[i:] + t[:i]

    for cfg in range(256):
        val = [-1.0 if (cfg >> k) & 1 else 1.0 for k in range(8)]
        want = set()
        for loop in _cube_polygons(val, 0.0):
            ks = [tuple(sorted(e)) for e in loop]
            for i in range(1, len(ks) - 1):
                want.add(rot((ks[0], ks[i], ks[i + 1])))
        got = {rot(tuple(E[tri[cfg, 3 * t + i]] for i in range(3))) for t in range(int(ntri[cfg]))}
        assert sorted(got) == sorted(want), cfg  # same polygons, same fan apex (the cut edge with the smallest corner pair)
        cut = {e for e in E if ((cfg >> e[0]) & 1) != ((cfg >> e[1]) & 1)}
        assert {e for t in got for e in t} == cut, cfg
        # the same surface; fans may start at a different polygon vertex, so compare as polygon edge sets instead
        def boundary(ts):
            d = {}
            for a, b, c in ts:
                for e in ((a, b), (b, c), (c, a)):
                    d[e] = d.get(e, 0) + 1
            return {e for e in d if (e[1], e[0]) not in d}, d
        bw, dw = boundary(want)
        bg, dg = boundary(got)
        assert bw == bg and all(n == 1 for n in dg.values()), cfg
        assert all(tri[cfg, 3 * int(ntri[cfg]):] == -1)


def test_vertex_set_is_every_sign_changing_edge():
    from oracle.mesh_oracle import edge_vertices, marching_cubes

    rng = np.random.default_rng(4)
    sdf = rng.standard_normal((6, 7, 5)).astype(np.float32)  # noise: every ambiguous configuration occurs
    tris, verts = marching_cubes(sdf)
    used = {k for t in tris for k in t}
    assert used == set(edge_vertices(sdf)) == set(verts)
    D = sdf.shape
    n = 0
    for x in range(D[0]):
        for y in range(D[1]):
            for z in range(D[2]):
                for d in ((1, 0, 0), (0, 1, 0), (0, 0, 1)):
                    q = (x + d[0], y + d[1], z + d[2])
                    if q[0] < D[0] and q[1] < D[1] and q[2] < D[2] and (sdf[x, y, z] < 0) != (sdf[q] < 0):
                        n += 1
    assert n == len(verts)
