"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU (numpy) restatement of the iso-surface extractor of neuralrecon-w_amd/csrc/ncw_mesh.hip: marching tetrahedra
on a regular grid, used in place of `skimage.measure.marching_cubes` at utils/visualization.py:114.
PARITY UNPINNED: skimage (Lewiner) is neither under /root/reference nor installable, so the triangulation of the
reference cannot be reproduced; this oracle pins the GPU kernel to an independent implementation of the SAME
algorithm (plain Python loops over cubes and tetrahedra: small grids only) and the tests add
geometry-level properties (vertices on the surface, watertightness, enclosed volume, orientation).
"""
import numpy as np

TETS = [(0, 1, 3, 7), (0, 3, 2, 7), (0, 2, 6, 7), (0, 6, 4, 7), (0, 4, 5, 7), (0, 5, 1, 7)]  # corner bit0=x,1=y,2=z


def marching_tetrahedra(sdf, level=0.0, mask=None):
    """sdf [Dx,Dy,Dz] float32 -> (triangles as a list of 3-tuples of vertex keys (lo_id, hi_id), dict key -> xyz)."""
    sdf = np.asarray(sdf, dtype=np.float32)
    Dx, Dy, Dz = sdf.shape
    pid = lambda x, y, z: (x * Dy + y) * Dz + z  # noqa: E731
    verts, tris = {}, []
    f32 = np.float32

    def vertex(ca, cb, base):
        (ia, pa, va), (ib, pb, vb) = ca, cb
        if ia > ib:
            (ia, pa, va), (ib, pb, vb) = (ib, pb, vb), (ia, pa, va)
        t = (f32(level) - va) / (vb - va)
        p = tuple(f32(pa[d]) + t * (f32(pb[d]) - f32(pa[d])) for d in range(3))
        verts[(ia, ib)] = p
        return (ia, ib)

    for x in range(Dx - 1):
        for y in range(Dy - 1):
            for z in range(Dz - 1):
                if mask is not None and not mask[x + 1, y + 1, z + 1]:
                    continue
                corner = []
                for k in range(8):
                    p = (x + (k & 1), y + ((k >> 1) & 1), z + ((k >> 2) & 1))
                    corner.append((pid(*p), p, sdf[p]))
                for tet in TETS:
                    ins = [c for c in tet if corner[c][2] < level]
                    out = [c for c in tet if not corner[c][2] < level]
                    if len(ins) in (0, 4):
                        continue
                    off = lambda c: np.array([c & 1, (c >> 1) & 1, (c >> 2) & 1], dtype=np.float32)  # noqa: E731
                    g = sum(off(c) for c in out) / len(out) - sum(off(c) for c in ins) / len(ins)
                    C = corner
                    if len(ins) == 1:
                        cand = [[vertex(C[ins[0]], C[o], None) for o in out]]
                    elif len(ins) == 3:
                        cand = [[vertex(C[out[0]], C[i], None) for i in ins]]
                    else:
                        ac, ad = vertex(C[ins[0]], C[out[0]], None), vertex(C[ins[0]], C[out[1]], None)
                        bd, bc = vertex(C[ins[1]], C[out[1]], None), vertex(C[ins[1]], C[out[0]], None)
                        cand = [[ac, ad, bd], [ac, bd, bc]]
                    for a, b, c in cand:
                        pa, pb, pc = (np.array(verts[k], dtype=np.float32) for k in (a, b, c))
                        n = np.cross(pb - pa, pc - pa)
                        if float(np.dot(n, g)) < 0:
                            b, c = c, b
                        tris.append((a, b, c))
    return tris, verts
