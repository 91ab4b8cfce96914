"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU (numpy, plain Python loops: small grids only) restatement of the iso-surface extractor that stands in for
`skimage.measure.marching_cubes(sdf, level, mask=...)` at utils/visualization.py:114: MARCHING CUBES.

What the reference's call guarantees independently of skimage's internals -- and what `edge_vertices` restates exactly --
is the VERTEX SET: one vertex on every sign-changing grid edge of an enabled cube, at the linear zero crossing.
PARITY UNPINNED for the triangulation of the ambiguous cube configurations: skimage (Lewiner's variant: interior tests, a
cell-centre vertex in a few sub-cases) is neither under /root/reference nor installable here.  `marching_cubes` below
triangulates each cube by tracing the iso-polygons over the cube's faces straight from the corner values -- no case table
-- with the same face rule as the product's generated tables (on an ambiguous face the two `inside` corners, value <
level, are kept apart); the GPU kernel is compared with it triangle by triangle, and the tests add geometry-level
properties (watertightness, orientation, enclosed volume, vertices on the surface).
"""
import numpy as np


def _pid(shape, x, y, z):
    return (x * shape[1] + y) * shape[2] + z


def edge_vertices(sdf, level=0.0, mask=None):
    """{(lo_id, hi_id): xyz float32} for every sign-changing grid edge that belongs to at least one enabled cube
    (cube (i,j,k) is enabled iff mask is None or mask[i+1,j+1,k+1]); the vertex is interpolated from the lower point id to
    the higher one in float32, like the kernel."""
    sdf = np.asarray(sdf, dtype=np.float32)
    Dx, Dy, Dz = sdf.shape
    f32 = np.float32
    out = {}
    for x in range(Dx - 1):
        for y in range(Dy - 1):
            for z in range(Dz - 1):
                if mask is not None and not mask[x + 1, y + 1, z + 1]:
                    continue
                pts = [(x + (k & 1), y + ((k >> 1) & 1), z + ((k >> 2) & 1)) for k in range(8)]
                for a in range(8):
                    for b in range(a + 1, 8):
                        if bin(a ^ b).count("1") != 1:
                            continue
                        pa, pb = pts[a], pts[b]
                        va, vb = sdf[pa], sdf[pb]
                        if (va < level) == (vb < level):
                            continue
                        ia, ib = _pid(sdf.shape, *pa), _pid(sdf.shape, *pb)
                        if ia > ib:
                            ia, ib, pa, pb, va, vb = ib, ia, pb, pa, vb, va
                        t = (f32(level) - va) / (vb - va)
                        out[(ia, ib)] = tuple(f32(pa[d]) + t * (f32(pb[d]) - f32(pa[d])) for d in range(3))
    return out


def _cube_polygons(val, level):
    """Iso-polygons of one cube from its 8 corner values (corner k at offset (k&1, k>>1&1, k>>2&1)): lists of cut edges
    (a, b) with a inside (value < level), b outside, in cyclic order, wound with the normal towards increasing values.
    Tracing rule (stated from the geometry, not from a table): on the face with outward normal n the arc between two cut
    edges runs along s x n, s = in-face direction from the arc's inside side to its outside side; on a face whose diagonal
    corners share a sign each inside corner is cut off on its own."""
    ins = [v < level for v in val]
    off = [np.array([k & 1, (k >> 1) & 1, (k >> 2) & 1], dtype=np.float64) for k in range(8)]
    nxt = {}

    def cut(a, b):  # canonical name of the cut edge between adjacent corners a, b
        return (a, b) if ins[a] else (b, a)

    def where(e):
        return (off[e[0]] + off[e[1]]) / 2

    for axis in range(3):
        u, v = [d for d in range(3) if d != axis]
        for side in (0, 1):
            n = np.zeros(3)
            n[axis] = 2 * side - 1
            ring = [side << axis | (cu << u) | (cv << v) for cu, cv in ((0, 0), (1, 0), (1, 1), (0, 1))]
            crossing = [i for i in range(4) if ins[ring[i]] != ins[ring[(i + 1) % 4]]]
            arcs = []
            if len(crossing) == 2:
                i, j = crossing
                s = np.mean([off[k] for k in ring if not ins[k]], axis=0) - np.mean([off[k] for k in ring if ins[k]], axis=0)
                arcs.append((cut(ring[i], ring[(i + 1) % 4]), cut(ring[j], ring[(j + 1) % 4]), s))
            elif len(crossing) == 4:
                c = np.mean([off[k] for k in ring], axis=0)
                for i in range(4):
                    if ins[ring[i]]:
                        arcs.append((cut(ring[i], ring[i - 1]), cut(ring[i], ring[(i + 1) % 4]), c - off[ring[i]]))
            for e, f, s in arcs:
                if float(np.dot(where(f) - where(e), np.cross(s, n))) > 0:
                    nxt[e] = f
                else:
                    nxt[f] = e
    polys, done = [], set()
    for start in sorted(nxt):
        if start in done:
            continue
        loop, cur = [], start
        while cur not in done:
            done.add(cur)
            loop.append(cur)
            cur = nxt[cur]
        k = min(range(len(loop)), key=lambda i: tuple(sorted(loop[i])))  # fan apex: the cut edge with the smallest corner pair
        polys.append(loop[k:] + loop[:k])
    return polys


def marching_cubes(sdf, level=0.0, mask=None):
    """sdf [Dx,Dy,Dz] float32 -> (triangles as a list of 3-tuples of vertex keys (lo_id, hi_id), dict key -> xyz)."""
    sdf = np.asarray(sdf, dtype=np.float32)
    Dx, Dy, Dz = sdf.shape
    verts = edge_vertices(sdf, level, mask)
    tris = []
    for x in range(Dx - 1):
        for y in range(Dy - 1):
            for z in range(Dz - 1):
                if mask is not None and not mask[x + 1, y + 1, z + 1]:
                    continue
                pts = [(x + (k & 1), y + ((k >> 1) & 1), z + ((k >> 2) & 1)) for k in range(8)]
                ids = [_pid(sdf.shape, *p) for p in pts]
                for loop in _cube_polygons([sdf[p] for p in pts], np.float32(level)):
                    keys = [(min(ids[a], ids[b]), max(ids[a], ids[b])) for a, b in loop]
                    for i in range(1, len(keys) - 1):
                        tris.append((keys[0], keys[i], keys[i + 1]))
    return tris, verts
