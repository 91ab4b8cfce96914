"""Marching-cubes case tables, GENERATED (not transcribed): for each of the 256 sign configurations of a cube the
triangles of the iso-surface as triples of cut-edge indices.

utils/visualization.py:114 calls `skimage.measure.marching_cubes(sdf, level=0, mask=...)`: one vertex per sign-changing grid
edge (linear interpolation), triangles inside each cube.  The VERTEX SET is algorithm-independent; the triangulation of the
ambiguous configurations is not (skimage's Lewiner variant resolves them with interior tests and, in a few sub-cases, adds
a cell-centre vertex; it is neither in the reference tree nor installable, so its triangulation cannot be pinned).  The
tables here resolve every ambiguous FACE the same way from the face's own corner signs (the two `inside` corners -- value <
level -- are kept apart), so neighbouring cubes agree on every shared face and the mesh is watertight by construction.

Conventions: corner k has offset (k & 1, (k >> 1) & 1, (k >> 2) & 1) = (+x, +y, +z); configuration bit k is set iff corner k
is inside (value < level); edge e joins EDGES[e] = (a, b), a < b (edges sorted by that pair); triangles are wound so that the
normal points towards increasing values (outwards for a signed distance); a polygon of more than three cut edges is a fan
around its cut edge with the smallest corner pair.  (Like every table-driven marching cubes, a fan diagonal may lie inside
an ambiguous face when both arcs of that face belong to one polygon: rare doubled edges on noise, none on smooth fields.)
"""
import numpy as np

CORNERS = [(k & 1, (k >> 1) & 1, (k >> 2) & 1) for k in range(8)]
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]  # 12 edges
_EDGE_ID = {e: i for i, e in enumerate(EDGES)}


def _faces():
    """[(ring of 4 corners in cyclic order, outward normal)] of the 6 cube faces."""
    out = []
    for d in range(3):
        u, v = [x for x in range(3) if x != d]
        for s in (0, 1):
            ring = []
            for (cu, cv) in ((0, 0), (1, 0), (1, 1), (0, 1)):  # corners in cyclic order around the face
                c = [0, 0, 0]
                c[d], c[u], c[v] = s, cu, cv
                ring.append(c[0] | (c[1] << 1) | (c[2] << 2))
            n = np.zeros(3)
            n[d] = 1.0 if s else -1.0
            out.append((ring, n))
    return out


FACES = _faces()


def _edge(a, b):
    return _EDGE_ID[(min(a, b), max(a, b))]


def _pos(k):
    return np.array(CORNERS[k], dtype=float)


def _mid(e):
    return (_pos(EDGES[e][0]) + _pos(EDGES[e][1])) / 2


def case_triangles(cfg):
    """-> list of (e0, e1, e2) for configuration cfg (bit k: corner k inside).

    The iso-polygons are traced over the cube's faces as DIRECTED arcs: on a face with outward normal n, the segment
    between two cut edges separates an inside side from an outside side (s = in-face direction inside -> outside); with
    the surface normal towards the outside (increasing values) and the polygon's interior inside the cube, the boundary
    runs along s x n.  Every cut edge then has exactly one outgoing and one incoming arc, and the loops come out
    consistently oriented across neighbouring cubes (a shared face sees the opposite n, hence the opposite direction)."""
    inside = [(cfg >> k) & 1 for k in range(8)]
    succ = {}

    def arc(e, f, s_dir, n):
        if float(np.dot(_mid(f) - _mid(e), np.cross(s_dir, n))) > 0:
            assert e not in succ
            succ[e] = f
        else:
            assert f not in succ
            succ[f] = e

    for ring, n in FACES:
        cut = [_edge(ring[i], ring[(i + 1) % 4]) for i in range(4) if inside[ring[i]] != inside[ring[(i + 1) % 4]]]
        if len(cut) == 2:
            ins = [k for k in ring if inside[k]]
            out = [k for k in ring if not inside[k]]
            s_dir = sum(_pos(k) for k in out) / len(out) - sum(_pos(k) for k in ins) / len(ins)
            arc(cut[0], cut[1], s_dir, n)
        elif len(cut) == 4:  # ambiguous face: keep the two inside corners apart (cut each one off on its own)
            centre = sum(_pos(k) for k in ring) / 4
            for i in range(4):
                if inside[ring[i]]:
                    arc(_edge(ring[i - 1], ring[i]), _edge(ring[i], ring[(i + 1) % 4]), centre - _pos(ring[i]), n)
    tris, seen = [], set()
    for start in sorted(succ):
        if start in seen:
            continue
        loop, cur = [], start
        while cur not in seen:
            seen.add(cur)
            loop.append(cur)
            cur = succ[cur]
        assert cur == start and len(loop) >= 3
        k = loop.index(min(loop))  # convention: the fan's apex is the loop's lowest edge index (= smallest corner pair)
        loop = loop[k:] + loop[:k]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
    assert len(seen) == sum(1 for a, b in EDGES if inside[a] != inside[b])
    return tris


def tables():
    """-> (tri [256, 16] int8: edge indices, -1 padded; ntri [256] int32; edges [12, 2] int32 corner pairs)."""
    tri = -np.ones((256, 16), dtype=np.int8)
    ntri = np.zeros(256, dtype=np.int32)
    for cfg in range(256):
        t = case_triangles(cfg)
        assert len(t) <= 5, (cfg, len(t))
        ntri[cfg] = len(t)
        for i, (a, b, c) in enumerate(t):
            tri[cfg, 3 * i:3 * i + 3] = (a, b, c)
    return tri, ntri, np.array(EDGES, dtype=np.int32)
