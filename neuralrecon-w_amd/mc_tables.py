"""Marching-cubes case tables, GENERATED (not transcribed): for each of the 256 sign configurations of a cube the
triangles of the iso-surface as triples of cut-edge indices.

utils/visualization.py:114 calls `skimage.measure.marching_cubes(sdf, level=0, mask=...)`: one vertex per sign-changing grid
edge (linear interpolation), triangles inside each cube.  The VERTEX SET is algorithm-independent; the triangulation of the
ambiguous configurations is not (skimage's Lewiner variant resolves them with interior tests and, in a few sub-cases, adds
a cell-centre vertex; it is neither in the reference tree nor installable, so its triangulation cannot be pinned).  The
tables here resolve every ambiguous FACE the same way from the face's own corner signs (the two `inside` corners -- value <
level -- are kept apart), so neighbouring cubes agree on every shared face and the mesh is watertight by construction.

Conventions: corner k has offset (k & 1, (k >> 1) & 1, (k >> 2) & 1) = (+x, +y, +z); configuration bit k is set iff corner k
is inside (value < level); edge e joins EDGES[e] = (a, b), a < b; triangles are wound so that the normal points towards
increasing values (outwards for a signed distance).
"""
import numpy as np

CORNERS = [(k & 1, (k >> 1) & 1, (k >> 2) & 1) for k in range(8)]
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]  # 12 edges
_EDGE_ID = {e: i for i, e in enumerate(EDGES)}


def _faces():
    out = []
    for d in range(3):
        u, v = [x for x in range(3) if x != d]
        for s in (0, 1):
            ring = []
            for (cu, cv) in ((0, 0), (1, 0), (1, 1), (0, 1)):  # corners in cyclic order around the face
                c = [0, 0, 0]
                c[d], c[u], c[v] = s, cu, cv
                ring.append(c[0] | (c[1] << 1) | (c[2] << 2))
            out.append(ring)
    return out


FACES = _faces()


def _edge(a, b):
    return _EDGE_ID[(min(a, b), max(a, b))]


def case_triangles(cfg):
    """-> list of (e0, e1, e2) for configuration cfg (bit k: corner k inside)."""
    inside = [(cfg >> k) & 1 for k in range(8)]
    links = {}

    def link(e, f):
        links.setdefault(e, []).append(f)
        links.setdefault(f, []).append(e)

    for ring in FACES:
        cut = [_edge(ring[i], ring[(i + 1) % 4]) for i in range(4) if inside[ring[i]] != inside[ring[(i + 1) % 4]]]
        if len(cut) == 2:
            link(cut[0], cut[1])
        elif len(cut) == 4:  # ambiguous face: keep the two inside corners apart
            for i in range(4):
                if inside[ring[i]]:
                    link(_edge(ring[i - 1], ring[i]), _edge(ring[i], ring[(i + 1) % 4]))
    assert all(len(v) == 2 for v in links.values())
    tris, seen = [], set()
    mid = lambda e: (np.array(CORNERS[EDGES[e][0]], float) + np.array(CORNERS[EDGES[e][1]], float)) / 2  # noqa: E731
    for start in sorted(links):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [f for f in links[cur] if f != prev]
            nxt = nxt[0] if nxt else links[cur][0]
            if links[cur][0] == links[cur][1]:
                nxt = links[cur][0]
            if nxt == start:
                break
            loop.append(nxt)
            seen.add(nxt)
            prev, cur = cur, nxt
        assert len(loop) >= 3
        # orientation: Newell normal of the loop (edge mid-points) against the inside -> outside direction of its edges
        P = [mid(e) for e in loop]
        n = sum(np.cross(P[i], P[(i + 1) % len(P)]) for i in range(len(P)))
        g = np.zeros(3)
        for e in loop:
            a, b = EDGES[e]
            ca, cb = np.array(CORNERS[a], float), np.array(CORNERS[b], float)
            g += (cb - ca) if inside[a] else (ca - cb)
        if float(np.dot(n, g)) < 0:
            loop = loop[::-1]
        for i in range(1, len(loop) - 1):
            tris.append((loop[0], loop[i], loop[i + 1]))
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
