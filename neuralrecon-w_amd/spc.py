"""Structured point clouds (SPC) in kaolin's published tensor format, on the device, and the ray / voxel query over them.

The reference stores its voxel guidance as kaolin SPCs and calls six kaolin functions on them
(tools/prepare_data/generate_voxel.py:149-150 `quantize_points` / `unbatched_points_to_octree`, :175-176 `scan_octrees` /
`generate_points`, :185 `to_dense`, :358-368 `unbatched_raytrace`).  kaolin is a CUDA-only package whose source is not part of
the reference tree; this module implements the DOCUMENTED tensor contracts of those six calls (torch ops for the data
structure, `ncw_ray_voxel_trace` for the ray query), so that `compat/kaolin` can present them under kaolin's names and the
reference's own `gen_octree` / `convert_to_dense` / `octree_to_spc` / `get_near_far` / `NeuconWSystem.octree_update` run
unedited on a ROCm machine.  Parity with kaolin itself is UNPINNED: the formats below follow kaolin's
documentation --

  * octree       uint8 [n_nodes]: one byte per NON-LEAF node, breadth first, nodes of a level in Morton order; bit c of a byte =
                 child c exists, c = 4 x + 2 y + z of the child's position inside its parent;
  * pyramid      int32 [2, L + 2]: row 0 = points per level 0 .. L (then 0), row 1 = their exclusive prefix sum (then the total);
  * exsum        int32 [n_nodes + 1]: exclusive prefix sum of the bytes' bit counts;
  * point_hierarchy int16 [total, 3]: integer coordinates of every node, level by level, Morton order inside a level;
  * nuggets      (ray_index [N] int32, point_index [N] int32 into point_hierarchy, depth [N, 1 | 2]) ordered by ray, then depth.
"""
import torch

from . import lib as L

_OFF = None


def _child_offsets(dev):
    c = torch.arange(8, device=dev)
    return torch.stack([(c >> 2) & 1, (c >> 1) & 1, c & 1], -1)  # child c -> (x, y, z) bit


def quantize_points(x, level):
    """kaolin.ops.spc.points.quantize_points: floor(clamp(2^level (x + 1) / 2, 0, 2^level - 1)) as int16 [N, 3]."""
    res = 2 ** int(level)
    return torch.floor(torch.clamp(res * (x + 1.0) / 2.0, 0, res - 1.0)).short()


def points_to_morton(q):
    """[N, 3] integer coordinates -> int64 Morton codes, x in the most significant bit of each triple."""
    q = q.long()
    m = torch.zeros(q.shape[0], dtype=torch.int64, device=q.device)
    for i in range(16):
        m |= ((q[:, 2] >> i) & 1) << (3 * i)
        m |= ((q[:, 1] >> i) & 1) << (3 * i + 1)
        m |= ((q[:, 0] >> i) & 1) << (3 * i + 2)
    return m


def morton_to_points(m):
    q = torch.zeros(m.shape[0], 3, dtype=torch.int64, device=m.device)
    for i in range(16):
        q[:, 2] |= ((m >> (3 * i)) & 1) << i
        q[:, 1] |= ((m >> (3 * i + 1)) & 1) << i
        q[:, 0] |= ((m >> (3 * i + 2)) & 1) << i
    return q.short()


def unbatched_points_to_octree(points, level, sorted=False):
    """kaolin.ops.spc.unbatched_points_to_octree: quantised points int16 [N, 3] (duplicates allowed) -> octree uint8."""
    level = int(level)
    m = points_to_morton(points)
    m = torch.unique(m)  # sorted
    levels = []
    for _ in range(level):
        parent, child = m >> 3, m & 7
        up, inv = torch.unique_consecutive(parent, return_inverse=True)
        byte = torch.zeros(up.shape[0], dtype=torch.int64, device=m.device)
        byte.index_add_(0, inv, torch.ones_like(child) << child)  # children are unique: the sum is the OR
        levels.append(byte.to(torch.uint8))
        m = up
    return torch.cat(levels[::-1])


_POP8 = None


def _popcount8(b):
    global _POP8
    if _POP8 is None or _POP8.device != b.device:
        t = torch.arange(256, device=b.device)
        _POP8 = sum(((t >> i) & 1) for i in range(8)).to(torch.int32)
    return _POP8[b.long()]


def scan_octrees(octree, lengths):
    """kaolin.ops.spc.scan_octrees for ONE octree (the reference's only use: lengths = [len(octree)]) ->
    (max_level, pyramid int32 [1, 2, max_level + 2] (on the CPU like kaolin's), exsum int32 [len + 1])."""
    if int(lengths.numel()) != 1 or int(lengths.reshape(-1)[0]) != octree.shape[0]:
        raise NotImplementedError("scan_octrees: one octree per call (lengths = [len(octree)])")
    pop = _popcount8(octree)
    exsum = torch.zeros(octree.shape[0] + 1, dtype=torch.int32, device=octree.device)
    exsum[1:] = torch.cumsum(pop, 0)
    ex = exsum.cpu()
    counts, at, n = [1], 0, 1
    while at < octree.shape[0]:
        nxt = int(ex[at + n] - ex[at])
        at += n
        n = nxt
        counts.append(n)
    max_level = len(counts) - 1
    pyramid = torch.zeros(1, 2, max_level + 2, dtype=torch.int32)
    pyramid[0, 0, :max_level + 1] = torch.tensor(counts, dtype=torch.int32)
    pyramid[0, 1, 1:] = torch.cumsum(pyramid[0, 0, :max_level + 1], 0)
    return max_level, pyramid, exsum


def generate_points(octree, pyramid, exsum):
    """kaolin.ops.spc.generate_points -> point_hierarchy int16 [total, 3] (all levels, root first)."""
    pyr = pyramid.reshape(2, -1)
    max_level = pyr.shape[1] - 2
    dev = octree.device
    off = _child_offsets(dev)
    pts = torch.zeros(1, 3, dtype=torch.int64, device=dev)
    out = [pts]
    at = 0
    shifts = torch.arange(8, device=dev)
    for l in range(max_level):
        n = int(pyr[0, l])
        mask = ((octree[at:at + n].long()[:, None] >> shifts[None, :]) & 1).bool()
        idx = mask.nonzero()  # row-major: by parent, then by child -> Morton order
        pts = pts[idx[:, 0]] * 2 + off[idx[:, 1]]
        out.append(pts)
        at += n
    return torch.cat(out).short()


def level_points(point_hierarchy, pyramid, level):
    pyr = pyramid.reshape(2, -1)
    lvl = int(level) if int(level) >= 0 else pyr.shape[1] - 2
    s = int(pyr[1, lvl])
    return point_hierarchy[s:s + int(pyr[0, lvl])], s, lvl


def to_dense(point_hierarchies, pyramids, input, level=-1):
    """kaolin.ops.spc.to_dense for one SPC: features `input` [n_level_points, C] -> [1, C, G, G, G] (x first)."""
    pts, _, lvl = level_points(point_hierarchies, pyramids, level)
    G = 1 << lvl
    C_ = input.shape[1]
    out = torch.zeros(1, C_, G, G, G, dtype=input.dtype, device=input.device)
    p = pts.long()
    out[0, :, p[:, 0], p[:, 1], p[:, 2]] = input.t()
    return out


def occupancy_bits(q, level):
    """Integer voxel coordinates [K, 3] at `level` -> (occ int32 [G^3 / 32], brick int32 [ceil((G / 8)^3 / 32)]): the bit masks
    `ncw_voxel_build` writes (csrc/ncw_voxel.hip), built with torch ops so that they exist on any device."""
    G = 1 << int(level)
    q = q.long()

    def pack(lin, n_bits):
        lin = torch.unique(lin)
        words = torch.zeros((n_bits + 31) // 32, dtype=torch.int64, device=q.device)
        words.index_add_(0, lin >> 5, torch.ones_like(lin) << (lin & 31))
        return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)

    Gb = max(G >> 3, 1)
    occ = pack((q[:, 0] * G + q[:, 1]) * G + q[:, 2], G * G * G)
    b = q >> 3
    brick = pack((b[:, 0] * Gb + b[:, 1]) * Gb + b[:, 2], Gb ** 3)
    return occ, brick


_OCC_CACHE = {}


def _occupancy_of(point_hierarchy, pyramid, level):
    """(occ, brick, Morton codes of the level's points, their offset in the hierarchy, level) of an SPC, cached per point-hierarchy
    TENSOR: the entry holds a weak reference to it and is only used while that very object is alive and unmodified (an address can be
    reused by another tensor of the same shape)."""
    import weakref

    key = (id(point_hierarchy), int(level))
    hit = _OCC_CACHE.get(key)
    if hit is not None and (hit[0]() is not point_hierarchy or hit[1] != point_hierarchy._version):
        hit = None
    if hit is None:
        pts, start, lvl = level_points(point_hierarchy, pyramid, level)
        occ, brick = occupancy_bits(pts, lvl)
        mort = points_to_morton(pts)  # ascending by construction
        for k in [k for k, v in _OCC_CACHE.items() if v[0]() is None]:
            del _OCC_CACHE[k]
        if len(_OCC_CACHE) > 8:
            _OCC_CACHE.clear()
        hit = _OCC_CACHE[key] = (weakref.ref(point_hierarchy), point_hierarchy._version, occ, brick, mort, start, lvl)
    return hit[2:]


def unbatched_raytrace(octree, point_hierarchy, pyramid, exsum, origin, direction, level, return_depth=True, with_exit=False):
    """kaolin.render.spc.unbatched_raytrace: rays (origin [R, 3] in the SPC's [-1, 1]^3 cube, direction [R, 3]) against the
    occupied voxels of `level` -> (ray_index int32 [N], point_index int32 [N], depth float32 [N, 2 if with_exit else 1]),
    ordered by ray and depth.  HIP only (`ncw_ray_voxel_trace`): CPU tensors raise."""
    if not origin.is_cuda:
        raise L.NeuconwHipError("spc.unbatched_raytrace needs GPU tensors; there is no CPU fallback")
    dev = origin.device
    occ, brick, mort, start, lvl = _occupancy_of(point_hierarchy, pyramid, level)
    if not 3 <= lvl <= 10:
        raise ValueError("octree level %d outside the supported 3..10" % lvl)
    o = origin.contiguous().float()
    d = direction.contiguous().float()
    R = o.shape[0]
    lib = L.get_lib()
    counts = torch.zeros(R, dtype=torch.int32, device=dev)
    st = L.stream_ptr(dev)
    L.check(lib.ncw_ray_voxel_trace(L.ptr(o), L.ptr(d), R, lvl, L.ptr(occ), L.ptr(brick), None, L.ptr(counts), None, None, None,
                                    st), "ncw_ray_voxel_trace")
    incl = torch.cumsum(counts, 0)
    n = int(incl[-1]) if R else 0
    offsets = (incl - counts).to(torch.int32)
    ray = torch.empty(n, dtype=torch.int32, device=dev)
    vox = torch.empty(n, dtype=torch.int32, device=dev)
    depth = torch.empty(n, 2, dtype=torch.float32, device=dev)
    if n:
        L.check(lib.ncw_ray_voxel_trace(L.ptr(o), L.ptr(d), R, lvl, L.ptr(occ), L.ptr(brick), L.ptr(offsets), None, L.ptr(ray),
                                        L.ptr(vox), L.ptr(depth), st), "ncw_ray_voxel_trace")
    G = 1 << lvl
    v = vox.long()
    q = torch.stack([v // (G * G), (v // G) % G, v % G], -1)
    pid = (torch.searchsorted(mort, points_to_morton(q)) + start).to(torch.int32)
    if not return_depth:
        return ray, pid
    return ray, pid, (depth if with_exit else depth[:, :1].contiguous())
