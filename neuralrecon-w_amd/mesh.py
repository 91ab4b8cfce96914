"""Mesh extraction on the GPU: the steps after the SDF grid sweep of config 5.

Reference: utils/visualization.py:37-159 `extract_mesh` (rank-0 CPU `skimage.measure.marching_cubes`, vertex
colours through `renderer.rgb`, trimesh export) driven by tools/extract_mesh.py:104-168.  Here the SDF grid never
leaves the GPU: `grid.sdf_grid` -> `isosurface` (marching cubes, csrc/ncw_mesh.hip + the generated case tables of
mc_tables.py) -> vertex welding with torch.unique -> optional colour pass -> binary PLY.  The VERTEX SET is the reference's
(one linear zero crossing per sign-changing grid edge of the enabled cubes: tests/test_gpu_mesh.py checks it edge by
edge); skimage's triangulation of the ambiguous cube configurations (Lewiner) is not reproducible without skimage, which
is neither in the reference tree nor installable: **triangulation parity is unpinned** there.  Also kept: the `mask`
semantics, the sparse mode (`gen_grid_spc`, tools/extract_mesh.py:60-102: SDF only inside the occupied octree voxels, cubes with
all 8 corners evaluated) and the coordinate chain `verts * voxel_size + vol_origin`, `* scene_radius + scene_origin` (:116-117).
Normals / winding point towards increasing SDF (outwards).
"""
import numpy as np
import torch

from . import grid as _grid
from . import lib as L


_TABLES = {}


def _mc_tables(dev):
    """The generated marching-cubes case tables (mc_tables.py) on `dev`."""
    t = _TABLES.get(str(dev))
    if t is None:
        from . import mc_tables

        tri, ntri, edges = mc_tables.tables()
        t = _TABLES[str(dev)] = (torch.from_numpy(tri).to(dev).contiguous(), torch.from_numpy(ntri).to(dev).contiguous(),
                                 torch.from_numpy(edges).to(dev).contiguous())
    return t


@torch.no_grad()
def isosurface(sdf, level=0.0, mask=None):
    """sdf [Dx,Dy,Dz] float32 on the GPU (x slowest) -> (verts [V,3] float32 in grid-index coordinates,
    faces [F,3] int64).  mask: bool [Dx,Dy,Dz] or None -- cube (i,j,k) is used iff mask[i+1,j+1,k+1]."""
    if not sdf.is_cuda:
        raise L.NeuconwHipError("mesh.isosurface: the SDF grid is not on a GPU; there is no CPU fallback")
    sdf = sdf.contiguous().float()
    Dx, Dy, Dz = sdf.shape
    dev = sdf.device
    m8 = None if mask is None else mask.to(device=dev, dtype=torch.uint8).contiguous()
    ncubes = (Dx - 1) * (Dy - 1) * (Dz - 1)
    counts = torch.empty(ncubes, dtype=torch.int32, device=dev)
    tri, ntri, edges = _mc_tables(dev)
    lib = L.get_lib()
    L.check(lib.ncw_mc_count(L.ptr(sdf), L.ptr(m8), Dx, Dy, Dz, float(level), L.ptr(ntri), L.ptr(counts), L.stream_ptr(dev)),
            "ncw_mc_count")
    incl = torch.cumsum(counts, 0, dtype=torch.int64)
    T = int(incl[-1])  # one device->host read: the mesh size
    if T == 0:
        return torch.zeros(0, 3, device=dev), torch.zeros(0, 3, dtype=torch.int64, device=dev)
    offsets = (incl - counts).contiguous()
    pos = torch.empty(T, 3, 3, device=dev, dtype=torch.float32)
    key = torch.empty(T, 3, device=dev, dtype=torch.int64)
    L.check(lib.ncw_mc_emit(L.ptr(sdf), L.ptr(m8), Dx, Dy, Dz, float(level), L.ptr(tri), L.ptr(ntri), L.ptr(edges),
                            L.ptr(offsets), L.ptr(pos), L.ptr(key), L.stream_ptr(dev)), "ncw_mc_emit")
    uniq, inv = torch.unique(key.reshape(-1), return_inverse=True)  # weld: one vertex per grid edge
    verts = torch.empty(uniq.shape[0], 3, device=dev, dtype=torch.float32)
    verts[inv] = pos.reshape(-1, 3)  # all copies of a vertex are bit-identical (interpolated lo -> hi)
    faces = inv.reshape(T, 3)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return verts, faces[ok]  # a value exactly on the level collapses an edge: drop the degenerate faces


@torch.no_grad()
def vertex_colors(renderer, verts_training, embedding_a, chunk=1 << 16):
    """utils/visualization.py:124-150: rgb at the vertices, viewed along +z, one appearance code for all."""
    V = verts_training.shape[0]
    dev = verts_training.device
    out = torch.empty(V, 3, device=dev)
    a = embedding_a.reshape(1, -1).to(dev).float()
    for i in range(0, V, chunk):
        p = verts_training[i:i + chunk].float()
        d = torch.zeros_like(p)
        d[:, 2] = 1
        with torch.enable_grad():  # renderer.rgb runs NeuconW.forward, which needs the SDF gradient
            out[i:i + chunk] = renderer.rgb(p.unsqueeze(1), d.unsqueeze(1), a.expand(p.shape[0], -1).unsqueeze(1)).detach()
    return out * 255


@torch.no_grad()
def gen_grid_spc(octree_data, eval_level):
    """tools/extract_mesh.py:60-102 `gen_grid_spc` on the device: the lower corners of the level-`eval_level` sub-voxels of
    the occupied voxels of the training octree (`voxel.octree_from_sfm`), in SfM coordinates.  Same arithmetic and dtypes as the
    reference: index * voxel in float32, + float64 volume origin, (float32 downstream).  Returns the reference's `sparse_data`
    dict (sparse_vol [K,3] float64 tensor on the GPU, voxel_size / vol_origin as numpy float64, dim)."""
    from . import voxel

    level = int(octree_data["level"])
    dense = voxel.dense_from_occupancy(octree_data)           # convert_to_dense(octree, level)
    dev = dense.device
    low_dim = dense.shape[0]
    sparse_ind = torch.nonzero(dense)                         # [n,3], lexicographic in (x,y,z)
    up_level = int(eval_level) - level
    if up_level < 0:
        raise ValueError("eval_level %d below the octree level %d" % (eval_level, level))
    up_times = 2 ** up_level
    eval_dim = int(low_dim * up_times)
    k = torch.arange(0, up_times, device=dev)
    up_kernel = torch.stack(torch.meshgrid(k, k, k, indexing="ij"), dim=-1).reshape(-1, 3)
    ind_up = sparse_ind.repeat_interleave(up_times ** 3, dim=0) * up_times + up_kernel.repeat([sparse_ind.shape[0], 1])
    octree_scale = np.float64(octree_data["scale"])
    so = octree_data["scene_origin"]
    octree_origin = np.asarray(so.detach().cpu().numpy() if torch.is_tensor(so) else so, dtype=np.float64).reshape(3)
    eval_voxel_size = 2 / (2 ** int(eval_level)) * octree_scale                       # :91
    vol_origin = octree_origin - octree_scale                                         # :92
    xyz_sfm = (ind_up * float(eval_voxel_size)).double() + torch.from_numpy(vol_origin).to(dev)  # :94 (f32 product, f64 sum)
    return {"sparse_vol": xyz_sfm, "voxel_size": eval_voxel_size, "dim": eval_dim, "vol_origin": vol_origin}


@torch.no_grad()
def sparse_volume(sparse_data, sdf_values):
    """utils/visualization.py:51-56,91-110: scatter the SDF of the sparse points into a dense [dim]^3 volume of ones and build
    the cube mask -- a grid point is valid iff it and its 7 lower neighbours were evaluated (the reference's rolls, wrap-around
    included).  Returns (sdf_dense, mask, ind)."""
    sparse_vol = sparse_data["sparse_vol"].float()
    dev = sparse_vol.device
    dim = int(sparse_data["dim"])
    vo64 = torch.from_numpy(np.asarray(sparse_data["vol_origin"], dtype=np.float64)).to(dev)
    ind = torch.round((sparse_vol - vo64) / float(sparse_data["voxel_size"])).long()  # :51 (float64 like the reference)
    sdf_dense = torch.ones(dim, dim, dim, device=dev, dtype=torch.float32)
    sdf_dense[ind[:, 0], ind[:, 1], ind[:, 2]] = sdf_values.reshape(-1).float()
    m = torch.zeros(dim, dim, dim, device=dev, dtype=torch.bool)
    m[ind[:, 0], ind[:, 1], ind[:, 2]] = True
    m = (m & torch.roll(m, shifts=1, dims=0) & torch.roll(m, shifts=1, dims=1) & torch.roll(m, shifts=1, dims=2)
         & torch.roll(m, shifts=[1, 1], dims=[0, 1]) & torch.roll(m, shifts=[1, 1], dims=[0, 2])
         & torch.roll(m, shifts=[1, 1], dims=[1, 2]) & torch.roll(m, shifts=[1, 1, 1], dims=[0, 1, 2]))
    return sdf_dense, m, ind


@torch.no_grad()
def extract_mesh(renderer, dim, scene_radius, scene_origin, origin=None, radius=1.0, with_color=False, embedding_a=None,
                 level=0.0, chunk_rgb=1 << 16, sparse_data=None, chunk=1 << 22, group=None):
    """utils/visualization.py:37-159.  Dense path: the [dim]^3 lattice over `origin` +- `radius` (training coordinates), swept
    on chip and sharded over the ranks (grid.sdf_grid).  Sparse path (`sparse_data` from gen_grid_spc): the SDF only at the
    sub-voxels of the occupied octree voxels (sharded like neuconw_system.py:236-256), marching cubes under the all-8-corners
    mask.  Returns dict(vertices [V,3] world coordinates, faces [F,3], vertices_training, colors [V,3] uint8 or None) as GPU
    tensors, on every rank (the reference returns the mesh on rank 0 only)."""
    dev = next(renderer.neuconw.parameters()).device
    so = torch.as_tensor(np.asarray(scene_origin, dtype=np.float32), device=dev).reshape(3)
    if sparse_data is None:
        origin = [0.0, 0.0, 0.0] if origin is None else [float(v) for v in origin]
        lo = tuple(o - radius for o in origin)
        hi = tuple(o + radius for o in origin)
        sdf = _grid.sdf_grid(renderer.neuconw.sdf_net, dim, lo, hi, prec=renderer.infer_prec, group=group).view(dim, dim, dim)
        verts, faces = isosurface(sdf, level)
        voxel_size = 2 * radius / (dim - 1)                                      # :44
        vol_origin = torch.tensor(lo, device=dev, dtype=torch.float32)           # :43
    else:
        from . import voxel

        sparse_vol = sparse_data["sparse_vol"].to(dev).float()
        xyz = ((sparse_vol - so) / float(scene_radius)).contiguous()             # :58
        sdf_sparse = voxel._sdf_sharded(renderer, xyz, int(chunk), group)
        sdf, mask, _ = sparse_volume(dict(sparse_data, sparse_vol=sparse_vol), sdf_sparse)
        verts, faces = isosurface(sdf, level, mask)
        vo_sfm = torch.from_numpy(np.asarray(sparse_data["vol_origin"], dtype=np.float64)).float().to(dev)  # :53
        vol_origin = (vo_sfm - so) / float(scene_radius)                         # :59
        voxel_size = float(sparse_data["voxel_size"]) / float(scene_radius)      # :61
    verts_t = verts * voxel_size + vol_origin                                    # :116
    verts_w = verts_t * float(scene_radius) + so                                 # :117
    colors = None
    if with_color:
        colors = vertex_colors(renderer, verts_t, embedding_a, chunk_rgb).clamp(0, 255).to(torch.uint8)
    return {"vertices": verts_w, "faces": faces, "vertices_training": verts_t, "colors": colors}


def write_ply(path, vertices, faces, colors=None):
    """Binary little-endian PLY (what trimesh's export writes for tools/extract_mesh.py:160-168)."""
    v = vertices.detach().cpu().numpy().astype("<f4")
    f = faces.detach().cpu().numpy().astype("<i4")
    hdr = ["ply", "format binary_little_endian 1.0", "element vertex %d" % v.shape[0], "property float x",
           "property float y", "property float z"]
    if colors is not None:
        hdr += ["property uchar red", "property uchar green", "property uchar blue"]
    hdr += ["element face %d" % f.shape[0], "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(hdr) + "\n").encode("ascii"))
        if colors is None:
            fh.write(v.tobytes())
        else:
            c = colors.detach().cpu().numpy().astype("u1")
            rec = np.empty(v.shape[0], dtype=[("p", "<f4", 3), ("c", "u1", 3)])
            rec["p"], rec["c"] = v, c
            fh.write(rec.tobytes())
        rec = np.empty(f.shape[0], dtype=[("n", "u1"), ("i", "<i4", 3)])
        rec["n"], rec["i"] = 3, f
        fh.write(rec.tobytes())
