"""Voxel guidance (BASELINE config 3): occupancy grids and ray / voxel near-far on the GPU.

Replaces the kaolin-backed helpers of the reference's tools/prepare_data/generate_voxel.py
(`gen_octree`, `get_near_far`) for the renderer's `octree_data` / `fine_octree_data` attributes.  The
dictionaries keep the reference's keys that the renderer reads (`scene_origin`, `scale`, `level`,
`voxel_size`) and replace the kaolin SPC tensors by `occ` / `brick` bit masks.
"""
import ctypes as C
import math

import torch

from . import lib as L
from . import spc


class OctreeData(dict):
    """`renderer.octree_data` / `renderer.fine_octree_data`.  Holds what the HIP kernels read (`occ` / `brick` bit masks) beside
    the reference's keys (`scene_origin`, `scale`, `level`, `voxel_size`).  The reference's kaolin-format members -- `octree`
    (generate_voxel.py:150) and `spc_data` = {points, pyramid, prefix} (neuconw_system.py:300-305) -- are derived on first
    access, so code written against the reference's dictionaries (its `surface_selection`, `convert_to_dense`, checkpointing of
    the octree) finds them, and a training step never pays for them."""

    def __missing__(self, key):
        if key in ("octree", "spc_data") and "occ" in self:
            q = voxels_from_occupancy(self)
            octree = spc.unbatched_points_to_octree(q, int(self["level"]))
            _, pyramid, prefix = spc.scan_octrees(octree, torch.tensor([octree.shape[0]], dtype=torch.int32))
            self["octree"] = octree
            self["spc_data"] = {"points": spc.generate_points(octree, pyramid, prefix), "pyramid": pyramid[0], "prefix": prefix}
            return self[key]
        if key in ("occ", "brick") and "octree" in self:
            ensure_occupancy(self)
            return self[key]
        raise KeyError(key)


def ensure_occupancy(octree_data):
    """A dictionary in the REFERENCE's format (built by its own `gen_octree` / `octree_update` over compat/kaolin: `octree`
    + optionally `spc_data`, no bit masks) gets the `occ` / `brick` masks the kernels read, once."""
    if dict.__contains__(octree_data, "occ"):
        return octree_data
    octree, level = octree_data["octree"], int(octree_data["level"])
    sd = octree_data.get("spc_data") if dict.__contains__(octree_data, "spc_data") else None
    if sd is None:
        _, pyramid, prefix = spc.scan_octrees(octree, torch.tensor([octree.shape[0]], dtype=torch.int32))
        points, pyramid = spc.generate_points(octree, pyramid, prefix), pyramid[0]
    else:
        points, pyramid = sd["points"], sd["pyramid"]
    q, _, _ = spc.level_points(points, pyramid, level)
    octree_data["occ"], octree_data["brick"] = spc.occupancy_bits(q, level)
    return octree_data


def voxels_from_occupancy(octree_data):
    """Integer coordinates [K, 3] (x, y, z; lexicographic) of the occupied voxels, decoded from the non-zero words only."""
    occ = ensure_occupancy(octree_data)["occ"]
    G = 1 << int(octree_data["level"])
    w = occ.nonzero().reshape(-1)
    bits = (occ[w].view(-1, 1) >> torch.arange(32, device=occ.device, dtype=torch.int32).view(1, 32)) & 1
    hit = bits.nonzero()
    lin = w[hit[:, 0]] * 32 + hit[:, 1]
    return torch.stack([lin // (G * G), (lin // G) % G, lin % G], -1)


def occupancy_from_points(points_sfm, scene_origin, scale, level, voxel_size=None):
    """Build octree_data from SfM-space points [N,3] (the occupied voxels are those containing a
    point), scene cube = scene_origin +- scale.  generate_voxel.py:75-171 (gen_octree) equivalent."""
    dev = points_sfm.device
    if not points_sfm.is_cuda:
        raise L.NeuconwHipError("voxel.occupancy_from_points needs GPU tensors")
    G = 1 << level
    origin = torch.as_tensor(scene_origin, dtype=torch.float32, device=dev).reshape(3)
    pn = ((points_sfm.float() - origin) / float(scale)).contiguous()
    occ = torch.zeros(G * G * G // 32, dtype=torch.int32, device=dev)
    gb = max(G // 8, 1)
    brick = torch.zeros((gb ** 3 + 31) // 32, dtype=torch.int32, device=dev)
    L.check(L.get_lib().ncw_voxel_build(L.ptr(pn), pn.shape[0], level, L.ptr(occ), L.ptr(brick), L.stream_ptr(dev)),
            "ncw_voxel_build")
    return OctreeData({"occ": occ, "brick": brick, "scene_origin": origin, "scale": float(scale), "level": int(level),
                       "voxel_size": float(voxel_size) if voxel_size is not None else 2.0 * float(scale) / G})


def occupancy_from_dense(dense_bool, scene_origin, scale, voxel_size=None):
    """dense_bool [G,G,G] (x slowest) -> octree_data (tests / synthetic shells of SURVEY 8d config 3)."""
    G = dense_bool.shape[0]
    level = int(round(math.log2(G)))
    idx = dense_bool.nonzero().float()
    centres = (idx + 0.5) * (2.0 / G) - 1.0
    origin = torch.as_tensor(scene_origin, dtype=torch.float32, device=dense_bool.device).reshape(3)
    return occupancy_from_points(centres * float(scale) + origin, scene_origin, scale, level, voxel_size)


def read_points3d_xyz(path, min_track_length):
    """COLMAP `points3D.bin` (utils/colmap_utils.py:264-291: u64 count, then per point `<QdddBBBd` + u64 track
    length + that many `<ii` track elements) -> float64 [N,3] xyz of the points whose track is LONGER than
    `min_track_length` (generate_voxel.py:55-58: `p.point2D_idxs.shape[0] > min_track_length`), file order."""
    import struct

    import numpy as np

    with open(path, "rb") as f:
        buf = f.read()
    (n,) = struct.unpack_from("<Q", buf, 0)
    off, keep = 8, []
    for _ in range(n):
        x, y, z = struct.unpack_from("<ddd", buf, off + 8)
        (track,) = struct.unpack_from("<Q", buf, off + 43)
        off += 51 + 8 * track
        if track > min_track_length:
            keep.append((x, y, z))
    if off != len(buf):
        raise ValueError("%s: %d trailing bytes after %d points (not a COLMAP points3D.bin?)" % (path, len(buf) - off, n))
    return np.asarray(keep, dtype=np.float64).reshape(-1, 3)


def sfm_cube(scene_config, radius=1.0):
    """generate_voxel.py:87-118: the evaluation box of the scene's config.yaml carried into SfM space by
    inv(sfm2gt); scene cube = centre +- scale with scale = (longest box edge) / 2 * radius.  float64 numpy."""
    import numpy as np

    gt_to_sfm = np.linalg.inv(np.array(scene_config["sfm2gt"], dtype=np.float64))
    v1 = gt_to_sfm[:3, :3] @ np.array(scene_config["eval_bbx"][0], dtype=np.float64) + gt_to_sfm[:3, 3]
    v2 = gt_to_sfm[:3, :3] @ np.array(scene_config["eval_bbx"][1], dtype=np.float64) + gt_to_sfm[:3, 3]
    lo, hi = np.minimum(v1, v2), np.maximum(v1, v2)
    return lo + (hi - lo) / 2, float(np.max(hi - lo) / 2 * radius)


def dilate_points(points, voxel_size):
    """generate_voxel.py:24-38 `expand_points`: every point plus its 26 neighbours at +-voxel_size (float64)."""
    import itertools

    import numpy as np

    offs = np.array(list(itertools.product((-1.0, 0.0, 1.0), repeat=3)), dtype=np.float64) * float(voxel_size)
    return (points[None, :, :] + offs[:, None, :]).reshape(-1, 3)


def octree_from_sfm(recontruct_path, min_track_length, voxel_size, device, sfm_path="sparse", expand=1, radius=1.0):
    """NeuconWRenderer.get_octree (renderer.py:137-155) -> generate_voxel.py:41-73 `gen_octree_from_sfm` + :75-171
    `gen_octree`: COLMAP points with a track longer than `min_track_length`, dilated once by +-voxel_size, cropped
    to the scene cube, voxelised at level floor(log2(2 scale / voxel_size)).  The file reading and the float64
    geometry are numpy like the reference; the voxelisation is `ncw_voxel_build` (octree_from_points).  The
    returned octree_data carries the keys the renderer and the octree refresh read (`scene_origin`, `scale`,
    `level`) with the kaolin SPC tensors replaced by the `occ` / `brick` bit masks."""
    import os

    import numpy as np
    import yaml

    with open(os.path.join(recontruct_path, "config.yaml"), "r") as f:
        scene_config = yaml.load(f, Loader=yaml.FullLoader)
    pts = read_points3d_xyz(os.path.join(recontruct_path, "dense", sfm_path, "points3D.bin"), min_track_length)
    if pts.shape[0] == 0:
        raise ValueError("no SfM point has a track longer than %d" % min_track_length)
    for _ in range(int(expand)):
        pts = dilate_points(pts, voxel_size)
    scene_origin, scale = sfm_cube(scene_config, radius)
    dev = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
    data = octree_from_points(torch.from_numpy(np.ascontiguousarray(pts)).to(dev), voxel_size, scene_origin, scale)
    data["scene_origin"] = torch.from_numpy(scene_origin).to(dev)  # float64, like renderer.py:144
    return data


def get_near_far(rays_o_sfm, rays_d, octree_data):
    """generate_voxel.py:311-439: (near, far) [R,1] in SfM units; 0 where the ray misses."""
    dev = rays_o_sfm.device
    R = rays_o_sfm.shape[0]
    o = rays_o_sfm.contiguous().float()
    d = rays_d.contiguous().float()
    near = torch.empty(R, device=dev, dtype=torch.float32)
    far = torch.empty(R, device=dev, dtype=torch.float32)
    ensure_occupancy(octree_data)
    so_host = octree_data.get("_scene_origin_host")
    if so_host is None:  # one device->host read per octree, not per step
        so = octree_data["scene_origin"]
        so_host = (C.c_float * 3)(*[float(v) for v in (so.tolist() if hasattr(so, "tolist") else so)])
        octree_data["_scene_origin_host"] = so_host
    L.check(L.get_lib().ncw_ray_voxel_near_far(L.ptr(o), L.ptr(d), R, so_host, float(octree_data["scale"]),
                                               int(octree_data["level"]), L.ptr(octree_data["occ"]),
                                               L.ptr(octree_data["brick"]), L.ptr(near), L.ptr(far),
                                               L.stream_ptr(dev)), "ncw_ray_voxel_near_far")
    return near.reshape(R, 1), far.reshape(R, 1)


# ---------------------------------------------------------------------------------------------------
# Octree refresh on the device: lightning_modules/neuconw_system.py:186-312
# (`surface_selection`, `octree_update`).  The reference bounces through the CPU (dense grid, nonzero,
# repeat_interleave, chunked sdf calls with .cpu() per chunk, numpy, kaolin); here the occupied coarse voxels
# are expanded, swept with the fused SDF kernel and re-voxelised without leaving the GPU.
# ---------------------------------------------------------------------------------------------------
def dense_from_occupancy(octree_data):
    """[G,G,G] bool (x slowest) from the bit mask: generate_voxel.py:181-186 `convert_to_dense`."""
    G = 1 << int(octree_data["level"])
    occ = ensure_occupancy(octree_data)["occ"]
    bits = (occ.view(-1, 1) >> torch.arange(32, device=occ.device, dtype=torch.int32).view(1, 32)) & 1
    return bits.reshape(-1)[: G * G * G].bool().view(G, G, G)


def shard_range(total, rank, world):
    """utils/visualization.py:27-35 `get_local_split`: pad to a multiple of world (a whole extra row per rank
    when not divisible), contiguous equal slices.  Returns (start, valid_count, padded_slice_length)."""
    padded = total if total % world == 0 else (total // world + 1) * world
    per = padded // world
    start = rank * per
    return start, max(0, min(per, total - start)), per


def _sdf_sharded(renderer, xyz_training, chunk, group=None, force_collective=False):
    """renderer.sdf over all points; with torch.distributed initialised each rank sweeps its slice and ONE
    all_gather assembles the result (neuconw_system.py:236-256)."""
    import torch.distributed as dist

    n = xyz_training.shape[0]
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    start, count, per = shard_range(n, rank, world)
    local = torch.zeros(per, device=xyz_training.device, dtype=torch.float32)
    for i in range(0, count, chunk):
        j = min(count, i + chunk)
        local[i:j] = renderer.sdf(xyz_training[start + i:start + j].reshape(-1, 1, 3)).reshape(-1)
    if world == 1 and not (force_collective and dist.is_available() and dist.is_initialized()):
        return local[:n]
    parts = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(parts, local, group=group)
    return torch.cat(parts)[:n]


@torch.no_grad()
def surface_selection(renderer, train_level, threshold, chunk=1 << 22, group=None, force_collective=False):
    """neuconw_system.py:186-264.  Returns (points_sfm [K,3] float32 on the GPU: the lower corners of the
    level-`train_level` sub-voxels of the occupied coarse voxels whose sdf <= threshold; train_voxel_size).
    Same arithmetic and operation order as the reference (float32 `ind * voxel + origin`), so the selected
    set is the reference's up to sdf rounding at the threshold."""
    od = renderer.octree_data
    if od is None:
        od = renderer.octree_data = renderer.get_octree(renderer.origin.device)
    dev = ensure_occupancy(od)["occ"].device
    level = int(od["level"])
    octree_origin = od["scene_origin"].float().to(dev).reshape(3)
    octree_scale = float(od["scale"])
    dense = dense_from_occupancy(od)
    sparse_ind = torch.nonzero(dense)  # [n,3], lexicographic in (x,y,z) like the reference
    up_times = 2 ** (int(train_level) - level)
    if up_times < 1:
        raise ValueError("train_level below the octree level")
    k = torch.arange(0, up_times, device=dev)
    up_kernel = torch.stack(torch.meshgrid(k, k, k, indexing="ij"), dim=-1).reshape(-1, 3)
    ind_up = (sparse_ind.repeat_interleave(up_times ** 3, dim=0) * up_times
              + up_kernel.repeat([sparse_ind.shape[0], 1]))
    train_voxel_size = 2 / (2 ** int(train_level)) * octree_scale
    vol_origin = octree_origin - octree_scale
    xyz_sfm = ind_up * train_voxel_size + vol_origin          # int64 * python float -> float32, then + float32
    scene_origin = renderer.origin.float().to(dev).reshape(3)
    xyz_training = (xyz_sfm - scene_origin) / renderer.radius
    sdf = _sdf_sharded(renderer, xyz_training.contiguous(), int(chunk), group, force_collective)
    return xyz_sfm[sdf <= threshold], train_voxel_size


@torch.no_grad()
def quantise_points(points_sfm, voxel_size, scene_origin, scale):
    """generate_voxel.py:113-150: normalise into the cube (float64 as numpy does), keep the points STRICTLY
    inside (-1,1)^3, level = floor(log2(2 scale / voxel_size)), kaolin's documented quantisation rule
    floor(clamp(2^level (x+1)/2, 0, 2^level - 1)).  Returns (q [K,3] int64, level).  Runs on any device."""
    dev = points_sfm.device
    origin64 = torch.as_tensor(scene_origin, dtype=torch.float64, device=dev).reshape(3)
    level = int(math.floor(math.log2(2 * float(scale) / float(voxel_size))))
    if not 3 <= level <= 10:
        raise ValueError("octree level %d outside the supported 3..10" % level)
    pn = (points_sfm.double() - origin64) / float(scale)
    inside = (pn > -1).all(-1) & (pn < 1).all(-1)
    res = 2 ** level
    q = torch.floor(torch.clamp(res * (pn[inside] + 1.0) / 2.0, 0, res - 1.0)).long()
    return q, level


@torch.no_grad()
def octree_from_points(points_sfm, voxel_size, scene_origin, scale):
    """generate_voxel.py:75-171 `gen_octree(expand=False)` for points already in SfM space (quantise_points).
    The kaolin octree/SPC tensors are replaced by the dense bit mask the ray kernel reads."""
    dev = points_sfm.device
    origin64 = torch.as_tensor(scene_origin, dtype=torch.float64, device=dev).reshape(3)
    q, level = quantise_points(points_sfm, voxel_size, scene_origin, scale)
    res = 2 ** level
    centres = ((q.float() + 0.5) * (2.0 / res) - 1.0)          # voxel centres quantise back to q exactly
    data = occupancy_from_points(centres * float(scale) + origin64.float(), origin64.float(), float(scale), level,
                                 voxel_size)
    return data


@torch.no_grad()
def octree_update(renderer, train_level, threshold, chunk=1 << 22, group=None):
    """neuconw_system.py:266-312: rebuild renderer.fine_octree_data from the current SDF."""
    renderer.fine_octree_data = None
    pts, train_voxel_size = surface_selection(renderer, train_level, threshold, chunk, group)
    od = renderer.octree_data
    data = octree_from_points(pts, train_voxel_size, od["scene_origin"], od["scale"])
    data["voxel_size"] = train_voxel_size
    renderer.fine_octree_data = data
    return data
