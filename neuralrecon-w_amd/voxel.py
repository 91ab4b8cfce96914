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
    return {"occ": occ, "brick": brick, "scene_origin": origin, "scale": float(scale), "level": int(level),
            "voxel_size": float(voxel_size) if voxel_size is not None else 2.0 * float(scale) / G}


def occupancy_from_dense(dense_bool, scene_origin, scale, voxel_size=None):
    """dense_bool [G,G,G] (x slowest) -> octree_data (tests / synthetic shells of SURVEY 8d config 3)."""
    G = dense_bool.shape[0]
    level = int(round(math.log2(G)))
    idx = dense_bool.nonzero().float()
    centres = (idx + 0.5) * (2.0 / G) - 1.0
    origin = torch.as_tensor(scene_origin, dtype=torch.float32, device=dense_bool.device).reshape(3)
    return occupancy_from_points(centres * float(scale) + origin, scene_origin, scale, level, voxel_size)


def octree_from_sfm(recontruct_path, min_track_length, voxel_size, device):
    raise NotImplementedError(
        "building the coarse octree from a COLMAP reconstruction (generate_voxel.py:41-73) needs the dataset "
        "readers, which are out of the hot-path scope; build octree_data with voxel.occupancy_from_points(...) "
        "from the SfM points and assign it to renderer.octree_data")


def get_near_far(rays_o_sfm, rays_d, octree_data):
    """generate_voxel.py:311-439: (near, far) [R,1] in SfM units; 0 where the ray misses."""
    dev = rays_o_sfm.device
    R = rays_o_sfm.shape[0]
    o = rays_o_sfm.contiguous().float()
    d = rays_d.contiguous().float()
    near = torch.empty(R, device=dev, dtype=torch.float32)
    far = torch.empty(R, device=dev, dtype=torch.float32)
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
