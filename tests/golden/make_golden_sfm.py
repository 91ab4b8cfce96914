"""Golden vector for the coarse octree construction (NeuconWRenderer.get_octree, renderer.py:137-155), produced by
RUNNING the reference's own `gen_octree_from_sfm` (tools/prepare_data/generate_voxel.py:41-171) on a small synthetic
COLMAP model written here (tests/golden/sfm_scene/{config.yaml,dense/sparse/points3D.bin}).

kaolin is absent, so its two calls are the seam: `spc.points.quantize_points` is replaced by a recorder that captures
the float64 `points_filtered` tensor and the level the reference hands to kaolin, `spc.unbatched_points_to_octree`
returns a dummy.  Everything up to that call -- the binary reader, the track-length filter, the 27-neighbour dilation
+ np.unique, the eval-box transform by inv(sfm2gt), the cube normalisation and strict (-1,1) crop, the level -- is the
reference's own code.   Run:  python tests/golden/make_golden_sfm.py
"""
import os
import struct
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

SCENE = os.path.join(HERE, "sfm_scene")
MIN_TRACK, VOXEL = 3, 0.3


def write_scene():
    rng = np.random.RandomState(11)
    n = 160
    # a noisy sphere shell + a few far outliers (outside the evaluation box after the transform)
    d = rng.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    xyz = d * (1.6 + 0.1 * rng.randn(n, 1)) + np.array([0.3, -0.2, 0.1])
    xyz[:8] *= 6.0
    tracks = rng.randint(1, 9, size=n)  # some at / below MIN_TRACK
    os.makedirs(os.path.join(SCENE, "dense", "sparse"), exist_ok=True)
    with open(os.path.join(SCENE, "dense", "sparse", "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", n))
        for i in range(n):
            f.write(struct.pack("<QdddBBBd", i + 1, *xyz[i], 10, 20, 30, 0.5))
            f.write(struct.pack("<Q", int(tracks[i])))
            for t in range(int(tracks[i])):
                f.write(struct.pack("<ii", t + 1, 7 * t + i))
    # sfm2gt: rotation about z by 20 degrees, scale 1.7, translation
    a = np.deg2rad(20.0)
    Rm = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]]) * 1.7
    T = np.eye(4)
    T[:3, :3] = Rm
    T[:3, 3] = [0.4, -0.1, 0.25]
    cfg = {"origin": [0.3, -0.2, 0.1], "radius": 2.4, "sfm2gt": T.tolist(),
           "eval_bbx": [[-3.2, -3.9, -3.0], [4.1, 3.3, 3.6]], "voxel_size": VOXEL, "min_track_length": MIN_TRACK}
    with open(os.path.join(SCENE, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)


def main():
    write_scene()
    ref_import.load()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    try:
        import importlib

        gv = importlib.import_module("tools.prepare_data.generate_voxel")
    finally:
        sys.path.remove(ref_import.REFERENCE_ROOT)
    cap = {}

    def quantize_points(points, level):
        cap["points_filtered"] = points.detach().cpu().numpy().copy()
        cap["level"] = int(level)
        return points

    gv.spc.points.quantize_points = quantize_points
    gv.spc.unbatched_points_to_octree = lambda q, level: torch.zeros(1, dtype=torch.uint8)
    octree, scene_origin, scale, level = gv.gen_octree_from_sfm(SCENE, MIN_TRACK, VOXEL, device="cpu")
    assert level == cap["level"]
    out = os.path.join(HERE, "sfm_octree.npz")
    np.savez_compressed(out, points_filtered=cap["points_filtered"], level=level, scene_origin=np.asarray(scene_origin),
                        scale=np.float64(scale), min_track_length=MIN_TRACK, voxel_size=np.float64(VOXEL))
    print("wrote", out, cap["points_filtered"].shape, cap["points_filtered"].dtype, level, scale, os.path.getsize(out))


if __name__ == "__main__":
    main()
