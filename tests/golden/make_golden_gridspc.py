"""Golden vector for the sparse evaluation grid of the mesh extraction (tools/extract_mesh.py:60-102 `gen_grid_spc`), produced
by RUNNING the reference's own function.

tools/extract_mesh.py imports PyTorch-Lightning at module scope (not installable), so the function is taken from the
reference's source file as text (ast) and executed with the two names it calls into kaolin-land replaced:
`gen_octree_from_sfm` returns the (origin, scale, level) of tests/golden/sfm_octree.npz -- themselves outputs of the reference's
own gen_octree_from_sfm (make_golden_sfm.py) -- and `convert_to_dense` returns the dense occupancy built from that golden's
quantised points with kaolin's documented rule (the one seam).  Everything else -- nonzero, repeat_interleave up-sampling, the
index -> SfM arithmetic with its mixed numpy-float64 / torch-float32 dtypes -- is the reference's code.
Run:  python tests/golden/make_golden_gridspc.py
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

EVAL_LEVEL_UP = 2  # eval_level = octree level + 2


def reference_function():
    src = open(os.path.join(ref_import.REFERENCE_ROOT, "tools", "extract_mesh.py")).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "gen_grid_spc"][0]
    return ast.get_source_segment(src, fn)


def dense_from_golden(z):
    level = int(z["level"])
    res = 2 ** level
    pn = torch.from_numpy(z["points_filtered"])  # float64, strictly inside (-1,1)^3
    q = torch.floor(torch.clamp(res * (pn + 1.0) / 2.0, 0, res - 1.0)).long()  # kaolin quantize_points
    dense = torch.zeros(res, res, res)
    dense[q[:, 0], q[:, 1], q[:, 2]] = 1.0
    return dense, level


def main():
    z = np.load(os.path.join(HERE, "sfm_octree.npz"))
    dense, level = dense_from_golden(z)
    ns = {"torch": torch, "np": np, "print": lambda *a, **k: None}
    ns["gen_octree_from_sfm"] = lambda data_path, mtl, vs, device=0: (None, np.asarray(z["scene_origin"]), z["scale"][()], level)
    ns["convert_to_dense"] = lambda octree, lvl: dense
    exec(reference_function(), ns)
    eval_level = level + EVAL_LEVEL_UP
    out = ns["gen_grid_spc"]({"min_track_length": int(z["min_track_length"]), "voxel_size": float(z["voxel_size"])},
                             "unused", eval_level, device="cpu")
    sv = out["sparse_vol"]
    path = os.path.join(HERE, "grid_spc.npz")
    np.savez_compressed(path, sparse_vol=sv.numpy(), sparse_vol_dtype=str(sv.dtype), voxel_size=np.float64(out["voxel_size"]),
                        dim=int(out["dim"]), vol_origin=np.asarray(out["vol_origin"]), eval_level=eval_level,
                        dense=dense.numpy().astype(np.uint8), level=level)
    print("wrote", path, sv.shape, sv.dtype, out["dim"], out["voxel_size"], os.path.getsize(path))


if __name__ == "__main__":
    main()
