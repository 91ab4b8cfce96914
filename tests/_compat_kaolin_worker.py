"""Fresh interpreter with `compat/` ahead on sys.path: the reference's OWN code (tools/prepare_data/generate_voxel.py,
lightning_modules/neuconw_system.py:186-312) runs UNEDITED over the kaolin-named boundary module.  CPU: the structure calls
are torch ops; `Tensor.cuda()` is the identity and `renderer.sdf` an analytic sphere SDF (the seams of
tests/golden/make_golden_octree.py, minus its `convert_to_dense` stub -- that one now runs for real).
Writes an .npz for tests/test_compat_kaolin.py.   usage: python tests/_compat_kaolin_worker.py OUT.npz"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "compat"), ROOT]
import kaolin  # noqa: E402
import kaolin.ops.spc  # noqa: E402,F401
import kaolin.render.spc  # noqa: E402,F401

assert os.path.join("compat", "kaolin") in kaolin.__file__
from oracle import ref_import  # noqa: E402

sysmod, _ = ref_import.load_system()  # (its MagicMock stubs skip names already in sys.modules: kaolin stays ours)
gv = sys.modules["tools.prepare_data.generate_voxel"]
assert gv.spc is kaolin.ops.spc and sysmod.convert_to_dense is gv.convert_to_dense
GOLDEN = os.path.join(HERE, "golden")
SCENE = os.path.join(GOLDEN, "sfm_scene")
out = {}

# 1. gen_octree_from_sfm -> convert_to_dense / octree_to_spc, all the reference's own (generate_voxel.py:41-186)
g = np.load(os.path.join(GOLDEN, "sfm_octree.npz"))
octree, origin, scale, level = gv.gen_octree_from_sfm(SCENE, int(g["min_track_length"]), float(g["voxel_size"]), device="cpu")
out.update(sfm_level=level, sfm_origin=origin, sfm_scale=scale, sfm_dense=gv.convert_to_dense(octree, level).numpy(),
           sfm_octree=octree.numpy())
points, pyramid, prefix = gv.octree_to_spc(octree)
out.update(sfm_points=points.numpy(), sfm_pyramid=pyramid.numpy())

# 2. NeuconWSystem.surface_selection + octree_update on the golden refresh case
z = np.load(os.path.join(GOLDEN, "octree_refresh.npz"))
dense = torch.from_numpy(z["dense"]).float()
lvl, train_level, threshold = int(z["level"]), int(z["train_level"]), float(z["threshold"])
oct0 = kaolin.ops.spc.unbatched_points_to_octree(torch.nonzero(dense > 0).short(), lvl)
renderer = types.SimpleNamespace(
    origin=torch.from_numpy(z["origin"]), radius=float(z["radius"]), recontruct_path=SCENE, fine_octree_data=None,
    sdf=lambda pts: pts.reshape(-1, 3).norm(dim=-1, keepdim=True) - 0.5,
    octree_data={"octree": oct0, "scene_origin": torch.from_numpy(z["octree_origin"]), "scale": float(z["octree_scale"]), "level": lvl})
me = types.SimpleNamespace(renderer=renderer, hparams=types.SimpleNamespace(num_gpus=1))
me.surface_selection = types.MethodType(sysmod.NeuconWSystem.surface_selection, me)
sysmod.get_world_size, sysmod.get_rank = (lambda: 1), (lambda: 0)
orig_cuda = torch.Tensor.cuda
torch.Tensor.cuda = lambda self, *a, **k: self
try:
    pts, tvs = sysmod.NeuconWSystem.surface_selection(me, train_level, threshold, device="cpu", chunk=4096)
    sysmod.NeuconWSystem.octree_update(me, train_level, threshold, device="cpu", chunk=4096)
finally:
    torch.Tensor.cuda = orig_cuda
fine = renderer.fine_octree_data
out.update(sel_pts=np.asarray(pts), sel_voxel=tvs, fine_octree=fine["octree"].numpy(), fine_level=fine["level"],
           fine_scale=fine["scale"], fine_origin=fine["scene_origin"].numpy(), fine_voxel=fine["voxel_size"],
           fine_points=fine["spc_data"]["points"].numpy(), fine_pyramid=fine["spc_data"]["pyramid"].numpy())
# 3. the product reads that dictionary (voxel.ensure_occupancy: bit masks from the SPC)
from neuralrecon_w_amd import voxel  # noqa: E402

voxel.ensure_occupancy(fine)
out.update(fine_occ=fine["occ"].numpy(), fine_brick=fine["brick"].numpy())
np.savez_compressed(sys.argv[1], **out)
print("ok")
