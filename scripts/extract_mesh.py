"""Mesh extraction driver: the reference's `tools/extract_mesh.py:24-168` command line without PyTorch-Lightning,
kaolin, skimage or trimesh -- one process per GPU.

    python scripts/extract_mesh.py --cfg_path config/train_brandenburg_gate.yaml --ckpt_path ckpts/exp/last.ckpt \
        --mesh_size 512 --mesh_origin "0, 0, 0" --mesh_radius 1.0 [--vertex_color] [--eval_level 10]
    (N GPUs: python -m torch.distributed.run --nproc-per-node N scripts/extract_mesh.py ...)

  * experiment yaml + scene config.yaml + the three state_dicts of the checkpoint exactly as the reference loads them
    (`load_ckpt` x3, :128-131);
  * --eval_level <= 0: the dense [mesh_size]^3 lattice over mesh_origin +- mesh_radius (training coordinates), generated on chip,
    each rank sweeping its contiguous slice with the fused SDF kernel, ONE all_gather (grid.sdf_grid);
    --eval_level > 0: the sparse grid of `gen_grid_spc` (:60-102) -- the sub-voxels of the occupied voxels of the training
    octree (COLMAP points, voxel.octree_from_sfm) at that level; the SDF only there, cubes with all 8 corners evaluated;
  * marching cubes, vertex welding, the vertex-colour pass (`renderer.rgb` viewed along +z with appearance code 1123, :148) and the
    binary PLY on the GPU of rank 0 (mesh.py); file name and directory follow :112-114,160-167.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd import config as C  # noqa: E402
from neuralrecon_w_amd import mesh, trainer, voxel  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg_path", required=True, help="config path")
    ap.add_argument("--root_dir", default=None, help="overrides DATASET.ROOT_DIR")
    ap.add_argument("--dataset_name", default="phototourism", choices=["phototourism"])
    ap.add_argument("--eval_level", type=int, default=-1, help="octree level of the sparse grid (<= 0: dense lattice)")
    ap.add_argument("--mesh_size", type=int, default=128, help="resolution of the dense lattice (N, N, N)")
    ap.add_argument("--mesh_origin", default="0, 0, 0", help="origin of the lattice in training coordinates, x, y, z")
    ap.add_argument("--mesh_radius", type=float, default=1.0)
    ap.add_argument("--vertex_color", action="store_true", help="colour the vertices with the radiance network")
    ap.add_argument("--chunk", type=int, default=1 << 22, help="points per SDF launch of the sparse mode")
    ap.add_argument("--chunk_rgb", type=int, default=1 << 16, help="vertices per launch of the colour pass")
    ap.add_argument("--ckpt_path", required=True, help="checkpoint in the reference's layout (trainer.save_checkpoint / PL)")
    ap.add_argument("--out_dir", default=None, help="default: results/<dataset_name>/<ckpt dir>_<ckpt name>/mesh")
    ap.add_argument("--prec", default=None, choices=["bf16", "f16", "f32"], help="default: fp32 for the colour pass, SDFNetwork.value_prec() for the SDF lattice")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("NCW_TRAIN_ONE_GPU_TEST"):  # plumbing test: the ranks share GPU 0 (collectives over gloo)
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("NCW_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    cfg = C.load_config(args.cfg_path, {"DATASET": {"ROOT_DIR": args.root_dir}} if args.root_dir else None)
    emb, neuconw, nerf, rdr, scene = C.build_system(cfg, dev)
    trainer.load_checkpoint(args.ckpt_path, emb, neuconw, nerf)
    for m in (emb, neuconw, nerf):
        m.eval()
    if args.prec:
        rdr.infer_prec = {"bf16": nw.PREC_BF16, "f16": nw.PREC_F16, "f32": nw.PREC_F32}[args.prec]
    origin = [float(c.strip()) for c in args.mesh_origin.split(",")]
    sparse = None
    if args.eval_level > 0:
        # the training octree straight from the scene's COLMAP points and config.yaml, as tools/extract_mesh.py:60-67 builds it
        od = voxel.octree_from_sfm(cfg["DATASET"]["ROOT_DIR"], scene["min_track_length"], scene["voxel_size"], dev)
        sparse = mesh.gen_grid_spc(od, args.eval_level)
        if rank == 0:
            print("evaluation level %d: dim %d, %d sparse points" % (args.eval_level, sparse["dim"], sparse["sparse_vol"].shape[0]))
    a = emb(torch.ones(1, device=dev, dtype=torch.long) * 1123) if args.vertex_color else None  # tools/extract_mesh.py:148
    m = mesh.extract_mesh(rdr, args.mesh_size, scene["radius"], scene["origin"], origin=origin, radius=args.mesh_radius,
                          with_color=args.vertex_color, embedding_a=a, chunk_rgb=args.chunk_rgb, sparse_data=sparse,
                          chunk=args.chunk)
    if rank == 0:
        save_name = "_".join(os.path.normpath(args.ckpt_path).split(os.sep)[-2:]).replace(".ckpt", "")
        out_dir = args.out_dir or os.path.join("results", args.dataset_name, save_name, "mesh")
        os.makedirs(out_dir, exist_ok=True)
        colored = "_colored" if args.vertex_color else ""
        name = ("extracted_mesh_level_%d%s.ply" % (args.eval_level, colored) if args.eval_level > 0 else
                "extracted_mesh_res_%d_radius_%s%s.ply" % (args.mesh_size, args.mesh_radius, colored))
        path = os.path.join(out_dir, name)
        mesh.write_ply(path, m["vertices"], m["faces"], m["colors"])
        print("%d vertices, %d faces -> %s" % (m["vertices"].shape[0], m["faces"].shape[0], path))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
