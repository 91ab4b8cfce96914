"""Golden vector for the ray-cache batch assembly, produced by RUNNING the reference's own code on a
small synthetic cache written here (tests/golden/raycache_scene/cache/splits/split_{0,1,2}/{rays,rgbs}1.npz):

  * `PhototourismDataset.__getitem__` (datasets/phototourism.py:709-726) on a bare instance whose buffers are filled
    by the reference's own npz-loading lines' equivalent (np.load(...)["arr_0"], torch.cat) -- per row, then stacked like
    the DataLoader's default collate;
  * the black-list filter of `NeuconWSystem.training_step` (neuconw_system.py:345-353) with the reference's
    `get_label_id_mapping()` and the shipped RAY_MASK_LIST;
  * `DataModule._get_local_split` (datasets/data.py:83-101) for world sizes 1, 2, 4, 8.
Run:  python tests/golden/make_golden_raycache.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

SCENE = os.path.join(HERE, "raycache_scene")
RAY_MASK_LIST = ["person", "car", "bicycle", "minibike"]  # config/train_brandenburg_gate.yaml


def write_cache():
    rng = np.random.RandomState(3)
    labels = np.array([0, 1, 2, 4, 6, 12, 20, 116, 127, 149], dtype=np.float32)
    for i, n in enumerate((40, 33, 27)):
        rays = rng.randn(n, 13).astype(np.float32)
        rays[:, 8] = rng.randint(0, 1500, n)            # ts
        rays[:, 9] = labels[rng.randint(0, len(labels), n)]
        rgbs = rng.rand(n, 3).astype(np.float32)
        d = os.path.join(SCENE, "cache", "splits", "split_%d" % i)
        os.makedirs(d, exist_ok=True)
        np.savez_compressed(os.path.join(d, "rays1.npz"), rays)
        np.savez_compressed(os.path.join(d, "rgbs1.npz"), rgbs)


def main():
    write_cache()
    ref_import.load()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    try:
        import importlib

        pt = importlib.import_module("datasets.phototourism")
        data = importlib.import_module("datasets.data")
        mu = importlib.import_module("datasets.mask_utils")
    finally:
        sys.path.remove(ref_import.REFERENCE_ROOT)
    names = ["split_0", "split_1", "split_2"]
    all_rays = torch.cat([torch.from_numpy(np.load(os.path.join(SCENE, "cache", "splits", n, "rays1.npz"))["arr_0"]) for n in names])
    all_rgbs = torch.cat([torch.from_numpy(np.load(os.path.join(SCENE, "cache", "splits", n, "rgbs1.npz"))["arr_0"]) for n in names])
    ds = object.__new__(pt.PhototourismDataset)  # bare instance: only the fields __getitem__ reads for split "train"
    ds.split, ds.with_semantics, ds.all_rays, ds.all_rgbs = "train", True, all_rays, all_rgbs
    idx = np.random.RandomState(5).permutation(all_rays.shape[0])[:64]
    samples = [ds[int(i)] for i in idx]
    batch = {k: torch.stack([s[k] for s in samples]) for k in samples[0]}  # default collate
    # neuconw_system.py:337-353
    ts, label = batch["ts"], batch["semantics"]
    ray_mask = torch.ones_like(ts, dtype=torch.bool)
    for label_name in RAY_MASK_LIST:
        ray_mask[mu.get_label_id_mapping()[label_name] == label] = False
    out = dict(idx=idx, rays=batch["rays"].numpy(), ts=batch["ts"].numpy(), semantics=batch["semantics"].numpy(),
               rgbs=batch["rgbs"].numpy(), ray_mask=ray_mask.numpy(),
               mask_ids=np.array([mu.get_label_id_mapping()[n] for n in RAY_MASK_LIST]))
    dm = object.__new__(data.DataModule)
    items = ["split_%d" % i for i in range(10)]
    for world in (1, 2, 4, 8):
        for rank in range(world):
            out["split_w%d_r%d" % (world, rank)] = np.array(dm._get_local_split(items, world, rank))
    path = os.path.join(HERE, "raycache.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, batch["rays"].shape, int(ray_mask.sum()), os.path.getsize(path))


if __name__ == "__main__":
    main()
