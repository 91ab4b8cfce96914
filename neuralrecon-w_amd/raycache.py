"""Ray-cache reader and on-device batch assembly: the step right before the hot path.

The reference trains from `<root>/<cache_dir>/splits/<name>/{rays,rgbs}<downscale>.npz` chunks written by
tools/prepare_data/prepare_data_cache.py:78-160 (`np.savez_compressed(chunk, array)` -> key `arr_0`;
rays [N,13] = o, d, near, far, ts, semantic label, 3 depth columns; rgbs [N,3]).  Each rank loads its share of the
chunks (datasets/data.py:83-119), keeps them in host memory, and a 16-worker DataLoader gathers rows, collates,
pins and copies every batch; NeuconWSystem.training_step then drops the black-listed labels with a boolean index.

Here the rank's chunks are uploaded to HBM ONCE (a brandenburg_gate cache is tens of GB; 288 GB per MI355X) and a
batch is one row-gather launch (`ncw_batch_assemble`): no loader workers, no per-step H2D copy.  `prefilter=True`
additionally removes the black-listed rays once at load time, so every batch has the full, fixed ray count and the
step stays free of device->host syncs (the reference's per-batch filter yields a data-dependent batch size).
h5 chunks (the cache writer's default) are read through h5py when it is importable; this image has none, and then an h5
cache is refused with a message naming the npz alternative."""
import ctypes as C
import os

import numpy as np
import torch

from . import lib as L
from .labels import label_id


def local_splits(names, world_size, rank, seed=6):
    """datasets/data.py:83-101 `_get_local_split`: permute with RandomState(seed), pad to a multiple of the world size
    with RandomState(seed).choice(..., replace=True), contiguous slice per rank.  Same numpy calls -> same assignment."""
    names = list(names)
    n = len(names)
    perm = np.random.RandomState(seed).permutation(names)
    if n % world_size == 0:
        padded = perm
    else:
        pad = np.random.RandomState(seed).choice(names, world_size - (n % world_size), replace=True)
        padded = np.concatenate([perm, pad])
    per = len(padded) // world_size
    return [str(x) for x in padded[per * rank: per * (rank + 1)]]


def list_splits(root_dir, cache_dir):
    """data.py:104-107: the sub-directories of <root>/<cache_dir>/splits, in os.walk order."""
    return next(os.walk(os.path.join(root_dir, cache_dir, "splits")))[1]


def _read_chunk(stem, cache_type, key):
    """One cache chunk as a numpy array.  npz: `arr_0` (phototourism.py:488-490); h5 (the DEFAULT `--cache_type` of
    tools/prepare_data/prepare_data_cache.py:36-40, datasets `rays` / `rgbs`, phototourism.py:491-495) through h5py when it
    is importable -- this image has none, and then the refusal says how to get a readable cache."""
    if cache_type == "npz":
        return np.load(stem + ".npz")["arr_0"]
    try:
        import h5py
    except ImportError:
        raise NotImplementedError(
            "ray cache %s.h5: h5py is not installed here; re-run the reference's prepare_data_cache.py with "
            "`--cache_type npz` (what config/defaults.py:89 reads) or install h5py" % stem)
    with h5py.File(stem + ".h5", "r") as f:
        return np.asarray(f[key][:])


class RayCache:
    """The rank's training rays, resident on `device`.

    batch(idx) -> dict(rays [B,11], ts [B] int64, semantics [B] int64, rgbs [B,3]) exactly as the reference's
    DataLoader delivers them, + `keep` [B] bool (False for RAY_MASK_LIST labels); filtered(batch) applies the
    reference's boolean index (neuconw_system.py:345-353)."""

    def __init__(self, root_dir, cache_dir, names, device, img_downscale=1, with_semantics=True, ray_mask_list=None,
                 prefilter=False):
        self.device = torch.device(device)
        self.with_semantics = bool(with_semantics)
        split_path = os.path.join(cache_dir, "splits")
        if not names:
            raise ValueError("no cache splits assigned to this rank")
        first = os.path.join(root_dir, split_path, names[0])
        cache_type = os.listdir(first)[0].split(".")[-1]  # phototourism.py:478-481
        if cache_type not in ("npz", "h5"):
            raise NotImplementedError("ray cache type %r: the reference writes npz or h5 chunks" % cache_type)
        rays, rgbs = [], []
        for nme in names:  # phototourism.py:482-513
            rays.append(torch.from_numpy(_read_chunk(os.path.join(root_dir, split_path, nme, "rays%d" % img_downscale), cache_type, "rays")))
            rgbs.append(torch.from_numpy(_read_chunk(os.path.join(root_dir, split_path, nme, "rgbs%d" % img_downscale), cache_type, "rgbs")))
        rays, rgbs = torch.cat(rays, 0), torch.cat(rgbs, 0)
        want = 13 if self.with_semantics else 12
        if rays.shape[1] != want or rgbs.shape[1] != 3 or rays.shape[0] != rgbs.shape[0]:
            raise ValueError("ray cache shape %s / %s: expected [N,%d] / [N,3]" % (tuple(rays.shape), tuple(rgbs.shape), want))
        self.all_rays = rays.float().contiguous().to(self.device)   # the ONE host->device copy
        self.all_rgbs = rgbs.float().contiguous().to(self.device)
        ids = [label_id(n) for n in (ray_mask_list or [])]
        if len(ids) > 4:
            raise NotImplementedError("RAY_MASK_LIST with more than 4 labels")
        self.mask_ids = ids
        self._ids_arr = (C.c_int * 4)(*(ids + [0] * (4 - len(ids))))
        self.prefiltered = False
        if prefilter and ids and self.with_semantics:
            keep = self.batch(None)["keep"]
            self.all_rays = self.all_rays[keep].contiguous()
            self.all_rgbs = self.all_rgbs[keep].contiguous()
            self.prefiltered = True

    def __len__(self):
        return int(self.all_rays.shape[0])

    def batch(self, idx):
        """idx: int64 tensor [B] on the device (None = the whole cache in order)."""
        dev = self.device
        if not self.all_rays.is_cuda:
            raise L.NeuconwHipError("RayCache: the cache is not on a GPU; batch assembly has no CPU fallback")
        n = len(self)
        B = n if idx is None else int(idx.shape[0])
        if idx is not None:
            idx = idx.to(dev, torch.int64).contiguous()
        rays = torch.empty(B, 11, device=dev)
        ts = torch.empty(B, dtype=torch.int64, device=dev)
        label = torch.empty(B, dtype=torch.int64, device=dev)
        rgbs = torch.empty(B, 3, device=dev)
        keep = torch.empty(B, dtype=torch.uint8, device=dev)
        L.check(L.get_lib().ncw_batch_assemble(L.ptr(self.all_rays), self.all_rays.shape[1], L.ptr(self.all_rgbs), L.ptr(idx),
                                               n, B, int(self.with_semantics), L.ptr(rays), L.ptr(ts), L.ptr(label),
                                               L.ptr(rgbs), self._ids_arr, len(self.mask_ids), L.ptr(keep),
                                               L.stream_ptr(dev)), "ncw_batch_assemble")
        out = {"rays": rays, "ts": ts, "rgbs": rgbs, "keep": keep.bool()}
        if self.with_semantics:
            out["semantics"] = label
        return out

    @staticmethod
    def filtered(batch):
        """neuconw_system.py:345-353: rays / ts / rgbs / label of the rays that are not black-listed (dynamic shape)."""
        k = batch["keep"]
        return batch["rays"][k], batch["ts"][k], batch["semantics"][k], batch["rgbs"][k]

    def epoch(self, batch_size, generator=None, drop_last=False, max_batches=None, skip_batches=0):
        """Shuffled batches of one epoch (DataLoader(shuffle=True): a uniform permutation; drawn on the device).
        max_batches: stop after that many (every rank of a data-parallel job must run the same number of steps);
        skip_batches: a resumed run drops the batches the interrupted epoch already consumed (same permutation when the
        generator is restored to its state at the start of that epoch)."""
        n = len(self)
        perm = torch.randperm(n, device=self.device, generator=generator)
        for i, s in enumerate(range(0, n, batch_size)):
            if (drop_last and s + batch_size > n) or (max_batches is not None and i >= max_batches):
                return
            if i < skip_batches:
                continue
            yield self.batch(perm[s:s + batch_size])
