"""Activation-stash arenas and batched weight-gradient launches (host side)."""
import ctypes as C

import torch

from . import lib as L


class StashArena:
    """One device buffer carved into stash vectors ([tile32][rb][4][64][4] elements each)."""

    def __init__(self, device, prec, n_points):
        self.device = torch.device(device)
        self.prec = prec
        self.n = int(n_points)
        self.tiles = (self.n + 31) // 32
        # kernels run whole workgroups (<= 8 waves = 8 tiles): pad so surplus waves store into padding
        self.tiles_alloc = (self.tiles + 7) // 8 * 8
        self.esize = 4 if prec == L.PREC_F32 else 2
        self._off = []
        self._bytes = 0
        self.buf = None

    def new(self, rb):
        off = self._bytes
        self._bytes += self.tiles_alloc * rb * 1024 * self.esize
        self._bytes = (self._bytes + 255) & ~255
        self._off.append((off, rb))
        return len(self._off) - 1

    def allocate(self, zero=False):
        fn = torch.zeros if zero else torch.empty
        self.buf = fn(max(self._bytes, 256), dtype=torch.uint8, device=self.device)
        return self

    def ptr(self, i):
        return self.buf.data_ptr() + self._off[i][0]

    def rb(self, i):
        return self._off[i][1]

    # ---- layout converters (tests / module boundary) -------------------------------------------
    def to_rows(self, i, F):
        out = torch.empty(self.n, F, device=self.device, dtype=torch.float32)
        L.check(L.get_lib().ncw_stash_to_rows(self.prec, self.ptr(i), self.n, F, self.rb(i), L.ptr(out),
                                              L.stream_ptr(self.device)), "ncw_stash_to_rows")
        return out

    def from_rows(self, i, rows):
        rows = rows.contiguous().float()
        assert rows.shape[0] == self.n
        L.check(L.get_lib().ncw_stash_from_rows(self.prec, L.ptr(rows), self.n, rows.shape[1], self.rb(i),
                                                self.ptr(i), L.stream_ptr(self.device)), "ncw_stash_from_rows")


class WgradBatch:
    """Collects (X, Y, dense) weight-gradient products and runs them in one ncw_wgrad launch."""

    def __init__(self, device, prec, n_points):
        self.device = torch.device(device)
        self.prec = prec
        self.n = int(n_points)
        self.items = []

    def add(self, x_ptr, rbx, y_ptr, rby, dense_ptr, ld, dbias_ptr=0):
        self.items.append((x_ptr, rbx, y_ptr, rby, dense_ptr, ld, dbias_ptr))

    _cache = {}  # content-addressed device tables: (items, prec, n, tile) -> (table, prefix, n_desc, wgs, ksplit)

    def _launch_group(self, items, tile, ksplit):
        """tile: None = f32 kernel (128x128), 0 = bf16 128x256, 1 = bf16 256x256."""
        if not items:
            return None
        key = (tuple(items), self.prec, self.n, tile, str(self.device))
        hit = WgradBatch._cache.get(key)
        if hit is None:
            xb, yb = (4, 4) if tile is None else ((4, 8) if tile == 0 else (8, 8))
            descs, prefix = [], [0]
            for (x, rbx, y, rby, dense, ld, db) in items:
                d = L.NcwWgradDesc()
                d.x, d.y, d.dense, d.dbias = x, y, dense, db
                d.rbx, d.rby, d.ld = rbx, rby, ld
                descs.append(d)
                prefix.append(prefix[-1] + ((rbx + xb - 1) // xb) * ((rby + yb - 1) // yb) * ksplit)
            arr = (L.NcwWgradDesc * len(descs))(*descs)
            tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            pre = torch.tensor(prefix, dtype=torch.int32, device=self.device)
            if len(WgradBatch._cache) > 32:
                WgradBatch._cache.clear()
            hit = WgradBatch._cache[key] = (tab, pre, len(descs), prefix[-1], ksplit)
        return (tile,) + hit

    def run(self):
        if not self.items or self.n == 0:
            return
        groups = self.__dict__.get("_groups")
        if groups is None:
            tiles = (self.n + 31) // 32
            chunk_tiles = 2 if self.prec == L.PREC_BF16 else 1
            ksplit = max(1, min(16, tiles // (8 * chunk_tiles)))
            if self.prec == L.PREC_BF16:
                big = [it for it in self.items if it[1] > 4]      # more than 4 X blocks: 256 x 256 tiles
                small = [it for it in self.items if it[1] <= 4]
                groups = [g for g in (self._launch_group(big, 1, ksplit), self._launch_group(small, 0, ksplit)) if g]
            else:
                groups = [self._launch_group(self.items, None, ksplit)]
            self._groups = groups
        lib = L.get_lib()
        for tile, tab, pre, nd, wgs, ks in groups:
            if tile is None:
                L.check(lib.ncw_wgrad(L.ptr(tab), L.ptr(pre), nd, wgs, ks, self.prec, self.n, L.stream_ptr(self.device)),
                        "ncw_wgrad")
            else:
                L.check(lib.ncw_wgrad_tiled(L.ptr(tab), L.ptr(pre), nd, wgs, ks, tile, self.n,
                                            L.stream_ptr(self.device)), "ncw_wgrad_tiled")


class StashCache:
    """Per-module cache of activation-stash arenas keyed by (prec, n, device).  An entry is LEASED:
    while a forward's stash is waiting for its backward it is `busy` and a concurrent forward
    (validation render, inference) gets a fresh arena instead of clobbering it."""

    def __init__(self):
        self._e = {}

    def acquire(self, key, builder):
        e = self._e.get(key)
        if e is not None and not e["busy"]:
            e["busy"] = True
            return e
        new = builder()
        new["busy"] = True
        self._e[key] = new  # keep the most recent one (a lease that is never returned just gets dropped)
        return new

    @staticmethod
    def release(e):
        if e is not None:
            e["busy"] = False
