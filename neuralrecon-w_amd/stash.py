"""Activation-stash arenas and batched weight-gradient launches (host side)."""

import os

import torch

from . import lib as L


class StashArena:
    """One device buffer carved into stash vectors ([tile32][rb][4][64][4] elements each)."""

    def __init__(self, device, prec, n_points):
        self.device = torch.device(device)
        self.prec = prec
        self.n = int(n_points)
        self.tiles = (self.n + 31) // 32
        # kernels run whole workgroups (8 waves = 8 tiles; the W = 512 kernels 2, 3 or 4 tiles): pad to a multiple of 24
        # so that surplus waves / tiles store into padding
        self.tiles_alloc = (self.tiles + 23) // 24 * 24
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
    """Collects (X, Y, dense) weight-gradient products and runs them in one launch.

    bf16: every product of every network of the step goes through ONE ncw_wgrad_tiled launch (256 x 256
    workgroup tiles).  The kernel is HBM-bound with one workgroup per CU, so the split-K factor is chosen
    PER PRODUCT such that the launch is ~3 rounds of equal-length workgroups over the 256 CUs
    (scripts/bench_wgrad.py: 5.2 TB/s of stash bytes, against 4.3 TB/s for one launch per network/shape).  f32: the exact kernel, one launch per point count."""

    TARGET_WGS = int(os.environ.get("NCW_WGRAD_TARGET_WGS", "768"))  # (env: tuning hook of scripts/diag/wgrad_rounds.sh)
    SEL_FRACTION = 0.125

    def __init__(self, device, prec, n_points, n_dev=None, sel_fraction=None):
        """n_dev: device int32[1] -- the products cover only the first min(n_points, n_dev[0]) points of their stashes (a
        selection made on the device, NcwPoints mode 4; 16-bit tiled launch only).  sel_fraction: the fraction of n_points the
        selection is EXPECTED to hold (the split-K plan balances the launch with it; the kernel clamps to the real count):
        the renderer passes what it last observed (stash.SelectionProbe), SEL_FRACTION otherwise."""
        self.device = torch.device(device)
        self.prec = prec
        self.n = int(n_points)
        self.n_dev = 0 if n_dev is None else int(n_dev.data_ptr())
        self._keep_dev = n_dev
        self.sel_fraction = float(self.SEL_FRACTION if sel_fraction is None else sel_fraction)
        self.items = []

    def add(self, x_ptr, rbx, y_ptr, rby, dense_ptr, ld, dbias_ptr=0, n=None):
        self.items.append((x_ptr, rbx, y_ptr, rby, dense_ptr, ld, dbias_ptr, self.n if n is None else int(n), self.n_dev,
                           self.sel_fraction if self.n_dev else 1.0))

    def extend(self, other):
        """Take over another batch's products (they keep their own point count)."""
        self.items.extend(other.items)
        self.__dict__.setdefault("_keep_other", []).append(other._keep_dev)

    _scratch = {}  # device -> K-slice slabs of the ordered (fp32) weight-gradient launches
    _cache = {}  # content-addressed device tables: (items, prec, tile) -> (table, prefix, n_desc, wgs, ksplit, n)

    @staticmethod
    def algorithmic_bytes(items, elem=2):
        """Stash bytes the products stream from HBM: every X and Y block once per product."""
        return sum((it[1] + it[3]) * 1024 * elem * ((it[7] + 31) // 32) for it in items)

    def _table(self, items, tile, ksplits, ksplit, n):
        key = (tuple(items), tuple(ksplits), self.prec, tile, ksplit, n, str(self.device))
        hit = WgradBatch._cache.get(key)
        if hit is None:
            xb, yb = (4, 4) if tile is None else ((4, 8) if tile == 0 else (8, 8))
            descs, prefix = [], [0]
            for (x, rbx, y, rby, dense, ld, db, ni, ndev, _frac), ksp in zip(items, ksplits):
                d = L.NcwWgradDesc()
                d.x, d.y, d.dense, d.dbias = x, y, dense, db
                d.rbx, d.rby, d.ld = rbx, rby, ld
                d.ksplit, d.n_points = (ksp, ni) if tile is not None else (0, 0)
                if ndev:
                    d.n_points_dev = ndev
                descs.append(d)
                prefix.append(prefix[-1] + ((rbx + xb - 1) // xb) * ((rby + yb - 1) // yb) * ksp)
            arr = (L.NcwWgradDesc * len(descs))(*descs)
            tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            pre = torch.tensor(prefix, dtype=torch.int32, device=self.device)
            if len(WgradBatch._cache) > 32:
                WgradBatch._cache.clear()
            hit = WgradBatch._cache[key] = (tab, pre, len(descs), prefix[-1], ksplit, n)
        return (tile,) + hit

    def _plan(self):
        items = [it for it in self.items if it[7] > 0]
        if not items:
            return []
        if self.prec != L.PREC_F32:
            # cost of one (product, 256x256 quad): stash blocks streamed x tiles
            def quads(it):
                return ((it[1] + 7) // 8) * ((it[3] + 7) // 8)
            def cost(it):
                # a quad costs the same whatever its real block count (missing blocks are re-reads of a
                # valid one, L2 hits): weighting by HBM bytes instead measured 1.55 ms vs 1.02 ms per step
                tiles = (it[7] + 31) // 32
                # a product sized on the device (dead-background elimination) is planned at the selection's EXPECTED share of
                # its stash (it[9]: observed by the renderer, stash.SelectionProbe): the split only balances the launch, the
                # kernel clamps to the real count.  (Round 5: a fixed 12.5 % under-planned the shipped 8 + 16 shape -- 18 % of
                # its samples are selected -- whose 17 single-workgroup background products then ran 1.5x longer than the rest:
                # weight-gradient launch 0.87 ms dense, 1.09 ms with the elimination; NOTEBOOK R5.4.)
                return max(1, int(tiles * it[9] + 0.999)) if it[8] else tiles
            total = sum(cost(it) * quads(it) for it in items)
            per_wg = max(1.0, total / self.TARGET_WGS)
            ksplits = []
            for it in items:
                tiles = (it[7] + 31) // 32
                ksplits.append(int(max(1, min(tiles // 8 if tiles >= 8 else 1, round(cost(it) / per_wg)))))
            return [self._table(items, 1, ksplits, 1, max(it[7] for it in items))]
        groups = []
        for n in sorted({it[7] for it in items}):
            sub = [it for it in items if it[7] == n]
            ksplit = max(1, min(16, ((n + 31) // 32) // 8))
            groups.append(self._table(sub, None, [ksplit] * len(sub), ksplit, n))
        return groups

    def run(self):
        groups = self.__dict__.get("_groups")
        if groups is None:
            groups = self._groups = self._plan()
        lib = L.get_lib()
        for tile, tab, pre, nd, wgs, ks, n in groups:
            if tile is None:
                # fp32 parity mode: order-fixed split-K (run-to-run reproducible); one scratch per process, grown on demand
                need = int(lib.ncw_wgrad_ordered_scratch_floats(wgs))
                scr = WgradBatch._scratch.get(str(self.device))
                if scr is None or scr.numel() < need:
                    scr = WgradBatch._scratch[str(self.device)] = torch.empty(need, device=self.device, dtype=torch.float32)
                L.check(lib.ncw_wgrad_ordered(L.ptr(tab), L.ptr(pre), nd, wgs, ks, self.prec, n, L.ptr(scr),
                                              L.stream_ptr(self.device)), "ncw_wgrad_ordered")
            else:
                fn = lib.ncw_wgrad_tiled_f16 if self.prec == L.PREC_F16 else lib.ncw_wgrad_tiled
                L.check(fn(L.ptr(tab), L.ptr(pre), nd, wgs, ks, tile, n, L.stream_ptr(self.device)), "ncw_wgrad_tiled")


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


class LeaseGuard:
    """Owns the stash leases of ONE forward.  The autograd node keeps it alive; when the node dies without a
    backward (a grad-enabled validation render, a skipped step) the leases are returned instead of staying busy
    forever -- the next forward would otherwise allocate a fresh multi-GB arena every step.  `consume()` is the
    backward: a second backward over the same stashes (retain_graph=True) would read buffers a later forward may
    already have overwritten, so it raises."""

    def __init__(self, leases):
        self.leases = [l for l in leases if l is not None]
        self.consumed = False

    def consume(self):
        if self.consumed:
            raise RuntimeError("NeuconWRenderer: backward called twice over one render(); the activation stashes are "
                               "released after the first backward (retain_graph=True is not supported)")
        self.consumed = True

    def release(self):
        for l in self.leases:
            StashCache.release(l)
        self.leases = []

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class SelectionProbe:
    """How many samples the dead-background elimination keeps (device int32[1], renderer / nerf.fwd_stash), observed WITHOUT a
    device->host synchronisation: every forward copies the count into one of two pinned words (non-blocking) behind an event;
    a later forward harvests whichever copy has completed.  `fraction` (count / n) is therefore a step or two old -- it only
    steers the split-K plan of the weight-gradient launch, which clamps to the real count on the device."""

    OVERPLAN = float(os.environ.get("NCW_SEL_OVERPLAN", "1.6"))

    def __init__(self):
        self.bufs = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(2)]
        self.evs, self.ns, self.i, self.fraction = [None, None], [0, 0], 0, None

    def observe(self, count_dev, n_points):
        if torch.cuda.is_current_stream_capturing():
            return
        for j in (0, 1):
            if self.evs[j] is not None and self.evs[j].query():
                self.fraction = float(self.bufs[j][0]) / max(1, self.ns[j])
                self.evs[j] = None
        i = self.i
        if self.evs[i] is None:
            self.bufs[i].copy_(count_dev, non_blocking=True)
            self.evs[i] = torch.cuda.Event()
            self.evs[i].record()
            self.ns[i] = int(n_points)
            self.i ^= 1

    def bucket(self, default):
        """(fraction to plan with, its bucket): re-planning happens when the observed share leaves its 1.25x bucket (with hysteresis)."""
        import math

        f = default if self.fraction is None else min(1.0, max(self.fraction, 1e-3))
        # planned share = OVERPLAN x observed: the background products are the LAST of the launch, and workgroups that turn out
        # shorter than planned drain the tail faster than exactly balanced ones (measured at the headline shape, 7.5 % selected:
        # planned at 7.5 % the launch takes 0.80 ms, at 12.5 % 0.69 ms; under-planning -- 12.5 % for the 18 % of the shipped
        # shape -- costs 1.09 against 0.87 ms).  NCW_SEL_OVERPLAN: tuning hook of scripts/r05.
        f = min(1.0, f * self.OVERPLAN)
        bf = math.log(f) / math.log(1.25)
        b = self.__dict__.get("_b")
        if b is None or abs(bf - b) > 0.75:  # hysteresis: a share that hovers around a bucket edge must not re-plan every step
            b = self._b = round(bf)
        return 1.25 ** b, b
