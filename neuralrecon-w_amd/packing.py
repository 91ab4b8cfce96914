"""Host-side weight-pack plans: which nn.Linear goes where in the MFMA-fragment-ordered arenas.

A plan owns (per precision)
  * a zero-initialised weight arena (packed matrices, bf16 or f32) and a bias arena (f32),
  * the device descriptor table consumed by ncw_pack_weights (ONE launch re-packs every layer),
  * a dense f32 gradient arena + descriptor table for ncw_unpack_grads (weight-norm backward).
Padding is written once (zeros) and never touched again.
"""

import torch

from . import lib as L


def _elem_size(prec):
    return 4 if prec == L.PREC_F32 else 2


class PackPlan:
    def __init__(self, device, prec):
        self.device = torch.device(device)
        self.prec = prec
        self._w_bytes = 0
        self._b_elems = 0
        self._g_elems = 0
        self._mats = []   # (byte offset, rb_out, rb_in)
        self._biases = []  # (elem offset, rb_out)
        self._dense = []  # (elem offset, rows, cols)  dense gradient matrices (forward orientation)
        self._dense_b = []
        self._pack = []   # python-side descriptor dicts
        self._unpack = []
        self.finalized = False

    # ---- allocation -----------------------------------------------------------------------
    def new_matrix(self, rb_out, rb_in):
        off = self._w_bytes
        self._w_bytes += rb_out * rb_in * 1024 * _elem_size(self.prec)
        self._w_bytes = (self._w_bytes + 255) & ~255
        self._mats.append((off, rb_out, rb_in))
        return len(self._mats) - 1

    def new_bias(self, rb_out):
        off = self._b_elems
        self._b_elems += rb_out * 32
        self._biases.append((off, rb_out))
        return len(self._biases) - 1

    def new_dense_grad(self, rb_out, rb_in, with_bias=True):
        off = self._g_elems
        self._g_elems += rb_out * 32 * rb_in * 32
        self._dense.append((off, rb_out * 32, rb_in * 32))
        boff = None
        if with_bias:
            boff = self._g_elems
            self._g_elems += rb_out * 32
        self._dense_b.append(boff)
        return len(self._dense) - 1

    # ---- descriptors ----------------------------------------------------------------------
    def add_pack(self, weight, g, bias, mat, bias_slot, segs, row0=0, nrows=None, drow0=0, transpose=False,
                 scale=1.0, residual=False):
        """weight: [out, in] parameter (weight_v when g is given).  segs: [(col0, ncols, dcol0)].
        residual: store h16(w - h16(w)) -- the low half of a split (hi + lo) 16-bit matrix."""
        nrows = weight.shape[0] - row0 if nrows is None else nrows
        self._pack.append(dict(weight=weight, g=g, bias=bias, mat=mat, bias_slot=bias_slot, segs=list(segs),
                               row0=row0, nrows=nrows, drow0=drow0, transpose=bool(transpose), scale=float(scale),
                               residual=bool(residual)))

    def add_unpack(self, weight, g, bias, dense, segs, row0=0, nrows=None, drow0=0, scale=1.0):
        nrows = weight.shape[0] - row0 if nrows is None else nrows
        self._unpack.append(dict(weight=weight, g=g, bias=bias, dense=dense, segs=list(segs), row0=row0,
                                 nrows=nrows, drow0=drow0, scale=float(scale)))

    # ---- finalisation ---------------------------------------------------------------------
    def finalize(self):
        dev = self.device
        self.w_arena = torch.zeros(max(self._w_bytes, 256), dtype=torch.uint8, device=dev)
        self.b_arena = torch.zeros(max(self._b_elems, 32), dtype=torch.float32, device=dev)
        self.g_arena = torch.zeros(max(self._g_elems, 32), dtype=torch.float32, device=dev)
        self.finalized = True
        self._build_pack_table()
        return self

    def mat_ptr(self, mat):
        return self.w_arena.data_ptr() + self._mats[mat][0]

    def bias_ptr(self, slot):
        return self.b_arena.data_ptr() + 4 * self._biases[slot][0]

    def dense_ptr(self, d):
        return self.g_arena.data_ptr() + 4 * self._dense[d][0]

    def dense_bias_ptr(self, d):
        off = self._dense_b[d]
        return 0 if off is None else self.g_arena.data_ptr() + 4 * off

    def dense_view(self, d):
        off, r, c = self._dense[d]
        return self.g_arena[off:off + r * c].view(r, c)

    def dense_ld(self, d):
        return self._dense[d][2]

    def param_key(self):
        return tuple(p["weight"].data_ptr() for p in self._pack)

    def _table(self, structs):
        n = len(structs)
        arr = (type(structs[0]) * n)(*structs)
        raw = bytes(arr)
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        return t

    def _build_pack_table(self):
        descs, prefix = [], [0]
        for p in self._pack:
            d = L.NcwPackDesc()
            w = p["weight"]
            assert w.dtype == torch.float32 and w.is_contiguous() and w.device == self.device, "fp32 contiguous params on the plan's device required"
            d.src = w.data_ptr()
            d.g = p["g"].data_ptr() if p["g"] is not None else 0
            d.bias = p["bias"].data_ptr() if (p["bias"] is not None and p["bias_slot"] is not None) else 0
            off, rb_out, rb_in = self._mats[p["mat"]]
            d.dst_w = self.w_arena.data_ptr() + off
            d.dst_b = self.bias_ptr(p["bias_slot"]) if p["bias_slot"] is not None else 0
            d.ld = w.shape[1]
            d.row0, d.nrows, d.drow0 = p["row0"], p["nrows"], p["drow0"]
            d.rb_out, d.rb_in = rb_out, rb_in
            d.transpose = 1 if p["transpose"] else 0
            d.prec = self.prec
            d.scale = p["scale"]
            d.residual = 1 if p.get("residual") else 0
            d.nseg = len(p["segs"])
            assert d.nseg <= L.MAX_SEGS
            for i, (c0, nc, dc0) in enumerate(p["segs"]):
                d.seg[i].col0, d.seg[i].ncols, d.seg[i].dcol0 = c0, nc, dc0
                lim_in = 32 * (rb_out if p["transpose"] else rb_in)
                assert dc0 + nc <= lim_in, "segment exceeds packed matrix"
            lim_out = 32 * (rb_in if p["transpose"] else rb_out)
            assert p["drow0"] + p["nrows"] <= lim_out, "rows exceed packed matrix"
            descs.append(d)
            prefix.append(prefix[-1] + p["nrows"])
        self._pack_tab = self._table(descs)
        self._pack_prefix = torch.tensor(prefix, dtype=torch.int32, device=self.device)
        self._pack_rows = prefix[-1]
        self._pack_n = len(descs)
        self._key = self.param_key()

    def pack(self):
        """(Re)pack all weights on the current stream."""
        if self.param_key() != self._key:  # parameter storage moved (e.g. .to(), load): rebuild
            self._build_pack_table()
        lib = L.get_lib()
        L.check(lib.ncw_pack_weights(L.ptr(self._pack_tab), L.ptr(self._pack_prefix), self._pack_n,
                                     self._pack_rows, L.stream_ptr(self.device)), "ncw_pack_weights")

    def unpack_grads(self, accumulate_into, accumulate=False, grad_mul=1.0, grad_mul_dev=None, launch=True):
        """launch=False: only build / look up the device table -> (tab, prefix, n, rows) (unpack_many launches several plans' tables at once).
        accumulate_into: dict id(param) -> grad tensor (same shape, fp32, contiguous).  Writes (or,
        with accumulate=True, adds) the parameter gradients from the dense gradient arena.  The device
        descriptor table is cached by content, so the steady state does no host->device copy.
        grad_mul (host float) and grad_mul_dev (1-element device tensor, read at run time) multiply everything written
        (1 / loss scale in the fp16 mode)."""
        gm_ptr = grad_mul_dev.data_ptr() if grad_mul_dev is not None else 0
        key = (bool(accumulate), float(grad_mul), gm_ptr) + tuple(accumulate_into[id(u["weight"])].data_ptr() for u in self._unpack) \
            + tuple(u["weight"].data_ptr() for u in self._unpack)
        cache = self.__dict__.setdefault("_unpack_cache", {})
        hit = cache.get(key)
        if hit is not None:
            tab, pre, n, rows = hit
            if not launch:
                return hit
            L.check(L.get_lib().ncw_unpack_grads(L.ptr(tab), L.ptr(pre), n, rows, L.stream_ptr(self.device)),
                    "ncw_unpack_grads")
            return tab, pre
        descs, prefix = [], [0]
        for u in self._unpack:
            d = L.NcwUnpackDesc()
            w = u["weight"]
            d.dw = self.dense_ptr(u["dense"])
            d.db = self.dense_bias_ptr(u["dense"]) if u["bias"] is not None else 0
            d.src = w.data_ptr()
            d.g = u["g"].data_ptr() if u["g"] is not None else 0
            d.d_src = accumulate_into[id(w)].data_ptr()
            d.d_g = accumulate_into[id(u["g"])].data_ptr() if u["g"] is not None else 0
            d.d_bias = accumulate_into[id(u["bias"])].data_ptr() if u["bias"] is not None else 0
            d.ld, d.ldw = w.shape[1], self.dense_ld(u["dense"])
            d.row0, d.nrows, d.drow0 = u["row0"], u["nrows"], u["drow0"]
            d.scale = u["scale"]
            d.grad_mul = float(grad_mul)
            d.grad_mul_dev = gm_ptr
            d.accumulate = 1 if accumulate else 0
            d.nseg = len(u["segs"])
            for i, (c0, nc, dc0) in enumerate(u["segs"]):
                d.seg[i].col0, d.seg[i].ncols, d.seg[i].dcol0 = c0, nc, dc0
            descs.append(d)
            prefix.append(prefix[-1] + u["nrows"])
        tab = self._table(descs)
        pre = torch.tensor(prefix, dtype=torch.int32, device=self.device)
        if len(cache) > 8:
            cache.clear()
        cache[key] = (tab, pre, len(descs), prefix[-1])
        if not launch:
            return cache[key]
        lib = L.get_lib()
        L.check(lib.ncw_unpack_grads(L.ptr(tab), L.ptr(pre), len(descs), prefix[-1], L.stream_ptr(self.device)),
                "ncw_unpack_grads")
        return tab, pre  # keep alive until the stream has consumed them


# ---- several plans in ONE launch ------------------------------------------------------------------------------------
# The descriptor tables hold absolute device pointers, so the tables of several plans concatenate into one launch: the three
# networks of a train step (SDF, colour, background NeRF) re-pack in one ncw_pack_weights and run their weight-norm backward in
# one ncw_unpack_grads instead of three each (launch-latency-sized kernels, back to back on one stream).
_MERGED = {}


def _merge_tables(parts):
    """parts: [(tab uint8, prefix int32 [n + 1], n, rows)] -> one (tab, prefix, n, rows); cached by the parts' storage."""
    key = tuple((t.data_ptr(), pre.data_ptr(), n, rows) for t, pre, n, rows in parts)
    hit = _MERGED.get(key)
    if hit is None:
        tab = torch.cat([t for t, _, _, _ in parts])
        pres, off = [], 0
        for i, (_, pre, n, rows) in enumerate(parts):
            pres.append((pre if i == 0 else pre[1:]) + off)
            off += rows
        prefix = torch.cat(pres).to(torch.int32).contiguous()
        if len(_MERGED) > 16:
            _MERGED.clear()
        hit = _MERGED[key] = (tab, prefix, sum(p[2] for p in parts), off, parts)  # (parts kept: their addresses are the key)
    return hit[:4]


def pack_many(owners):
    """owners: [(module with _param_version(), its PackPlan)].  Re-packs every STALE plan in one launch on the current stream and
    marks it fresh (what module.packed(prec) does one plan at a time).  Fewer than two stale plans: nothing (packed() does it)."""
    stale = [(m, p) for m, p in owners if p.packed_version != (m._param_version(), p.param_key())]
    if len(stale) < 2:
        return
    dev = stale[0][1].device
    if any(p.device != dev for _, p in stale):
        return
    for _, p in stale:
        if p.param_key() != p._key:
            p._build_pack_table()
    tab, prefix, n, rows = _merge_tables([(p._pack_tab, p._pack_prefix, p._pack_n, p._pack_rows) for _, p in stale])
    L.check(L.get_lib().ncw_pack_weights(L.ptr(tab), L.ptr(prefix), n, rows, L.stream_ptr(dev)), "ncw_pack_weights")
    for m, p in stale:
        p.packed_version = (m._param_version(), p.param_key())


def unpack_many(plans, accumulate_into, accumulate=False, grad_mul=1.0, grad_mul_dev=None):
    """PackPlan.unpack_grads of several plans in one launch -> objects to keep alive until the stream has consumed them."""
    parts = [pl.unpack_grads(accumulate_into, accumulate=accumulate, grad_mul=grad_mul, grad_mul_dev=grad_mul_dev, launch=False)
             for pl in plans]
    if len(parts) == 1:
        tab, prefix, n, rows = parts[0]
    else:
        tab, prefix, n, rows = _merge_tables(parts)
    L.check(L.get_lib().ncw_unpack_grads(L.ptr(tab), L.ptr(prefix), n, rows, L.stream_ptr(plans[0].device)), "ncw_unpack_grads")
    return tab, prefix, parts
