"""Host-side mirror of the reference's models/neuconw.py (NeuconW, SDFNetwork, RenderingNetwork,
SingleVarianceNetwork): same constructor arguments, same `state_dict` keys, same method names --
but every forward/backward runs in the hand-written gfx950 kernels of libneuconw_hip.so.
These nn.Modules only OWN the parameters (so optimisers, DDP and checkpoints see ordinary
nn.Parameters); they contain no torch compute on the hot path.
"""
import ctypes
import math
import os

import numpy as np
import torch
from torch import nn

from . import lib as L
from .packing import PackPlan
from .stash import StashArena, StashCache


def points_struct(x=None, rays_o=None, rays_d=None, z=None, sample_dist=None, mode=0, idx=None, count=None):
    """Build the NcwPoints host struct (keeps the tensors alive on the returned object).  mode 4: a device-made selection
    (idx int32 [n], count int32 [1]) of the mode-2 ray samples (rayops.bg_select)."""
    p = L.NcwPoints()
    keep = []

    def _p(t):
        if t is None:
            return 0
        t = t.contiguous().float()
        keep.append(t)
        return t.data_ptr()

    p.x, p.rays_o, p.rays_d, p.z, p.sample_dist = _p(x), _p(rays_o), _p(rays_d), _p(z), _p(sample_dist)
    p.per_ray = int(z.shape[1]) if z is not None else 1
    p.mode = mode
    if mode == 4:
        assert idx.dtype == torch.int32 and count.dtype == torch.int32 and idx.is_contiguous()
        keep += [idx, count]
        p.idx, p.count = idx.data_ptr(), count.data_ptr()
    p._keep = keep
    return p


class WNLinear(nn.Module):
    """Parameter holder with the state_dict layout of nn.utils.weight_norm(nn.Linear)
    (keys bias, weight_g [out,1], weight_v [out,in]; models/neuconw.py:104-105,256-257)."""

    def __init__(self, lin: nn.Linear):
        super().__init__()
        self.in_features, self.out_features = lin.in_features, lin.out_features
        w = lin.weight.detach()
        self.bias = nn.Parameter(lin.bias.detach().clone())
        self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(w.clone())


class PlainLinear(nn.Module):
    def __init__(self, lin: nn.Linear):
        super().__init__()
        self.in_features, self.out_features = lin.in_features, lin.out_features
        self.weight = nn.Parameter(lin.weight.detach().clone())
        self.bias = nn.Parameter(lin.bias.detach().clone())


def _wvb(m):
    """(weight-or-v, g-or-None, bias) of a WNLinear / PlainLinear / nn.Linear."""
    if hasattr(m, "weight_v"):
        return m.weight_v, m.weight_g, m.bias
    return m.weight, None, m.bias


def _prec_of(v):
    v = v.lower()
    if v in ("f32", "fp32", "0"):
        return L.PREC_F32
    return L.PREC_F16 if v in ("f16", "fp16", "half", "2") else L.PREC_BF16


def default_prec():
    """Precision of the TRAINING passes (the three MLPs forward/backward, the sampler's SDF queries): NEUCONW_PREC =
    f16 (default) | bf16 | f32.  fp16 and bf16 run the same kernels (csrc/ncw_common.h: one source, compiled per 16-bit
    type); in fp16 the SDF VALUE chain additionally runs in split precision at W = 256 / 512 (SDFNetwork.split_value: fp16
    hi + lo operand pairs, fp32-like SDF values), which is what keeps the rendered outputs within 1e-4 of the fp64 oracle
    where NeuS trains (inv_s in the hundreds; bf16: 1e-2 .. 6e-2, tests/test_gpu_fullsize.py), at the price of fp16's
    range: the backward runs under a dynamic loss scale (renderer.loss_scale, adapted by trainer.FlatAdam) and the
    optimiser skips a step whose gradient norm is not finite.  Everything in the path is bounded well inside +-65504
    (points in the unit sphere, weight-normed layers, f32 outputs and accumulators); bf16 stays available for scenes
    where that is in doubt, f32 is the bitwise reproducible parity mode."""
    import os

    return _prec_of(os.environ.get("NEUCONW_PREC", "f16"))


def default_infer_prec():
    """Precision of the geometry-critical inference-only entry points (`gradient()`, `NeuconW.forward` / `renderer.rgb`, and
    `sdf()` / grid.sdf_grid / voxel.surface_selection / mesh.extract_mesh at widths without a split-precision value path): fp32
    like the reference evaluates them, unless NEUCONW_INFER_PREC says otherwise.  bf16 SDF values carry ~5e-3 absolute error
    (tests/test_gpu_sdf.py), which moves a zero level set / an `sdf <= threshold` selection by a fraction of a 512^3 voxel
    (~4e-3).  SDF VALUES alone default to `SDFNetwork.value_prec()`."""
    import os

    return _prec_of(os.environ.get("NEUCONW_INFER_PREC", "f32"))


class SDFNetwork(nn.Module):
    """models/neuconw.py:183-296.  d_in must be 3, multires 6 (the only shipped encoding)."""

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=6, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        if d_in != 3 or multires != 6:
            raise NotImplementedError("HIP SDF kernels are specialised for d_in=3, multires=6")
        if d_hidden % 32 != 0 or d_hidden // 32 not in (2, 8, 16) or d_out != d_hidden + 1:
            raise NotImplementedError("HIP SDF kernels support d_hidden in {64,256,512}, d_out=d_hidden+1")
        if not weight_norm:
            raise NotImplementedError("reference always uses weight_norm=True")
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        dims[0] = 3 + 3 * 2 * multires
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in) if skip_in is not None else ()
        if len(self.skip_in) > 1 or any(s <= 0 or s >= self.num_layers - 2 for s in self.skip_in):
            raise NotImplementedError("at most one skip connection into a hidden layer is supported")
        self.scale = float(scale)
        self.multires = multires
        self.d_hidden = d_hidden
        self.d_enc = dims[0]
        # same construction order / RNG consumption as the reference (neuconw.py:213-259)
        for l in range(0, self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                if l == self.num_layers - 2:
                    sign = -1.0 if inside_outside else 1.0
                    torch.nn.init.normal_(lin.weight, mean=sign * np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -sign * bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            setattr(self, "lin" + str(l), WNLinear(lin))
        self._plans = {}

    # ---- pack plan --------------------------------------------------------------------------
    @property
    def n_lin(self):
        return self.num_layers - 1

    def split_value(self, prec):
        """Split-precision VALUE path (NcwSdfNet.w_lo): fp16 mode at W = 256 / 512 evaluates sdf with hi + lo pairs of fp16
        weights and activations (three MFMAs per product) -- fp32-like SDF values, which sigmoid(sdf * inv_s) needs once
        inv_s is in the hundreds.  NEUCONW_SDF_SPLIT=0 (or `self.sdf_split = False`) switches it off."""
        import os

        on = self.__dict__.get("sdf_split")
        if on is None:
            on = os.environ.get("NEUCONW_SDF_SPLIT", "1") not in ("0", "")
        return bool(on) and prec == L.PREC_F16 and self.d_hidden in (256, 512) and self.n_lin >= 3

    def value_prec(self):
        """Default precision of the VALUE-ONLY evaluations (`sdf()`, the grid sweep, the octree refresh, the mesh lattice):
        at W = 256 / 512 the split-precision fp16 chain -- SDF values 5-9e-7 of the fp64 oracle, the same as the exact-fp32
        kernels (tests/test_gpu_sdf.py), 4-6x faster (512^3 at W = 512: 1.35 s against 5.9 s, itself 53 % of the f32 MFMA peak) -- otherwise fp32.
        NEUCONW_INFER_PREC (f32 / f16 / bf16) overrides."""
        import os

        env = os.environ.get("NEUCONW_INFER_PREC")
        if env:
            return _prec_of(env)
        return L.PREC_F16 if self.split_value(L.PREC_F16) else L.PREC_F32

    def plan(self, prec):
        dev = self.lin0.bias.device
        split = self.split_value(prec)
        key = (prec, str(dev), split, self.__dict__.get("adj_split"))
        p = self._plans.get(key)
        if p is not None:
            return p
        RB, W, E = self.d_hidden // 32, self.d_hidden, self.d_enc
        Lm = self.n_lin
        skip = self.skip_in[0] if self.skip_in else -1
        plan = PackPlan(dev, prec)
        net = L.NcwSdfNet()
        slots, lo, lo_t = {}, {}, {}
        # the adjoint sweep's transposed residuals (NcwSdfNet.wt_lo; csrc/ncw_split.hip sdf_fwdSA, ncw_sdf16.hip sdf_fwdS16<., ADJ>):
        # `.adj_split` / NEUCONW_SDF_ADJ_SPLIT = 0 | False: single-rounded operands in the adjoint sweep (round 4's kernels);
        # 1 | True: W^T as hi + lo pairs, t_l single fp16 (round 5); 2: t_l as a pair too (round 6, W = 512 only).
        # Default: 1 at W = 256, 2 at W = 512 -- at the shipped 8 + 16 shape ONE sample carries most of a ray's weight and the
        # colour network reads that sample's normal: with W^T alone as a pair the timed batch's worst ray on trained weights went
        # 3.2e-4 -> 6.2e-4 (profiles/r05/emul_timed_batch_shipped.log), with both operands as pairs AND the colour network's
        # activations as pairs the worst of 256 rays is at 2e-5 (profiles/r06/emul_timed_batch_shipped_tangent*.log).
        adj = self.__dict__.get("adj_split")
        if adj is None:
            env = os.environ.get("NEUCONW_SDF_ADJ_SPLIT")
            adj = (2 if RB == 16 else 1) if env is None else (int(env) if env.isdigit() else 1)
        adj = int(adj) if (split and RB in (8, 16)) else 0
        if RB == 8:
            adj = min(adj, 1)
        for l in range(Lm):
            v, g, b = _wvb(getattr(self, "lin%d" % l))
            n_out, n_in = v.shape
            if l == 0:
                rb_in, segs, scale = 2, [(0, E, 0)], 1.0
            elif l == skip:
                rb_in, segs, scale = RB + 2, [(0, W - E, 0), (W - E, E, 32 * RB)], 1.0 / math.sqrt(2.0)
            else:
                rb_in, segs, scale = RB, [(0, n_in, 0)], 1.0
            if l < Lm - 1:
                m, bs = plan.new_matrix(RB, rb_in), plan.new_bias(RB)
                mt = plan.new_matrix(rb_in, RB)
                dn = plan.new_dense_grad(RB, rb_in)
                plan.add_pack(v, g, b, m, bs, segs, scale=scale)
                plan.add_pack(v, g, None, mt, None, segs, transpose=True, scale=scale)
                plan.add_unpack(v, g, b, dn, segs, scale=scale)
                slots[l] = (m, bs, mt, dn)
                if split:
                    lo[l] = plan.new_matrix(RB, rb_in)
                    plan.add_pack(v, g, None, lo[l], None, segs, scale=scale, residual=True)
                if adj:
                    lo_t[l] = plan.new_matrix(rb_in, RB)
                    plan.add_pack(v, g, None, lo_t[l], None, segs, transpose=True, scale=scale, residual=True)
            else:  # last Linear: row 0 = sdf, rows 1..W = feature vector (neuconw.py:279)
                m, bs = plan.new_matrix(1, RB), plan.new_bias(1)
                mt = plan.new_matrix(RB, 1)
                mf, bf = plan.new_matrix(RB, RB), plan.new_bias(RB)
                mft = plan.new_matrix(RB, RB)
                dn = plan.new_dense_grad(1, RB)
                dnf = plan.new_dense_grad(RB, RB)
                plan.add_pack(v, g, b, m, bs, segs, row0=0, nrows=1)
                plan.add_pack(v, g, None, mt, None, segs, row0=0, nrows=1, transpose=True)
                plan.add_pack(v, g, b, mf, bf, segs, row0=1, nrows=W)
                plan.add_pack(v, g, None, mft, None, segs, row0=1, nrows=W, transpose=True)
                plan.add_unpack(v, g, b, dn, segs, row0=0, nrows=1)
                plan.add_unpack(v, g, b, dnf, segs, row0=1, nrows=W)
                slots[l] = (m, bs, mt, dn, mf, bf, mft, dnf)
                if split:
                    lo[l] = plan.new_matrix(1, RB)
                    plan.add_pack(v, g, None, lo[l], None, segs, row0=0, nrows=1, residual=True)
                if adj:
                    lo_t[l] = plan.new_matrix(RB, 1)
                    plan.add_pack(v, g, None, lo_t[l], None, segs, row0=0, nrows=1, transpose=True, residual=True)
        plan.finalize()
        for l, m_lo in lo.items():
            net.w_lo[l] = plan.mat_ptr(m_lo)
        for l, m_lo in lo_t.items():
            net.wt_lo[l] = plan.mat_ptr(m_lo)
        for l in range(Lm):
            s = slots[l]
            net.w[l], net.b[l], net.wt[l] = plan.mat_ptr(s[0]), plan.bias_ptr(s[1]), plan.mat_ptr(s[2])
        s = slots[Lm - 1]
        net.w_feat, net.b_feat, net.wt_feat = plan.mat_ptr(s[4]), plan.bias_ptr(s[5]), plan.mat_ptr(s[6])
        net.n_layers, net.skip_layer, net.rb, net.multires, net.scale = Lm, skip, RB, self.multires, self.scale
        net.adj_mode = adj
        plan.net = net
        plan.slots = slots
        plan.packed_version = None
        self._plans[key] = plan
        return plan

    def _param_version(self):
        # _ncw_version_srcs: base tensors whose in-place updates change these parameters without touching their
        # own version counters (trainer.FlatParams re-seats p.data into one flat buffer)
        return tuple(p._version for p in self.parameters()) + \
            tuple(t._version for t in self.__dict__.get("_ncw_version_srcs", ()))

    def packed(self, prec):
        """Pack plan with weights up to date on the current stream."""
        plan = self.plan(prec)
        ver = (self._param_version(), plan.param_key())
        if plan.packed_version != ver:
            plan.pack()
            plan.packed_version = (self._param_version(), plan.param_key())
        return plan

    # ---- reference API ----------------------------------------------------------------------
    @torch.no_grad()
    def sdf(self, x, prec=None):
        """SDFNetwork.sdf (neuconw.py:281-282): x[..., 3] -> [N, 1]; no autograd (the reference only
        calls it under no_grad: renderer.py:825, neuconw_system.py:245-249, visualization.py:75-80).  prec None = value_prec()."""
        prec = self.value_prec() if prec is None else prec
        if not x.is_cuda:
            raise L.NeuconwHipError("SDFNetwork.sdf: input is not on a GPU; the hot path has no CPU fallback")
        xf = x.reshape(-1, 3).float().contiguous()
        out = torch.empty(xf.shape[0], device=x.device, dtype=torch.float32)
        plan = self.packed(prec)
        lib = L.get_lib()
        L.check(lib.ncw_sdf_infer(plan.net, prec, L.ptr(xf), xf.shape[0], L.ptr(out), L.stream_ptr(x.device)),
                "ncw_sdf_infer")
        return out.reshape(-1, 1)

    # ---- training path: forward with input gradient, backward, weight gradients ---------------------
    def fwd_stash(self, pts, n, prec, train=True):
        """ncw_sdf_fwd: returns (sdf [n], grad [n,3], ctx).  ctx carries the activation stash.
        train=False: the forward-only render (validation / novel views / vertex colours: rendering/renderer.py:785-916 under
        no_grad, :951-961): the same outputs bit for bit, but the arena holds only what the launch itself re-reads -- h_l, the
        scratch of the analytic adjoint sweep -- and `feat`, the colour network's input: (L - 1 + 1) x W x 2 B per point instead
        of the training stash's 21 KB (NcwSdfStash.t[0] == NULL selects the kernels that store nothing else)."""
        dev = self.lin0.bias.device
        plan = self.packed(prec)
        RB, Lm = self.d_hidden // 32, self.n_lin
        h_lo = int(plan.net.adj_mode) == 2  # the adjoint sweep takes phi' from h as an fp16 hi + lo pair (csrc/ncw_sdf16.hip)

        def build_render():
            ar = StashArena(dev, prec, n)
            ids = dict(feat=ar.new(RB))
            ids["h"] = {l: ar.new(RB) for l in range(1, Lm)}
            ids["s"] = {l: ar.new(RB) for l in range(1, Lm)} if h_lo else {}
            ar.allocate()
            st = L.NcwSdfStash()
            st.feat = ar.ptr(ids["feat"])
            for l, i in ids["h"].items():
                st.h[l] = ar.ptr(i)
            for l, i in ids["s"].items():
                st.s[l] = ar.ptr(i)
            return dict(arena=ar, ids=ids, stash=st)

        def build():
            ar = StashArena(dev, prec, n)
            ids = dict(gamma=ar.new(2), feat=ar.new(RB), dfeat=ar.new(RB), zsdf=ar.new(1), one=ar.new(1))
            ids["h"] = {l: ar.new(RB) for l in range(1, Lm)}
            # Softplus' is recomputed from h (s = 1 - exp(-100 h)): no stash vector of its own.  adj_mode 2 (W = 512): the slots hold the fp16
            # RESIDUALS of h (NcwSdfStash.s = h_lo), written by the split value chain and read by the adjoint sweep of the same launch
            ids["s"] = {l: ar.new(RB) for l in range(1, Lm)} if h_lo else {}
            ids["t"] = {l: ar.new(RB) for l in range(Lm - 1)}
            ids["qbar"] = {l: ar.new(2 if l == 0 else RB) for l in range(Lm)}
            ids["zbar"] = {l: ar.new(RB) for l in range(Lm - 1)}
            ar.allocate()
            st = L.NcwSdfStash()
            st.gamma, st.feat, st.dfeat = ar.ptr(ids["gamma"]), ar.ptr(ids["feat"]), ar.ptr(ids["dfeat"])
            st.zsdf, st.one = ar.ptr(ids["zsdf"]), ar.ptr(ids["one"])
            for k in ("h", "s", "t", "qbar", "zbar"):
                for l, i in ids[k].items():
                    getattr(st, k)[l] = ar.ptr(i)
            return dict(arena=ar, ids=ids, stash=st)

        ent = self.__dict__.setdefault("_stash_cache", StashCache()).acquire((prec, n, str(dev), h_lo, bool(train)),
                                                                             build if train else build_render)
        ar, ids, st = ent["arena"], ent["ids"], ent["stash"]
        sdf = torch.empty(n, device=dev, dtype=torch.float32)
        grad = torch.empty(n, 3, device=dev, dtype=torch.float32)
        L.check(L.get_lib().ncw_sdf_fwd(plan.net, prec, pts, n, L.ptr(sdf), L.ptr(grad), st, L.stream_ptr(dev)),
                "ncw_sdf_fwd")
        ctx = dict(arena=ar, ids=ids, stash=st, pts=pts, n=n, prec=prec, plan=plan, lease=ent)
        return sdf, grad, ctx

    def bwd_stash(self, ctx, d_sdf, d_grad):
        """ncw_sdf_bwd: consumes d_sdf [n], d_grad [n,3] and ctx's dfeat stash."""
        dev = self.lin0.bias.device
        d_sdf = d_sdf.contiguous().float()
        d_grad = d_grad.contiguous().float()
        L.check(L.get_lib().ncw_sdf_bwd(ctx["plan"].net, ctx["prec"], ctx["pts"], ctx["n"], L.ptr(d_sdf),
                                        L.ptr(d_grad), ctx["stash"], L.stream_ptr(dev)), "ncw_sdf_bwd")
        ctx["_keep_bwd"] = (d_sdf, d_grad)

    def add_wgrads(self, ctx, batch):
        """Queue every weight-gradient product of the SDF net (forward + adjoint terms) on `batch`."""
        plan, ar, ids = ctx["plan"], ctx["arena"], ctx["ids"]
        RB, Lm = self.d_hidden // 32, self.n_lin
        skip = self.skip_in[0] if self.skip_in else -1
        P = ar.ptr
        for l in range(Lm - 1):
            dn = plan.slots[l][3]
            ld = plan.dense_ld(dn)
            y_f, rby = (P(ids["gamma"]), 2) if l == 0 else (P(ids["h"][l]), RB)
            batch.add(P(ids["zbar"][l]), RB, y_f, rby, plan.dense_ptr(dn), ld, plan.dense_bias_ptr(dn))
            batch.add(P(ids["t"][l]), RB, P(ids["qbar"][l]), rby, plan.dense_ptr(dn), ld)
            if l == skip:
                off = 4 * 32 * RB
                batch.add(P(ids["zbar"][l]), RB, P(ids["gamma"]), 2, plan.dense_ptr(dn) + off, ld)
                batch.add(P(ids["t"][l]), RB, P(ids["qbar"][0]), 2, plan.dense_ptr(dn) + off, ld)
        s = plan.slots[Lm - 1]
        dn, dnf = s[3], s[7]
        hl = P(ids["h"][Lm - 1])
        batch.add(P(ids["dfeat"]), RB, hl, RB, plan.dense_ptr(dnf), plan.dense_ld(dnf), plan.dense_bias_ptr(dnf))
        batch.add(P(ids["zsdf"]), 1, hl, RB, plan.dense_ptr(dn), plan.dense_ld(dn), plan.dense_bias_ptr(dn))
        batch.add(P(ids["one"]), 1, P(ids["qbar"][Lm - 1]), RB, plan.dense_ptr(dn), plan.dense_ld(dn))


class _PackedNet(nn.Module):
    """Shared plan/pack caching for the parameter-holder modules."""

    def _init_plans(self):
        self._plans = {}

    def _plan_switches(self):
        """Mutable attributes `_build_plan` reads (part of the plan cache key)."""
        return ()

    def _param_version(self):
        # _ncw_version_srcs: base tensors whose in-place updates change these parameters without touching their
        # own version counters (trainer.FlatParams re-seats p.data into one flat buffer)
        return tuple(p._version for p in self.parameters()) + \
            tuple(t._version for t in self.__dict__.get("_ncw_version_srcs", ()))

    def _first_param(self):
        return next(self.parameters())

    def plan(self, prec):
        dev = self._first_param().device
        # switches read when the plan is BUILT (residual matrices present or not) belong to the key: flipping `.refine` /
        # `.weight_split` after the first forward selects (or builds) the matching plan instead of being a silent no-op
        key = (prec, str(dev)) + tuple(self._plan_switches())
        p = self._plans.get(key)
        if p is None:
            p = self._build_plan(prec, dev)
            p.packed_version = None
            self._plans[key] = p
        return p

    def packed(self, prec):
        plan = self.plan(prec)
        ver = (self._param_version(), plan.param_key())
        if plan.packed_version != ver:
            plan.pack()
            plan.packed_version = (self._param_version(), plan.param_key())
        return plan


class RenderingNetwork(_PackedNet):
    """models/neuconw.py:59-170, `encode_apperence=True`, mode "idr" (the only shipped configuration)."""

    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, head_channels=128, in_channels_dir_a=48,
                 static_head_layers=2, weight_norm=True, multires_view=4, squeeze_out=True, encode_apperence=True):
        super().__init__()
        if mode != "idr" or not encode_apperence or d_in != 9 or d_out != 3 or multires_view != 4 or not weight_norm \
                or not squeeze_out:
            raise NotImplementedError("HIP colour kernels implement mode='idr', encode_apperence=True, d_in=9, d_out=3")
        if (d_feature // 32, head_channels // 32, d_hidden // 32) not in ((2, 1, 2), (2, 4, 8), (8, 4, 8), (16, 4, 8)) \
                or d_feature % 32 or head_channels % 32 or d_hidden % 32:
            raise NotImplementedError("unsupported colour-net widths %s" % ((d_feature, head_channels, d_hidden),))
        if in_channels_dir_a > 69 or static_head_layers > 4 or n_layers > 7:
            raise NotImplementedError("n_a <= 69, static_head_layers <= 4, n_layers <= 7")
        self.d_feature, self.head_channels, self.d_hidden = d_feature, head_channels, d_hidden
        self.n_a, self.n_head, self.mode = in_channels_dir_a, static_head_layers, mode
        dims = [d_in + head_channels - 3] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):  # same order / RNG use as the reference (:99-107)
            setattr(self, "lin" + str(l), WNLinear(nn.Linear(dims[l], dims[l + 1])))
        from collections import OrderedDict

        od = OrderedDict()
        od["static_linear_0"] = PlainLinear(nn.Linear(d_feature + in_channels_dir_a + 27, head_channels))
        for i in range(1, static_head_layers):
            od["static_linear_%d" % i] = PlainLinear(nn.Linear(head_channels, head_channels))
        self.static_encoding = nn.Sequential(od)
        self.xyz_encoding_final = PlainLinear(nn.Linear(d_feature, d_feature))
        # 16-bit modes: per-ray fp32 evaluation of the head's view-direction / appearance-code columns (fwd_stash);
        # NEUCONW_COLOR_RAY_BIAS=0 / .ray_bias = False = those columns as 16-bit MFMA operands like the rest
        self.ray_bias = os.environ.get("NEUCONW_COLOR_RAY_BIAS", "1") != "0"
        # fp16 mode: the forward evaluates every Linear of this network with its weights as fp16 hi + lo pairs (two MFMAs per
        # product: W_hi x + W_lo x); NEUCONW_COLOR_WSPLIT=0 / .weight_split = False = one rounding per weight (set before the
        # first forward: the packed-weight plan is built once per precision)
        self.weight_split = os.environ.get("NEUCONW_COLOR_WSPLIT", "1") != "0"
        # fp16 mode, with weight_split: the ACTIVATIONS of every layer as hi + lo pairs too (NcwColorNet.act_split: the W_hi pass of the
        # weight ring feeds every fragment to two MFMAs; forward only).  None = the default: ON at d_feature = 256 and 512.  At 512 (the
        # shipped width: 8 + 16 samples per ray, one sample carries a ray) it is half of what brings the trained-weights colour from
        # 3.1e-4 to 2e-5; at 256 (headline) ten ray batches measured, same box, alternating: worst trained-weights colour 8.4e-5 -> 5.2e-5,
        # mean 5.2e-5 -> 3.4e-5, for +0.03 ms per step (4.136 -> 4.167 ms; profiles/r06/bench_seed*_asplit{0,1}.json).
        # NEUCONW_COLOR_ASPLIT=0 / 1 or `.act_split = False / True` override.
        env = os.environ.get("NEUCONW_COLOR_ASPLIT")
        self.act_split = None if env is None else (env not in ("0", ""))
        self._init_plans()

    def _plan_switches(self):
        return (bool(self.weight_split), self.act_split)

    @property
    def n_lin(self):
        return self.num_layers - 1

    def _build_plan(self, prec, dev):
        RBF, RBH, RBC = self.d_feature // 32, self.head_channels // 32, self.d_hidden // 32
        W, HC, A = self.d_feature, self.head_channels, self.n_a
        plan = PackPlan(dev, prec)
        net = L.NcwColorNet()
        sl = {}

        split = prec == L.PREC_F16 and self.weight_split  # forward matrices as fp16 hi + lo pairs (ncw_color_fwd)

        def full(name, mod, rb_out, rb_in, segs):
            v, g, b = _wvb(mod)
            m, bs, mt = plan.new_matrix(rb_out, rb_in), plan.new_bias(rb_out), plan.new_matrix(rb_in, rb_out)
            dn = plan.new_dense_grad(rb_out, rb_in)
            plan.add_pack(v, g, b, m, bs, segs)
            plan.add_pack(v, g, None, mt, None, segs, transpose=True)
            plan.add_unpack(v, g, b, dn, segs)
            lo = None
            if split:
                lo = plan.new_matrix(rb_out, rb_in)
                plan.add_pack(v, g, None, lo, None, segs, residual=True)
            sl[name] = (m, bs, mt, dn, lo)

        full("f", self.xyz_encoding_final, RBF, RBF, [(0, W, 0)])
        full("e0", self.static_encoding[0], RBH, RBF + 3, [(0, W, 0), (W, 27 + A, 32 * RBF)])
        for i in range(1, self.n_head):
            full("e%d" % i, self.static_encoding[i], RBH, RBH, [(0, HC, 0)])
        full("l0", self.lin0, RBC, RBH + 1, [(6, HC, 0), (0, 6, 32 * RBH)])
        for l in range(1, self.n_lin - 1):
            full("l%d" % l, getattr(self, "lin%d" % l), RBC, RBC, [(0, self.d_hidden, 0)])
        full("l%d" % (self.n_lin - 1), getattr(self, "lin%d" % (self.n_lin - 1)), 1, RBC, [(0, self.d_hidden, 0)])
        plan.finalize()
        net.w_f, net.b_f, net.wt_f = plan.mat_ptr(sl["f"][0]), plan.bias_ptr(sl["f"][1]), plan.mat_ptr(sl["f"][2])
        net.w_f_lo = plan.mat_ptr(sl["f"][4]) if split else None
        for i in range(self.n_head):
            s = sl["e%d" % i]
            net.w_e[i], net.b_e[i], net.wt_e[i] = plan.mat_ptr(s[0]), plan.bias_ptr(s[1]), plan.mat_ptr(s[2])
            net.w_e_lo[i] = plan.mat_ptr(s[4]) if split else None
        for l in range(self.n_lin):
            s = sl["l%d" % l]
            net.w_l[l], net.b_l[l], net.wt_l[l] = plan.mat_ptr(s[0]), plan.bias_ptr(s[1]), plan.mat_ptr(s[2])
            net.w_l_lo[l] = plan.mat_ptr(s[4]) if split else None
        net.n_head, net.n_lin, net.rbf, net.rbh, net.rbc, net.n_a = self.n_head, self.n_lin, RBF, RBH, RBC, A
        asplit = True if self.act_split is None else bool(self.act_split)
        net.act_split = 1 if (split and asplit and (RBF, RBH, RBC) in ((8, 4, 8), (16, 4, 8))) else 0
        plan.net, plan.slots = net, sl
        return plan

    def fwd_stash(self, pts, n, prec, normals, a, feat_ptr, train=True):
        """train=False: the forward-only render -- NOTHING is stashed (NcwColorStash.aux1 == NULL selects color_render_kernel);
        the same rgb bit for bit."""
        dev = self._first_param().device
        plan = self.packed(prec)
        RBF, RBH, RBC = self.d_feature // 32, self.head_channels // 32, self.d_hidden // 32

        def build_render():
            return dict(arena=StashArena(dev, prec, n).allocate(), ids={}, stash=L.NcwColorStash())

        def build():
            ar = StashArena(dev, prec, n)
            ids = dict(aux1=ar.new(3), aux2=ar.new(1), f=ar.new(RBF), zf=ar.new(RBF), zo=ar.new(1))
            ids["e"] = [ar.new(RBH) for _ in range(self.n_head)]
            ids["ze"] = [ar.new(RBH) for _ in range(self.n_head)]
            ids["x"] = [ar.new(RBC) for _ in range(self.n_lin - 1)]
            ids["zx"] = [ar.new(RBC) for _ in range(self.n_lin - 1)]
            ar.allocate()
            st = L.NcwColorStash()
            for k in ("aux1", "aux2", "f", "zf", "zo"):
                setattr(st, k, ar.ptr(ids[k]))
            for k in ("e", "ze", "x", "zx"):
                for i, v in enumerate(ids[k]):
                    getattr(st, k)[i] = ar.ptr(v)
            return dict(arena=ar, ids=ids, stash=st)

        ent = self.__dict__.setdefault("_stash_cache", StashCache()).acquire((prec, n, str(dev), bool(train)),
                                                                             build if train else build_render)
        ar, ids, st = ent["arena"], ent["ids"], ent["stash"]
        rgb = torch.empty(n, 3, device=dev, dtype=torch.float32)
        normals = normals.contiguous().float()
        a = a.contiguous().float()
        # 16-bit modes: the per-RAY part of the head's first layer -- W_e0[:, view-dir | appearance columns] . [gamma_4(d) | a]
        # (models/neuconw.py:131-140) -- is evaluated once per ray in fp32 (ncw_aux_ray_bias) and added to that layer's bias;
        # rounding `a` and gamma(d) to 16 bits is coherent along a ray and was the largest term of the fp16 mode's colour
        # error on trained weights (scripts/diag/emul_color16.py).  The backward / weight gradients are unchanged.
        st.aux_bias = None
        lin0 = self.static_encoding[0]
        if prec != L.PREC_F32 and self.ray_bias and pts.rays_d and not hasattr(lin0, "weight_v"):
            R, no = a.shape[0], 32 * RBH
            ab = ent.get("aux_bias")
            if ab is None or ab.shape[0] != R:
                ab = ent["aux_bias"] = torch.empty(R, no, device=dev, dtype=torch.float32)
            w0 = lin0.weight.detach()
            assert w0.is_contiguous() and w0.dtype == torch.float32 and w0.shape == (self.head_channels, self.d_feature + 27 + self.n_a)
            L.check(L.get_lib().ncw_aux_ray_bias(L.ptr(w0), w0.shape[1], self.d_feature, self.head_channels,
                                                 ctypes.c_void_p(pts.rays_d), L.ptr(a), self.n_a, R, L.ptr(ab), no, L.stream_ptr(dev)),
                    "ncw_aux_ray_bias")
            st.aux_bias = ab.data_ptr()
        L.check(L.get_lib().ncw_color_fwd(plan.net, prec, pts, n, L.ptr(normals), L.ptr(a), feat_ptr, L.ptr(rgb), st,
                                          L.stream_ptr(dev)), "ncw_color_fwd")
        return rgb, dict(arena=ar, ids=ids, stash=st, pts=pts, n=n, prec=prec, plan=plan, rgb=rgb, feat_ptr=feat_ptr,
                         keep=(normals, a), lease=ent)

    def bwd_stash(self, ctx, d_rgb, d_grad, d_a, dfeat_ptr, d_a_rows=None):
        """d_grad [n,3] is updated in place (+= d normals); d_a [R,n_a] accumulates (atomics) -- or, with
        d_a_rows [n,n_a], every point's row is stored instead (the caller reduces them: ncw_ray_sum_rows)."""
        dev = self._first_param().device
        d_rgb = d_rgb.contiguous().float()
        assert d_grad.is_contiguous() and (d_a is None or d_a.is_contiguous())
        assert (d_a is None) != (d_a_rows is None), "exactly one of d_a / d_a_rows"
        L.check(L.get_lib().ncw_color_bwd(ctx["plan"].net, ctx["prec"], ctx["pts"], ctx["n"], L.ptr(ctx["rgb"]),
                                          L.ptr(d_rgb), L.ptr(d_grad), L.ptr(d_a), L.ptr(d_a_rows), dfeat_ptr, ctx["stash"],
                                          L.stream_ptr(dev)), "ncw_color_bwd")
        ctx["_keep_bwd"] = d_rgb

    def add_wgrads(self, ctx, batch):
        plan, ar, ids, sl = ctx["plan"], ctx["arena"], ctx["ids"], ctx["plan"].slots
        RBF, RBH, RBC = self.d_feature // 32, self.head_channels // 32, self.d_hidden // 32
        P = ar.ptr

        def dn(name):
            d = sl[name][3]
            return plan.dense_ptr(d), plan.dense_ld(d), plan.dense_bias_ptr(d)

        dp, ld, db = dn("f")
        batch.add(P(ids["zf"]), RBF, ctx["feat_ptr"], RBF, dp, ld, db)
        dp, ld, db = dn("e0")
        batch.add(P(ids["ze"][0]), RBH, P(ids["f"]), RBF, dp, ld, db)
        batch.add(P(ids["ze"][0]), RBH, P(ids["aux1"]), 3, dp + 4 * 32 * RBF, ld)
        for i in range(1, self.n_head):
            dp, ld, db = dn("e%d" % i)
            batch.add(P(ids["ze"][i]), RBH, P(ids["e"][i - 1]), RBH, dp, ld, db)
        dp, ld, db = dn("l0")
        batch.add(P(ids["zx"][0]), RBC, P(ids["e"][self.n_head - 1]), RBH, dp, ld, db)
        batch.add(P(ids["zx"][0]), RBC, P(ids["aux2"]), 1, dp + 4 * 32 * RBH, ld)
        for l in range(1, self.n_lin - 1):
            dp, ld, db = dn("l%d" % l)
            batch.add(P(ids["zx"][l]), RBC, P(ids["x"][l - 1]), RBC, dp, ld, db)
        dp, ld, db = dn("l%d" % (self.n_lin - 1))
        batch.add(P(ids["zo"]), 1, P(ids["x"][self.n_lin - 2]), RBC, dp, ld, db)


class SingleVarianceNetwork(nn.Module):
    """models/neuconw.py:173-179: inv_s = exp(10 * variance)."""

    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))

    def inv_s(self):
        return torch.exp(self.variance * 10.0).clamp(1e-6, 1e6).reshape(1)

    def forward(self, x):
        return torch.ones([len(x), 1], device=x.device) * torch.exp(self.variance * 10.0)


class NeuconW(nn.Module):
    """models/neuconw.py:299-376.  Same ctor, same state_dict keys (incl. the dead
    `xyz_encoding_final = nn.Linear(512, 512)`, :319)."""

    def __init__(self, sdfNet_config, colorNet_config, SNet_config, in_channels_a, encode_a):
        super().__init__()
        self.sdfNet_config, self.colorNet_config, self.SNet_config = sdfNet_config, colorNet_config, SNet_config
        self.in_channels_a, self.encode_a = in_channels_a, encode_a
        self.sdf_net = SDFNetwork(**sdfNet_config)
        self.xyz_encoding_final = nn.Linear(512, 512)  # never used, never receives a gradient (reference :319)
        self.deviation_network = SingleVarianceNetwork(**SNet_config)
        self.color_net = RenderingNetwork(**colorNet_config, in_channels_dir_a=in_channels_a,
                                          encode_apperence=encode_a)

    def sdf(self, input_xyz, prec=None):
        return self.sdf_net.sdf(input_xyz, prec)

    @torch.no_grad()
    def gradient(self, x, prec=None):
        prec = default_infer_prec() if prec is None else prec
        xf = x.reshape(-1, 3).float().contiguous()
        _, grad, c = self.sdf_net.fwd_stash(points_struct(x=xf), xf.shape[0], prec, train=False)
        StashCache.release(c["lease"])
        return grad

    @torch.no_grad()
    def forward(self, x, prec=None):
        """x [R,S,3+3+A] -> (rgb [R,S,3], inv_s [1,1], sdf [R,S], grad [R,S,3]) -- inference
        (renderer.rgb(), visualisation).  The differentiable training path is NeuconWRenderer.render."""
        prec = default_infer_prec() if prec is None else prec
        R, S, _ = x.shape
        n = R * S
        xyz = x[..., 0:3].reshape(n, 3).float().contiguous()
        dirs = x[..., 3:6].reshape(n, 3).float().contiguous()
        a = x[..., 6:].reshape(n, -1).float().contiguous()
        pts = points_struct(x=xyz, rays_d=dirs)
        sdf, grad, sctx = self.sdf_net.fwd_stash(pts, n, prec, train=False)
        rgb, cctx = self.color_net.fwd_stash(pts, n, prec, grad, a, sctx["arena"].ptr(sctx["ids"]["feat"]), train=False)
        StashCache.release(sctx["lease"])
        StashCache.release(cctx["lease"])
        inv_s = self.deviation_network.inv_s().reshape(1, 1)
        return rgb.reshape(R, S, 3), inv_s, sdf.reshape(R, S), grad.reshape(R, S, 3)
