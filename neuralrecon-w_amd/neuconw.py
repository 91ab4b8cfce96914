"""Host-side mirror of the reference's models/neuconw.py (NeuconW, SDFNetwork, RenderingNetwork,
SingleVarianceNetwork): same constructor arguments, same `state_dict` keys, same method names --
but every forward/backward runs in the hand-written gfx950 kernels of libneuconw_hip.so.
These nn.Modules only OWN the parameters (so optimisers, DDP and checkpoints see ordinary
nn.Parameters); they contain no torch compute on the hot path.
"""
import math

import numpy as np
import torch
from torch import nn

from . import lib as L
from .packing import PackPlan
from .stash import StashArena, WgradBatch


def points_struct(x=None, rays_o=None, rays_d=None, z=None, sample_dist=None, mode=0):
    """Build the NcwPoints host struct (keeps the tensors alive on the returned object)."""
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


def default_prec():
    import os

    v = os.environ.get("NEUCONW_PREC", "bf16").lower()
    return L.PREC_F32 if v in ("f32", "fp32", "0") else L.PREC_BF16


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

    def plan(self, prec):
        dev = self.lin0.bias.device
        key = (prec, str(dev))
        p = self._plans.get(key)
        if p is not None:
            return p
        RB, W, E = self.d_hidden // 32, self.d_hidden, self.d_enc
        Lm = self.n_lin
        skip = self.skip_in[0] if self.skip_in else -1
        plan = PackPlan(dev, prec)
        net = L.NcwSdfNet()
        slots = {}
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
        plan.finalize()
        for l in range(Lm):
            s = slots[l]
            net.w[l], net.b[l], net.wt[l] = plan.mat_ptr(s[0]), plan.bias_ptr(s[1]), plan.mat_ptr(s[2])
        s = slots[Lm - 1]
        net.w_feat, net.b_feat, net.wt_feat = plan.mat_ptr(s[4]), plan.bias_ptr(s[5]), plan.mat_ptr(s[6])
        net.n_layers, net.skip_layer, net.rb, net.multires, net.scale = Lm, skip, RB, self.multires, self.scale
        plan.net = net
        plan.slots = slots
        plan.packed_version = None
        self._plans[key] = plan
        return plan

    def _param_version(self):
        return tuple(p._version for p in self.parameters())

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
        calls it under no_grad: renderer.py:825, neuconw_system.py:245-249, visualization.py:75-80)."""
        prec = default_prec() if prec is None else prec
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
    def fwd_stash(self, pts, n, prec):
        """ncw_sdf_fwd: returns (sdf [n], grad [n,3], ctx).  ctx carries the activation stash."""
        dev = self.lin0.bias.device
        plan = self.packed(prec)
        RB, Lm = self.d_hidden // 32, self.n_lin
        ar = StashArena(dev, prec, n)
        ids = dict(gamma=ar.new(2), feat=ar.new(RB), dfeat=ar.new(RB), zsdf=ar.new(1), one=ar.new(1))
        ids["h"] = {l: ar.new(RB) for l in range(1, Lm)}
        ids["s"] = {l: ar.new(RB) for l in range(Lm - 1)}
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
        sdf = torch.empty(n, device=dev, dtype=torch.float32)
        grad = torch.empty(n, 3, device=dev, dtype=torch.float32)
        L.check(L.get_lib().ncw_sdf_fwd(plan.net, prec, pts, n, L.ptr(sdf), L.ptr(grad), st, L.stream_ptr(dev)),
                "ncw_sdf_fwd")
        ctx = dict(arena=ar, ids=ids, stash=st, pts=pts, n=n, prec=prec, plan=plan)
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
