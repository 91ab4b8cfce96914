"""Host-side mirror of the reference's models/nerf.py NeRF (background network): same ctor,
same state_dict keys; compute in libneuconw_hip.so (ncw_nerf_fwd / ncw_nerf_bwd)."""
import ctypes
import os
from collections import OrderedDict

import torch
from torch import nn

from . import lib as L
from .neuconw import _PackedNet, _wvb, default_prec, points_struct
from .packing import PackPlan
from .stash import StashArena, StashCache


class NeRF(_PackedNet):
    """models/nerf.py:86-183 with use_viewdirs=True, encode_appearance=True, d_in=4 (inverted
    sphere), multires=10, multires_view=4 -- the configuration neuconw_system.py:90-103 builds."""

    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=[4],
                 in_channels_a=48, in_channels_dir=27, encode_appearance=False, use_viewdirs=False):
        super().__init__()
        if not (use_viewdirs and encode_appearance and d_in == 4 and d_in_view == 3 and multires == 10
                and multires_view == 4):
            raise NotImplementedError("HIP NeRF kernels implement the background configuration of "
                                      "lightning_modules/neuconw_system.py:90-103 only")
        if W not in (64, 256) or D < 2 or D > 8 or len(skips) != 1 or not (0 <= skips[0] < D - 1) or in_channels_a > 69:
            raise NotImplementedError("W in {64,256}, 2<=D<=8, one skip, n_a<=69")
        self.D, self.W, self.skips = D, W, list(skips)
        self.in_channels_a, self.in_channels_dir = in_channels_a, in_channels_dir
        self.input_ch, self.input_ch_view = 4 + 4 * 2 * multires, 3 + 3 * 2 * multires_view
        self.encode_appearance, self.use_viewdirs = encode_appearance, use_viewdirs
        # same construction order as the reference (nerf.py:127-154)
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)]
            + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W) for i in range(D - 1)])
        od = OrderedDict([("static_linear_0", nn.Linear(W + in_channels_dir + in_channels_a, W // 2))])
        for i in range(1, D // 2):
            od["static_linear_%d" % i] = nn.Linear(W // 2, W // 2)
        self.apperence_encoding = nn.Sequential(od)
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])  # dead (nerf.py:143,175-179)
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        # fwd_stash: per-ray fp32 head columns (16-bit modes); NEUCONW_NERF_RAY_BIAS overrides NEUCONW_COLOR_RAY_BIAS for this net
        self.ray_bias = os.environ.get("NEUCONW_NERF_RAY_BIAS", os.environ.get("NEUCONW_COLOR_RAY_BIAS", "1")) != "0"
        # fp16 mode, W = 256: the forward re-evaluates the samples the compositor can use in split precision (ncw_nerf_refine: every
        # operand as an fp16 hi + lo pair) over the plain-fp16 outputs; NEUCONW_NERF_REFINE=0 / .refine = False = plain fp16 only
        self.refine = os.environ.get("NEUCONW_NERF_REFINE", "1") != "0"
        self._init_plans()

    @property
    def n_head(self):
        return len(self.apperence_encoding)

    def _plan_switches(self):
        return (bool(self.refine),)

    def _build_plan(self, prec, dev):
        RBN, RBH, W, A, E = self.W // 32, self.W // 64, self.W, self.in_channels_a, self.input_ch
        plan = PackPlan(dev, prec)
        net = L.NcwNerfNet()
        sl = {}

        split = prec == L.PREC_F16 and self.refine and RBN == 8  # residual matrices of the forward (ncw_nerf_refine)

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

        full("p0", self.pts_linears[0], RBN, 3, [(0, E, 0)])
        for i in range(1, self.D):
            if i == self.skips[0] + 1:  # input = cat([gamma(p), h])  (nerf.py:166-167)
                full("p%d" % i, self.pts_linears[i], RBN, RBN + 3, [(E, W, 0), (0, E, 32 * RBN)])
            else:
                full("p%d" % i, self.pts_linears[i], RBN, RBN, [(0, W, 0)])
        full("alpha", self.alpha_linear, 1, RBN, [(0, W, 0)])
        full("feat", self.feature_linear, RBN, RBN, [(0, W, 0)])
        full("a0", self.apperence_encoding[0], RBH, RBN + 3, [(0, W, 0), (W, 27 + A, 32 * RBN)])
        for i in range(1, self.n_head):
            full("a%d" % i, self.apperence_encoding[i], RBH, RBH, [(0, W // 2, 0)])
        full("rgb", self.rgb_linear, 1, RBH, [(0, W // 2, 0)])
        plan.finalize()

        def trip(name):
            s = sl[name]
            return plan.mat_ptr(s[0]), plan.bias_ptr(s[1]), plan.mat_ptr(s[2])

        for i in range(self.D):
            net.w_p[i], net.b_p[i], net.wt_p[i] = trip("p%d" % i)
        net.w_alpha, net.b_alpha, net.wt_alpha = trip("alpha")
        net.w_feat, net.b_feat, net.wt_feat = trip("feat")
        for i in range(self.n_head):
            net.w_a[i], net.b_a[i], net.wt_a[i] = trip("a%d" % i)
        net.w_rgb, net.b_rgb, net.wt_rgb = trip("rgb")
        if split:
            for i in range(self.D):
                net.w_p_lo[i] = plan.mat_ptr(sl["p%d" % i][4])
            for i in range(self.n_head):
                net.w_a_lo[i] = plan.mat_ptr(sl["a%d" % i][4])
            net.w_alpha_lo, net.w_feat_lo, net.w_rgb_lo = (plan.mat_ptr(sl[k][4]) for k in ("alpha", "feat", "rgb"))
        plan.has_lo = split
        net.D, net.skip, net.rbn, net.rbh, net.n_head, net.n_a = self.D, self.skips[0], RBN, RBH, self.n_head, A
        plan.net, plan.slots = net, sl
        return plan

    def supports_selection(self, prec):
        """Point selections (NcwPoints mode 4, dead-background elimination): every background kernel takes them (the
        W = 256 16-bit weights-stationary kernels, and the generic weights-through-LDS kernels of the fp32 parity mode /
        other widths / the reproducible d_a_rows path)."""
        return True

    def fwd_stash(self, pts, n, prec, a, x4=None, select=None, train=True, refine=None):
        """refine = (z, O) (like `select`; fp16 mode at W = 256 with .refine on): after the plain forward the samples the compositor
        can use are re-evaluated in split precision over its outputs (ncw_nerf_refine) -- also when `select` is None (the dense
        evaluation of every sample like the reference; the list is then made for the refinement alone).
        train=False: the forward-only render -- nothing is stashed (NcwNerfStash.gp == NULL selects the render kernels); the same
        density / rgb bit for bit.
        select = (z, O): pts are the R x (S + O) mode-2 samples of z_feed, z [R, S] the primary samples; evaluate only the
        columns the compositor can use -- i < S where primary sample i is outside the unit sphere, and the O outside
        samples (ncw_bg_select).  density / rgb stay
        dense [n] (zero where skipped: the compositor selects, never multiplies, there); the stashes are compact and the
        weight-gradient products are sized by the device count ctx["sel_count"]."""
        dev = self._first_param().device
        plan = self.packed(prec)
        RBN, RBH = self.W // 32, self.W // 64

        def build_render():
            return dict(arena=StashArena(dev, prec, n).allocate(), ids={}, stash=L.NcwNerfStash())

        def build():
            ar = StashArena(dev, prec, n)
            ids = dict(gp=ar.new(3), aux1=ar.new(3), featn=ar.new(RBN), zalpha=ar.new(1), zfeat=ar.new(RBN),
                       zrgb=ar.new(1))
            ids["h"] = {i: ar.new(RBN) for i in range(1, self.D + 1)}
            ids["zp"] = [ar.new(RBN) for _ in range(self.D)]
            ids["e"] = [ar.new(RBH) for _ in range(self.n_head)]
            ids["ze"] = [ar.new(RBH) for _ in range(self.n_head)]
            ar.allocate()
            st = L.NcwNerfStash()
            for k in ("gp", "aux1", "featn", "zalpha", "zfeat", "zrgb"):
                setattr(st, k, ar.ptr(ids[k]))
            for i, v in ids["h"].items():
                st.h[i] = ar.ptr(v)
            for k in ("zp", "e", "ze"):
                for i, v in enumerate(ids[k]):
                    getattr(st, k)[i] = ar.ptr(v)
            return dict(arena=ar, ids=ids, stash=st)

        ent = self.__dict__.setdefault("_stash_cache", StashCache()).acquire((prec, n, str(dev), bool(train)),
                                                                             build if train else build_render)
        ar, ids, st = ent["arena"], ent["ids"], ent["stash"]
        sel_count = None
        do_refine = (refine is not None and prec == L.PREC_F16 and self.refine and getattr(plan, "has_lo", False) and x4 is None)
        pts_sel = None
        if select is not None or do_refine:
            z_prim, O_ = select if select is not None else refine
            z_prim = z_prim.contiguous().float()
            S_ = int(z_prim.shape[1])
            assert x4 is None and pts.mode == 2 and pts.per_ray == S_ + O_ and self.supports_selection(prec)
            if "sel_idx" not in ent:  # per lease: the weight-gradient table caches the count's address
                ent["sel_idx"] = torch.empty(n, device=dev, dtype=torch.int32)
                ent["sel_count"] = torch.zeros(1, device=dev, dtype=torch.int32)
                ent["sel_offs"] = torch.empty(n // (S_ + O_) + 1, device=dev, dtype=torch.int32)
            sel_count = ent["sel_count"]
            L.check(L.get_lib().ncw_bg_select(pts.rays_o, pts.rays_d, L.ptr(z_prim), pts.sample_dist, n // (S_ + O_), S_, O_,
                                              L.ptr(ent["sel_idx"]), L.ptr(ent["sel_offs"]), L.ptr(sel_count),
                                              L.stream_ptr(dev)), "ncw_bg_select")
            if train and select is not None:  # the weight-gradient launch plans its split-K with the observed share (stash.SelectionProbe)
                from .stash import SelectionProbe

                probe = self.__dict__.get("_sel_probe")
                if probe is None and not torch.cuda.is_current_stream_capturing():  # (pinned words: not allocatable under capture)
                    probe = self._sel_probe = SelectionProbe()
                if probe is not None:
                    probe.observe(sel_count, n)
                # before anything has been observed: the outside samples + ~5 % of the primary ones
                default = (O_ + 0.05 * S_) / float(S_ + O_)
                ent["sel_plan"] = probe.bucket(default) if probe is not None else ent.get("sel_plan", (default, None))
            keep_src = pts
            pts_sel = points_struct(mode=4, idx=ent["sel_idx"], count=sel_count)
            pts_sel.rays_o, pts_sel.rays_d, pts_sel.z, pts_sel.sample_dist = keep_src.rays_o, keep_src.rays_d, keep_src.z, keep_src.sample_dist
            pts_sel.per_ray = keep_src.per_ray
            pts_sel._keep = pts_sel._keep + [keep_src, z_prim]
        if select is not None:
            pts = pts_sel
            density = torch.zeros(n, device=dev, dtype=torch.float32)
            rgb = torch.zeros(n, 3, device=dev, dtype=torch.float32)
        else:
            sel_count = None  # (a list made for the refinement alone does not size the weight-gradient products)
            density = torch.empty(n, device=dev, dtype=torch.float32)
            rgb = torch.empty(n, 3, device=dev, dtype=torch.float32)
        a = a.contiguous().float()
        x4c = x4.contiguous().float() if x4 is not None else None
        # 16-bit modes: the appearance head's view-direction / appearance-code columns once per ray in fp32 (as
        # RenderingNetwork.fwd_stash; models/nerf.py:131-139,173-174).  Forward only.
        st.aux_bias = None
        lin0 = self.apperence_encoding[0]
        if prec != L.PREC_F32 and self.ray_bias and pts.rays_d and hasattr(lin0, "weight"):
            R, no = a.shape[0], 32 * RBH
            ab = ent.get("aux_bias")
            if ab is None or ab.shape[0] != R:
                ab = ent["aux_bias"] = torch.empty(R, no, device=dev, dtype=torch.float32)
            w0 = lin0.weight.detach()
            assert w0.is_contiguous() and w0.dtype == torch.float32 and w0.shape == (self.W // 2, self.W + 27 + self.in_channels_a)
            L.check(L.get_lib().ncw_aux_ray_bias(L.ptr(w0), w0.shape[1], self.W, self.W // 2, ctypes.c_void_p(pts.rays_d), L.ptr(a),
                                                 self.in_channels_a, R, L.ptr(ab), no, L.stream_ptr(dev)), "ncw_aux_ray_bias")
            st.aux_bias = ab.data_ptr()
        L.check(L.get_lib().ncw_nerf_fwd(plan.net, prec, pts, L.ptr(x4c), n, L.ptr(a), L.ptr(density), L.ptr(rgb), st,
                                         L.stream_ptr(dev)), "ncw_nerf_fwd")
        if do_refine and st.aux_bias:  # (the head's per-ray fp32 columns are its input: .ray_bias on)
            L.check(L.get_lib().ncw_nerf_refine(plan.net, prec, pts_sel, n, ctypes.c_void_p(st.aux_bias), L.ptr(density), L.ptr(rgb),
                                                L.stream_ptr(dev)), "ncw_nerf_refine")
        return density, rgb, dict(arena=ar, ids=ids, stash=st, pts=pts, n=n, prec=prec, plan=plan, keep=(a, x4c),
                                  lease=ent, sel_count=sel_count)

    def bwd_stash(self, ctx, d_density, d_rgb, d_a, d_a_rows=None):
        """d_a [R,n_a] accumulates with atomics -- or, with d_a_rows [n,n_a], every point's row is stored instead
        (the caller reduces them in order: ncw_ray_sum_rows)."""
        dev = self._first_param().device
        d_density = d_density.contiguous().float()
        d_rgb = d_rgb.contiguous().float()
        L.check(L.get_lib().ncw_nerf_bwd(ctx["plan"].net, ctx["prec"], ctx["pts"], ctx["n"], L.ptr(d_density),
                                         L.ptr(d_rgb), L.ptr(d_a), L.ptr(d_a_rows), ctx["stash"], L.stream_ptr(dev)),
                "ncw_nerf_bwd")
        ctx["_keep_bwd"] = (d_density, d_rgb)

    def add_wgrads(self, ctx, batch):
        plan, ar, ids, sl = ctx["plan"], ctx["arena"], ctx["ids"], ctx["plan"].slots
        RBN, RBH = self.W // 32, self.W // 64
        P = ar.ptr

        def dn(name):
            d = sl[name][3]
            return plan.dense_ptr(d), plan.dense_ld(d), plan.dense_bias_ptr(d)

        dp, ld, db = dn("p0")
        batch.add(P(ids["zp"][0]), RBN, P(ids["gp"]), 3, dp, ld, db)
        for i in range(1, self.D):
            dp, ld, db = dn("p%d" % i)
            batch.add(P(ids["zp"][i]), RBN, P(ids["h"][i]), RBN, dp, ld, db)
            if i == self.skips[0] + 1:
                batch.add(P(ids["zp"][i]), RBN, P(ids["gp"]), 3, dp + 4 * 32 * RBN, ld)
        hD = P(ids["h"][self.D])
        dp, ld, db = dn("alpha")
        batch.add(P(ids["zalpha"]), 1, hD, RBN, dp, ld, db)
        dp, ld, db = dn("feat")
        batch.add(P(ids["zfeat"]), RBN, hD, RBN, dp, ld, db)
        dp, ld, db = dn("a0")
        batch.add(P(ids["ze"][0]), RBH, P(ids["featn"]), RBN, dp, ld, db)
        batch.add(P(ids["ze"][0]), RBH, P(ids["aux1"]), 3, dp + 4 * 32 * RBN, ld)
        for i in range(1, self.n_head):
            dp, ld, db = dn("a%d" % i)
            batch.add(P(ids["ze"][i]), RBH, P(ids["e"][i - 1]), RBH, dp, ld, db)
        dp, ld, db = dn("rgb")
        batch.add(P(ids["zrgb"]), 1, P(ids["e"][self.n_head - 1]), RBH, dp, ld, db)

    @torch.no_grad()
    def forward(self, input_pts, input_views, embedding_a, prec=None):
        """NeRF.forward(pts4 [N,4], views [N,3], a [N,A]) -> (alpha [N,1], rgb [N,3])  (nerf.py:156-182);
        inference entry point -- training goes through NeuconWRenderer.render."""
        prec = default_prec() if prec is None else prec
        n = input_pts.shape[0]
        pts = points_struct(x=input_pts[:, :3].contiguous(), rays_d=input_views.contiguous().float())
        density, rgb, c = self.fwd_stash(pts, n, prec, embedding_a, x4=input_pts, train=False)
        StashCache.release(c["lease"])
        return density.reshape(n, 1), rgb
