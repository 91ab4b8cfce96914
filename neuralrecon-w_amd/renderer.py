"""Host-side mirror of the reference's rendering/renderer.py `NeuconWRenderer`: same constructor
keywords, same `render()` output dictionary, same helper methods (`sdf`, `rgb`, `get_octree`) and
mutable attributes -- the body runs the hand-written gfx950 kernels of libneuconw_hip.so.

Only per-RAY glue (ray normalisation, the embedding lookup) stays in torch; the per-ray loss terms render()
returns (gradient_error, mask_error, sfm_depth_loss) are one C-ABI launch (`_RayTailFn`); everything per ray-SAMPLE (sampling, the three MLPs forward and backward,
compositing) is HIP.  Autograd sees ONE node (`_RenderFn`) whose backward launches the fused
backward kernels and returns the gradients of every parameter.
"""
import ctypes as C
import os

import numpy as np
import torch
import yaml

from . import lib as L
from . import rayops
from .neuconw import default_infer_prec, default_prec, points_struct
from .packing import pack_many, unpack_many
from .stash import LeaseGuard, WgradBatch

from .labels import LABEL_IDS, label_id as _label_id  # noqa: E402  (ADE20K ids: datasets/mask_utils.py)

SKY_LABEL_ID = LABEL_IDS["sky"]


class _RenderFn(torch.autograd.Function):
    """render_core_outside + render_core (renderer.py:157-228, 570-783) as one autograd node."""

    @staticmethod
    def forward(ctx, rdr, train, rays_o, rays_d, z, z_out, sample_dist, cos_anneal, background_rgb, a_embedded, variance,
                *params):
        # train: somebody may come back for a backward (grad mode on and a differentiable input: decided by render(), since
        # inside forward() grad mode is always off and needs_input_grad ignores it).  False = the forward-only render of the
        # reference's validation / novel-view path (lightning_modules/neuconw_system.py:404-458 under no_grad): the MLP
        # launches stash nothing but the SDF network's h_l scratch + feat (5 KB instead of 36 KB per sample).
        prec = rdr.prec
        neuconw, nerf = rdr.neuconw, rdr.nerf
        R, S = z.shape
        dev = z.device
        ctx.set_materialize_grads(False)  # outputs the loss does not use arrive as None, not as freshly filled zeros
        inv_s = torch.empty(1, device=dev, dtype=torch.float32)   # clamp(exp(10 variance), 1e-6, 1e6): one launch
        s_val = torch.empty(1, device=dev, dtype=torch.float32)
        L.check(L.get_lib().ncw_inv_s_fwd(L.ptr(variance.detach().reshape(1).float().contiguous()), L.ptr(inv_s), L.ptr(s_val),
                                          L.stream_ptr(dev)), "ncw_inv_s_fwd")
        a_det = a_embedded.detach().contiguous().float()
        use_bg = rdr.render_bg and rdr.n_outside > 0 and z_out is not None
        z_feed = density = bg_rgb = nctx = None
        if use_bg:
            M = S + z_out.shape[1]
            # Dead-background elimination: with trim_sphere the compositor takes the background NeRF only where a primary
            # sample's section mid-point is OUTSIDE the unit sphere (renderer.py:637,693-708: everything else is
            # multiplied by 1 - inside_sphere = 0, forward and backward) and at the n_outside samples; the reference
            # evaluates the NeRF on all S + O samples all the same.  Identical outputs and gradients; bg_dense=True
            # evaluates everything like the reference.  (Columns are paired with primary samples by index, as there.)
            select = refine = None
            if rdr.trim_sphere and nerf.supports_selection(prec):
                refine = (z, M - S)  # fp16 mode: the samples the compositor can use are re-evaluated in split precision (ncw_nerf_refine)
                if not rdr.bg_dense:
                    select = refine
            # The background NeRF is independent of the SDF / colour chain until the compositor: with `bg_stream` (a
            # second HIP stream, renderer.use_bg_stream) its launches overlap the ramps, tails and partial last rounds of the
            # SDF / colour launches (1056 workgroups of 128 background points = 4.125 rounds of 256 CUs: the fifth round
            # keeps 32 CUs busy) instead of queueing behind them.
            side = rdr._bg_stream(dev) if rdr.use_bg_stream else None
            if side is not None:
                main = torch.cuda.current_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    # (the merge of the primary and outside depths is only read by the NeRF and, after the join, by the compositor: a
                    # launch-latency-sized kernel off the SDF / colour chain)
                    z_feed, _ = rayops.sort_merge(z, z_out)
                    pts_bg = points_struct(rays_o=rays_o, rays_d=rays_d, z=z_feed, sample_dist=sample_dist, mode=2)
                    density, bg_rgb, nctx = nerf.fwd_stash(pts_bg, R * M, prec, a_det, select=select, train=train, refine=refine)
                # allocated under the side stream, consumed by the compositor on the main stream after the join below: tell
                # the caching allocator, so that freeing them can never hand the memory out while the main stream still reads it
                density.record_stream(main)
                bg_rgb.record_stream(main)
                z_feed.record_stream(main)
            else:
                z_feed, _ = rayops.sort_merge(z, z_out)
                pts_bg = points_struct(rays_o=rays_o, rays_d=rays_d, z=z_feed, sample_dist=sample_dist, mode=2)
                density, bg_rgb, nctx = nerf.fwd_stash(pts_bg, R * M, prec, a_det, select=select, train=train, refine=refine)
            assert z_feed.shape[1] == M
            density, bg_rgb = density.view(R, M), bg_rgb.view(R, M, 3)
        try:
            pts_in = points_struct(rays_o=rays_o, rays_d=rays_d, z=z, sample_dist=sample_dist, mode=2)
            sdf, grad, sctx = neuconw.sdf_net.fwd_stash(pts_in, R * S, prec, train=train)
            feat_ptr = sctx["arena"].ptr(sctx["ids"]["feat"])
            rgb, cctx = neuconw.color_net.fwd_stash(pts_in, R * S, prec, grad, a_det, feat_ptr, train=train)
        finally:  # the main stream ALWAYS rejoins the side stream (also when the SDF / colour chain raised)
            if use_bg and rdr.use_bg_stream:
                torch.cuda.current_stream(dev).wait_stream(rdr._bg_stream(dev))  # join before the compositor
        comp = rayops.CompositeCtx(rays_o, rays_d, z, sample_dist, sdf.view(R, S), grad.view(R, S, 3),
                                   rgb.view(R, S, 3), inv_s, cos_anneal, z_feed, density, bg_rgb, background_rgb,
                                   rdr.trim_sphere)
        o = comp.forward()
        ctx.rdr, ctx.comp, ctx.sctx, ctx.cctx, ctx.nctx = rdr, comp, sctx, cctx, nctx
        ctx.inv_s, ctx.n_params, ctx.use_bg = inv_s, len(params), use_bg
        ctx.a_shape = a_embedded.shape
        ctx.variance = variance
        ctx.params = params
        extras = (o["color_sphere"], o["color_bg"], o["weights"], o["cdf"], o["inside"], o["normals"],
                  sdf.view(R, S), grad.view(R, S, 3), o["mid_z"], o["dists"], o["eik"][1], inv_s, s_val,
                  o["weights_max"])
        ctx.mark_non_differentiable(*extras)
        # the leases live exactly as long as this autograd node: returned by backward(), or when the node is dropped
        ctx.guard = LeaseGuard([c["lease"] for c in (sctx, cctx, nctx) if c is not None])
        ctx.train = train
        if not train:  # forward-only render: nothing was stashed, nobody comes back
            ctx.guard.release()
        return (o["color"], o["weights_sum"], o["depth"], o["eik"][0]) + extras

    @staticmethod
    def backward(ctx, d_color, d_wsum, d_depth, d_eik, *unused):
        rdr, comp, sctx, cctx, nctx = ctx.rdr, ctx.comp, ctx.sctx, ctx.cctx, ctx.nctx
        neuconw, nerf = rdr.neuconw, rdr.nerf
        prec = rdr.prec
        dev = ctx.inv_s.device
        if not ctx.train:
            raise RuntimeError("NeuconWRenderer: backward through a forward-only render (it ran under torch.no_grad(), or no "
                               "input required a gradient): nothing was stashed")
        ctx.guard.consume()
        # fp16 mode: the per-point adjoints are rounded to fp16 inside the MLP backward kernels (MFMA operands, delta
        # stashes), whose normal range ends at 6e-5 -- the loss's 1/(3R) alone puts them below it.  The compositor
        # backward multiplies its upstream by the (power-of-two) loss scale; it is divided back out of the parameter
        # gradients (unpack), d_a and d_var below, all in f32.
        # The scale is a DEVICE scalar (rdr.loss_scale: {scale, 1 / scale}), so the optimiser can halve it after an
        # overflowed step and grow it back (trainer.FlatAdam / ncw_adam_step_dev) without a device->host round trip.
        sc = rdr.loss_scale.tensor(dev) if prec == L.PREC_F16 else None
        sc_mul, sc_inv = (sc[0:1], sc[1:2]) if sc is not None else (None, None)
        g = comp.backward(d_color, d_wsum, d_depth, d_eik, grad_scale_dev=sc_mul)
        R, S = comp.R, comp.S
        d_grad = g["d_grad"].view(R * S, 3)
        dfeat_ptr = sctx["arena"].ptr(sctx["ids"]["dfeat"])
        n_a = int(ctx.a_shape[-1])
        # reproducible mode (default in fp32): per-point appearance-code adjoints + an order-fixed per-ray sum
        # instead of f32 atomics (the weight-gradient split-K is order-fixed in fp32 too, stash.WgradBatch.run)
        ordered = rdr.reproducible if rdr.reproducible is not None else (prec == L.PREC_F32)
        lib = L.get_lib()
        M = comp.S + comp.O
        # the background backward on the second stream (see forward): issued BEFORE the colour / SDF backward of the main
        # stream so that both are in flight together; joined before the weight-gradient launch.  (Atomics path only: the
        # order-fixed d_a_rows reduction of the reproducible mode stays on one stream.)
        side = rdr._bg_stream(dev) if (ctx.use_bg and rdr.use_bg_stream and not ordered) else None
        d_a_bg = None
        if side is not None:
            d_a_bg = torch.zeros(ctx.a_shape, device=dev, dtype=torch.float32)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                nerf.bwd_stash(nctx, g["d_density"].view(R * M), g["d_bg_rgb"].view(R * M, 3), d_a_bg)
            for t_ in (g["d_density"], g["d_bg_rgb"]):  # main-stream tensors read by the side stream
                t_.record_stream(side)
        try:
            if ordered:
                d_a = torch.empty(ctx.a_shape, device=dev, dtype=torch.float32)
                rows = torch.empty(R * S, n_a, device=dev, dtype=torch.float32)
                neuconw.color_net.bwd_stash(cctx, g["d_rgb"].view(R * S, 3), d_grad, None, dfeat_ptr, d_a_rows=rows)
                L.check(lib.ncw_ray_sum_rows(L.ptr(rows), R, S, n_a, L.ptr(d_a), 0, L.stream_ptr(dev)), "ncw_ray_sum_rows")
            else:
                d_a = torch.zeros(ctx.a_shape, device=dev, dtype=torch.float32)
                neuconw.color_net.bwd_stash(cctx, g["d_rgb"].view(R * S, 3), d_grad, d_a, dfeat_ptr)
            neuconw.sdf_net.bwd_stash(sctx, g["d_sdf"].view(R * S), d_grad)
        finally:  # the main stream always rejoins the side stream
            if side is not None:
                torch.cuda.current_stream(dev).wait_stream(side)
        plans = [sctx["plan"], cctx["plan"]]
        if side is not None:
            d_a.add_(d_a_bg)
            plans.append(nctx["plan"])
        elif ctx.use_bg:
            if ordered:
                # with the elimination only the selected samples write their row (at the ray sample's slot): the others
                # are the exact zeros the dense evaluation would have produced (their cotangents are zero)
                rows_bg = (torch.zeros if nctx.get("sel_count") is not None else torch.empty)(R * M, n_a, device=dev,
                                                                                             dtype=torch.float32)
                nerf.bwd_stash(nctx, g["d_density"].view(R * M), g["d_bg_rgb"].view(R * M, 3), None, d_a_rows=rows_bg)
                L.check(lib.ncw_ray_sum_rows(L.ptr(rows_bg), R, M, n_a, L.ptr(d_a), 1, L.stream_ptr(dev)),
                        "ncw_ray_sum_rows")
            else:
                nerf.bwd_stash(nctx, g["d_density"].view(R * M), g["d_bg_rgb"].view(R * M, 3), d_a)
            plans.append(nctx["plan"])
        # every weight-gradient product of the step (SDF, colour, background NeRF) in ONE launch; the product
        # list only depends on the (cached) stash arenas: build it once per lease combination
        sel = nctx.get("sel_count") if ctx.use_bg else None
        # the selection's expected share of the background samples (observed a step or two ago, stash.SelectionProbe): the
        # product table is re-planned when it leaves its bucket
        sel_frac, sel_bucket = nctx["lease"].get("sel_plan", (None, None)) if sel is not None else (None, None)
        tag = (cctx["arena"].buf.data_ptr(), nctx["arena"].buf.data_ptr() if ctx.use_bg else None, prec,
               None if sel is None else sel.data_ptr(), sel_bucket)
        batch = sctx["lease"].get("wgrad_batch")
        if batch is None or batch.tag != tag:
            batch = WgradBatch(dev, prec, R * S)
            batch.tag = tag
            neuconw.sdf_net.add_wgrads(sctx, batch)
            neuconw.color_net.add_wgrads(cctx, batch)
            if ctx.use_bg:
                b_bg = WgradBatch(dev, prec, R * (comp.S + comp.O), n_dev=nctx.get("sel_count"), sel_fraction=sel_frac)
                nerf.add_wgrads(nctx, b_bg)
                batch.extend(b_bg)
            sctx["lease"]["wgrad_batch"] = batch
        torch._foreach_zero_([p.g_arena for p in plans])  # one multi-tensor launch for the three dense gradient arenas
        batch.run()
        # Parameter gradients.  Default: returned THROUGH autograd (AccumulateGrad runs, so DistributedDataParallel /
        # Lightning reducer hooks, torch.autograd.grad and post-accumulate hooks all see them -- the reference trains
        # under accelerator='ddp', train.py:53-55).  Only when a trainer has adopted a flat gradient buffer
        # (trainer.FlatParams -> adopt_grad_buffer) and every .grad still IS its view of that buffer does the
        # weight-norm backward write straight into it (one launch, no per-parameter adds).
        out = [None] * len(ctx.params)
        direct = False
        if rdr.__dict__.get("_gv_adopted", False):
            flat, views = rdr._grad_views(ctx.params)
            direct = all(p.grad is not None and p.grad.data_ptr() == views[id(p)].data_ptr() for p in ctx.params)
        if direct:
            keep = unpack_many(plans, views, accumulate=True, grad_mul_dev=sc_inv)  # one launch for the three networks
        else:
            tmp = torch.zeros(sum(p.numel() for p in ctx.params), device=dev, dtype=torch.float32)
            tviews, off = {}, 0
            out = []
            for p in ctx.params:
                v = tmp[off:off + p.numel()].view(p.shape)
                tviews[id(p)] = v
                out.append(v)
                off += p.numel()
            keep = unpack_many(plans, tviews, grad_mul_dev=sc_inv)
        ctx._keep = (keep, batch)
        d_var = torch.empty(1, device=dev, dtype=torch.float32)  # 10 inv_s [clamp inactive] sum_r d_inv_s[r], fixed order
        L.check(lib.ncw_inv_s_bwd(L.ptr(g["d_inv_s"]), R, L.ptr(ctx.inv_s), L.ptr(d_var), L.stream_ptr(dev)), "ncw_inv_s_bwd")
        d_var = d_var.reshape(ctx.variance.shape)
        if sc_inv is not None:
            d_a.mul_(sc_inv)
            d_var = d_var * sc_inv
        ctx.guard.release()
        return (None, None, None, None, None, None, None, None, None, d_a, d_var) + tuple(out)


class _EmbedFn(torch.autograd.Function):
    """embeddings["a"](ts) (renderer.py:808) with the backward as one atomic scatter-add launch (`ncw_scatter_add_rows`,
    ~3 us) instead of torch's embedding_dense_backward (77 us for 1024 rays: it serialises repeated indices)."""

    @staticmethod
    def forward(ctx, weight, ts, direct_grad):
        ctx.save_for_backward(ts)
        ctx.shape, ctx.direct_grad = weight.shape, direct_grad
        return weight.detach().index_select(0, ts)

    @staticmethod
    def backward(ctx, d_a):
        (ts,) = ctx.saved_tensors
        d_a = d_a.contiguous().float()
        # direct_grad: the weight's .grad inside a trainer's flat gradient buffer (already zeroed by FlatParams.zero_grad):
        # accumulate straight into it -- no 960 KB zero fill + add_ through AccumulateGrad
        dw = ctx.direct_grad if ctx.direct_grad is not None else torch.zeros(ctx.shape, device=d_a.device, dtype=torch.float32)
        L.check(L.get_lib().ncw_scatter_add_rows(L.ptr(d_a), L.ptr(ts), d_a.shape[0], d_a.shape[1], dw.shape[0], L.ptr(dw),
                                                 L.stream_ptr(d_a.device)), "ncw_scatter_add_rows")
        return (None if ctx.direct_grad is not None else dw), None, None


class _RayTailFn(torch.autograd.Function):
    """gradient_error, mask_error and the sync-free sfm_depth_loss of render() (renderer.py:763-765, 869-877,
    892-897) as one forward and one backward launch (ncw_ray_tail_fwd/bwd) instead of ~55 tiny torch kernels."""

    @staticmethod
    def forward(ctx, wsum, depth, eik_num, eik_den, label, depth_gt, depth_w, ids, has_mask, has_depth):
        R, dev = wsum.shape[0], wsum.device
        f = lambda t: t.detach().contiguous().float()  # noqa: E731
        wsum_c, depth_c, num_c, den_c = f(wsum), f(depth), f(eik_num), f(eik_den)
        label_c = label.contiguous().long() if (has_mask and label is not None) else None
        gt_c, w_c = (f(depth_gt), f(depth_w)) if has_depth else (None, None)
        mask_error = torch.empty(R, device=dev) if has_mask else None
        sfm = torch.empty(R, device=dev) if has_depth else None
        scal = torch.empty(3, device=dev)
        ids_arr = (C.c_int * 4)(*(list(ids) + [0] * (4 - len(ids))))
        L.check(L.get_lib().ncw_ray_tail_fwd(L.ptr(wsum_c), L.ptr(label_c), ids_arr, len(ids), L.ptr(depth_c),
                                             L.ptr(gt_c), L.ptr(w_c), L.ptr(num_c), L.ptr(den_c), R, L.ptr(mask_error),
                                             L.ptr(sfm), L.ptr(scal), L.stream_ptr(dev)), "ncw_ray_tail_fwd")
        ctx.keep = (wsum_c, depth_c, label_c, gt_c, w_c, scal, ids_arr, len(ids), has_mask, has_depth)
        return mask_error, sfm, scal[0:1]

    @staticmethod
    def backward(ctx, d_me, d_sfm, d_ge):
        wsum_c, depth_c, label_c, gt_c, w_c, scal, ids_arr, n_ids, has_mask, has_depth = ctx.keep
        R, dev = wsum_c.shape[0], wsum_c.device
        g = lambda t: None if t is None else t.contiguous().float()  # noqa: E731
        d_me, d_sfm, d_ge = (g(d_me) if has_mask else None), (g(d_sfm) if has_depth else None), g(d_ge)
        d_wsum, d_depth, d_num = (torch.empty(R, device=dev) for _ in range(3))
        L.check(L.get_lib().ncw_ray_tail_bwd(L.ptr(wsum_c), L.ptr(label_c), ids_arr, n_ids, L.ptr(depth_c), L.ptr(gt_c),
                                             L.ptr(w_c), R, L.ptr(scal), L.ptr(d_me), L.ptr(d_sfm), L.ptr(d_ge),
                                             L.ptr(d_wsum), L.ptr(d_depth), L.ptr(d_num), L.stream_ptr(dev)),
                "ncw_ray_tail_bwd")
        return d_wsum, d_depth, d_num, None, None, None, None, None, None, None


class LossScale:
    """The fp16 mode's loss scale as device data: float[2] = {scale, 1 / scale} (powers of two).  The compositor backward
    multiplies its upstream cotangents by [0] (NcwCompositeGrad.grad_scale_dev), the weight-norm backward / d_a / d_var
    multiply by [1] (NcwUnpackDesc.grad_mul_dev); `ncw_adam_step_dev` updates both in place.  The reference trains in fp32
    and has no counterpart (train.py:48-62)."""

    def __init__(self, init):
        self.init, self.buf = float(init), None

    def tensor(self, device):
        if self.buf is None or self.buf.device != torch.device(device):
            self.buf = torch.tensor([self.init, 1.0 / self.init], device=device, dtype=torch.float32)
        return self.buf

    def set(self, value):
        self.init = float(value)
        if self.buf is not None:
            self.buf.copy_(torch.tensor([self.init, 1.0 / self.init]))

    def value(self):
        """Current scale (synchronises when it lives on the device)."""
        return self.init if self.buf is None else float(self.buf[0])


class NeuconWRenderer:
    @property
    def grad_scale(self):
        return self.loss_scale.value()

    @grad_scale.setter
    def grad_scale(self, v):
        self.loss_scale.set(v)

    def __init__(self, nerf, neuconw, embeddings, n_samples, n_importance, n_outside, up_sample_steps, perturb,
                 origin, radius, s_val_base=0, spc_options=None, sample_range=None, boundary_samples=None,
                 nerf_far_override=False, render_bg=True, trim_sphere=True, save_sample=False,
                 save_step_sample=False, mesh_mask_list=None, floor_normal=False, depth_loss=False,
                 floor_labels=None, prec=None, infer_prec=None):
        self.nerf, self.neuconw, self.embeddings = nerf, neuconw, embeddings
        self.n_samples, self.n_importance, self.n_outside = n_samples, n_importance, n_outside
        self.up_sample_steps, self.perturb, self.s_val_base = up_sample_steps, perturb, s_val_base
        self.boundary_samples, self.nerf_far_override = boundary_samples, nerf_far_override
        self.octree_data, self.sample_range, self.fine_octree_data = None, sample_range, None
        spc_options = spc_options or {}
        if self.nerf_far_override:  # same coupling as the reference (renderer.py:96-99)
            self.recontruct_path = spc_options["recontruct_path"]
            self.min_track_length = spc_options["min_track_length"]
            self.voxel_size = spc_options["voxel_size"]
        self.sfm_to_gt = torch.eye(4)
        scene_config_path = os.path.join(spc_options.get("recontruct_path", ""), "config.yaml")
        if os.path.isfile(scene_config_path):  # renderer.py:103-112
            with open(scene_config_path, "r") as f:
                scene_config = yaml.load(f, Loader=yaml.FullLoader)
            origin, radius = scene_config["origin"], scene_config["radius"]
            self.sfm_to_gt = torch.from_numpy(np.array(scene_config["sfm2gt"]))
        self.origin = torch.from_numpy(np.array(origin, dtype=np.float64))
        self.radius = radius
        self.render_bg, self.floor_normal, self.floor_labels = render_bg, floor_normal, floor_labels
        self.depth_loss, self.save_sample, self.trim_sphere = depth_loss, save_sample, trim_sphere
        self.mesh_mask_list = mesh_mask_list
        self.save_step_sample = save_step_sample
        if floor_normal:
            raise NotImplementedError("floor_normal loss is not on the HIP path (all scene yamls ship FLOOR_NORMAL: False)")
        if save_sample or save_step_sample:
            raise NotImplementedError("debug PLY dumps (open3d) are out of scope")
        self.prec = default_prec() if prec is None else prec
        # inference-only helpers (sdf(), rgb(): octree refresh, grid sweep, mesh colours) run in fp32 like the
        # reference unless told otherwise; the training passes and the sampler follow `prec`
        # None = the defaults: fp32 for rgb() / NeuconW.forward, SDFNetwork.value_prec() for sdf() and the sweeps built on it
        self.infer_prec = infer_prec
        if self.n_samples + self.n_importance + (self.boundary_samples or 0) + self.n_outside > 1088:
            raise ValueError("n_samples + n_importance + boundary_samples + n_outside = %d > 1088: the per-ray kernels keep a "
                             "ray's samples in LDS (csrc/ncw_rays.hip: 512 in the standard kernels, 1088 in the large-ray ones -- "
                             "config/defaults.py's own 512 + 512 + 32 fits)."
                             % (self.n_samples + self.n_importance + (self.boundary_samples or 0) + self.n_outside))
        # bg_dense=True: evaluate the background NeRF on every sample like the reference does, instead of only where the
        # compositor can use it (dead-background elimination, _RenderFn.forward: every precision, incl. the reproducible
        # fp32 parity mode); NEUCONW_BG_DENSE=1 sets the default
        self.bg_dense = os.environ.get("NEUCONW_BG_DENSE", "0") not in ("0", "")
        # loss scale of the fp16 mode (prec = PREC_F16; unused otherwise): a power of two kept on the device, see
        # _RenderFn.backward.  `grad_scale` (property) reads / sets it; trainer.FlatAdam adapts it (halves after a step
        # with a non-finite gradient norm, doubles after `growth_interval` clean steps).
        self.loss_scale = LossScale(float(os.environ.get("NEUCONW_F16_LOSS_SCALE", "1024")))
        # use_bg_stream: run the background NeRF's launches on a second HIP stream beside the SDF / colour chain (forward and
        # backward; joined before the compositor / the weight-gradient launch).  NEUCONW_BG_STREAM=0 keeps one stream.
        self.use_bg_stream = os.environ.get("NEUCONW_BG_STREAM", "1") not in ("0", "")
        # sync_free=True keeps render() free of device->host synchronisations (see sfm_depth_loss below);
        # the default reproduces the reference's output shapes exactly.
        self.sync_free = False
        # reproducible: None = on in fp32 (the parity mode is bitwise run-to-run reproducible), off in bf16 (f32 atomics
        # for the appearance-code gradient); True / False force it
        self.reproducible = None

    def _bg_stream(self, device):
        st = self.__dict__.get("_bg_streams")
        if st is None:
            st = self._bg_streams = {}
        key = str(device)
        if key not in st:
            st[key] = torch.cuda.Stream(device=device)
        return st[key]

    # ---- sampler (renderer.py:458-568, under no_grad) -------------------------------------------
    def _sdf_rays(self, rays_o, rays_d, z):
        R, n = z.shape
        out = torch.empty(R, n, device=z.device, dtype=torch.float32)
        sp = self.__dict__.get("sampler_prec")  # None = the training precision; see scripts/diag/sampler_prec.py
        sp = self.prec if sp is None else sp
        plan = self.neuconw.sdf_net.packed(sp)
        L.check(L.get_lib().ncw_sdf_infer_rays(plan.net, sp, L.ptr(rays_o), L.ptr(rays_d), L.ptr(z), R, n,
                                               L.ptr(out), L.stream_ptr(z.device)), "ncw_sdf_infer_rays")
        return out

    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s, step=0):
        return rayops.upsample(rays_o, rays_d, z_vals, sdf, n_importance, inv_s)

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False):
        if last:
            return rayops.sort_merge(z_vals, new_z_vals)[0], sdf
        new_sdf = self._sdf_rays(rays_o, rays_d, new_z_vals.contiguous())
        return rayops.sort_merge(z_vals, new_z_vals, sdf, new_sdf)

    @torch.no_grad()
    def sparse_sampler(self, rays_o, rays_d, near, far, perturb, _rand=None):
        dev = rays_o.device
        R = rays_o.shape[0]
        rays_o, rays_d = rays_o.contiguous().float(), rays_d.contiguous().float()
        if self.nerf_far_override:
            if self.octree_data is None:
                self.octree_data = self.get_octree(dev)
            near, far, _ = self.get_near_far_octree(self.octree_data, rays_o, rays_d, near, far)
        s_near, s_far = near, far
        if self.fine_octree_data is not None:
            s_near, s_far, _ = self.get_near_far_sdf(self.fine_octree_data, rays_o, rays_d, near, far)
        n_out = self.n_outside if (self.render_bg and self.n_outside > 0) else 0
        rs = ro = None
        if perturb > 0:
            if _rand is not None:
                rs, ro = _rand
            else:  # same draw order as renderer.py:499,506-508
                rs = torch.rand([R, 1], device=dev)
                ro = torch.rand([R, n_out], device=dev) if n_out > 0 else None
        z, z_out, sample_dist = rayops.sample_coarse(near, far, s_near, s_far, self.n_samples, n_out, rs, ro)
        n_samples = self.n_samples
        if self.n_importance > 0:
            sdf = self._sdf_rays(rays_o, rays_d, z)
            for i in range(self.up_sample_steps):
                z_new = rayops.upsample(rays_o, rays_d, z, sdf, self.n_importance // self.up_sample_steps,
                                        64 * 2 ** (self.s_val_base + i))
                z, sdf = self.cat_z_vals(rays_o, rays_d, z, z_new, sdf, last=(i + 1 == self.up_sample_steps))
            n_samples = self.n_samples + self.n_importance
        if self.fine_octree_data is not None and self.boundary_samples and self.boundary_samples > 0:
            zb = rayops.boundary(near, far, z, self.boundary_samples)
            z, _ = rayops.sort_merge(zb, z)
        return n_samples, z, z_out, sample_dist

    # ---- voxel guidance (renderer.py:137-155, 380-456): see voxel.py ---------------------------------
    def get_octree(self, device):
        from . import voxel

        return voxel.octree_from_sfm(self.recontruct_path, self.min_track_length, self.voxel_size, device)

    def get_near_far_octree(self, octree_data, rays_o, rays_d, near, far):
        from . import voxel

        o_sfm = (rays_o * self.radius).view(-1, 3) + self.origin.to(rays_o.device).float()
        vn, vf = voxel.get_near_far(o_sfm, rays_d, octree_data)
        hit = vn > 0
        near = torch.where(hit, vn / self.radius, near)
        far = torch.where(hit, (vf + self.voxel_size) / self.radius, far)
        return near, far, hit

    def get_near_far_sdf(self, octree_data, rays_o, rays_d, near, far):
        from . import voxel

        o_sfm = (rays_o * self.radius).view(-1, 3) + self.origin.to(rays_o.device).float()
        surf, _ = voxel.get_near_far(o_sfm, rays_d, octree_data)
        miss = surf <= 0
        rng = self.sample_range * octree_data["voxel_size"]
        v_near = torch.where(miss, near, (surf - rng) / self.radius)
        v_far = torch.where(miss, far, (surf + rng) / self.radius)
        return v_near, v_far, ~miss

    def _grad_views(self, params):
        key = tuple(id(p) for p in params)
        c = self.__dict__.get("_gv")
        if c is None or c[0] != key or c[1].device != params[0].device:
            flat = torch.zeros(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
            views, off = {}, 0
            for p in params:
                views[id(p)] = flat[off:off + p.numel()].view(p.shape)
                off += p.numel()
            self._gv = c = (key, flat, views)
        return c[1], c[2]

    def adopt_grad_buffer(self, params, flat):
        """Use `flat` (fp32, numel = sum of `params`) as the persistent flat gradient buffer: the trainer's
        flat gradient storage (trainer.FlatParams) -- the weight-norm backward then writes straight into it."""
        params = list(params)
        assert [id(p) for p in params] == [id(p) for p in self._params()], "adopt_grad_buffer: parameter order"
        assert flat.numel() == sum(p.numel() for p in params) and flat.dtype == torch.float32
        views, off = {}, 0
        for p in params:
            views[id(p)] = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._gv = (tuple(id(p) for p in params), flat, views)
        self._gv_adopted = True

    def flat_grad_buffer(self):
        """The persistent flat gradient buffer of (sdf, colour, background) parameters, or None."""
        c = self.__dict__.get("_gv")
        return None if c is None else c[1]

    # ---- render (renderer.py:785-916) ---------------------------------------------------------------
    def _params(self):
        ps = list(self.neuconw.sdf_net.parameters()) + list(self.neuconw.color_net.parameters())
        nf = self.nerf
        ps += [p for n, p in nf.named_parameters() if not n.startswith("views_linears")]
        return ps

    def render(self, rays, ts, label, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0, _rand=None, _z_override=None):
        """_rand / _z_override are test hooks: the sampler's uniforms, and primary sample depths [R, S] that replace the sampler's
        (parity of the MLPs + compositor at FIXED sample positions, bench.py `parity.fixed_z`)."""
        device = rays.device
        if not rays.is_cuda:
            raise L.NeuconwHipError("NeuconWRenderer.render: rays are not on a GPU; the hot path has no CPU fallback")
        if self.origin.device != device:
            self.origin = self.origin.to(device).float()
            self.sfm_to_gt = self.sfm_to_gt.to(device).float()
        # the three networks' weight-norm forward + MFMA-order packing in ONE launch when more than one of them is stale (after an
        # optimiser step: all three); each module's packed() then finds its plan fresh
        if self.__dict__.get("sampler_prec") is None:
            nets = [self.neuconw.sdf_net, self.neuconw.color_net] + ([self.nerf] if (self.render_bg and self.n_outside > 0) else [])
            pack_many([(m, m.plan(self.prec)) for m in nets])
        # renderer.py:793-806 (ray normalisation into unit-sphere units) as one launch
        R = rays.shape[0]
        rays_c = rays.contiguous().float()
        rays_o, rays_d = torch.empty(R, 3, device=device), torch.empty(R, 3, device=device)
        near, far = torch.empty(R, 1, device=device), torch.empty(R, 1, device=device)
        depth_gt, depth_weight = torch.empty(R, device=device), torch.empty(R, device=device)
        if self.__dict__.get("_origin_host") is None or self._origin_host[1] is not self.origin:
            self._origin_host = ((C.c_float * 3)(*[float(v) for v in self.origin.reshape(-1).tolist()]), self.origin)
        L.check(L.get_lib().ncw_ray_prologue(L.ptr(rays_c), rays_c.shape[1], R, self._origin_host[0], float(self.radius),
                                             L.ptr(rays_o), L.ptr(rays_d), L.ptr(near), L.ptr(far), L.ptr(depth_gt),
                                             L.ptr(depth_weight), L.stream_ptr(device)), "ncw_ray_prologue")
        emb_a = self.embeddings["a"]
        ordered = self.reproducible if self.reproducible is not None else (self.prec == L.PREC_F32)
        if (not ordered and isinstance(emb_a, torch.nn.Embedding) and emb_a.padding_idx is None and emb_a.max_norm is None
                and not emb_a.sparse and emb_a.weight.dtype == torch.float32 and torch.is_grad_enabled()
                and emb_a.weight.requires_grad and ts.dtype == torch.int64 and ts.dim() == 1):
            g = emb_a.weight.grad
            direct = g if (self.__dict__.get("_gv_adopted", False) and g is not None and g.is_contiguous()
                           and g.dtype == torch.float32) else None
            a_embedded = _EmbedFn.apply(emb_a.weight, ts.contiguous(), direct)
        else:  # fp32 / reproducible mode, or an unusual embedding: torch's own (deterministic) lookup + backward
            a_embedded = emb_a(ts)
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        n_samples, z_vals, z_vals_outside, sample_dist = self.sparse_sampler(rays_o, rays_d, near, far, perturb, _rand)
        if _z_override is not None:
            assert tuple(_z_override.shape) == tuple(z_vals.shape), (tuple(_z_override.shape), tuple(z_vals.shape))
            z_vals = _z_override.to(device=device, dtype=torch.float32).contiguous()
        bgc = None
        if background_rgb is not None:
            if background_rgb.numel() != 3:
                raise ValueError("background_rgb must hold ONE colour (3 values; every reference call site passes "
                                 "torch.ones/zeros([1, 3])): got shape %s" % (tuple(background_rgb.shape),))
            bgc = background_rgb.reshape(3)
        params = self._params()
        variance = self.neuconw.deviation_network.variance
        train = torch.is_grad_enabled() and (a_embedded.requires_grad or variance.requires_grad or any(p.requires_grad for p in params))
        outs = _RenderFn.apply(self, bool(train), rays_o, rays_d, z_vals, z_vals_outside, sample_dist,
                               cos_anneal_ratio if torch.is_tensor(cos_anneal_ratio) else float(cos_anneal_ratio), bgc,
                               a_embedded, variance, *params)
        (color, wsum, depth, eik_num, color_sphere, color_bg, weights, cdf, inside, normals, sdf, gradients, mid_z,
         dists, eik_den, inv_s, s_val, weights_max) = outs
        weights_sum = wsum.unsqueeze(-1)
        has_mask = self.mesh_mask_list is not None
        dense_depth = bool(self.depth_loss and self.sync_free)
        ids = tuple(_label_id(name) for name in self.mesh_mask_list) if has_mask else ()
        # gradient_error (renderer.py:763-765, batch-global scalar), mask_error (:869-877) and -- in sync-free mode --
        # sfm_depth_loss in one launch.  Sync-free sfm_depth_loss: the reference returns the SELECTED entries (a
        # data-dependent shape, :892-897) and the loss takes their mean; here every ray keeps an entry, scaled so that
        # `.mean()` over all R entries equals the reference's mean over the selected ones (0 when none is selected).
        me, sfm_dense, gradient_error = _RayTailFn.apply(wsum, depth, eik_num, eik_den, label, depth_gt, depth_weight,
                                                         ids, has_mask, dense_depth)
        mask_error = me.unsqueeze(-1) if has_mask else torch.zeros_like(weights_sum)
        if dense_depth:
            sfm_depth_loss = sfm_dense
        elif self.depth_loss and torch.sum(depth_weight > 0) > 0:  # renderer.py:892-897
            sfm_depth_loss = (((depth - depth_gt) ** 2) * depth_weight)[depth_weight > 0]
        else:
            sfm_depth_loss = torch.zeros_like(depth)
        return {
            "color": color, "color_sphere": color_sphere, "color_bg": color_bg, "s_val": s_val.reshape(1, 1),
            "cdf_fine": cdf, "gradients": gradients, "mask_error": mask_error, "weights": weights,
            "weights_sum": weights_sum, "weights_max": weights_max.unsqueeze(-1),
            "gradient_error": gradient_error, "inside_sphere": inside,
            "depth": depth, "floor_normal_error": (zn := torch.zeros_like(normals)), "floor_y_error": zn,
            "sfm_depth_loss": sfm_depth_loss,
        }

    # ---- helpers used by NeuconWSystem / extract_mesh (renderer.py:947-961) ---------------------------
    def sdf(self, pts):
        return self.neuconw.sdf(pts, self.infer_prec)

    def rgb(self, pts, rays_d, a_embedded):
        num_points = pts.shape[0]
        rgb, _, _, _ = self.neuconw(torch.cat([pts, rays_d, a_embedded], -1),
                                    default_infer_prec() if self.infer_prec is None else self.infer_prec)
        return rgb.reshape(num_points, 3)
