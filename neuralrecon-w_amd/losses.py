"""NeuconWLoss (the reference's losses.py:21-43) as ONE forward and ONE backward launch (`ncw_loss_fwd` / `ncw_loss_bwd`)
instead of the ~25 tiny torch kernels of sub / abs / sum / div / mean x3 / mul / add and their autograd backward.

    loss = coef * ( sum|color - rgbs| / (R + 1e-5)  +  igr_weight * gradient_error
                    + [MESH_MASK_LIST] mask_weight * mean(mask_error)  +  [DEPTH_LOSS] depth_weight * mean(sfm_depth_loss) )

The reference returns the terms as a dict and `training_step` sums them (neuconw_system.py:355-357); `__call__` here
returns that sum, `terms()` the dict (plain torch, for logging).  `floor_normal` is not on the HIP path (renderer refuses it)."""
import torch

from . import lib as L


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rgbs, ge, me, sfm, w):
        dev = color.device
        f = lambda t: None if t is None else t.detach().reshape(-1).float().contiguous()  # noqa: E731
        color_c, rgbs_c = color.detach().float().contiguous(), rgbs.detach().float().contiguous()
        assert ge is None or ge.numel() == 1  # NeuconWLoss.__call__ reduces anything else first
        ge_c, me_c, sfm_c = f(ge), f(me), f(sfm)
        R = color_c.shape[0]
        n_me, n_sfm = (0 if me_c is None else me_c.numel()), (0 if sfm_c is None else sfm_c.numel())
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        L.check(L.get_lib().ncw_loss_fwd(L.ptr(color_c), L.ptr(rgbs_c), R, L.ptr(ge_c), L.ptr(me_c), n_me, L.ptr(sfm_c), n_sfm,
                                         w[0], w[1], w[2], w[3], L.ptr(loss), L.stream_ptr(dev)), "ncw_loss_fwd")
        ctx.keep = (color_c, rgbs_c, R, n_me, n_sfm, w, ge is not None, None if me is None else me.shape,
                    None if sfm is None else sfm.shape)
        ctx.set_materialize_grads(False)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, d_loss):
        color_c, rgbs_c, R, n_me, n_sfm, w, has_ge, me_shape, sfm_shape = ctx.keep
        if d_loss is None:
            return None, None, None, None, None, None
        dev = color_c.device
        d_loss = d_loss.reshape(1).float().contiguous()
        d_color = torch.empty_like(color_c)
        d_ge = torch.empty(1, device=dev) if has_ge else None
        d_me = torch.empty(n_me, device=dev) if n_me else None
        d_sfm = torch.empty(n_sfm, device=dev) if n_sfm else None
        L.check(L.get_lib().ncw_loss_bwd(L.ptr(d_loss), L.ptr(color_c), L.ptr(rgbs_c), R, n_me, n_sfm, w[0], w[1], w[2], w[3],
                                         L.ptr(d_color), L.ptr(d_ge), L.ptr(d_me), L.ptr(d_sfm), L.stream_ptr(dev)), "ncw_loss_bwd")
        return (d_color, None, d_ge, None if d_me is None else d_me.reshape(me_shape),
                None if d_sfm is None else d_sfm.reshape(sfm_shape), None)


class NeuconWLoss:
    """Same constructor keywords as the reference's NeuconWLoss (losses.py:11); `config` may be the reference's CfgNode, our
    config dict (neuralrecon_w_amd.config) or None (then `use_mask` / `use_depth` decide which terms exist)."""

    def __init__(self, coef=1, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, floor_weight=0.01, config=None,
                 use_mask=True, use_depth=True):
        self.coef, self.igr_weight, self.mask_weight, self.depth_weight = float(coef), float(igr_weight), float(mask_weight), float(depth_weight)
        if config is not None:
            n = config["NEUCONW"] if isinstance(config, dict) else config.NEUCONW
            get = (lambda k: n[k]) if isinstance(n, dict) else (lambda k: getattr(n, k))
            use_mask, use_depth = get("MESH_MASK_LIST") is not None, bool(get("DEPTH_LOSS"))
        self.use_mask, self.use_depth = bool(use_mask), bool(use_depth)

    def __call__(self, inputs, targets):
        if not inputs["color"].is_cuda:
            raise L.NeuconwHipError("NeuconWLoss: tensors are not on a GPU; the hot path has no CPU fallback")
        w = (self.coef, self.igr_weight, self.mask_weight, self.depth_weight)
        ge = inputs["gradient_error"]
        if ge.numel() != 1:  # losses.py:29 takes .mean(); render() returns the batch-global scalar (renderer.py:763-765),
            ge = ge.mean()   # any other shape is reduced here by torch (with its autograd) before the fused launch
        return _LossFn.apply(inputs["color"], targets, ge, inputs["mask_error"] if self.use_mask else None,
                             inputs["sfm_depth_loss"] if self.use_depth else None, w)

    def terms(self, inputs, targets):
        """The reference's return value (a dict of weighted terms), in plain torch."""
        R = targets.shape[0]
        ret = {"color_loss": (inputs["color"] - targets).abs().sum() / (R + 1e-5),
               "normal_loss": self.igr_weight * inputs["gradient_error"].mean()}
        if self.use_mask:
            ret["mask_error"] = self.mask_weight * inputs["mask_error"].mean()
        if self.use_depth:
            ret["sfm_depth_loss"] = self.depth_weight * inputs["sfm_depth_loss"].mean()
        return {k: self.coef * v for k, v in ret.items()}
