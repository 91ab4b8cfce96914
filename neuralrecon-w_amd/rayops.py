"""Thin python wrappers over the per-ray C-ABI kernels (sampler + compositor).  Tensors in,
tensors out; every call is asynchronous on torch's current HIP stream."""
import torch

from . import lib as L


def _f(t):
    return t.contiguous().float()


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise L.NeuconwHipError("hot-path tensors must live on the GPU (no CPU fallback)")


def sample_coarse(near, far, s_near, s_far, n_samples, n_outside, rand_shift=None, rand_out=None):
    """renderer.py:488-514 -> z [R,n], z_out [R,O] (or None), sample_dist [R,1]."""
    _chk_cuda(near, far)
    R = near.shape[0]
    dev = near.device
    near, far, s_near, s_far = (_f(t).reshape(-1) for t in (near, far, s_near, s_far))
    z = torch.empty(R, n_samples, device=dev)
    zo = torch.empty(R, max(n_outside, 0), device=dev)
    sd = torch.empty(R, device=dev)
    rs = _f(rand_shift).reshape(-1) if rand_shift is not None else None
    ro = _f(rand_out) if rand_out is not None else None
    lib = L.get_lib()
    L.check(lib.ncw_sample_coarse(L.ptr(near), L.ptr(far), L.ptr(s_near), L.ptr(s_far), R, n_samples, n_outside,
                                  L.ptr(rs), L.ptr(ro), L.ptr(z), L.ptr(zo), L.ptr(sd), L.stream_ptr(dev)),
            "ncw_sample_coarse")
    return z, (zo if n_outside > 0 else None), sd.reshape(R, 1)


def upsample(rays_o, rays_d, z, sdf, n_new, inv_s):
    """renderer.py:257-341 + sample_pdf(det=True) -> z_new [R, n_new]."""
    _chk_cuda(z, sdf)
    R, n = z.shape
    out = torch.empty(R, n_new, device=z.device)
    lib = L.get_lib()
    L.check(lib.ncw_upsample(L.ptr(_f(rays_o)), L.ptr(_f(rays_d)), L.ptr(_f(z)), L.ptr(_f(sdf)), R, n, float(inv_s),
                             n_new, L.ptr(out), L.stream_ptr(z.device)), "ncw_upsample")
    return out


def sort_merge(a, b, pa=None, pb=None):
    """stable sort(cat([a, b], -1)) per ray (+ payload permuted alike): renderer.py:343-363."""
    _chk_cuda(a, b)
    R, na = a.shape
    nb = b.shape[1]
    out = torch.empty(R, na + nb, device=a.device)
    pout = torch.empty(R, na + nb, device=a.device) if pa is not None else None
    lib = L.get_lib()
    L.check(lib.ncw_sort_merge(L.ptr(_f(a)), na, L.ptr(_f(b)), nb, L.ptr(_f(pa) if pa is not None else None),
                               L.ptr(_f(pb) if pb is not None else None), R, L.ptr(out), L.ptr(pout),
                               L.stream_ptr(a.device)), "ncw_sort_merge")
    return out, pout


def boundary(near, far, z, nb):
    """renderer.py:549-565 -> [R, nb] boundary samples (unsorted)."""
    _chk_cuda(z)
    R, n = z.shape
    out = torch.empty(R, nb, device=z.device)
    lib = L.get_lib()
    L.check(lib.ncw_boundary(L.ptr(_f(near).reshape(-1)), L.ptr(_f(far).reshape(-1)), L.ptr(_f(z)), n, R, nb,
                             L.ptr(out), L.stream_ptr(z.device)), "ncw_boundary")
    return out


class CompositeCtx:
    """Holds the (contiguous) inputs of one compositor call so forward and backward see the same
    buffers.  All tensors f32 on the GPU."""

    def __init__(self, rays_o, rays_d, z, sample_dist, sdf, grad, rgb, inv_s, cos_anneal, z_feed=None, density=None,
                 bg_rgb=None, background_rgb=None, trim_sphere=True):
        _chk_cuda(z, sdf, grad, rgb)
        self.dev = z.device
        self.R, self.S = z.shape
        self.has_bg = z_feed is not None
        self.O = (z_feed.shape[1] - self.S) if self.has_bg else 0
        self.t = dict(rays_o=_f(rays_o), rays_d=_f(rays_d), z=_f(z), z_feed=_f(z_feed) if self.has_bg else None,
                      sample_dist=_f(sample_dist).reshape(-1), sdf=_f(sdf), grad=_f(grad), rgb=_f(rgb),
                      density=_f(density) if self.has_bg else None, bg_rgb=_f(bg_rgb) if self.has_bg else None,
                      inv_s=_f(inv_s).reshape(1),
                      background_rgb=_f(background_rgb).reshape(3) if background_rgb is not None else None)
        # cos_anneal: python float, or a 1-element device tensor (read by the kernels at run time: graph replay)
        self.t["cos_anneal_dev"] = _f(cos_anneal).reshape(1) if torch.is_tensor(cos_anneal) else None
        s = L.NcwCompositeIn()
        for k, v in self.t.items():
            setattr(s, k, v.data_ptr() if v is not None else 0)
        s.cos_anneal = 0.0 if torch.is_tensor(cos_anneal) else float(cos_anneal)
        s.R, s.S, s.O, s.has_bg, s.trim_sphere = self.R, self.S, self.O, int(self.has_bg), int(bool(trim_sphere))
        self.cin = s

    def forward(self):
        R, S, M, dev = self.R, self.S, self.S + self.O, self.dev
        o = dict(color=torch.empty(R, 3, device=dev), color_sphere=torch.empty(R, 3, device=dev),
                 color_bg=torch.empty(R, 3, device=dev), weights=torch.empty(R, M, device=dev),
                 weights_sum=torch.empty(R, device=dev), cdf=torch.empty(R, S, device=dev),
                 inside=torch.empty(R, S, device=dev), depth=torch.empty(R, device=dev),
                 normals=torch.empty(R, 3, device=dev), eik=torch.empty(2, R, device=dev),
                 mid_z=torch.empty(R, S, device=dev), dists=torch.empty(R, S, device=dev),
                 bg_alpha=torch.empty(R, M, device=dev) if self.has_bg else None,
                 weights_max=torch.empty(R, device=dev))
        s = L.NcwCompositeOut()
        for k, v in o.items():
            setattr(s, k, v.data_ptr() if v is not None else 0)
        lib = L.get_lib()
        L.check(lib.ncw_composite_fwd(self.cin, s, L.stream_ptr(dev)), "ncw_composite_fwd")
        return o

    def backward(self, d_color, d_weights_sum, d_depth, d_eik_num, grad_scale=1.0, grad_scale_dev=None):
        R, S, M, dev = self.R, self.S, self.S + self.O, self.dev
        z = lambda t, shape: _f(t).reshape(shape) if t is not None else torch.zeros(shape, device=dev)  # noqa: E731
        ups = dict(d_color=z(d_color, (R, 3)), d_weights_sum=z(d_weights_sum, (R,)), d_depth=z(d_depth, (R,)),
                   d_eik_num=z(d_eik_num, (R,)))
        g = dict(d_sdf=torch.empty(R, S, device=dev), d_grad=torch.empty(R, S, 3, device=dev),
                 d_rgb=torch.empty(R, S, 3, device=dev),
                 d_density=torch.empty(R, M, device=dev) if self.has_bg else None,
                 d_bg_rgb=torch.empty(R, M, 3, device=dev) if self.has_bg else None,
                 d_inv_s=torch.empty(R, device=dev))
        s = L.NcwCompositeGrad()
        for k, v in list(ups.items()) + list(g.items()):
            setattr(s, k, v.data_ptr() if v is not None else 0)
        s.grad_scale = float(grad_scale)  # every returned adjoint carries this factor (fp16 loss scaling)
        # ... times this device scalar (the dynamic loss scale the optimiser adapts, trainer.FlatAdam), if given
        s.grad_scale_dev = grad_scale_dev.data_ptr() if grad_scale_dev is not None else 0
        lib = L.get_lib()
        L.check(lib.ncw_composite_bwd(self.cin, s, L.stream_ptr(dev)), "ncw_composite_bwd")
        self._keep = ups
        return g  # d_inv_s: per-ray terms [R]; the caller reduces them in a fixed order (ncw_inv_s_bwd / .sum())
