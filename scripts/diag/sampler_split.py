"""GPU: does the SAMPLER need the split-precision SDF, or only the final evaluation?  The composed step vs the fp64 oracle with
the sampler's SDF queries in plain fp16 (sdf_split off during sparse_sampler) and everything else as the product runs it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neuralrecon_w_amd as nw  # noqa: E402
from neuralrecon_w_amd import renderer as R  # noqa: E402
from tests._parity import run_case  # noqa: E402

orig = R.NeuconWRenderer._sdf_rays


def plain_sampler(self, rays_o, rays_d, z):
    net = self.neuconw.sdf_net
    old = net.__dict__.get("sdf_split")
    net.sdf_split = False
    try:
        return orig(self, rays_o, rays_d, z)
    finally:
        net.sdf_split = old


for variance in (0.3, 0.5, 0.6, 0.7):
    for name, fn in (("split sampler", orig), ("plain sampler", plain_sampler)):
        R.NeuconWRenderer._sdf_rays = fn
        r = run_case(256, 64, 64, nw.PREC_F16, 64, variance=variance, with_grads=False)
        print("variance %.1f %-14s" % (variance, name), {k: "%.2e" % v for k, v in r["errs"].items()}, flush=True)
R.NeuconWRenderer._sdf_rays = orig
