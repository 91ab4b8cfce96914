"""Config 5: on-chip grid generation + SDF sweep against the oracle; shard/chunk invariance."""
import pytest
import torch

from tests._util import rel_err

pytestmark = pytest.mark.gpu


def test_sdf_grid_vs_oracle_and_shard_invariance():
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import grid
    from oracle import neuconw_oracle as O
    from tests.test_gpu_sdf import _mk

    net = _mk(64, 8, (4,), seed=2)
    dim = 21
    bmin, bmax, origin, radius = (-1.2, -0.9, -1.0), (1.1, 1.0, 0.8), (0.05, -0.02, 0.01), 1.3
    g = grid.sdf_grid(net, dim, bmin, bmax, origin, radius, prec=nw.PREC_F32).cpu()
    lin = [torch.linspace(bmin[a], bmax[a], dim) for a in range(3)]
    xx, yy, zz = torch.meshgrid(*lin, indexing="ij")  # utils/visualization.py:46-50
    pts = (torch.stack([xx, yy, zz], -1).reshape(-1, 3) - torch.tensor(origin)) / radius
    sd = {"sdf_net." + k: v.detach().cpu() for k, v in net.state_dict().items()}
    ref = O.sdf_net(sd, pts, with_grad=False)[0].reshape(dim, dim, dim)
    assert rel_err(g, ref) < 1e-4
    # 3 ragged shards + small chunks == one sweep (bit-exact: a point's value does not depend on its tile mates)
    total = dim ** 3
    parts = []
    for r in range(3):
        s, c, per = grid.local_range(total, r, 3)
        parts.append(grid.sdf_grid_range(net, dim, bmin, bmax, s, c, origin, radius, prec=nw.PREC_F32, chunk=1000))
    assert torch.equal(torch.cat(parts).cpu(), g.reshape(-1))


@pytest.mark.parametrize("W", [256, 512])
def test_sdf_grid_default_precision_at_real_widths(W):
    """Config 5 at the widths and in the precision it ships with: `grid.sdf_grid` with prec=None -> SDFNetwork.value_prec() =
    the split-precision fp16 value chain at W = 256 / 512 (csrc/ncw_split.hip / ncw_sdf16.hip through NcwPoints mode 3: the
    lattice point from the linear index) against the fp64 oracle on a 48^3 lattice of an off-centre box, with ragged 3-way
    shards and small chunks (utils/visualization.py:27-89).  Bound: 5e-6 like test_split_precision_value_path."""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import grid
    from oracle import neuconw_oracle as O
    from tests.test_gpu_sdf import _mk

    net = _mk(W, 8, (4,), seed=3)
    assert net.value_prec() == nw.PREC_F16 and net.split_value(nw.PREC_F16)
    dim = 48
    bmin, bmax, origin, radius = (-1.2, -0.9, -1.0), (1.1, 1.0, 0.8), (0.05, -0.02, 0.01), 1.3
    g = grid.sdf_grid(net, dim, bmin, bmax, origin, radius).cpu()   # prec=None: the product default
    lin = [torch.linspace(bmin[a], bmax[a], dim, dtype=torch.float64) for a in range(3)]
    xx, yy, zz = torch.meshgrid(*lin, indexing="ij")  # utils/visualization.py:46-50
    pts = (torch.stack([xx, yy, zz], -1).reshape(-1, 3) - torch.tensor(origin, dtype=torch.float64)) / radius
    sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    ref = O.sdf_net(sd, pts, with_grad=False)[0].reshape(dim, dim, dim)
    e = rel_err(g, ref)
    e_abs = float((g.double() - ref).abs().max())
    # the exact-fp32 kernels on the same lattice, and the plain fp16 chain the default replaced
    e32 = rel_err(grid.sdf_grid(net, dim, bmin, bmax, origin, radius, prec=nw.PREC_F32).cpu(), ref)
    net.sdf_split = False
    e16 = rel_err(grid.sdf_grid(net, dim, bmin, bmax, origin, radius, prec=nw.PREC_F16).cpu(), ref)
    net.sdf_split = None
    print("sdf_grid W=%d 48^3 default precision: rel %.2e abs %.2e  (exact-fp32 kernels %.2e, plain fp16 %.2e)" % (W, e, e_abs, e32, e16))
    assert e < 5e-6 and e16 > 20 * e
    total = dim ** 3
    parts = []
    for r in range(3):  # ragged: 110592 = 3 x 36864, chunk 5000 is not a multiple of the 64 / 96 / 128-point workgroups
        s, c, per = grid.local_range(total + 0, r, 3)
        parts.append(grid.sdf_grid_range(net, dim, bmin, bmax, s, c, origin, radius, chunk=5000))
    assert torch.equal(torch.cat(parts).cpu(), g.reshape(-1))
    # an uneven split (world 5: the last shard is short) assembles the same lattice
    parts = [grid.sdf_grid_range(net, dim, bmin, bmax, *grid.local_range(total, r, 5)[:2], origin, radius) for r in range(5)]
    assert torch.equal(torch.cat(parts).cpu(), g.reshape(-1))
