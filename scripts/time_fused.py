"""Time the SDF fused entry points alone (bench-shape SDF net, 131072 points) through the module API.
Safe to run against the timing-experiment library variants (garbage weights only make garbage numbers)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
from neuralrecon_w_amd.neuconw import points_struct
dev = torch.device("cuda:0")
W, n = 256, 131072
net = nw.SDFNetwork(d_in=3, d_out=W + 1, d_hidden=W, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                    geometric_init=True, weight_norm=True, inside_outside=False).to(dev)
x = (torch.rand(n, 3, device=dev) * 2 - 1) * 0.9
w_sdf, w_grad = torch.randn(n, device=dev), torch.randn(n, 3, device=dev)
pts = points_struct(x=x)
prec = nw.PREC_BF16
def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
ctx = [None]
def fwd():
    from neuralrecon_w_amd.stash import StashCache
    if ctx[0] is not None: StashCache.release(ctx[0]["lease"])
    ctx[0] = net.fwd_stash(pts, n, prec)[2]
def bwd():
    net.bwd_stash(ctx[0], w_sdf, w_grad)
t_inf = timeit(lambda: net.sdf(x, prec=prec))
t_fwd = timeit(fwd)
t_bwd = timeit(bwd)
print("sdf_infer %.3f ms   sdf_fwd %.3f ms   sdf_bwd %.3f ms" % (t_inf, t_fwd, t_bwd))
