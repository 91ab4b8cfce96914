import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
for W in (256, 512):
    net = nw.SDFNetwork(d_in=3, d_out=W+1, d_hidden=W, n_layers=8, skip_in=(4,)).to(dev)
    x = (torch.rand(131072, 3, device=dev) * 2 - 1)
    macs = 39*W + 6*W*W + W*(W-39) + W*W + W  # sdf-only (approx real MACs)
    for prec, name in ((nw.PREC_BF16, "bf16"), (nw.PREC_F32, "f32")):
        for _ in range(3): net.sdf(x, prec=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): net.sdf(x, prec=prec)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("W=%d %s: %.3f ms  %.1f TFLOP/s  %.1f Mpts/s" % (W, name, ms, 2*macs*131072/ms/1e9, 131072/ms/1e3))
