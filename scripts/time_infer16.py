"""sdf_infer at W = 512, 16-bit plain path: time per launch at several sizes (A/B: NCW_PP16=0 -> burst kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
W = 512
net = nw.SDFNetwork(d_in=3, d_out=W + 1, d_hidden=W, n_layers=8, skip_in=(4,)).to(dev)
net.sdf_split = False
macs = 39 * W + 6 * W * W + W * (W - 39) + W
for n in (24576, 49152, 131072, 1048576):
    x = torch.rand(n, 3, device=dev) * 2 - 1
    for prec, name in ((nw.PREC_BF16, "bf16"), (nw.PREC_F16, "f16")):
        for _ in range(3):
            net.sdf(x, prec=prec)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        k = 20
        e0.record()
        for _ in range(k):
            net.sdf(x, prec=prec)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / k
        print("n=%8d %s: %.4f ms  %.0f TFLOP/s (%.1f %% of 2.5 PF)  PP16=%s" % (n, name, ms, 2 * macs * n / ms / 1e9, 2 * macs * n / ms / 1e9 / 25, os.environ.get("NCW_PP16", "1")))
