import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
print(torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,)).to(dev)
for n in (128, 128 * 64, 128 * 128, 128 * 256, 128 * 257, 128 * 512, 128 * 768, 128 * 1024, 128 * 2048, 128 * 4096):
    x = (torch.rand(n, 3, device=dev) * 2 - 1)
    for _ in range(5): net.sdf(x, prec=nw.PREC_BF16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k = 50
    e0.record()
    for _ in range(k): net.sdf(x, prec=nw.PREC_BF16)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / k
    print("n=%7d (%5d WGs): %.4f ms  %.0f TFLOP/s" % (n, n // 128, ms, 2 * 459008 * n / ms / 1e9))
