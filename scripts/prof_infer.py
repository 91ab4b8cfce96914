import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,)).to(dev)
x = (torch.rand(131072, 3, device=dev) * 2 - 1)
for _ in range(5): net.sdf(x, prec=nw.PREC_BF16)
torch.cuda.synchronize()
