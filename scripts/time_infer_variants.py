"""Time ncw_sdf_infer variants (NCW_SDF_INFER8 = 2: weights-stationary, 3: ping-pong) in subprocesses + check parity."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, torch
sys.path.insert(0, %r)
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
torch.manual_seed(0)
W = 256
net = nw.SDFNetwork(d_in=3, d_out=W+1, d_hidden=W, n_layers=8, skip_in=(4,)).to(dev)
x = (torch.rand(131072, 3, device=dev) * 2 - 1)
ref = net.sdf(x, prec=nw.PREC_F32)
got = net.sdf(x, prec=nw.PREC_BF16)
err = float((got - ref).abs().max() / ref.abs().max())
macs = 459008
for _ in range(5): net.sdf(x, prec=nw.PREC_BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n): net.sdf(x, prec=nw.PREC_BF16)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("variant %%s: %%.4f ms  %%.1f TFLOP/s  rel err vs f32 %%.2e" %% (os.environ.get("NCW_SDF_INFER8"), ms, 2*macs*131072/ms/1e9, err))
''' % ROOT
for v in sys.argv[1:] or ["2", "3"]:
    env = dict(os.environ, NCW_SDF_INFER8=v)
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-1500:])
