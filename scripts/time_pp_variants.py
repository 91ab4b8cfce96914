"""Time sdf_inferC (NCW_SDF_INFER8=3) through the library variants built by scripts/pp_variants.sh; reports the
error against the fp32 kernel too (experiment variants compute garbage: ignore their error)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, torch
sys.path.insert(0, %r)
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,)).to(dev)
with torch.no_grad():
    for n_, p in net.named_parameters():
        if n_.endswith("weight_g"): p.mul_(1.0 + 0.1 * torch.randn_like(p))
x = (torch.rand(131072, 3, device=dev) * 2 - 1)
ref = net.sdf(x, prec=nw.PREC_F32)
got = net.sdf(x, prec=nw.PREC_BF16)
err = float((got - ref).abs().max() / ref.abs().max())
for _ in range(5): net.sdf(x, prec=nw.PREC_BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n): net.sdf(x, prec=nw.PREC_BF16)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print("%%-10s %%.4f ms  %%.0f TFLOP/s   rel err vs f32 %%.2e" %% (os.environ.get("PP_TAG"), ms, 2*459008*131072/ms/1e9, err))
''' % ROOT
for tag in sys.argv[1:]:
    v = "3"
    if tag.startswith("B:"):
        v, tag = "2", tag[2:]
    env = dict(os.environ, NCW_SDF_INFER8=v, PP_TAG=("B:" if v == "2" else "") + tag,
               NEUCONW_HIP_LIB=os.path.join(ROOT, "neuralrecon-w_amd", "libneuconw_hip_%s.so" % tag))
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-800:])
