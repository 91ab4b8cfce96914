"""GPU: which ATen kernels / copies the timed TrainStep still launches besides the C-ABI kernels, with the Python line that issues each
(torch.profiler, 3 profiled steps after warm-up).  VERDICT r4 weak 10: the torch glue inside the timed step."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import neuralrecon_w_amd as nw  # noqa: E402

dev = torch.device("cuda", 0)
emb, neuconw, nerf, rdr = bench.build_models(dev, nw.PREC_F16)
rdr.bg_dense = "--elim" not in sys.argv
train = nw.TrainStep(rdr, [emb, neuconw, nerf], bench.loss_fn, lr=1e-4, eps=1e-7, clip=0.99)
rays, ts, label, rgbs = bench.synth_batch(1024, 1000, dev)
bg = torch.zeros(1, 3, device=dev)
for i in range(5):
    train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(3):
        train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and len(ev.kernels) > 0:
        st = [s for s in ev.stack if "neuralrecon" in s or "bench.py" in s or "losses" in s]
        rows.append((ev.name, tuple(str(s) for s in ev.input_shapes)[:3], st[0] if st else (ev.stack[0] if ev.stack else "?"),
                     sum(k.duration for k in ev.kernels), [k.name[:50] for k in ev.kernels]))
import collections  # noqa: E402

agg = collections.OrderedDict()
for name, shp, where, dur, ks in rows:
    key = (name, where.split("/")[-1][:90])
    a = agg.setdefault(key, [0, 0.0, shp, ks])
    a[0] += 1
    a[1] += dur
print("ATen ops that launch device work inside the step (per step = count / 3):")
for (name, where), (n, dur, shp, ks) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-28s x%.1f/step  %6.1f us/step  %-50s %s  %s" % (name, n / 3.0, dur / 3.0, where, shp, ks[0]))
