"""Which torch ops launch the small kernels of a bench step (torch.profiler, one step, grouped by op + shapes + python caller)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import neuralrecon_w_amd as nw
dev = torch.device("cuda:0")
emb, neuconw, nerf, rdr = bench.build_models(dev, nw.PREC_BF16)
train = nw.TrainStep(rdr, [emb, neuconw, nerf], bench.loss_fn, lr=1e-4, eps=1e-7, clip=0.99)
rays, ts, label, rgbs = bench.synth_batch(1024, 1000, dev)
bg = torch.zeros(1, 3, device=dev)
for i in range(3): train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    dt = getattr(ev, "self_device_time_total", 0)
    if dt > 0 and ev.key.startswith("aten::"):
        st = [f for f in (ev.stack or []) if "/root/repo" in f or "bench.py" in f]
        rows.append((ev.count, ev.key, str(ev.input_shapes)[:50], dt, (st[0].split("repo/")[-1][:80] if st else "?")))
tot = 0
for n, name, shp, dt, where in sorted(rows, key=lambda r: -r[0]):
    print("%3d  %-24s %-52s %7.1f us  %s" % (n, name, shp, dt, where)); tot += n
print("total aten ops with GPU time:", tot)
