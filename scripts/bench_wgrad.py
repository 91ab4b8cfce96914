"""Time the merged weight-gradient launch of the bench-shape step under different split-K plans."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import neuralrecon_w_amd as nw
stash = importlib.import_module(nw.NeuconWRenderer.__module__.rsplit(".", 1)[0] + ".stash")
dev = torch.device("cuda:0")
emb, neuconw, nerf, rdr = bench.build_models(dev, nw.PREC_BF16)
rays, ts, label, rgbs = bench.synth_batch(1024, 1000, dev)
out = rdr.render(rays, ts, label, background_rgb=torch.zeros(1, 3, device=dev), cos_anneal_ratio=0.0)
bench.loss_fn(out, rgbs).backward()
torch.cuda.synchronize()
wb = [e["wgrad_batch"] for e in neuconw.sdf_net._stash_cache._e.values() if e.get("wgrad_batch") is not None][0]
gb = stash.WgradBatch.algorithmic_bytes(wb.items) / 1e9
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for tgt in (256, 384, 512, 768, 1024, 1536, 2048):
    stash.WgradBatch.TARGET_WGS = tgt
    wb.__dict__.pop("_groups", None)
    ms = timeit(wb.run)
    print(f"target={tgt} wgs={wb._groups[0][4]} {ms:.3f} ms  {gb/ms:.2f} TB/s")
# one launch per sub-list (the previous plan): SDF+colour big, small, NeRF big, small with a uniform ksplit 16
subs = {}
for it in wb.items:
    subs.setdefault((it[7], it[1] > 4), []).append(it)
batches = []
for (n, big), items in subs.items():
    b = stash.WgradBatch(dev, nw.PREC_BF16, n); b.items = items
    b._groups = [b._table(items, 1 if big else 0, [16] * len(items), 16, n)]
    batches.append(b)
ms = timeit(lambda: [b.run() for b in batches])
print(f"4 launches, uniform ksplit 16: {ms:.3f} ms {gb/ms:.2f} TB/s")
for b in batches:
    ms = timeit(b.run)
    g = stash.WgradBatch.algorithmic_bytes(b.items) / 1e9
    print(f"   n={b.n} products={len(b.items)} wgs={b._groups[0][4]} {ms:.3f} ms {g/ms:.2f} TB/s")
