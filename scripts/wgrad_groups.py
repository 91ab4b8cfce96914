"""Print the weight-gradient launch groups of one bench-shape training step (products, workgroups, split-K)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import neuralrecon_w_amd as nw
stash = importlib.import_module(nw.NeuconWRenderer.__module__.rsplit(".", 1)[0] + ".stash")
orig = stash.WgradBatch.run
def run(self):
    orig(self)
    for tile, tab, pre, nd, wgs, ks, n in self._groups:
        rb = [(it[1], it[3], it[7]) for it in self.items]
        print(f"n={self.n} tile={tile} products={nd} wgs={wgs} ksplit={ks} wgs/product={pre.cpu().diff().tolist()} shapes={rb}")
stash.WgradBatch.run = run
dev = torch.device("cuda:0")
emb, neuconw, nerf, rdr = bench.build_models(dev, nw.PREC_BF16)
rays, ts, label, rgbs = bench.synth_batch(1024, 1000, dev)
out = rdr.render(rays, ts, label, background_rgb=torch.zeros(1, 3, device=dev), cos_anneal_ratio=0.0)
bench.loss_fn(out, rgbs).backward()
torch.cuda.synchronize()
