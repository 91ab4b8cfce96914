"""Does the default training precision TRAIN like the fp32 parity mode?  

A fixed 'teacher' scene (the benchmark networks from one seed, SDF weight_v jittered so that the surface is not the
geometric initialisation's sphere, variance 0.5 = inv_s 148) renders the target colours of a pool of rays once, in fp32.
A 'student' (another seed) is then trained on random batches of that pool with the reference's recipe (TrainStep: render +
NeuconWLoss + backward + clip 0.99 + Adam eps 1e-7, cos-anneal) -- once per precision, same seeds, same batches -- and the
held-out image error (PSNR of the rendered colours against the teacher's on rays never trained on), the loss, the variance
network's inv_s and the number of skipped steps are logged along the way.

    python scripts/diag/train_equivalence.py --steps 1500 --out gpurun_out/train_equivalence.json
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import neuralrecon_w_amd as nw  # noqa: E402
from tests._parity import perturb_weights  # noqa: E402


def rays_pool(n, seed, dev):
    rays, ts, label, _ = bench.synth_batch(n, seed, dev)
    ts = ts % 16  # a handful of appearance codes, each seen often
    label = torch.zeros_like(label)
    return rays, ts, label


@torch.no_grad()
def render_colors(rdr, rays, ts, label, chunk=2048):
    out = []
    bg = torch.zeros(1, 3, device=rays.device)
    for i in range(0, rays.shape[0], chunk):
        with torch.enable_grad():
            o = rdr.render(rays[i:i + chunk], ts[i:i + chunk], label[i:i + chunk], perturb_overwrite=0, background_rgb=bg,
                           cos_anneal_ratio=1.0)
        out.append(o["color"].detach())
    return torch.cat(out)


def psnr(a, b):
    return -10.0 * math.log10(float(((a - b) ** 2).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--pool", type=int, default=65536)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--precs", default="f32,f16,bf16")
    ap.add_argument("--log-every", type=int, default=100)
    ap.add_argument("--out", default="")
    ap.add_argument("--shipped", action="store_true", help="the shipped yaml's shape: SDF 8 x 512, 8 + 16 samples per ray (bench.py --config shipped)")
    args = ap.parse_args()
    if args.shipped:  # where the forward-only refinements differ most from the backward's function (adjoint sweep with both operands and phi' as pairs)
        bench.__dict__.update(W_SDF=512, N_SAMPLES=8, N_IMPORTANCE=16)
    dev = torch.device("cuda:0")
    # ---- teacher: targets in fp32 ---------------------------------------------------------------------------------
    emb_t, neuconw_t, nerf_t, rdr_t = bench.build_models(dev, nw.PREC_F32, seed=100)
    perturb_weights(neuconw_t, 0.0, 0.04, seed=3)
    with torch.no_grad():
        neuconw_t.deviation_network.variance.fill_(0.5)
    rays, ts, label = rays_pool(args.pool, 4242, dev)
    target = render_colors(rdr_t, rays, ts, label)
    rays_v, ts_v, label_v = rays_pool(8192, 777, dev)
    target_v = render_colors(rdr_t, rays_v, ts_v, label_v)
    del rdr_t, emb_t, neuconw_t, nerf_t
    result = {"steps": args.steps, "rays": args.rays, "pool": args.pool, "lr": args.lr, "shape": "shipped (W = 512, 8 + 16)" if args.shipped else "headline (W = 256, 64 + 64)",
              "runs": {}}
    for name in args.precs.split(","):
        prec = {"f32": nw.PREC_F32, "f16": nw.PREC_F16, "bf16": nw.PREC_BF16, "f16_noextras": nw.PREC_F16,
                "f32b": nw.PREC_F32}[name]  # f32b: fp32 again with ANOTHER batch order -- the run-to-run spread the others are read against
        emb, neuconw, nerf, rdr = bench.build_models(dev, prec, seed=7)
        if name == "f16_noextras":
            # the forward-only refinements of the fp16 mode OFF (tests/_parity.py run_case(forward_extras=False)): the forward
            # then computes exactly the function the single-rounded fp16 backward differentiates.  Set before the first forward.
            neuconw.color_net.ray_bias = False
            neuconw.color_net.weight_split = False
            neuconw.color_net.act_split = False
            nerf.ray_bias = False
            nerf.refine = False
            neuconw.sdf_net.adj_split = False
            if hasattr(neuconw.sdf_net, "tangent"):
                neuconw.sdf_net.tangent = False
        step_fn = nw.TrainStep(rdr, [emb, neuconw, nerf], bench.loss_fn, lr=args.lr, eps=1e-7, clip=0.99)
        g = torch.Generator(device=dev)
        g.manual_seed(100 if name == "f32b" else 99)
        bg = torch.zeros(1, 3, device=dev)
        log = []
        t0 = time.perf_counter()
        for it in range(args.steps + 1):
            if it % args.log_every == 0 or it == args.steps:
                pv = psnr(render_colors(rdr, rays_v, ts_v, label_v), target_v)
                inv_s = float(torch.exp(neuconw.deviation_network.variance.detach() * 10.0))
                log.append({"step": it, "psnr_heldout": round(pv, 3), "inv_s": round(inv_s, 4),
                            "variance": float(neuconw.deviation_network.variance.detach()),
                            "loss": None if it == 0 else round(float(loss.detach()), 5),
                            "skipped": int(getattr(step_fn.opt, "skipped_steps", 0)),
                            "loss_scale": float(rdr.grad_scale) if name.startswith("f16") else None})
                print("%s step %5d  held-out PSNR %.3f dB  inv_s %.1f  loss %s  skipped %d" %
                      (name, it, pv, inv_s, log[-1]["loss"], log[-1]["skipped"]), flush=True)
            if it == args.steps:
                break
            idx = torch.randint(0, args.pool, (args.rays,), device=dev, generator=g)
            loss, out = step_fn(rays[idx], ts[idx], label[idx], target[idx], background_rgb=bg,
                                cos_anneal_ratio=min(1.0, it / float(args.steps)))
        torch.cuda.synchronize()
        result["runs"][name] = {"log": log, "seconds": round(time.perf_counter() - t0, 2)}
        del step_fn, rdr, emb, neuconw, nerf
        torch.cuda.empty_cache()
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)
    ref = result["runs"].get("f32")
    if ref:
        for name, r in result["runs"].items():
            d = [abs(a["psnr_heldout"] - b["psnr_heldout"]) for a, b in zip(r["log"], ref["log"])]
            print("%s: final held-out PSNR %.3f dB (fp32 %.3f), max |dPSNR| along the run %.3f dB, %s s" %
                  (name, r["log"][-1]["psnr_heldout"], ref["log"][-1]["psnr_heldout"], max(d), r["seconds"]))


if __name__ == "__main__":
    main()
