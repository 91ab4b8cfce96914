"""GPU: the composed train step vs the fp64 oracle at the operating points NeuS actually trains at:
variance in {0.3, 0.5, 0.6, 0.7} (inv_s 20 / 148 / 403 / 1097) x {sphere-like init, weight_v jittered 5 %} x
{f32, f16, bf16}, W = 256, 64 + 64 samples (the headline shape), R rays.

    python scripts/diag/trained_point_parity.py [--R 16] [--json out.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import neuralrecon_w_amd as nw  # noqa: E402
from tests._parity import run_case  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--ns", type=int, default=64)
ap.add_argument("--ni", type=int, default=64)
ap.add_argument("--W", type=int, default=256)
ap.add_argument("--json", default=None)
ap.add_argument("--precs", default="f32,f16,bf16")
args = ap.parse_args()
P = {"f32": nw.PREC_F32, "f16": nw.PREC_F16, "bf16": nw.PREC_BF16}
rows = []
for v_jit in (0.0, 0.05):
    for variance in (0.3, 0.5, 0.6, 0.7):
        for pn in args.precs.split(","):
            r = run_case(args.W, args.ns, args.ni, P[pn], args.R, variance=variance, v_jit=v_jit)
            row = dict(prec=pn, variance=variance, inv_s=round(r["inv_s"]), v_jit=v_jit, loss_err=abs(r["loss"] - r["loss_ref"]),
                       grad_worst=r["grad_worst"], **r["errs"])
            rows.append(row)
            print("v_jit %.2f variance %.1f (inv_s %4d) %-4s colour %.1e depth %.1e wsum %.1e eik %.1e loss %.1e grads %.1e"
                  % (v_jit, variance, row["inv_s"], pn, row["color"], row["depth"], row["weights_sum"], row["gradient_error"],
                     row["loss_err"], row["grad_worst"]), flush=True)
if args.json:
    json.dump(rows, open(args.json, "w"), indent=1)
