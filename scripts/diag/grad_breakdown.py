"""GPU: per-parameter gradient error of one composed step against the fp64 oracle (tests/_parity.run_case), the largest
entries first -- which tensor carries `grad_worst`.

    python scripts/diag/grad_breakdown.py [--train_steps 40 --variance 0.6 --prec f16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--train_steps", type=int, default=40)
ap.add_argument("--variance", type=float, default=0.6)
ap.add_argument("--prec", default="f16")
ap.add_argument("--R", type=int, default=16)
ap.add_argument("--v_jit", type=float, default=0.0)
ap.add_argument("--reps", type=int, default=1)
args = ap.parse_args()
import neuralrecon_w_amd as nw  # noqa: E402
from tests._parity import run_case  # noqa: E402

prec = {"f32": nw.PREC_F32, "bf16": nw.PREC_BF16, "f16": nw.PREC_F16}[args.prec]
for rep in range(args.reps):
    r = run_case(256, 64, 64, prec, args.R, variance=args.variance, train_steps=args.train_steps, v_jit=args.v_jit)
    print("env %s rep %d: errs %s  grad_worst %.3e" % ({k: v for k, v in os.environ.items() if k.startswith("NEUCONW_")}, rep,
                                                                         {k: "%.2e" % v for k, v in r["errs"].items()}, r["grad_worst"]))
    for k, e in sorted(r["grad_errs"].items(), key=lambda kv: -kv[1])[:8]:
        print("    %-58s %.3e" % (k, e))
