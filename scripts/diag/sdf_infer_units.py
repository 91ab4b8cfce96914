"""GPU: the value-only SDF kernels after the move to t-units (round 6): time per launch (HIP events on the launch
stream), accuracy against the fp64 oracle and -- with --pmc -- VALU / MFMA instruction counts of sdf_inferC from rocprofv3.

    python scripts/diag/sdf_infer_units.py [--pmc] [--one]        (--one: a single 1 M-point bf16 launch, for the counter passes)
"""
import collections
import csv
import glob
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(W):
    import neuralrecon_w_amd as nw
    from tests._parity import perturb_weights

    torch.manual_seed(0)
    net = nw.SDFNetwork(d_in=3, d_out=W + 1, d_hidden=W, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1, geometric_init=True,
                        weight_norm=True, inside_outside=False)

    class Holder(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.sdf_net = n

    perturb_weights(Holder(net), 0.1, 0.02)
    return net.cuda()


def points(N):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, 3, generator=g)
    return (x / x.norm(dim=-1, keepdim=True) * torch.rand(N, 1, generator=g) ** (1 / 3)).float()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2], t[0]


def main():
    import neuralrecon_w_amd as nw
    from oracle import neuconw_oracle as O

    if "--one" in sys.argv:
        net = build(256)
        x = points(1 << 20).cuda()
        net.sdf_split = False
        for _ in range(3):
            net.sdf(x, nw.PREC_BF16)
        torch.cuda.synchronize()
        return
    for W in (256, 512):
        net = build(W)
        sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
        for N in (131072, 1 << 20):
            x = points(N)
            xc = x.cuda()
            M = 8192
            ref = O.sdf_net(sd, x[:M].double(), "sdf_net.", with_grad=False)[0]
            for name, prec, split in (("bf16 plain", nw.PREC_BF16, False), ("f16 plain", nw.PREC_F16, False), ("f16 split", nw.PREC_F16, True)):
                net.sdf_split = split
                s = net.sdf(xc, prec).reshape(-1)
                err = float((s[:M].cpu().double() - ref).abs().max())
                med, best = timed(lambda: net.sdf(xc, prec))
                flop = 2.0 * N * sum(p.shape[0] * p.shape[1] for n_, p in net.named_parameters() if n_.endswith("weight_v") or n_.endswith(".weight"))
                print("W=%d %-10s %8d points: %.4f ms (best %.4f)  %.0f TFLOP/s algorithmic = %.3f of 2.5 PFLOP/s   max|sdf - fp64| %.2e"
                      % (W, name, N, med, best, flop / med / 1e9, flop / med / 1e9 / 2500.0, err), flush=True)
    if "--pmc" in sys.argv:
        for gi, grp in enumerate(["SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES", "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"]):
            d = "/tmp/sdf_units_pmc_%d" % gi
            subprocess.run(["rocprofv3", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                            os.path.abspath(__file__), "--one"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True)
            acc = collections.defaultdict(lambda: [0.0, 0])
            for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if "sdf_inferC" in row["Kernel_Name"]:
                        a = acc[row["Counter_Name"]]
                        a[0] += float(row["Counter_Value"])
                        a[1] += 1
            c = {k: v[0] / max(v[1], 1) for k, v in acc.items()}
            print("sdf_inferC, 1,048,576 points bf16, per launch:", {k: "%.3e" % v for k, v in c.items()}, flush=True)
            if c.get("SQ_INSTS_MFMA"):
                print("   SQ_INSTS_VALU / SQ_INSTS_MFMA = %.2f   SQ_INSTS_LDS / SQ_INSTS_MFMA = %.2f   (round 5, z-units: 8.8 / 1.29)"
                      % (c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"], c.get("SQ_INSTS_LDS", 0) / c["SQ_INSTS_MFMA"]), flush=True)
            if c.get("SQ_WAVE_CYCLES"):
                print("   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.2f   matrix-pipe busy = %.2f   (round 5: 0.33 / 0.43)"
                      % (c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                         c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (c.get("SQ_BUSY_CYCLES", 1) / 8.0)), flush=True)


if __name__ == "__main__":
    main()
