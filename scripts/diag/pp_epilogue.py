"""GPU: the round-4 epilogue experiment on the plain fp16 `sdf_infer` (csrc/ncw_pp.hip sdf_inferC, W = 256): the f32 Softplus
epilogue against the two packed-fp16 forms (NCW_PP_EPI = f32 | pk16 | poly16, read once per process, so one subprocess per
variant).  Per variant: ms per 131,072 and per 1,048,576 points (HIP events, median of 20), algorithmic TFLOP/s and fraction of
the 2.5 PFLOP/s dense peak (SDF value chain: 459,008 MAC per point), max |sdf - fp64 oracle| on 8,192 points.

The packed variants exist only in a PROBE library (the product build has the f32 epilogue alone):
    NCW_BUILD_TAG=epi python -m neuralrecon_w_amd.build && NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_epi.so \
    python scripts/diag/pp_epilogue.py            (driver)      python scripts/diag/pp_epilogue.py --one   (one variant)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one():
    import torch

    import neuralrecon_w_amd as nw
    from oracle import neuconw_oracle as O
    from tests._parity import perturb_weights

    torch.manual_seed(0)
    net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                        geometric_init=True, weight_norm=True, inside_outside=False)

    class Holder(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.sdf_net = n

    perturb_weights(Holder(net), 0.1, 0.0)
    net = net.cuda()
    net.sdf_split = False  # the PLAIN 16-bit value chain (the split path has its own kernels)
    res = {"epi": os.environ.get("NCW_PP_EPI", "f32")}
    g = torch.Generator().manual_seed(1)
    for N in (131072, 1048576):
        x = torch.randn(N, 3, generator=g)
        x = (x / x.norm(dim=-1, keepdim=True) * torch.rand(N, 1, generator=g) ** (1 / 3)).float()
        xc = x.cuda()
        for prec, pname in ((nw.PREC_F16, "f16"), (nw.PREC_BF16, "bf16")):
            for _ in range(5):
                s = net.sdf(xc, prec)
            ms = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                s = net.sdf(xc, prec)
                e1.record()
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            ms.sort()
            t = ms[len(ms) // 2]
            tf = 2.0 * 459008 * N / (t * 1e-3) / 1e12
            row = {"ms": round(t, 4), "min_ms": round(ms[0], 4), "tflops": round(tf, 1), "frac_mfma": round(tf / 2500.0, 4)}
            if N == 131072:
                M = 8192
                sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
                ref, _, _ = O.sdf_net(sd, x[:M].double(), "sdf_net.", with_grad=False)
                row["max_abs_err_vs_fp64"] = float((s.reshape(-1)[:M].cpu().double() - ref).abs().max())
            res["%s_%d" % (pname, N)] = row
    print(json.dumps(res))


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
    else:
        for epi in ("f32", "pk16", "poly16"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, NCW_PP_EPI=epi),
                               capture_output=True, text=True, cwd=ROOT)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(lines[-1] if lines else "FAILED %s: %s" % (epi, r.stderr[-1500:]), flush=True)
