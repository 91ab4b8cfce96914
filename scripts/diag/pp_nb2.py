"""GPU, PROBE library only: the plain bf16 `sdf_infer` (csrc/ncw_pp.hip sdf_inferC, W = 256) with ONE output block per wave (8 waves,
the product) against TWO (NCW_PP_NB=2: four 512-register waves, every B fragment read from LDS feeds two MFMAs -- half the LDS
reads per MFMA), the file compiled with `-mllvm -amdgpu-mfma-vgpr-form` so that the accumulators stay in VGPRs and the weight slices
sit in AGPRs as MFMA srcA (round 2 measured NB = 2 WITHOUT that flag: 0.188 vs 0.165 ms -- accumulators in AGPRs, one
v_accvgpr_read per epilogue value).  VERDICT r4 item 5.

    NCW_BUILD_TAG=nb2 NCW_FLAGS2="-mllvm -amdgpu-mfma-vgpr-form" NCW_FILES2=ncw_pp.hip python -m neuralrecon_w_amd.build
    NEUCONW_HIP_LIB=neuralrecon-w_amd/libneuconw_hip_nb2.so python scripts/diag/pp_nb2.py
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

    torch.manual_seed(0)
    net = nw.SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                        geometric_init=True, weight_norm=True, inside_outside=False)
    with torch.no_grad():
        for n_, p_ in net.named_parameters():
            if n_.endswith("weight_g"):
                p_.mul_(1.0 + 0.1 * torch.randn_like(p_))
    net = net.cuda()
    res = {"nb": os.environ.get("NCW_PP_NB", "1")}
    g = torch.Generator().manual_seed(1)
    for N in (131072, 1048576):
        x = torch.randn(N, 3, generator=g)
        x = (x / x.norm(dim=-1, keepdim=True) * torch.rand(N, 1, generator=g) ** (1 / 3)).float()
        xc = x.cuda()
        for _ in range(5):
            s = net.sdf(xc, nw.PREC_BF16)
        ms = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s = net.sdf(xc, nw.PREC_BF16)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        t = ms[len(ms) // 2]
        res["ms_%d" % N] = round(t, 4)
        res["frac_mfma_%d" % N] = round(2.0 * 459008 * N / (t * 1e-3) / 2.5e15, 4)
    xs = x[:8192]
    sd = {"sdf_net." + k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    ref = O.sdf_net(sd, xs.double(), with_grad=False)[0]
    res["max_abs_err_vs_fp64"] = float((net.sdf(xs.cuda(), nw.PREC_BF16).cpu()[:, 0].double() - ref).abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    if "--one" in sys.argv:
        one()
    else:
        for nb in ("1", "2", "1", "2"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, NCW_PP_NB=nb),
                               capture_output=True, text=True)
            ls = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(ls[-1] if ls else ("FAILED nb=%s: " % nb) + r.stderr[-400:], flush=True)
