"""Where do the tolerances at TRAINED sharpness come from?  From the reference itself.

At inv_s = exp(10 variance) in the hundreds the hierarchical sampler (searchsorted over sigmoid(sdf * 1024) weights,
rendering/renderer.py:15-48,257-341) turns 1e-7 SDF differences into a moved sample on single rays, so two fp32-accurate
evaluations of the same function differ by far more than 1e-4 there.  scripts/diag/port_over_reference.py ran the REAL
reference (fp32, CPU) and the fp64 oracle on the EXACT inputs of tests/test_gpu_fullsize.py::test_train_step_vs_oracle_at_
trained_operating_points (profiles/r04/port_over_reference.json, `at_test_inputs`):

    (variance, v_jit)   reference fp32 vs oracle fp64: colour / depth / weights_sum ; worst parameter gradient
    (0.5, 0)            8.6e-6 / 2.3e-5 / 9.6e-6 ; 1.1e-3
    (0.6, 0)            5.7e-4 / 7.8e-4 / 6.5e-4 ; 2.1e-3
    (0.7, 0)            1.4e-4 / 1.9e-4 / 1.6e-4 ; 2.5e-2
    (0.6, 0.05)         8.8e-7 / 8.0e-7 / 8.5e-7 ; 2.6e-4

i.e. the reference's own arithmetic is 6-8e-4 away from the exact result on these rays at inv_s 403, and its gradients 2e-3 ..
2.5e-2.  The GPU tolerances (TRAINED_TOL) must sit at or above that floor and not far above it; this file checks that against the
committed measurement and, where /root/reference exists, re-measures one point live."""
import json
import os

import pytest
import torch

from tests._util import ROOT

FLOOR = os.path.join(ROOT, "profiles", "r04", "port_over_reference.json")


def _floor_rows():
    with open(FLOOR) as f:
        return {(r["variance"], r["v_jit"]): r for r in json.load(f)["at_test_inputs"]}


def test_trained_tolerances_follow_the_reference_floor():
    import importlib

    T = importlib.import_module("tests.test_gpu_fullsize").TRAINED_TOL
    rows = _floor_rows()
    assert set(rows) == set(T)
    for key, r in rows.items():
        e = r["reference_fp32_vs_oracle_fp64"]
        floor_out = max(e["color"], e["depth"], e["weights_sum"])
        floor_grad = r["param_grad_worst_rel_to_network_max"]
        for prec in ("f32", "f16"):
            tol_out, tol_grad, _ = T[key][prec]
            # never tighter than what the reference itself achieves on these inputs, never looser than 1e-4 (the north-star
            # bar) or 2.5x the reference's own deviation, whichever is larger
            assert tol_out >= min(floor_out, 1e-4), (key, prec, tol_out, floor_out)
            assert tol_out <= max(1.2e-4, 2.5 * floor_out), (key, prec, tol_out, floor_out)
            assert tol_grad <= max(4e-3, 2.5 * floor_grad), (key, prec, tol_grad, floor_grad)
    # the point of the file: at inv_s 403 on the sphere SDF the REFERENCE misses 1e-4 by a wide margin
    assert max(rows[(0.6, 0.0)]["reference_fp32_vs_oracle_fp64"][k] for k in ("color", "depth", "weights_sum")) > 4e-4


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not mounted")
def test_live_reference_fp32_is_off_the_fp64_oracle_at_inv_s_403():
    """Re-measures the (0.6, 0) row forward-only with the real reference (about 10 s): same number as the committed file."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden as G
    from oracle import neuconw_oracle as O
    from oracle import ref_import
    from tests._build import build_system, state_dict_cpu
    from tests._parity import CFG, perturb_weights
    from tests._util import rel_err, synth_rays

    ns = ref_import.load()
    e_, n_, f_, _ = build_system(W=256, n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128, seed=5, device="cpu", prec=0,
                                 n_samples=64, n_importance=64)
    perturb_weights(n_, 0.1, 0.0)
    with torch.no_grad():
        n_.deviation_network.variance.fill_(0.6)
    sd = state_dict_cpu(e_, n_, f_, torch.float32)
    emb, neuconw, nerf, rdr = G.build_reference(ns, 256, 8, (4,), n_a=48, n_vocab=100, nerf_w=256, color_hidden=256, head=128,
                                                seed=0, n_samples=64, n_importance=64)
    with torch.no_grad():
        emb.weight.copy_(sd["embedding_a.weight"])
        neuconw.load_state_dict({k[len("neuconw."):]: v for k, v in sd.items() if k.startswith("neuconw.")}, strict=False)
        nerf.load_state_dict({k[len("nerf."):]: v for k, v in sd.items() if k.startswith("nerf.")})
    r, t, lb, c = synth_rays(16, 77, 100)
    out = rdr.render(r.clone(), t, lb, perturb_overwrite=0, background_rgb=torch.zeros(1, 3), cos_anneal_ratio=0.3)
    with torch.no_grad():
        sd64 = {k: v.double() for k, v in sd.items()}
        o64 = O.render(sd64, dict(CFG, n_samples=64, n_importance=64), r.double(), t, lb, 0.3, torch.zeros(1, 3, dtype=torch.float64))
    errs = {k: rel_err(out[k].detach(), o64[k]) for k in ("color", "depth", "weights_sum")}
    print(errs)
    want = _floor_rows()[(0.6, 0.0)]["reference_fp32_vs_oracle_fp64"]
    for k in errs:
        assert 0.5 * want[k] < errs[k] < 2.0 * want[k], (k, errs[k], want[k])
    assert max(errs.values()) > 4e-4
