"""End-to-end GPU parity of NeuconWRenderer.render (+ loss + backward) through the C ABI against the
golden vectors produced by the REAL reference, in the fp32 parity mode."""
import pytest
import torch

from tests._build import build_system, load_golden_weights, loss_from_outputs, named_params
from tests._util import load_golden, rel_err

pytestmark = pytest.mark.gpu

PER_SAMPLE = ("weights", "weights_max", "cdf_fine", "gradients")


@pytest.mark.parametrize("name,ns,ni,perturb", [("render_w64_det", 16, 16, False),
                                                ("render_w64_perturb", 16, 16, True),
                                                ("render_w64_shipped_shape", 8, 16, False)])
def test_render_matches_reference_golden(name, ns, ni, perturb):
    import neuralrecon_w_amd as nw

    sd, grads, outs, m = load_golden(name)
    emb, neuconw, nerf, rdr = build_system(prec=nw.PREC_F32, n_samples=ns, n_importance=ni)
    load_golden_weights(sd, emb, neuconw, nerf)
    emb, neuconw, nerf = emb.cuda(), neuconw.cuda(), nerf.cuda()
    rand = (m["rand_shift"].cuda(), m["rand_out"].cuda()) if perturb else None
    out = rdr.render(m["rays"].cuda(), m["ts"].cuda(), m["label"].cuda(), perturb_overwrite=1 if perturb else 0,
                     background_rgb=torch.zeros(1, 3).cuda(), cos_anneal_ratio=0.25, _rand=rand)
    for k, ref in outs.items():
        got = out[k].detach().cpu()
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        tol = 5e-4 if k in PER_SAMPLE else 1e-4  # see tests/test_oracle_golden.py for the per-sample bound
        assert rel_err(got, ref) < tol, (k, rel_err(got, ref))
    loss = loss_from_outputs(out, m["rgbs"].cuda())
    assert abs(float(loss) - float(m["loss"])) < 2e-5
    loss.backward()
    params = named_params(emb, neuconw, nerf)
    worst = 0.0
    for k, gref in grads.items():
        g = params[k].grad
        assert g is not None, k
        e = rel_err(g.cpu(), gref)
        worst = max(worst, e)
        assert e < 1e-3, (k, e)
    print(name, "worst param-grad rel err vs reference: %.2e" % worst)
    # parameters the reference never touches must stay untouched here too
    assert neuconw.xyz_encoding_final.weight.grad is None
