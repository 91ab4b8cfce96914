"""GPU: the fp16 mode (prec = PREC_F16: the bf16 kernels compiled a second time with the 16-bit type switched, csrc/ncw_common.h)
and its loss scale (NeuconWRenderer.loss_scale, device {scale, 1/scale} -> NcwCompositeGrad.grad_scale_dev ->
NcwUnpackDesc.grad_mul_dev; adapted by trainer.FlatAdam / ncw_adam_step_dev).  Parity against
the oracle is in test_gpu_fullsize.py / test_gpu_sdf.py / test_gpu_sdf_train.py; here: what the loss scale is for, that it
cancels, that training in fp16 follows training in fp32, and the optimiser's non-finite guard."""
import pytest
import torch

from tests._build import build_system, loss_from_outputs, named_params
from tests._util import rel_err, synth_rays

pytestmark = pytest.mark.gpu


def _grads(prec, scale=None, R=64, seed=5):
    emb, neuconw, nerf, rdr = build_system(seed=seed, prec=prec)
    if scale is not None:
        rdr.grad_scale = scale
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(R, seed=11, n_vocab=64)]
    out = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=torch.zeros(1, 3, device="cuda"),
                     cos_anneal_ratio=0.4)
    loss = loss_from_outputs(out, rgbs)
    loss.backward()
    return float(loss), {k: p.grad.detach().clone() for k, p in named_params(emb, neuconw, nerf).items() if p.grad is not None}


def _worst(ga, gb):
    """largest parameter-gradient difference in units of the largest gradient of the same tensor"""
    return max(rel_err(ga[k], gb[k]) for k in ga if float(gb[k].abs().max()) > 0)


def test_loss_scale_cancels_and_is_needed():
    import neuralrecon_w_amd as nw

    l32, g32 = _grads(nw.PREC_F32)
    l_a, g_a = _grads(nw.PREC_F16, 1024.0)
    l_b, g_b = _grads(nw.PREC_F16, 4096.0)
    l_1, g_1 = _grads(nw.PREC_F16, 1.0)
    assert abs(l_a - l32) < 2e-4 and l_a == l_b == l_1  # the forward does not see the scale
    for g in (g_a, g_b, g_1):
        assert all(bool(torch.isfinite(v).all()) for v in g.values())
    e_a, e_b, e_1 = _worst(g_a, g32), _worst(g_b, g32), _worst(g_1, g32)
    print("fp16 parameter gradients vs fp32: scale 1024 %.2e, 4096 %.2e, unscaled %.2e" % (e_a, e_b, e_1))
    # a power-of-two scale only moves the exponent: the two scaled runs agree with fp32 (and with each other) alike ...
    # (measured 7.6e-2 / 7.7e-2 on the worst tensor of this W = 64, 8 + 8 sample system -- few samples, sharp sigmoid -- and
    # 0.49 unscaled)
    assert e_a < 0.15 and e_b < 0.15 and abs(e_a - e_b) < 0.02
    # ... and without it the per-point adjoints (1 / (3 R) of an O(1) residual, times the compositing weights) sit in
    # fp16's subnormals: the gradients lose most of their bits
    assert e_1 > 3 * max(e_a, e_b)


def test_fp16_training_follows_fp32():
    """20 optimiser steps from the same initial parameters on the same rays.  Step 1 sees the same parameters, so its loss
    is the fp32 mode's to 2e-4 (measured 6.8e-5); afterwards Adam turns every ~0 gradient whose sign the rounding decides into a +-lr
    step, so the trajectories separate (measured: max |f16 - f32| 5.1e-3 .. 8.7e-3, |bf16 - f32| 6.3e-3 .. 8.2e-3 over 20
    steps on this W = 64 system, run to run) while the curves stay together: the fp16 run must descend like the fp32 run."""
    import neuralrecon_w_amd as nw

    R, steps = 128, 20
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(R, seed=21, n_vocab=64)]
    bg = torch.zeros(1, 3, device="cuda")
    curves = {}
    for name, prec in (("f32", nw.PREC_F32), ("f16", nw.PREC_F16), ("bf16", nw.PREC_BF16)):
        emb, neuconw, nerf, rdr = build_system(seed=8, prec=prec)
        train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99)
        curves[name] = [float(train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.05 * i,
                                    perturb_overwrite=0)[0]) for i in range(steps)]
    d16 = max(abs(a - b) for a, b in zip(curves["f16"], curves["f32"]))
    dbf = max(abs(a - b) for a, b in zip(curves["bf16"], curves["f32"]))
    print("loss f32 %.5f -> %.5f; max |f16 - f32| %.2e, max |bf16 - f32| %.2e" % (curves["f32"][0], curves["f32"][-1], d16, dbf))
    assert curves["f32"][-1] < curves["f32"][0] - 0.1
    assert abs(curves["f16"][0] - curves["f32"][0]) < 2e-4
    assert d16 < 3e-2, (d16, curves)
    assert abs(curves["f16"][-1] - curves["f32"][-1]) < 3e-2


def test_adam_step_skips_a_non_finite_gradient_norm():
    """A skipped step writes nothing, does not advance Adam's bias-correction step (device counter), and halves the loss
    scale; `growth_interval` clean steps double it again (torch.cuda.amp.GradScaler's policy)."""
    from neuralrecon_w_amd.trainer import FlatAdam, FlatParams

    torch.manual_seed(0)
    lin = torch.nn.Linear(1000, 1, bias=False).cuda()
    fp = FlatParams([lin])
    scale = torch.tensor([1024.0, 1.0 / 1024.0], device="cuda")
    opt = FlatAdam(fp, lr=1e-2, eps=1e-7, clip=0.99, loss_scale=scale, growth_interval=2)
    fp.flat_grad.copy_(torch.randn_like(fp.flat_grad))
    opt.step()
    before = fp.flat.detach().clone()
    m_before = opt.exp_avg.clone()
    fp.flat_grad.copy_(torch.randn_like(fp.flat_grad))
    fp.flat_grad[17] = float("inf")  # what one overflowed fp16 adjoint does to the norm
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(fp.flat, before) and torch.equal(opt.exp_avg, m_before)  # nothing written
    assert opt.step_count == 1 and opt.skipped_steps == 1
    assert scale.tolist() == [512.0, 1.0 / 512.0]
    fp.flat_grad.copy_(torch.randn_like(fp.flat_grad))
    fp.flat_grad[3] = float("nan")
    opt.step()
    assert opt.step_count == 1 and opt.skipped_steps == 2 and float(scale[0]) == 256.0
    for _ in range(2):
        fp.flat_grad.copy_(torch.randn_like(fp.flat_grad))
        opt.step()
    assert bool(torch.isfinite(fp.flat).all()) and not torch.equal(fp.flat, before)
    assert opt.step_count == 3 and scale.tolist() == [512.0, 1.0 / 512.0]  # two clean steps: doubled once


def test_dynamic_loss_scale_recovers_from_an_overflowing_scale():
    """fp16 TrainStep started with a loss scale that overflows fp16 (2^24 on O(1e-4) adjoints): the first steps come
    out non-finite and are skipped while the device-resident scale halves; training then proceeds.  No host sync is
    involved: the compositor backward and the weight-norm backward read the scale from the device."""
    import neuralrecon_w_amd as nw
    from tests._build import build_system, loss_from_outputs
    from tests._util import synth_rays

    emb, neuconw, nerf, rdr = build_system(seed=3, prec=nw.PREC_F16)
    rdr.sync_free = True
    rdr.grad_scale = 2.0 ** 40
    train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99)
    assert train.opt.loss_scale.data_ptr() == rdr.loss_scale.tensor(train.fp.flat.device).data_ptr()
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(64, seed=12, n_vocab=64)]
    bg = torch.zeros(1, 3, device="cuda")
    p0 = train.fp.flat.detach().clone()
    losses = []
    for i in range(60):
        loss, _ = train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.0, perturb_overwrite=0)
        losses.append(float(loss))
    skipped, applied, final = train.opt.skipped_steps, train.opt.step_count, rdr.grad_scale
    print("skipped %d applied %d final scale 2^%d; loss %.4f -> %.4f" % (skipped, applied, round(__import__("math").log2(final)),
                                                                         losses[0], losses[-1]))
    assert skipped >= 5 and applied >= 10 and skipped + applied == 60
    assert final < 2.0 ** 36 and bool(torch.isfinite(train.fp.flat).all()) and not torch.equal(train.fp.flat, p0)
    assert losses[-1] < losses[0]
