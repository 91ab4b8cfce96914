"""GPU: trainer.TrainStep (flat parameters, one fill / norm / Adam per step, gradients written straight into
the flat buffer by the C-ABI weight-norm backward) follows the stock per-tensor recipe of the reference
(Adam eps 1e-7 + clip 0.99, train.py:61) step for step."""
import pytest
import torch

from tests._build import build_system, loss_from_outputs, named_params
from tests._util import synth_rays

pytestmark = pytest.mark.gpu


def test_train_step_matches_per_tensor_recipe():
    import neuralrecon_w_amd as nw

    R, steps = 64, 4
    sys_a = build_system(seed=5, prec=nw.PREC_F32)
    sys_b = build_system(seed=5, prec=nw.PREC_F32)
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(R, seed=11, n_vocab=64)]
    bg = torch.zeros(1, 3, device="cuda")
    # a) stock: per-tensor Adam, zero_grad(set_to_none), clip over the parameter list
    emb, neuconw, nerf, rdr = sys_a
    params = [p for m in (emb, neuconw, nerf) for p in m.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, eps=1e-7)
    losses_a, first_a = [], None
    for i in range(steps):
        opt.zero_grad(set_to_none=True)
        out = rdr.render(rays, ts, label, perturb_overwrite=0, background_rgb=bg, cos_anneal_ratio=0.1 * i)
        loss = loss_from_outputs(out, rgbs)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.99)
        opt.step()
        losses_a.append(float(loss))
        if i == 0:
            first_a = {k: p.detach().clone() for k, p in named_params(emb, neuconw, nerf).items()}
    # b) flat
    emb2, neuconw2, nerf2, rdr2 = sys_b
    keys = list(neuconw2.state_dict()) + list(nerf2.state_dict())
    train = nw.TrainStep(rdr2, [emb2, neuconw2, nerf2], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99)
    assert list(neuconw2.state_dict()) + list(nerf2.state_dict()) == keys
    losses_b, first_b = [], None
    for i in range(steps):
        loss, _ = train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1 * i, perturb_overwrite=0)
        losses_b.append(float(loss))
        if i == 0:
            first_b = {k: p.detach().clone() for k, p in named_params(emb2, neuconw2, nerf2).items()}
    assert rdr2.flat_grad_buffer().data_ptr() == train.fp.flat_grad.data_ptr()  # renderer writes into the flat buffer
    for a, b in zip(losses_a, losses_b):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (losses_a, losses_b)
    # Step 1: both runs start from the same parameters and the fp32 mode is bitwise reproducible
    # (tests/test_gpu_repro.py), so they see IDENTICAL gradients; what separates them is the last bit of torch's
    # per-tensor clip / Adam vs the flat C-ABI update (3e-7 per step, test_flat_adam_matches_torch_adam).
    worst1 = max(float((first_a[k] - first_b[k]).abs().max()) for k in first_a)
    assert worst1 < 2e-6, worst1
    # Steps 2..4: the 1e-7 parameter differences re-enter the gradients, and Adam turns a ~0 gradient whose SIGN
    # they decide into a +-lr-sized step (the round-1 flake: 3.9e-5 on one element).  A bookkeeping error (stale packed
    # weights, wrong step count, a lost gradient slice) moves the BULK by >= 1e-4: bound the distribution, not the max.
    pa, pb = named_params(emb, neuconw, nerf), named_params(emb2, neuconw2, nerf2)
    diffs = torch.cat([(pa[k] - pb[k]).abs().reshape(-1) for k in pa])
    assert float(diffs.max()) < 1e-3, float(diffs.max())
    assert float(diffs.median()) < 2e-6, float(diffs.median())
    assert float((diffs > 2e-5).float().mean()) < 2e-3, float((diffs > 2e-5).float().mean())


def test_captured_step_replays_like_eager():
    """TrainStep(capture=True): HIP-graph replay of the whole step follows the eager step, including the
    per-step cos_anneal_ratio (a device scalar inside the graph) and Adam's bias correction."""
    import neuralrecon_w_amd as nw

    R, steps = 64, 7
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(R, seed=12, n_vocab=64)]
    bg = torch.zeros(1, 3, device="cuda")
    runs = []
    for capture in (False, True):
        emb, neuconw, nerf, rdr = build_system(seed=6, prec=nw.PREC_F32)
        rdr.sync_free = True
        train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99,
                             capture=capture, capture_warmup=3)
        losses = []
        for i in range(steps):
            loss, out = train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.15 * i,
                              perturb_overwrite=0)
            losses.append(float(loss))
        if capture:
            assert train._graphs is not None  # steps 3.. were graph replays
        assert train.native  # both: FlatAdam through the C ABI; its step counter / bias corrections live on the device
        nstep = train.opt.step_count
        runs.append((losses, named_params(emb, neuconw, nerf), float(out["color"].abs().sum()), nstep))
    (la, pa, ca, na), (lb, pb, cb, nb) = runs
    assert na == nb == steps  # Adam's bias correction advanced once per replay
    for a, b in zip(la, lb):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (la, lb)
    assert la[0] != la[-1]
    # The weight-gradient atomics are order-dependent in the last bit and Adam turns the sign of a ~0
    # gradient into a +-lr step, so single elements may differ by a fraction of lr; a bookkeeping error (stale
    # packed weights, a wrong step count: >= 16 % of EVERY update, i.e. >= 1e-4 here) would move the bulk.
    diffs = torch.cat([(pa[k] - pb[k]).abs().reshape(-1) for k in pa])
    assert float(diffs.max()) < 1e-3, float(diffs.max())
    assert float(diffs.median()) < 5e-6, float(diffs.median())
    assert float((diffs > 1e-4).float().mean()) < 2e-3, float((diffs > 1e-4).float().mean())
    assert abs(ca - cb) <= 1e-4 * max(1.0, abs(ca))


def test_flat_adam_matches_torch_adam():
    """ncw_adam_step_dev (clip + Adam, step state on the device) against clip_grad_norm_ + torch.optim.Adam on the same
    flat data."""
    from neuralrecon_w_amd.trainer import FlatAdam, FlatParams

    torch.manual_seed(0)
    n = 100003  # ragged: not a multiple of 4
    lin = torch.nn.Linear(n, 1, bias=False).cuda()
    ref = torch.nn.Parameter(lin.weight.detach().clone())
    fp = FlatParams([lin])
    opt_n = FlatAdam(fp, lr=3e-3, eps=1e-7, clip=0.99)
    opt_t = torch.optim.Adam([ref], lr=3e-3, eps=1e-7)
    for i in range(6):
        g = torch.randn(1, n, device="cuda") * (10.0 if i % 2 else 1e-3)  # clipped and unclipped steps
        fp.flat_grad.copy_(g.reshape(-1))
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 0.99)
        opt_t.step()
        opt_n.step()
        assert torch.allclose(fp.flat_grad.view_as(ref), ref.grad, rtol=1e-6, atol=0)  # clipped gradient written back
        assert torch.allclose(lin.weight, ref, rtol=0, atol=3e-7), float((lin.weight - ref).abs().max())
    assert opt_n.step_count == 6
    opt_t2 = FlatAdam(fp, lr=3e-3, eps=1e-7, clip=None)  # no clipping: NULL norm pointer
    fp.flat_grad.fill_(0.5)
    before = fp.flat.detach().clone()
    opt_t2.step()
    assert torch.allclose(fp.flat.detach(), before - 3e-3, atol=1e-6)  # first Adam step = -lr * sign(g)


def test_captured_step_follows_lr_schedule():
    """The learning rate is device data (FlatAdam.lr_dev): assigning `opt.lr` between HIP-graph replays -- how
    scripts/train.py applies the per-epoch cosine / steplr schedule (train.py:21-25 `get_scheduler`) -- changes the replayed
    update.  With lr = 0 from the first replay on, the parameters must stop moving while Adam's step counter keeps
    advancing; restoring it moves them again."""
    import neuralrecon_w_amd as nw

    R = 64
    rays, ts, label, rgbs = [t.cuda() for t in synth_rays(R, seed=13, n_vocab=64)]
    bg = torch.zeros(1, 3, device="cuda")
    emb, neuconw, nerf, rdr = build_system(seed=7, prec=nw.PREC_F32)
    train = nw.TrainStep(rdr, [emb, neuconw, nerf], loss_from_outputs, lr=1e-3, eps=1e-7, clip=0.99, capture=True,
                         capture_warmup=2)
    for i in range(4):  # 2 eager + capture/replay + 1 replay at lr 1e-3
        train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1, perturb_overwrite=0)
    assert train._graphs is not None
    p0 = train.fp.flat.detach().clone()
    train.opt.lr = 0.0
    for i in range(2):
        train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1, perturb_overwrite=0)
    assert torch.equal(train.fp.flat.detach(), p0), "a replay ignored opt.lr = 0: the capture baked the lr in"
    assert train.opt.step_count == 6
    train.opt.lr = 2e-3
    train(rays, ts, label, rgbs, background_rgb=bg, cos_anneal_ratio=0.1, perturb_overwrite=0)
    moved = float((train.fp.flat.detach() - p0).abs().max())
    assert 1e-4 < moved <= 2e-3 * 1.5, moved  # an Adam step is <= ~lr per element
    assert abs(train.opt.lr - 2e-3) < 1e-12 and abs(float(train.opt.lr_dev) - 2e-3) < 1e-9


@pytest.mark.parametrize("n", [4, 1001, 262144, 2088895])
def test_grad_norm_launch_matches_torch_and_rearms(n):
    """ncw_grad_norm (FlatAdam's total_norm, train.py:61 clip_grad_norm_): one launch, fixed order -- equal to torch's norm to fp32
    rounding, bitwise repeatable, re-armed after every launch, non-finite entries propagate."""
    import ctypes as C

    from neuralrecon_w_amd import lib as L

    lib = L.get_lib()
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n + 4, generator=g) * 3e-3).cuda()[:n]  # (a 16-byte aligned view)
    scratch = torch.zeros(int(lib.ncw_grad_norm_scratch_floats()), device="cuda")
    out = torch.empty(1, device="cuda")
    vals = []
    for _ in range(3):
        L.check(lib.ncw_grad_norm(L.ptr(x), n, L.ptr(scratch), L.ptr(out), L.stream_ptr(x.device)), "ncw_grad_norm")
        vals.append(float(out))
    ref = float(torch.linalg.vector_norm(x.double()))
    assert vals[0] == vals[1] == vals[2] and abs(vals[0] - ref) <= 2e-6 * ref, (vals, ref)
    x[n // 2] = float("inf")
    L.check(lib.ncw_grad_norm(L.ptr(x), n, L.ptr(scratch), L.ptr(out), L.stream_ptr(x.device)), "ncw_grad_norm")
    assert float(out) == float("inf")
    x[n // 2] = float("nan")
    L.check(lib.ncw_grad_norm(L.ptr(x), n, L.ptr(scratch), L.ptr(out), L.stream_ptr(x.device)), "ncw_grad_norm")
    assert float(out) != float(out)


def test_merged_pack_and_unpack_equal_per_plan_launches():
    """packing.pack_many / unpack_many (the three networks' tables concatenated: ONE ncw_pack_weights, ONE ncw_unpack_grads per step)
    against one launch per plan: bitwise the same arenas / parameter gradients."""
    import neuralrecon_w_amd as nw
    from neuralrecon_w_amd import packing

    emb, neuconw, nerf, rdr = build_system(seed=3, prec=nw.PREC_F16)
    mods = [neuconw.sdf_net, neuconw.color_net, nerf]
    plans = [m.plan(rdr.prec) for m in mods]
    for p in plans:
        p.pack()
    ref = [(p.w_arena.clone(), p.b_arena.clone()) for p in plans]
    for p in plans:
        p.w_arena.zero_()
        p.b_arena.zero_()
        p.packed_version = None
    packing.pack_many(list(zip(mods, plans)))
    for p, (w, b), m in zip(plans, ref, mods):
        assert torch.equal(p.w_arena, w) and torch.equal(p.b_arena, b)
        assert p.packed_version == (m._param_version(), p.param_key())  # module.packed() will not launch again
    g = torch.Generator(device="cuda").manual_seed(1)
    for p in plans:
        p.g_arena.copy_(torch.randn(p.g_arena.shape, device="cuda", generator=g))
    params = [q for m in mods for q in m.parameters()]
    one = {id(q): torch.zeros_like(q) for q in params}
    many = {id(q): torch.zeros_like(q) for q in params}
    scale = torch.full((1,), 0.5, device="cuda")
    keep = [p.unpack_grads(one, accumulate=True, grad_mul_dev=scale) for p in plans]
    keep2 = packing.unpack_many(plans, many, accumulate=True, grad_mul_dev=scale)
    torch.cuda.synchronize()
    assert any(float(one[id(q)].abs().max()) > 0 for q in params)
    for q in params:
        assert torch.equal(one[id(q)], many[id(q)])
    del keep, keep2
