"""Generates the committed golden vectors in tests/golden/*.npz by RUNNING THE REAL REFERENCE
(/root/reference, imported on CPU under the stub recipe of oracle/ref_import.py).

Run in the build container only:   python tests/golden/make_golden.py
The GPU box has no /root/reference; tests there read the committed .npz files.

Every fixture stores: the reference state_dict (so no RNG/init reproduction is needed), the
inputs, and the reference outputs (+ parameter gradients of the reference loss where stated).
Networks are deliberately small (W=64) so the fixtures stay < 1 MB each; the full-width
(W=256/512) comparisons against the live reference run in tests/test_oracle_vs_reference.py.
"""
import os
import sys
import tempfile

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

torch.set_num_threads(8)


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def synth_rays(R, seed, n_vocab, dtype=torch.float32):
    """SURVEY.md 8(d) ray distribution (unit-sphere units, origin 0 / radius 1)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.tensor([0.0, 0.0, -2.0]) + 0.1 * torch.randn(R, 3, generator=g)
    d = torch.tensor([0.0, 0.0, 1.0]) + 0.1 * torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    near = torch.full((R, 1), 1.0)
    far = torch.full((R, 1), 3.0)
    depth_gt = torch.full((R, 1), 2.0)
    depth_w = (torch.rand(R, 1, generator=g) < 0.2).float()
    rays = torch.cat([o, d, near, far, depth_gt, depth_w], -1).to(dtype)
    ts = torch.randint(0, n_vocab, (R,), generator=g)
    label = torch.where(torch.rand(R, generator=g) < 0.1, torch.tensor(2), torch.tensor(0))
    rgbs = torch.rand(R, 3, generator=g).to(dtype)
    return rays, ts, label, rgbs


def build_reference(ns, W, n_layers, skip_in, n_a=16, n_vocab=64, nerf_w=64, color_hidden=64, head=32,
                    seed=0, **renderer_kw):
    torch.manual_seed(seed)
    sdf_cfg = dict(d_in=3, d_out=W + 1, d_hidden=W, n_layers=n_layers, skip_in=skip_in, multires=6, bias=0.5,
                   scale=1, geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=W, mode="idr", d_out=3, d_hidden=color_hidden, n_layers=4,
                     head_channels=head, static_head_layers=2, weight_norm=True, multires_view=4)
    emb = torch.nn.Embedding(n_vocab, n_a)
    neuconw = ns.NeuconW(sdfNet_config=sdf_cfg, colorNet_config=color_cfg, SNet_config=dict(init_val=0.3),
                         in_channels_a=n_a, encode_a=True)
    nerf = ns.NeRF(D=8, d_in=4, d_in_view=3, W=nerf_w, multires=10, multires_view=4, output_ch=4, skips=[4],
                   encode_appearance=True, in_channels_a=n_a, in_channels_dir=27, use_viewdirs=True)
    # exercise weight-norm: jitter weight_g
    with torch.no_grad():
        for n, p in neuconw.named_parameters():
            if n.endswith("weight_g"):
                p.mul_(1.0 + 0.1 * torch.randn_like(p))
        # make the colour / background nets non-degenerate but tame
    scene = tempfile.mkdtemp()
    with open(os.path.join(scene, "config.yaml"), "w") as f:
        yaml.safe_dump({"origin": [0.0, 0.0, 0.0], "radius": 1.0, "sfm2gt": np.eye(4).tolist()}, f)
    kw = dict(n_samples=16, n_importance=16, n_outside=4, up_sample_steps=2, perturb=1.0, origin=[0, 0, 0],
              radius=1.0, s_val_base=3, spc_options={"recontruct_path": scene, "voxel_size": 0.1,
                                                      "min_track_length": 1},
              sample_range=16, boundary_samples=0, nerf_far_override=False, render_bg=True, trim_sphere=True,
              mesh_mask_list=["sky"], floor_normal=False, depth_loss=True, floor_labels=["road"])
    kw.update(renderer_kw)
    renderer = ns.NeuconWRenderer(nerf=nerf, neuconw=neuconw, embeddings={"a": emb}, **kw)
    return emb, neuconw, nerf, renderer


def full_state_dict(emb, neuconw, nerf):
    sd = {"embedding_a.weight": emb.weight}
    # neuconw.xyz_encoding_final (a dead, hard-coded 512x512 Linear, models/neuconw.py:319) is
    # dropped from the fixtures: it never influences any output and never receives a gradient.
    sd.update({"neuconw." + k: v for k, v in neuconw.state_dict().items()
               if not k.startswith("xyz_encoding_final")})
    sd.update({"nerf." + k: v for k, v in nerf.state_dict().items()})
    return sd


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def golden_units(ns):
    """Per-function vectors: sample_pdf, SDFNetwork fwd/gradient, RenderingNetwork, NeRF."""
    torch.manual_seed(1)
    emb, neuconw, nerf, renderer = build_reference(ns, 64, 8, (4,), seed=1)
    N = 96
    x = (torch.rand(N, 3) * 2 - 1) * 0.9
    dirs = torch.nn.functional.normalize(torch.randn(N, 3), dim=-1)
    a = torch.randn(N, 16) * 0.5
    sdf_full = neuconw.sdf_net(x)
    grad = neuconw.sdf_net.gradient(x.clone()).detach()
    rgb, _, _ = neuconw.color_net(x, grad, dirs, sdf_full[:, 1:], a)
    p4 = torch.cat([torch.nn.functional.normalize(torch.randn(N, 3), dim=-1), torch.rand(N, 1)], -1)
    dens, bg_rgb = nerf(p4, dirs, a)
    # sample_pdf
    bins = torch.sort(torch.rand(40, 17) * 2 + 1, -1)[0]
    w = torch.rand(40, 16) ** 4
    w[3] = 0.0  # all-zero weights row -> uniform pdf branch
    w[5, :8] = 0.0
    spdf = ns.sample_pdf(bins, w, 12, det=True)
    # up_sample + cat_z_vals
    rays, ts, label, rgbs = synth_rays(40, 11, 64)
    z = rays[:, 6:7] + (rays[:, 7:8] - rays[:, 6:7]) * torch.linspace(0, 1, 17)[None]
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    with torch.no_grad():
        sdfz = neuconw.sdf(pts.reshape(-1, 3)).reshape(40, 17)
        znew = renderer.up_sample(rays[:, 0:3], rays[:, 3:6], z, sdfz, 8, 64 * 2 ** 3, 0)
        zcat, sdfcat = renderer.cat_z_vals(rays[:, 0:3], rays[:, 3:6], z, znew, sdfz, last=False)
    sd = full_state_dict(emb, neuconw, nerf)
    save("units_w64", **{"sd/" + k: v for k, v in sd.items()}, x=x, dirs=dirs, a=a, sdf=sdf_full[:, 0],
         feat=sdf_full[:, 1:], grad=grad, rgb=rgb, p4=p4, density=dens, bg_rgb=bg_rgb, pdf_bins=bins,
         pdf_w=w, pdf_out=spdf, us_rays=rays, us_z=z, us_sdf=sdfz, us_znew=znew, us_zcat=zcat,
         us_sdfcat=sdfcat)


def golden_cfg1(ns):
    """BASELINE config 1: 64 rays x 32 uniform samples, 2-layer 64-wide SDF MLP, fp32, reference
    CPU path.  sparse_sampler crashes with n_importance==0 upstream, so the reference
    is driven through render_core_outside + render_core with explicit uniform z."""
    emb, neuconw, nerf, renderer = build_reference(ns, 64, 2, (), seed=2, n_samples=32, n_importance=0)
    R, S, O = 64, 32, 4
    rays, ts, label, rgbs = synth_rays(R, 21, 64)
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    z = near + (far - near) * torch.linspace(0, 1, S)[None]
    sample_dist = (far - near) / S
    z_out = far / torch.flip(torch.linspace(1e-3, 1 - 1 / (O + 1.0), O), dims=[-1]) + 1.0 / S
    a = emb(ts)
    zf, _ = torch.sort(torch.cat([z, z_out], -1), -1)
    ro = renderer.render_core_outside(o, d, zf, sample_dist, nerf, a_embedded=a)
    bg_alpha0 = ro["alpha"].detach().clone()
    rc = renderer.render_core(o, d, z, sample_dist, a, cos_anneal_ratio=0.3, background_alpha=ro["alpha"],
                              background_sampled_color=ro["sampled_color"], background_rgb=torch.zeros(1, 3))
    loss = (rc["color"] - rgbs).abs().sum() / R + 0.1 * rc["gradient_error"] + 0.05 * rc["weights_sum"].mean() \
        + 0.05 * rc["depth"].mean()
    params = dict(full_state_dict(emb, neuconw, nerf))
    named = [(k, v) for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [v for _, v in named], allow_unused=True)
    arrays = {"sd/" + k: v for k, v in params.items()}
    for (k, v), g in zip(named, grads):
        if g is not None:
            arrays["grad/" + k] = g
    save("cfg1_r64_s32", **arrays, rays=rays, ts=ts, label=label, rgbs=rgbs, z=z, z_out=z_out,
         sample_dist=sample_dist, bg_alpha=bg_alpha0, bg_rgb=ro["sampled_color"], loss=loss,
         **{"out/" + k: v for k, v in rc.items() if isinstance(v, torch.Tensor)})


def golden_render(ns, name, perturb, **kw):
    """Whole NeuconWRenderer.render + NeuconWLoss + backward on a small 8-layer network."""
    emb, neuconw, nerf, renderer = build_reference(ns, 64, 8, (4,), seed=3, **kw)
    R = 48
    rays, ts, label, rgbs = synth_rays(R, 31, 64)
    arrays = {}
    if perturb:
        torch.manual_seed(777)
        arrays["rand_shift"] = torch.rand(R, 1)
        arrays["rand_out"] = torch.rand(R, renderer.n_outside)
        torch.manual_seed(777)
    out = renderer.render(rays.clone(), ts, label, perturb_overwrite=1 if perturb else 0,
                          background_rgb=torch.zeros(1, 3), cos_anneal_ratio=0.25)
    cfg = AttrDict(NEUCONW=AttrDict(MESH_MASK_LIST=["sky"], DEPTH_LOSS=True, FLOOR_NORMAL=False))
    loss_fn = ns.NeuconWLoss(coef=1.0, igr_weight=0.1, mask_weight=0.1, depth_weight=0.1, floor_weight=0.01,
                             config=cfg)
    ld = loss_fn(out, rgbs)
    loss = sum(ld.values())
    params = dict(full_state_dict(emb, neuconw, nerf))
    named = [(k, v) for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [v for _, v in named], allow_unused=True)
    arrays.update({"sd/" + k: v for k, v in params.items()})
    for (k, v), g in zip(named, grads):
        if g is not None:
            arrays["grad/" + k] = g
    arrays.update({"out/" + k: v for k, v in out.items()})
    arrays.update({"loss/" + k: v for k, v in ld.items()})
    save(name, **arrays, rays=rays, ts=ts, label=label, rgbs=rgbs, loss=loss)


def main():
    ns = ref_import.load()
    golden_units(ns)
    golden_cfg1(ns)
    golden_render(ns, "render_w64_det", perturb=False)
    golden_render(ns, "render_w64_perturb", perturb=True)
    golden_render(ns, "render_w64_shipped_shape", perturb=False, n_samples=8, n_importance=16)


if __name__ == "__main__":
    main()
