"""Builders shared by the GPU tests: our modules configured like the golden fixtures / BASELINE configs."""
import torch


def build_system(W=64, n_layers=8, skip_in=(4,), n_a=16, n_vocab=64, nerf_w=64, color_hidden=64, head=32, seed=0,
                 device="cuda", prec=None, **renderer_kw):
    import neuralrecon_w_amd as nw

    torch.manual_seed(seed)
    sdf_cfg = dict(d_in=3, d_out=W + 1, d_hidden=W, n_layers=n_layers, skip_in=skip_in, multires=6, bias=0.5,
                   scale=1, geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=W, mode="idr", d_out=3, d_hidden=color_hidden, n_layers=4,
                     head_channels=head, static_head_layers=2, weight_norm=True, multires_view=4)
    emb = torch.nn.Embedding(n_vocab, n_a)
    neuconw = nw.NeuconW(sdfNet_config=sdf_cfg, colorNet_config=color_cfg, SNet_config=dict(init_val=0.3),
                         in_channels_a=n_a, encode_a=True)
    nerf = nw.NeRF(D=8, d_in=4, d_in_view=3, W=nerf_w, multires=10, multires_view=4, output_ch=4, skips=[4],
                   encode_appearance=True, in_channels_a=n_a, in_channels_dir=27, use_viewdirs=True)
    emb, neuconw, nerf = emb.to(device), neuconw.to(device), nerf.to(device)
    kw = dict(n_samples=16, n_importance=16, n_outside=4, up_sample_steps=2, perturb=1.0, origin=[0, 0, 0],
              radius=1.0, s_val_base=3, spc_options={"recontruct_path": "/nonexistent", "voxel_size": 0.1,
                                                      "min_track_length": 1},
              sample_range=16, boundary_samples=0, nerf_far_override=False, render_bg=True, trim_sphere=True,
              mesh_mask_list=["sky"], floor_normal=False, depth_loss=True, floor_labels=["road"], prec=prec)
    kw.update(renderer_kw)
    renderer = nw.NeuconWRenderer(nerf=nerf, neuconw=neuconw, embeddings={"a": emb}, **kw)
    return emb, neuconw, nerf, renderer


def load_golden_weights(sd, emb, neuconw, nerf):
    emb.load_state_dict({"weight": sd["embedding_a.weight"]})
    missing = neuconw.load_state_dict({k[len("neuconw."):]: v for k, v in sd.items() if k.startswith("neuconw.")},
                                      strict=False)
    assert all(k.startswith("xyz_encoding_final") for k in missing.missing_keys), missing
    nerf.load_state_dict({k[len("nerf."):]: v for k, v in sd.items() if k.startswith("nerf.")})


def state_dict_cpu(emb, neuconw, nerf, dtype=torch.float64):
    sd = {"embedding_a.weight": emb.weight.detach().cpu().to(dtype)}
    sd.update({"neuconw." + k: v.detach().cpu().to(dtype) for k, v in neuconw.state_dict().items()
               if not k.startswith("xyz_encoding_final")})
    sd.update({"nerf." + k: v.detach().cpu().to(dtype) for k, v in nerf.state_dict().items()})
    return sd


def named_params(emb, neuconw, nerf):
    out = {"embedding_a.weight": emb.weight}
    out.update({"neuconw." + k: p for k, p in neuconw.named_parameters()})
    out.update({"nerf." + k: p for k, p in nerf.named_parameters()})
    return out


def loss_from_outputs(out, rgbs, igr=0.1, mask_w=0.1, depth_w=0.1):
    """NeuconWLoss (losses.py:21-43) -- host-side torch glue, as in the reference."""
    R = rgbs.shape[0]
    loss = (out["color"] - rgbs).abs().sum() / (R + 1e-5)
    loss = loss + igr * out["gradient_error"].mean()
    loss = loss + mask_w * out["mask_error"].mean()
    loss = loss + depth_w * out["sfm_depth_loss"].mean()
    return loss
