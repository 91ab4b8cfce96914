"""Checkpoint I/O in the reference's PyTorch-Lightning layout: files written by
trainer.save_checkpoint load through the REFERENCE's own utils.load_ckpt into the REFERENCE's modules (when the
reference tree is present) and back into ours, including flat-parameter storage."""
import os

import pytest
import torch


def _ours(W=64, seed=0):
    import neuralrecon_w_amd as nw

    torch.manual_seed(seed)
    sdf_cfg = dict(d_in=3, d_out=W + 1, d_hidden=W, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                   geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=W, mode="idr", d_out=3, d_hidden=64, n_layers=4, head_channels=32,
                     static_head_layers=2, weight_norm=True, multires_view=4)
    emb = torch.nn.Embedding(32, 16)
    neuconw = nw.NeuconW(sdfNet_config=sdf_cfg, colorNet_config=color_cfg, SNet_config=dict(init_val=0.3),
                         in_channels_a=16, encode_a=True)
    nerf = nw.NeRF(D=8, d_in=4, d_in_view=3, W=64, multires=10, multires_view=4, output_ch=4, skips=[4],
                   encode_appearance=True, in_channels_a=16, in_channels_dir=27, use_viewdirs=True)
    return emb, neuconw, nerf, sdf_cfg, color_cfg


def test_roundtrip_with_flat_params(tmp_path):
    from neuralrecon_w_amd import trainer

    emb, neuconw, nerf, *_ = _ours(seed=1)
    path = os.path.join(tmp_path, "iter_10.ckpt")
    ck = trainer.save_checkpoint(path, emb, neuconw, nerf, global_step=10)
    assert set(k.split(".")[0] for k in ck["state_dict"]) == {"embedding_a", "neuconw", "nerf"}
    emb2, neuconw2, nerf2, *_ = _ours(seed=2)
    fp = trainer.FlatParams([emb2, neuconw2, nerf2])
    before = fp._version
    ptr = neuconw2.sdf_net.lin0.weight_v.data_ptr()
    got = trainer.load_checkpoint(path, emb2, neuconw2, nerf2, flat_params=fp)
    assert got["global_step"] == 10 and fp._version != before
    assert neuconw2.sdf_net.lin0.weight_v.data_ptr() == ptr  # still a view of the flat buffer
    for a, b in ((emb, emb2), (neuconw, neuconw2), (nerf, nerf2)):
        for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert k == k2 and torch.equal(v, v2), k
    off, n = fp.slices[id(neuconw2.sdf_net.lin0.weight_v)]
    assert torch.equal(fp.flat.data[off:off + n].view_as(neuconw.sdf_net.lin0.weight_v), neuconw.sdf_net.lin0.weight_v)


def test_reference_load_ckpt_reads_our_checkpoint(tmp_path):
    from oracle import ref_import

    if not ref_import.available():
        pytest.skip("reference tree not mounted")
    import importlib
    import sys

    ref = ref_import.load()
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    try:
        ref_utils = importlib.import_module("utils")
    finally:
        sys.path.remove(ref_import.REFERENCE_ROOT)
    from neuralrecon_w_amd import trainer

    emb, neuconw, nerf, sdf_cfg, color_cfg = _ours(seed=3)
    path = os.path.join(tmp_path, "ours.ckpt")
    trainer.save_checkpoint(path, emb, neuconw, nerf, global_step=3)
    r_emb = torch.nn.Embedding(32, 16)
    r_neuconw = ref.NeuconW(sdfNet_config=sdf_cfg, colorNet_config=color_cfg, SNet_config=dict(init_val=0.3),
                            in_channels_a=16, encode_a=True)
    r_nerf = ref.NeRF(D=8, d_in=4, d_in_view=3, W=64, multires=10, multires_view=4, output_ch=4, skips=[4],
                      encode_appearance=True, in_channels_a=16, in_channels_dir=27, use_viewdirs=True)
    ref_utils.load_ckpt(r_emb, path, model_name="embedding_a")  # tools/extract_mesh.py:132-134
    ref_utils.load_ckpt(r_neuconw, path, model_name="neuconw")
    ref_utils.load_ckpt(r_nerf, path, model_name="nerf")
    for ours, theirs in ((emb, r_emb), (neuconw, r_neuconw), (nerf, r_nerf)):
        so, st = ours.state_dict(), theirs.state_dict()
        assert list(so) == list(st)
        for k in so:
            assert torch.equal(so[k], st[k]), k
    # and the other direction: a checkpoint written from the reference's modules loads into ours
    torch.save({"state_dict": {**{"embedding_a." + k: v for k, v in r_emb.state_dict().items()},
                               **{"neuconw." + k: v for k, v in r_neuconw.state_dict().items()},
                               **{"nerf." + k: v for k, v in r_nerf.state_dict().items()}}}, path)
    emb2, neuconw2, nerf2, *_ = _ours(seed=4)
    trainer.load_checkpoint(path, emb2, neuconw2, nerf2)
    assert torch.equal(neuconw2.sdf_net.lin8.weight_v, r_neuconw.sdf_net.lin8.weight_v)


def test_optimizer_state_is_torch_adam_compatible(tmp_path):
    """trainer.save_checkpoint writes FlatAdam's moments in torch.optim.Adam's own state_dict layout, in the reference's
    parameter order (utils/__init__.py:10-31 over [embedding_a, {neuconw, nerf}]): a stock Adam -- what PyTorch-Lightning
    restores `optimizer_states[0]` into -- loads it, and FlatAdam reads it back."""
    from neuralrecon_w_amd import trainer

    emb, neuconw, nerf, *_ = _ours(seed=3)
    fp = trainer.FlatParams([emb, neuconw, nerf])
    opt = trainer.FlatAdam(fp, lr=2e-4, eps=1e-7, clip=0.99)
    g = torch.Generator().manual_seed(0)
    opt.exp_avg.copy_(torch.randn(opt.exp_avg.shape, generator=g))
    opt.exp_avg_sq.copy_(torch.rand(opt.exp_avg_sq.shape, generator=g))
    opt.step_count = 7
    path = os.path.join(tmp_path, "iter_7.ckpt")
    ck = trainer.save_checkpoint(path, emb, neuconw, nerf, optimizer=opt, global_step=7)
    order = trainer.reference_param_order(emb, neuconw, nerf)
    sd = torch.load(path, map_location="cpu")["optimizer_states"][0]
    assert sd["param_groups"][0]["params"] == list(range(len(order))) and sd["param_groups"][0]["eps"] == 1e-7
    stock = torch.optim.Adam(order, lr=1.0, eps=1e-3)
    stock.load_state_dict(sd)  # the stock optimiser accepts it ...
    assert stock.param_groups[0]["lr"] == 2e-4 and stock.param_groups[0]["eps"] == 1e-7
    p = neuconw.sdf_net.lin3.weight_v
    off, k = fp.slices[id(p)]
    st = stock.state[p]
    assert float(st["step"]) == 7.0 and torch.equal(st["exp_avg"], opt.exp_avg[off:off + k].view_as(p))
    assert torch.equal(st["exp_avg_sq"], opt.exp_avg_sq[off:off + k].view_as(p))
    # ... and FlatAdam reads a stock Adam's state back (the reference -> us direction)
    opt2 = trainer.FlatAdam(fp, lr=2e-4, eps=1e-7, clip=0.99)
    opt2.load_state_dict(stock.state_dict(), order)
    assert opt2.step_count == 7 and torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
    assert ck["global_step"] == 7
