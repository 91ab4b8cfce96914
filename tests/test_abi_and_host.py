"""CPU-side checks: the C-ABI library loads and exports every symbol include/neuconw_hip.h declares,
the ctypes structs mirror the header, the drop-in modules expose the reference's state_dict keys,
and the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

from tests._util import ROOT, load_golden


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "neuconw_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|int64_t|const char\*)\s+(ncw_\w+)\s*\(", src, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from neuralrecon_w_amd import build, lib as L

    path = build.LIB
    if not os.path.isfile(path):
        build.build(verbose=False)
    cdll = ctypes.CDLL(path)
    declared = _header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(cdll, name), "missing export: " + name
    assert sorted(L.exported_symbols()) == declared, "lib.py bindings out of sync with the header"
    assert cdll.ncw_abi_version() == L.ABI_VERSION


def test_library_is_built_from_this_tree(monkeypatch):
    """The shipped .so carries the sha256 of the sources it was built from; the binding recomputes it from csrc/ + include/
    and refuses a library built from anything else (a GPU box runs the PREBUILT library: this is what ties the kernels under
    test to the tree).  Also: a different hash is really refused."""
    from neuralrecon_w_amd import build, lib as L

    cdll = ctypes.CDLL(build.LIB)
    cdll.ncw_source_hash.restype = ctypes.c_char_p
    assert cdll.ncw_source_hash().decode() == build.source_hash(), "libneuconw_hip.so is stale: rebuild (__graft_entry__.build())"
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 64)
    monkeypatch.delenv("NEUCONW_HIP_LIB", raising=False)
    with pytest.raises(L.NeuconwHipError, match="built from other sources"):
        L.get_lib()
    monkeypatch.undo()
    L.get_lib()  # the real tree loads


def test_probe_hooks_cannot_enter_the_product_build():
    """NCW_EXP_* timing hooks (scripts/probes/ncw_exp_hooks.h) compile only under NCW_BUILD_TAG (a probe library beside the
    product); csrc/ncw_common.h #errors on them otherwise and build.py refuses the flag."""
    import subprocess
    import sys

    env = dict(os.environ, NCW_EXTRA_HIPCC_FLAGS="-DNCW_EXP_NOLOAD")
    env.pop("NCW_BUILD_TAG", None)
    r = subprocess.run([sys.executable, "-c", "import neuralrecon_w_amd.build"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "NCW_BUILD_TAG" in r.stderr
    src = open(os.path.join(ROOT, "neuralrecon-w_amd", "csrc", "ncw_common.h")).read()
    assert "#error" in src and "NCW_PROBE_BUILD" in src


def test_struct_sizes_match_c_layout():
    """sizeof of every ctypes mirror equals what the C compiler lays out (checked with gcc)."""
    import subprocess
    import tempfile

    from neuralrecon_w_amd import lib as L

    names = ["NcwSeg", "NcwPackDesc", "NcwUnpackDesc", "NcwSdfNet", "NcwPoints", "NcwSdfStash", "NcwWgradDesc",
             "NcwColorNet", "NcwColorStash", "NcwNerfNet", "NcwNerfStash", "NcwCompositeIn", "NcwCompositeOut",
             "NcwCompositeGrad"]
    prog = '#include <stdio.h>\n#include "neuconw_hip.h"\nint main(){' + "".join(
        'printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(prog)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split()
    sizes = dict(zip(out[0::2], map(int, out[1::2])))
    for n in names:
        assert ctypes.sizeof(getattr(L, n)) == sizes[n], (n, ctypes.sizeof(getattr(L, n)), sizes[n])


def test_state_dict_keys_match_reference_checkpoint_format():
    """Golden fixtures carry the real reference's state_dict: our modules must load it unchanged."""
    from tests._build import build_system, load_golden_weights

    sd, _, _, _ = load_golden("render_w64_det")
    emb, neuconw, nerf, _ = build_system(device="cpu")
    load_golden_weights(sd, emb, neuconw, nerf)
    ours = {"neuconw." + k for k in neuconw.state_dict()} | {"nerf." + k for k in nerf.state_dict()}
    theirs = {k for k in sd if not k.startswith("embedding_a")}
    assert theirs <= ours
    extra = ours - theirs
    assert all(k.startswith("neuconw.xyz_encoding_final") for k in extra), extra  # dropped from the fixtures
    # full-size checkpoint shape of the shipped config (84 tensors incl. embedding)
    import neuralrecon_w_amd as nw
    sdf_cfg = dict(d_in=3, d_out=513, d_hidden=512, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                   geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=512, mode="idr", d_out=3, d_hidden=256, n_layers=4, head_channels=128,
                     static_head_layers=2, weight_norm=True, multires_view=4)
    big = nw.NeuconW(sdf_cfg, color_cfg, dict(init_val=0.3), 48, True)
    bg = nw.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                 encode_appearance=True, in_channels_a=48, in_channels_dir=27, use_viewdirs=True)
    assert len(big.state_dict()) + len(bg.state_dict()) + 1 == 84
    assert sum(p.numel() for p in big.parameters()) == 2957627
    assert sum(p.numel() for p in bg.parameters()) == 698628
    assert big.sdf_net.lin3.weight_v.shape == (473, 512) and big.sdf_net.lin8.weight_v.shape == (513, 512)


def test_no_cpu_fallback():
    import neuralrecon_w_amd as nw
    from tests._build import build_system

    emb, neuconw, nerf, rdr = build_system(device="cpu")
    with pytest.raises(nw.NeuconwHipError):
        neuconw.sdf(torch.zeros(4, 3))
    with pytest.raises(nw.NeuconwHipError):
        rdr.render(torch.zeros(4, 10), torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))


def test_pack_plan_descriptors_cover_every_parameter():
    """Host logic of the pack plan: every Linear of the three networks is packed (forward +
    transposed) and has exactly one gradient-unpack descriptor per source row."""
    from neuralrecon_w_amd import lib as L
    from tests._build import build_system

    emb, neuconw, nerf, _ = build_system(device="cpu")
    for mod in (neuconw.sdf_net, neuconw.color_net, nerf):
        plan = mod.plan(L.PREC_BF16)
        packed = {p["weight"].data_ptr() for p in plan._pack}
        rows = {}
        for u in plan._unpack:
            rows.setdefault(u["weight"].data_ptr(), 0)
            rows[u["weight"].data_ptr()] += u["nrows"]
        for name, p in mod.named_parameters():
            if p.dim() == 2 and not name.startswith("views_linears") and not name.endswith("weight_g"):
                assert p.data_ptr() in packed, name
                assert rows[p.data_ptr()] == p.shape[0], name


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree not mounted")
def test_same_seed_same_init_as_reference():
    """Constructing our modules consumes the RNG exactly like the reference: same seed -> same weights."""
    import neuralrecon_w_amd as nw
    from oracle import ref_import

    ns = ref_import.load()
    sdf_cfg = dict(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1,
                   geometric_init=True, weight_norm=True, inside_outside=False)
    color_cfg = dict(d_in=9, d_feature=256, mode="idr", d_out=3, d_hidden=256, n_layers=4, head_channels=128,
                     static_head_layers=2, weight_norm=True, multires_view=4)
    torch.manual_seed(7)
    ref = ns.NeuconW(sdf_cfg, color_cfg, dict(init_val=0.3), 48, True)
    ref_bg = ns.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                     encode_appearance=True, in_channels_a=48, in_channels_dir=27, use_viewdirs=True)
    torch.manual_seed(7)
    ours = nw.NeuconW(sdf_cfg, color_cfg, dict(init_val=0.3), 48, True)
    ours_bg = nw.NeRF(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4],
                      encode_appearance=True, in_channels_a=48, in_channels_dir=27, use_viewdirs=True)
    a, b = ref.state_dict(), ours.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.allclose(a[k], b[k], rtol=0, atol=1e-7), k
    a, b = ref_bg.state_dict(), ours_bg.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
