"""Shared helpers for the test-suite (fixture loading, tolerances, synthetic batches)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name, dtype=torch.float32):
    """-> (sd, grads, outs, misc) ; sd/grad/out/loss-prefixed arrays are split out."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd, grads, outs, misc = {}, {}, {}, {}
    for k in z.files:
        v = torch.from_numpy(np.asarray(z[k]))
        if v.is_floating_point():
            v = v.to(dtype)
        if k.startswith("sd/"):
            sd[k[3:]] = v
        elif k.startswith("grad/"):
            grads[k[5:]] = v
        elif k.startswith("out/"):
            outs[k[4:]] = v
        else:
            misc[k] = v
    return sd, grads, outs, misc


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def rel_err(a, b):
    """max |a-b| / (max|b| + tiny): the '1e-4 rel fp32' measure of BASELINE.json north_star."""
    a = a.detach().double().reshape(-1)
    b = b.detach().double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def synth_rays(R, seed, n_vocab, dtype=torch.float32):
    """SURVEY.md 8(d) synthetic ray distribution (same generator as tests/golden/make_golden.py)."""
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


def compat_kaolin():
    """(kaolin.ops.spc, kaolin.render.spc) of compat/kaolin, whatever `kaolin` currently is in sys.modules (oracle/ref_import.py
    parks MagicMock stubs there for the reference imports of other tests): imported under a clean slate, previous entries restored."""
    import importlib
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "kaolin" or k.startswith("kaolin.")}
    sys.path.insert(0, os.path.join(root, "compat"))
    try:
        ops_spc = importlib.import_module("kaolin.ops.spc")
        render_spc = importlib.import_module("kaolin.render.spc")
    finally:
        sys.path.remove(os.path.join(root, "compat"))
        for k in [k for k in sys.modules if k == "kaolin" or k.startswith("kaolin.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return ops_spc, render_spc
