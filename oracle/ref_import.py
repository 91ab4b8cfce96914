"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Imports the *real* reference implementation (zju3dv/NeuralRecon-W, mounted read-only at
/root/reference) on CPU inside this container, so that

  * tests/golden/make_golden.py can generate the committed golden vectors, and
  * the `not gpu` tests can pin oracle/neuconw_oracle.py against the live reference at the
    full network widths (W=256 / W=512) whenever /root/reference is present.

/root/reference does not exist on the GPU box: nothing under `-m gpu`, `smoke()` or `bench.py`
may call into this module.

The reference's renderer imports open3d / kaolin / cv2 / ... at module scope
(rendering/renderer.py:1-12); none are installed, so they are stubbed with MagicMock exactly as
SURVEY.md section 8(c) prescribes.  models/neuconw.py and models/nerf.py import only numpy/torch.
"""
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = os.environ.get("NEUCONW_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "open3d", "kaolin", "kaolin.ops", "kaolin.ops.spc", "kaolin.render", "kaolin.render.spc",
    "cv2", "torchvision", "torchvision.transforms", "h5py", "torch_optimizer", "trimesh",
    "skimage", "skimage.measure", "loguru", "kornia", "kornia.losses", "lpips",
    "test_tube", "pytorch_lightning.loggers", "pytorch_lightning.callbacks",
]


class CfgNode(dict):
    """Stand-in for yacs.config.CfgNode (yacs is not installed): a dict with attribute access, `clone()` and
    `merge_from_file()` with yacs' value decoding (strings go through literal_eval: "(4,)" -> (4,), "1e-4" ->
    1e-4).  Enough for the reference's config/defaults.py + config/*.yaml."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy

        return copy.deepcopy(self)

    @staticmethod
    def _decode(v):
        import ast

        if isinstance(v, dict):
            n = CfgNode()
            for k, x in v.items():
                n[k] = CfgNode._decode(x)
            return n
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k]._merge(v)
            else:
                self[k] = v

    def merge_from_file(self, path):
        import yaml

        with open(path, "r") as f:
            self._merge(CfgNode._decode(yaml.safe_load(f)))

    def freeze(self):
        pass

    def defrost(self):
        pass


def _install_yacs():
    if "yacs.config" in sys.modules and getattr(sys.modules["yacs.config"], "CfgNode", None) is CfgNode:
        return
    y, yc = types.ModuleType("yacs"), types.ModuleType("yacs.config")
    yc.CfgNode = CfgNode
    y.config = yc
    sys.modules["yacs"], sys.modules["yacs.config"] = y, yc


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "rendering", "renderer.py"))


_loaded = {}


def load():
    """Returns a namespace with the reference's NeuconW, NeRF, NeuconWRenderer, sample_pdf,
    NeuconWLoss classes (the real ones, executed from /root/reference)."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    import torch  # noqa: F401

    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock()
    _install_yacs()
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")

        class _LM(torch.nn.Module):
            global_step = 0

            def save_hyperparameters(self, h):
                self.hparams = h

            def log(self, *a, **k):
                pass

        pl.LightningModule = _LM
        pl.LightningDataModule = object
        sys.modules["pytorch_lightning"] = pl

    # the reference's top-level package names (models, rendering, datasets, tools, utils, losses)
    # are generic; import them under the reference root and then drop the root from sys.path.
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import importlib

        neuconw_mod = importlib.import_module("models.neuconw")
        nerf_mod = importlib.import_module("models.nerf")
        renderer_mod = importlib.import_module("rendering.renderer")
        losses_mod = importlib.import_module("losses")
    finally:
        sys.path.remove(REFERENCE_ROOT)

    ns = types.SimpleNamespace(
        NeuconW=neuconw_mod.NeuconW,
        SDFNetwork=neuconw_mod.SDFNetwork,
        RenderingNetwork=neuconw_mod.RenderingNetwork,
        get_embedder=neuconw_mod.get_embedder,
        NeRF=nerf_mod.NeRF,
        NeuconWRenderer=renderer_mod.NeuconWRenderer,
        sample_pdf=renderer_mod.sample_pdf,
        NeuconWLoss=losses_mod.NeuconWLoss,
        renderer_mod=renderer_mod,
    )
    _loaded["ns"] = ns
    return ns


def load_system():
    """Returns (module lightning_modules.neuconw_system, function get_cfg_defaults) of the real reference.  The
    drop-in test swaps the three classes the module imported (neuconw_system.py:7-12) for ours -- the same edit
    INTEGRATION.md section 1 asks a maintainer to make in the import lines."""
    load()
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import importlib

        sysmod = importlib.import_module("lightning_modules.neuconw_system")
        defaults = importlib.import_module("config.defaults")
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return sysmod, defaults.get_cfg_defaults
