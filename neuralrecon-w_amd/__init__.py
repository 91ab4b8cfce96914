"""neuralrecon-w_amd: MI355X (gfx950) native volume-rendering hot path of NeuralRecon-W.

Drop-in for the reference's `models.neuconw.NeuconW`, `models.nerf.NeRF` and
`rendering.renderer.NeuconWRenderer`.  All per-sample compute is in
libneuconw_hip.so (hand-written HIP, C ABI in include/neuconw_hip.h).
"""
from . import lib  # noqa: F401
from . import mesh  # noqa: F401
from .lib import PREC_BF16, PREC_F16, PREC_F32, NeuconwHipError  # noqa: F401
from .nerf import NeRF  # noqa: F401
from .neuconw import NeuconW, RenderingNetwork, SDFNetwork, SingleVarianceNetwork  # noqa: F401
from .renderer import NeuconWRenderer  # noqa: F401
from .losses import NeuconWLoss  # noqa: F401
from .trainer import FlatAdam, FlatParams, TrainStep  # noqa: F401
