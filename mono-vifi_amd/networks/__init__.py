"""Networks of the drop-in trainer (plain PyTorch-ROCm modules; reference: networks/)."""
from . import dhrnet, litemono, monodepth2, posenet  # noqa: F401
from .fusion_module import FusionModule  # noqa: F401
from .ifrnet import IFRNet  # noqa: F401
