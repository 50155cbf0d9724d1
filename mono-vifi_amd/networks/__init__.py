"""Networks of the drop-in trainer (plain PyTorch-ROCm modules; reference: networks/)."""
from . import monodepth2, posenet  # noqa: F401
from .fusion_module import FusionModule  # noqa: F401
from .ifrnet import IFRNet  # noqa: F401
