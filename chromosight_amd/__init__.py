"""chromosight_amd -- MI355X (gfx950) implementation of chromosight's sliding-window
Pearson-correlation hot path (normxcorr2 / xcorr2 + per-diagonal detrend), behind the
reference's own Python call surface:

    chromosight_amd.utils.detection      ~ chromosight.utils.detection
    chromosight_amd.utils.preprocessing  ~ chromosight.utils.preprocessing
    chromosight_amd.utils.stats          ~ chromosight.utils.stats
    chromosight_amd.kernels              ~ chromosight.kernels

All arithmetic of the path runs in hand-written HIP kernels (chromosight_amd/csrc) reached
through the C ABI of include/chromosight_hip.h; there is no CPU fallback.
"""
from . import kernels  # noqa: F401
from .engine import get_precision, set_precision  # noqa: F401

__version__ = "0.1.0"
