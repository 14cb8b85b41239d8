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
import os as _os

# HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 by default).  A genome step keeps five to seven streams
# busy at once (staging, the tile kernels, the side lanes of the mask tables, the 1-D pattern's chain on its worker context) and an
# RCCL communicator brings its own: with four queues the chains that are meant to overlap share a queue and run one after the
# other -- a rank's share of 8 took 0.94 ms instead of 0.57 as soon as ncclCommInitRank had run in the process
# (tools/rccl_probe.py, profiles/r06_rccl_hw_queues.txt).  Read by the HIP runtime when it initialises: set here, before the
# library's first call; a process that initialised HIP earlier sets it itself (bench.py does, ahead of torch).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import kernels  # noqa: F401,E402
from .engine import get_precision, set_precision  # noqa: F401,E402

__version__ = "0.1.0"
