"""Pattern templates and their default parameters (data needed by the hot path).

Mirrors the attribute surface of the reference's `chromosight.kernels`
(reference chromosight/kernels/__init__.py:21-44): one module-level dict per
pattern (`loops`, `borders`, `hairpins`, ...) with the JSON parameters and the
template matrices pre-loaded under the "kernels" key, plus `kernel_names`.
The template matrices are stored in `templates.npz` (packed by
tools/pack_kernels.py from the reference's float64 text tables).
"""
import pathlib
import sys

import numpy as np

_TEMPLATES = np.load(pathlib.Path(__file__).with_name("templates.npz"))

# parameter tables: reference chromosight/kernels/<name>.json
_CONFIGS = {
    "loops": dict(
        kernels=["artificial_template_loops_type1"],
        min_dist=20000, max_dist=2000000, max_iterations=1, max_perc_zero=10.0,
        max_perc_undetected=50.0, min_separation=5000, pearson=0.3, resolution=2000,
    ),
    "loops_small": dict(
        kernels=["artificial_template_loops_small"],
        min_dist=20000, max_dist=2000000, max_iterations=1, max_perc_zero=10.0,
        max_perc_undetected=50.0, min_separation=5000, pearson=0.5, resolution=2000,
    ),
    "borders": dict(
        kernels=[
            "artificial_template_borders_type1",
            "artificial_template_borders_type2",
            "artificial_template_borders_type3",
        ],
        max_dist=0, min_dist=0, max_iterations=1, max_perc_zero=10.0,
        max_perc_undetected=75.0, min_separation=5000, pearson=0.15, resolution=5000,
    ),
    "hairpins": dict(
        kernels=["artificial_template_hairpin"],
        max_dist=0, min_dist=0, max_iterations=1, max_perc_zero=10.0,
        max_perc_undetected=75.0, min_separation=5000, pearson=0.1, resolution=10000,
    ),
    "centromeres": dict(
        kernels=["centromeres_1"],
        min_dist=0, max_dist=0, max_iterations=1, max_perc_zero=30.0,
        max_perc_undetected=50.0, pearson=0.5, min_separation=100000, resolution=2000,
    ),
    "stripes_left": dict(
        kernels=["stripes_left_1"],
        min_dist=60000, max_dist=2000000, max_iterations=1, max_perc_zero=80.0,
        max_perc_undetected=50.0, min_separation=40000, pearson=0.17, resolution=10000,
    ),
    "stripes_right": dict(
        kernels=["stripes_right_1"],
        min_dist=60000, max_dist=2000000, max_iterations=1, max_perc_zero=80.0,
        max_perc_undetected=50.0, min_separation=40000, pearson=0.17, resolution=10000,
    ),
}


def load_kernel_config(name):
    """Return a fresh config dict for a built-in pattern, templates loaded as
    float64 arrays (same keys as reference utils/io.py:81-205 produces)."""
    if name not in _CONFIGS:
        raise KeyError(f"unknown pattern {name!r}; choose from {sorted(_CONFIGS)}")
    cfg = dict(_CONFIGS[name])
    cfg["name"] = name
    cfg["kernels"] = [np.array(_TEMPLATES[k], dtype=np.float64) for k in cfg["kernels"]]
    return cfg


kernel_names = sorted(_CONFIGS)
_module = sys.modules[__name__]
for _name in kernel_names:
    setattr(_module, _name, load_kernel_config(_name))
