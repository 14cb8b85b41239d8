#!/usr/bin/env python3
"""Pack the pattern template matrices (float64 text tables, data only) that the
hot path needs into one binary file: chromosight_amd/kernels/templates.npz.

Source data: /root/reference/chromosight/kernels/*.txt.
The per-pattern parameters live in chromosight_amd/kernels/__init__.py.
Run in the authoring container only.
"""
import pathlib
import numpy as np

SRC = pathlib.Path("/root/reference/chromosight/kernels")
DST = pathlib.Path(__file__).resolve().parents[1] / "chromosight_amd" / "kernels" / "templates.npz"

arrays = {}
for txt in sorted(SRC.glob("*.txt")):
    arrays[txt.stem.replace(".", "_")] = np.loadtxt(txt)
np.savez_compressed(DST, **arrays)
print({k: v.shape for k, v in arrays.items()})
