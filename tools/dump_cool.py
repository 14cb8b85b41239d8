#!/usr/bin/env python3
"""Decode a .cool (HDF5) file into a plain .npz without cooler/h5py.

Test tooling only (authoring container): neither `cooler` nor `h5py` is
installed here or on the GPU box, so the fixture `.cool` files the reference's
tests use (`/root/reference/data_test/example.cool`, reference
`tests/test_preprocessing.py:10`) are decoded once with the HDF5 command line
tool `h5dump -b` and committed as small `.npz` fixtures under `tests/golden/`.

Datasets read (cooler schema v3): pixels/{bin1_id,bin2_id,count},
bins/{chrom,start,end,weight}, chroms/{name,length}, indexes/chrom_offset,
attribute bin-size.
"""
import re
import subprocess
import sys
import tempfile
import pathlib

import numpy as np

H5DUMP = "/opt/conda/bin/h5dump"


def _dump(path, dataset, dtype):
    with tempfile.TemporaryDirectory() as tmp:
        out = pathlib.Path(tmp) / "d.bin"
        subprocess.run(
            [H5DUMP, "-d", dataset, "-b", "LE", "-o", str(out), str(path)],
            check=True,
            stdout=subprocess.DEVNULL,
        )
        return np.fromfile(out, dtype=dtype)


def _strings(path, dataset):
    txt = subprocess.run(
        [H5DUMP, "-d", dataset, str(path)], check=True, capture_output=True, text=True
    ).stdout
    data = txt[txt.index("DATA {"):]
    return [x.split("\\000")[0] for x in re.findall(r'"([^"]*)"', data)]


def _attr_int(path, name):
    txt = subprocess.run(
        [H5DUMP, "-a", "/" + name, str(path)], check=True, capture_output=True, text=True
    ).stdout
    return int(re.search(r"\(0\):\s*(-?\d+)", txt).group(1))


def dump_cool(path):
    path = pathlib.Path(path)
    d = {
        "bin1_id": _dump(path, "pixels/bin1_id", "<i8").astype(np.int32),
        "bin2_id": _dump(path, "pixels/bin2_id", "<i8").astype(np.int32),
        "count": _dump(path, "pixels/count", "<i4"),
        "weight": _dump(path, "bins/weight", "<f8"),
        "bin_start": _dump(path, "bins/start", "<i4").astype(np.int64),
        "bin_end": _dump(path, "bins/end", "<i4").astype(np.int64),
        "chrom_offset": _dump(path, "indexes/chrom_offset", "<i8"),
        "chrom_names": np.array(_strings(path, "chroms/name")),
        "binsize": np.int64(_attr_int(path, "bin-size")),
    }
    return d


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    d = dump_cool(src)
    np.savez_compressed(dst, **d)
    print({k: (v.shape, v.dtype) for k, v in d.items()})
