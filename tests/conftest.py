import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu`)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(GOLDEN / f"{name}.npz", allow_pickle=False))
        return cache[name]

    return load


@pytest.fixture(scope="session")
def templates():
    import chromosight_amd.kernels as ck
    return {
        "loops": ck.loops["kernels"][0],
        "small": ck.loops_small["kernels"][0],
        "hairpin": ck.hairpins["kernels"][0],
        "borders": ck.borders["kernels"],
    }
