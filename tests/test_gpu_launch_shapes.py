"""Launch shaping of the streaming kernel must not change results: forced strip heights
(CHROMOSIGHT_HIP_STRIP_H), the two-height tiling of single-generation launches
(CHROMOSIGHT_HIP_SPLIT, auto-selected at C2 size) and ragged map sizes, against the C oracle."""
import os

import numpy as np
import pytest

from chromosight_amd.utils import detection as cud
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def run_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("shape", [(333, 517), (97, 130), (640, 256)])
@pytest.mark.parametrize("env", [{"CHROMOSIGHT_HIP_STRIP_H": "8"}, {"CHROMOSIGHT_HIP_STRIP_H": "22"},
                                 {"CHROMOSIGHT_HIP_STRIP_H": "200"}, {"CHROMOSIGHT_HIP_SPLIT": "20,12"},
                                 {"CHROMOSIGHT_HIP_SPLIT": "82,46"}, {"CHROMOSIGHT_HIP_SPLIT": "6,30"}],
                         ids=["h8", "h22", "h200", "split20-12", "split82-46", "split6-30"])
def test_dense_shapes(shape, env):
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    sig = rng.gamma(4, 0.25, size=shape)
    kern = rng.random((17, 17)) + 0.5
    kern = (kern + kern[::-1]) / 2          # the symmetric-template kernel, as the loops template uses
    for full in (False, True):
        got, _ = run_env(env, lambda: cud.normxcorr2(sig, kern, full=full))
        want, _ = c_oracle.normxcorr2(sig, kern, full=full)
        assert np.abs(got - want).max() < 1e-5, (shape, env, full)


def test_auto_split_ragged_rows():
    """3500 x 4096: 28 x 32 strips of 128 rows would be one wave per SIMD; the model picks a height
    with two waves per SIMD, where the two-height tiling applies with a ragged last pair."""
    rng = np.random.default_rng(11)
    sig = rng.gamma(4, 0.25, size=(3500, 4096)).astype(np.float32)
    import chromosight_amd.kernels as ck
    kern = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    got, _ = cud.normxcorr2(sig, kern, full=False)
    ref, _ = run_env({"CHROMOSIGHT_HIP_SPLIT": "0", "CHROMOSIGHT_HIP_STRIP_H": "32"}, lambda: cud.normxcorr2(sig, kern, full=False))
    assert np.abs(got - ref).max() < 1e-5
    rows = slice(3300, 3500)
    want, _ = c_oracle.normxcorr2(sig[3200:].astype(np.float64), kern, full=False)
    assert np.abs(got[rows] - want[100:300]).max() < 1e-5


@pytest.mark.parametrize("env", [{"CHROMOSIGHT_HIP_SPLIT": "20,12"}, {"CHROMOSIGHT_HIP_STRIP_H": "10"}], ids=["split", "h10"])
def test_factorised_mask_path_under_forced_shapes(env):
    """The per-bin mask path (row / column tables, frame corrections) with a two-height tiling and
    with short strips: same answers as with the default launch shape."""
    import importlib.util
    import pathlib
    spec = importlib.util.spec_from_file_location("regular_mask_cases",
                                                  pathlib.Path(__file__).with_name("test_gpu_regular_mask.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    run_env(env, lambda: mod.test_inter_dense_regular_vs_oracle("f32"))
    run_env(env, lambda: mod.test_band_regular_vs_general_and_oracle(mod.CASES[1]))
