"""End-to-end known answers of the reference on its own test file (data_test/example.cool,
decoded to tests/golden/example_cool.npz): the committed outputs
docs/notebooks/detect/example_{loops,borders,hairpins}.tsv (commands in
docs/notebooks/plot_output.ipynb:13-15) and the "89 patterns detected" of `chromosight test`
(chromosight/cli/chromosight.py:185-199)."""
import copy
import io
import time

import numpy as np
import pandas as pd
import pytest

import chromosight_amd.kernels as ck
from chromosight_amd import pipeline

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cool(golden):
    return golden("example_cool")


def load_tsv(name):
    from conftest import GOLDEN
    return pd.read_csv(GOLDEN / f"{name}.tsv", sep="\t")


def as_written(table):
    """Round-trip through the reference's writer format (io.py:208-226: %.10f)."""
    buf = io.StringIO()
    table.to_csv(buf, sep="\t", index=None, float_format="%.10f")
    buf.seek(0)
    return pd.read_csv(buf, sep="\t")


def test_default_loops_89_patterns(cool):
    cfg = copy.deepcopy(ck.loops)
    t0 = time.perf_counter()
    table = pipeline.detect(cool, cfg)
    dt = time.perf_counter() - t0
    assert len(table) == 89
    print(f"detect (loops, example.cool, 3 chromosomes): {dt:.3f} s")


@pytest.mark.parametrize("name,overrides", [
    ("example_loops", dict(pattern="loops", min_dist=8000, max_dist=50000, pearson=0.35)),
    ("example_borders", dict(pattern="borders")),
    ("example_hairpins", dict(pattern="hairpins")),
])
def test_committed_outputs(cool, name, overrides):
    overrides = dict(overrides)
    cfg = copy.deepcopy(getattr(ck, overrides.pop("pattern")))
    cfg.update(overrides)
    got = as_written(pipeline.detect(cool, cfg))
    ref = load_tsv(name)
    assert len(got) == len(ref)
    for col in ("chrom1", "start1", "end1", "chrom2", "start2", "end2", "bin1", "bin2", "kernel_id", "iteration"):
        assert got[col].tolist() == ref[col].tolist(), col
    assert np.allclose(got["score"], ref["score"], rtol=0, atol=2e-10)
    assert np.allclose(got["pvalue"], ref["pvalue"], rtol=0, atol=2e-10)
    assert np.allclose(got["qvalue"], ref["qvalue"], rtol=0, atol=2e-10)
