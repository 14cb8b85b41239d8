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


def test_c5_yeast_quantify_inter(golden):
    """Config C5: quantify with the 11x11 (resized) borders templates on the real 17-chromosome
    yeast map, intra blocks at the cohesin-peak pairs and inter blocks at seeded positions; scores
    within 1e-5 of the reference run (BASELINE.json configs[4]); block assembly, detrend and
    median scaling are redone here from the decoded .cool."""
    cool = golden("yeast_cool")
    g = golden("yeast_quantify")
    max_dist = int(g["max_dist"])
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0,
               max_dist=int(g["cfg_max_dist_bp"]), min_dist=0)
    n_rows = n_inter = 0
    worst = 0.0
    for bi in range(int(g["n_blocks"])):
        ca, cb = (int(x) for x in g[f"b{bi}_chroms"])
        coords = g[f"b{bi}_coords"]
        for ki in range(3):
            key = f"b{bi}_k{ki}_table"
            if key not in g:
                continue
            ref = g[key]
            tab, _ = pipeline.quantify_block(cool, ca, cb, coords, cfg, g[f"kernel{ki}"], max_dist, 11)
            if ref.shape[0] == 0:
                assert tab is None or len(tab) == 0
                continue
            got = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
            assert np.array_equal(got[:, :2], ref[:, :2])
            assert np.array_equal(np.isnan(got[:, 2]), np.isnan(ref[:, 2])), (bi, ki)
            ok = ~np.isnan(ref[:, 2])
            if ok.any():
                worst = max(worst, np.abs(got[ok, 2] - ref[ok, 2]).max())
            assert np.allclose(got[ok, 3], ref[ok, 3], rtol=1e-5, atol=1e-300)
            n_rows += ref.shape[0]
            n_inter += ref.shape[0] if ca != cb else 0
    assert worst < 1e-5
    assert n_rows > 6000 and n_inter > 500
    print(f"C5: {n_rows} scored positions ({n_inter} inter), max |score - reference| = {worst:.2e}")
