"""End-to-end known answers of the reference on its own test file (data_test/example.cool,
decoded to tests/golden/example_cool.npz): the committed outputs
docs/notebooks/detect/example_{loops,borders,hairpins}.tsv (commands in
docs/notebooks/plot_output.ipynb:13-15) and the "89 patterns detected" of `chromosight test`
(chromosight/cli/chromosight.py:185-199)."""
import copy
import io
import time

import os

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


def test_written_files_match_reference_bytes(cool, tmp_path):
    """detect -> chromosight_amd.io.write_patterns: the .tsv the reference committed, line by line; a
    score / p-value may differ in its 10th decimal (1 unit of the last printed digit), nothing else may."""
    from conftest import GOLDEN
    from chromosight_amd import io as cio
    total = exact = 0
    for name, overrides in [("example_loops", dict(pattern="loops", min_dist=8000, max_dist=50000, pearson=0.35)),
                            ("example_borders", dict(pattern="borders")), ("example_hairpins", dict(pattern="hairpins"))]:
        overrides = dict(overrides)
        cfg = copy.deepcopy(getattr(ck, overrides.pop("pattern")))
        cfg.update(overrides)
        table = pipeline.detect_to_files(cool, cfg, str(tmp_path / name), win_fmt="npy")
        wins = np.load(tmp_path / f"{name}.npy")
        assert wins.shape == (len(table),) + np.shape(cfg["kernels"][0])
        assert np.isfinite(wins).any(axis=(1, 2)).all()           # every kept pattern has a window with data
        got = (tmp_path / f"{name}.tsv").read_text().splitlines()
        ref = (GOLDEN / f"{name}.tsv").read_text().splitlines()
        assert len(got) == len(ref) and got[0] == ref[0]
        for a, b in zip(got[1:], ref[1:]):
            total += 1
            if a == b:
                exact += 1
                continue
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[:10] == fb[:10], (a, b)
            for x, y in zip(fa[10:], fb[10:]):
                assert abs(float(x) - float(y)) <= 1.5e-10, (a, b)
    print(f"{exact} of {total} lines byte-identical to the reference's committed .tsv files")
    assert exact >= 0.95 * total
    # quantify reads detect's output back (load_bed2d), windows are written in both formats
    bed = cio.load_bed2d(str(tmp_path / "example_loops.tsv"))
    n_loops = len((GOLDEN / "example_loops.tsv").read_text().splitlines()) - 1
    assert list(bed.columns) == cio.BED2D_COLUMNS and len(bed) == n_loops
    wins = np.arange(2 * 3 * 3, dtype=float).reshape(2, 3, 3)
    wins[0, 0, 0] = np.nan
    cio.save_windows(wins, str(tmp_path / "w"), fmt="json")
    cio.save_windows(wins, str(tmp_path / "w"), fmt="npy")
    import json
    back = json.loads((tmp_path / "w.json").read_text())
    assert sorted(back) == ["0", "1"] and back["1"] == wins[1].tolist()
    assert np.array_equal(np.load(tmp_path / "w.npy"), wins, equal_nan=True)


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


def test_quantify_harness_inter_best_of_kernels(golden):
    """pipeline.quantify = cmd_quantify (cli/chromosight.py:264-470): every position scored with the
    three 11x11 borders templates on intra and inter blocks staged on the device, the reference's
    per-coordinate selection among templates (sort by score, last row of each group), whole-genome bins,
    output order.  Expected values: the reference's per-block, per-template tables (yeast fixture)."""
    cool = golden("yeast_cool")
    g = golden("yeast_quantify")
    off = cool["chrom_offset"]
    names = [str(n) for n in cool["chrom_names"]]
    binsize = int(cool["binsize"])
    rows, want = [], []
    for bi in range(int(g["n_blocks"])):
        ca, cb = (int(x) for x in g[f"b{bi}_chroms"])
        coords = g[f"b{bi}_coords"]
        scores = np.full((coords.shape[0], 3), np.nan)
        for ki in range(3):
            key = f"b{bi}_k{ki}_table"
            if key in g and g[key].shape[0]:
                scores[:, ki] = g[key][:, 2]
        for (r, c), sc in zip(coords, scores):
            rows.append((names[ca], int(r) * binsize, (int(r) + 1) * binsize, names[cb], int(c) * binsize, (int(c) + 1) * binsize))
            want.append((int(off[ca] + r), int(off[cb] + c)) + tuple(sc))
    positions = pd.DataFrame(rows, columns=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
    want = pd.DataFrame(want, columns=["bin1", "bin2", "s0", "s1", "s2"])
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
               kernels=[g[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
    table, windows = pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=int(g["cfg_max_dist_bp"]))
    # the reference's selection rule applied to its own per-template scores
    long = pd.concat([pd.DataFrame({"bin1": want.bin1, "bin2": want.bin2, "score": want[f"s{k}"]}) for k in range(3)],
                     axis=0).reset_index(drop=True)
    exp = long.sort_values("score", ascending=True).groupby(["bin1", "bin2"], sort=False).tail(1)
    exp = exp.sort_values(["bin1", "bin2"]).reset_index(drop=True)
    assert len(table) == len(exp) == len(want.drop_duplicates(["bin1", "bin2"]))
    assert table["bin1"].tolist() == exp["bin1"].tolist() and table["bin2"].tolist() == exp["bin2"].tolist()
    a, b = table["score"].to_numpy(dtype=np.float64), exp["score"].to_numpy(dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    assert np.nanmax(np.abs(a - b)) < 1e-9
    assert windows.shape == (len(table), 11, 11)
    assert np.isnan(table["pvalue"].to_numpy(dtype=np.float64)[np.isnan(a)]).all()
    print(f"quantify: {len(table)} positions, {int(np.isnan(a).sum())} without a valid score")


def _yeast_positions(cool, g):
    names = [str(n) for n in cool["chrom_names"]]
    binsize = int(cool["binsize"])
    rows = []
    for bi in range(int(g["n_blocks"])):
        ca, cb = (int(x) for x in g[f"b{bi}_chroms"])
        for r, c in g[f"b{bi}_coords"]:
            rows.append((names[ca], int(r) * binsize, (int(r) + 1) * binsize, names[cb], int(c) * binsize, (int(c) + 1) * binsize))
    return pd.DataFrame(rows, columns=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])


def _same_quantify_tables(table, windows, ref, ref_windows):
    """Same positions in the same order; windows, scores, p- and q-values to rounding (the two scoring kernels sum a
    window in different orders: a wave per pixel, or a lane per pixel on runs of neighbours)."""
    assert len(table) == len(ref) and list(table.columns) == list(ref.columns)
    for col in ref.columns:
        a, b = table[col].to_numpy(), ref[col].to_numpy()
        if a.dtype == object:
            assert (a == b).all(), col
        elif col in ("score", "pvalue", "qvalue"):
            assert np.array_equal(np.isnan(a.astype(float)), np.isnan(b.astype(float))), col
            assert np.allclose(a.astype(float), b.astype(float), rtol=1e-9, atol=1e-12, equal_nan=True), col
        else:
            assert np.array_equal(a, b, equal_nan=True), col
    # (two stagings of a block differ at the 1e-16 level: the distance law is summed in LDS in arrival order)
    assert np.array_equal(np.isnan(windows), np.isnan(ref_windows))
    assert np.allclose(windows, ref_windows, rtol=0, atol=1e-12, equal_nan=True)


def test_quantify_batched_equals_block_by_block(golden, monkeypatch):
    """One native call per template over all sub-matrices (cs_quantify_blocks) == one call per sub-matrix and template
    (cs_quantify_pixels): the same table (scores to rounding) and the same windows."""
    cool = golden("yeast_cool")
    g = golden("yeast_quantify")
    positions = _yeast_positions(cool, g)
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
               kernels=[g[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
    md = int(g["cfg_max_dist_bp"])
    batched, win_b = pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=md)
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_QUANTIFY_BATCH", "1")
    single, win_s = pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=md)
    assert len(batched) == len(single) > 2000
    _same_quantify_tables(batched, win_b, single, win_s)


QUANTIFY_WORKER = r"""
import os, sys, numpy as np, pandas as pd
sys.path.insert(0, os.environ["CS_ROOT"])
import torch.distributed as dist
from chromosight_amd import parallel
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
cool = dict(np.load(os.path.join(os.environ["CS_ROOT"], "tests", "golden", "yeast_cool.npz"), allow_pickle=True))
g = dict(np.load(os.path.join(os.environ["CS_ROOT"], "tests", "golden", "yeast_quantify.npz"), allow_pickle=True))
positions = pd.read_pickle(os.environ["CS_POS"])
cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
           kernels=[g[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
table, windows = parallel.quantify_genome(cool, positions, cfg, inter=True, max_dist_bp=int(g["cfg_max_dist_bp"]))
table.to_pickle(os.environ["CS_OUT"] + f".{dist.get_rank()}.pkl")
np.save(os.environ["CS_OUT"] + f".{dist.get_rank()}.npy", windows)
dist.destroy_process_group()
"""


def test_quantify_two_ranks_equal_single_process(golden, tmp_path):
    """parallel.quantify_genome on 2 ranks (gloo rendezvous, both ranks on this GPU): the sub-matrices that hold a position are
    dealt to the ranks (cli/chromosight.py:396-410 pools over them), one exchange carries scores and windows, and BOTH ranks
    return the single-process table (which test_quantify_harness_inter_best_of_kernels pins against the reference)."""
    import subprocess
    import sys
    cool = golden("yeast_cool")
    g = golden("yeast_quantify")
    positions = _yeast_positions(cool, g)
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
               kernels=[g[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
    single, win_s = pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=int(g["cfg_max_dist_bp"]))
    pos_file = tmp_path / "positions.pkl"
    positions.to_pickle(pos_file)
    script = tmp_path / "worker.py"
    script.write_text(QUANTIFY_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "q"
    env = dict(os.environ, CS_ROOT=root, CS_OUT=str(out), CS_POS=str(pos_file), CHROMOSIGHT_HIP_DEVICE="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(2):
        _same_quantify_tables(pd.read_pickle(str(out) + f".{r}.pkl"), np.load(str(out) + f".{r}.npy"), single, win_s)


def test_detect_inter_and_subsample_run(golden):
    """--inter and --subsample through pipeline.detect: inter blocks are staged (median scaling) and
    scanned on the device; a seeded subsample is reproducible and keeps the requested proportion."""
    cool = golden("yeast_cool")
    cfg = copy.deepcopy(ck.loops)
    dcool = pipeline.DeviceCool(cool)
    sub_a = dcool.subsampled(0.5, seed=3, inter=True)
    sub_b = dcool.subsampled(0.5, seed=3, inter=True)
    assert np.array_equal(sub_a.host["count"], sub_b.host["count"])
    kept = sub_a.host["count"].sum() / np.asarray(cool["count"]).sum()
    assert 0.4 < kept < 0.62            # intra blocks draw from both triangles, only the upper one is stored
    small = {k: v for k, v in cool.items()}
    table = pipeline.detect(small, cfg, inter=True)
    assert len(table) > 0
    inter_rows = table[table.chrom1 != table.chrom2]
    intra_only = pipeline.detect(small, cfg, inter=False)
    assert len(table) - len(inter_rows) <= len(intra_only) + len(inter_rows)
    print(f"detect --inter on the yeast map: {len(table)} patterns ({len(inter_rows)} inter-chromosomal), "
          f"{len(intra_only)} without --inter")
