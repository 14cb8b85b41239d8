"""The device-resident block pipeline (chromosight_amd/pipeline.py DeviceCool / detect_block,
parallel.detect_genome): genome-wide pixel table in HBM -> band extents -> distance law -> fused
detrend + CSR->band tiler -> correlation -> device foci and validation statistics.

Checked against (a) the reference's own captures for data_test/example.cool (laws, prepared blocks,
pattern tables) and (b) an independent CPU pipeline made of the pinned oracles
(oracle/detrend_oracle.py, oracle/oracle.c, oracle/foci_oracle.py) on the synthetic C4 genome of
BASELINE.md -- 200 000 bins in 23 blocks, loops + the three borders templates."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from chromosight_amd._lib import LAYOUT_BAND, get_device
from oracle import c_oracle, detrend_oracle, foci_oracle
from tools.synthetic_genome import make_cool

pytestmark = pytest.mark.gpu


def download_block(dcool, block):
    """Staged block -> dense numpy (float64)."""
    dev = dcool.dev
    n = block.shape[0]
    sig = block.sig
    dt = np.float64 if sig.dtype == 1 else np.float32
    host = np.empty((n, sig.ld), dtype=dt)
    dev._check(dev.lib.cs_memcpy_d2h(dev.ctx, host.ctypes.data, sig.d_ptr, host.nbytes, None))
    if sig.layout in (LAYOUT_BAND, 3):               # (3: CS_LAYOUT_BAND_PADDED, a band with zero slots behind its diagonals)
        out = np.zeros((n, n))
        for d in range(sig.band_w):
            off = sig.band_lo + d
            idx = np.arange(max(0, -off), min(n, n - off))
            out[idx, idx + off] = host[idx, d]
        return out
    return host[:, :n].astype(np.float64)


def coo(g, prefix):
    return sp.coo_matrix((g[f"{prefix}_val"], (g[f"{prefix}_row"], g[f"{prefix}_col"])), shape=tuple(g[f"{prefix}_shape"]))


def test_fused_tiler_matches_reference_prepared_blocks(golden):
    """cs_csr_band_extent + cs_distance_law_csr/_finish + cs_csr_to_band(law) on views of the genome
    CSR == ContactMap.create_mat of the reference (detrend, >= 10 -> 1, trim, NaN -> 0), <= 1e-11."""
    dcool = pipeline.DeviceCool(golden("example_cool"))
    g = golden("example_blocks")
    for ci in range(3):
        for pname in ("loops", "borders", "hairpins"):
            max_dist = int(g[f"chr{ci}_{pname}_max_dist"])
            keep = int(g[f"chr{ci}_{pname}_keep"])
            n = dcool.chrom_size(ci)
            largest = keep - min(max_dist, n)
            block = dcool.stage_intra(ci, max_dist, largest)
            assert block.keep == keep
            got = download_block(dcool, block)
            ref = coo(g, f"chr{ci}_{pname}_prepared").toarray()
            assert np.abs(got - ref).max() < 1e-11, (ci, pname)
            assert np.array_equal(dcool.block_bins(ci), g[f"chr{ci}_det"])


def test_batched_staging_matches_reference_and_the_block_by_block_path(golden):
    """cs_stage_blocks (all chromosomes by three launches: law pass, finish, detrend / tiler with float64 + float32
    outputs) == the reference's prepared blocks (<= 1e-11), == the block-by-block kernels bit for bit up to the
    summation order of the law (<= 1e-13 relative), float32 copy == the float64 band rounded."""
    for name, max_dists in (("example_cool", (2000, 1, 60)), ("yeast_cool", (1000, 1, 100))):
        dcool = pipeline.DeviceCool(golden(name))
        chroms = list(range(dcool.n_chrom))
        for max_dist in max_dists:
            fast = dcool.stage_blocks(chroms, max_dist, 17)
            assert all(b.sig32 is not None for b in fast), "the batched entry did not serve the call"
            for ci, blk in zip(chroms, fast):
                slow = dcool.stage_intra(ci, max_dist, 17)
                a, b = download_block(dcool, blk), download_block(dcool, slow)
                assert blk.sig.layout == slow.sig.layout and blk.sig.ld == slow.sig.ld and blk.keep == slow.keep
                scale = max(np.abs(b).max(), 1e-300)
                assert np.abs(a - b).max() <= 1e-13 * scale, (name, max_dist, ci)
                assert np.array_equal(a == 0, b == 0)
                n = blk.shape[0]
                ld = int(blk.sig.ld)
                host32 = blk.buffer32.download().view(np.float32).reshape(-1)[:n * ld].reshape(n, ld)
                host64 = blk.buffer.download().view(np.float64).reshape(-1)[:n * ld].reshape(n, ld)
                assert np.array_equal(host32, host64.astype(np.float32)), (name, max_dist, ci)
                assert not host64[:, (blk.sig.band_w if blk.sig.layout == 1 else n):].any()          # zeroed padding
    # float32-only staging (map-level callers): the same values rounded once
    dcool = pipeline.DeviceCool(golden("yeast_cool"))
    for b64, b32 in zip(dcool.stage_blocks([0, 3, 11], 100, 17), dcool.stage_blocks([0, 3, 11], 100, 17, band_dtype=np.float32)):
        assert b32.sig.dtype == 0 and b32.sig.ld == b64.sig.ld
        assert np.array_equal(download_block(dcool, b32), download_block(dcool, b64).astype(np.float32).astype(np.float64))
    dcool = pipeline.DeviceCool(golden("example_cool"))
    g = golden("example_blocks")
    for pname in ("loops", "borders", "hairpins"):
        max_dist = int(g[f"chr0_{pname}_max_dist"])
        largest = int(g[f"chr0_{pname}_keep"]) - min(max_dist, dcool.chrom_size(0))
        for ci, blk in enumerate(dcool.stage_blocks([0, 1, 2], max_dist, largest)):
            ref = coo(g, f"chr{ci}_{pname}_prepared").toarray()
            assert np.abs(download_block(dcool, blk) - ref).max() < 1e-11, (ci, pname)


def test_inter_blocks_staged_together_equal_one_by_one(golden):
    """DeviceCool.stage_inter_many (the medians of all blocks from one native call, cs_csr_median_many: two synchronisations in
    all) == stage_inter block by block (cs_csr_median each): the same maps bit for bit; a pair whose block stores nothing gets
    the NaN median and an all-zero map either way."""
    dcool = pipeline.DeviceCool(golden("yeast_cool"))
    pairs = [(0, 1), (2, 7), (3, 15), (0, 16), (10, 11), (5, 6), (1, 2)]
    many = dcool.stage_inter_many(pairs)
    for (ca, cb), blk in zip(pairs, many):
        one = dcool.stage_inter(ca, cb, resident=True)
        a, b = download_block(dcool, blk), download_block(dcool, one)
        assert blk.shape == one.shape and blk.inter and a.shape == b.shape
        assert np.array_equal(a, b), (ca, cb)
        assert np.count_nonzero(a) > 0


def test_batched_staging_with_a_long_distance_law():
    """Laws of 3000+ diagonals: the law pass then needs more than the default 64 KB of dynamic LDS (only the
    tiler had asked for it) -- cs_stage_blocks == the block-by-block kernels up to the summation order of the law."""
    cool, _ = make_cool(15_500, 3_400, 2000, seed=11, chrom_sizes=[8_000, 7_500], loops_per_10k=0)
    dcool = pipeline.DeviceCool(cool)
    fast = dcool.stage_blocks([0, 1], 3_400, 17)
    assert all(b.sig32 is not None for b in fast), "the batched entry did not serve the call"
    for ci, blk in enumerate(fast):
        assert blk.sig.band_w > 3_100
        a, b = download_block(dcool, blk), download_block(dcool, dcool.stage_intra(ci, 3_400, 17))
        assert np.abs(a - b).max() <= 1e-13 * max(np.abs(b).max(), 1e-300), ci
        assert np.array_equal(a == 0, b == 0)


def test_device_blocks_reproduce_reference_tables(golden, templates):
    """detect_block on the staged example chromosomes: tables bit-identical to the reference's
    pattern_detector captures (coordinates and order), scores <= 1e-9, windows <= 1e-12."""
    dcool = pipeline.DeviceCool(golden("example_cool"))
    g = golden("example_blocks")
    pats = {
        "loops": (dict(pearson=0.3, max_perc_undetected=50.0, max_perc_zero=10.0, max_dist=2000000), [templates["loops"]]),
        "borders": (dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), templates["borders"]),
        "hairpins": (dict(pearson=0.1, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0), [templates["hairpin"]]),
    }
    n_rows = 0
    for ci in range(3):
        for pname, (cfg, kernels) in pats.items():
            max_dist = int(g[f"chr{ci}_{pname}_max_dist"])
            largest = int(g[f"chr{ci}_{pname}_keep"]) - min(max_dist, dcool.chrom_size(ci))
            block = dcool.stage_intra(ci, max_dist, largest, resident=True)
            for ki, kern in enumerate(kernels):
                tag = f"chr{ci}_{pname}{ki}"
                tab, wins = pipeline.detect_block(dcool, block, cfg, kern)
                ref_tab = g[f"{tag}_table"]
                if ref_tab.shape[0] == 0:
                    assert tab is None or len(tab) == 0
                    continue
                got = tab[["bin1", "bin2", "score", "pvalue"]].to_numpy(dtype=np.float64)
                assert np.array_equal(got[:, :2], ref_tab[:, :2]), tag
                assert np.abs(got[:, 2] - ref_tab[:, 2]).max() < 1e-9, tag
                assert np.allclose(got[:, 3], ref_tab[:, 3], rtol=1e-6, atol=1e-300), tag
                assert np.allclose(wins, g[f"{tag}_windows"], equal_nan=True, rtol=0, atol=1e-12), tag
                n_rows += ref_tab.shape[0]
    assert n_rows > 200


# ------------------------------------------------------------------------------------------------
# synthetic genome: device pipeline == CPU pipeline made of the pinned oracles
# ------------------------------------------------------------------------------------------------
def oracle_block_tables(cool, ci, cfg, max_dist, kernels, binsize):
    """(bin1, bin2, score) tables of one chromosome, one per template, from the CPU oracles."""
    largest = max(k.shape[0] for k in kernels)
    off = cool["chrom_offset"]
    n = int(off[ci + 1] - off[ci])
    md = max_dist
    keep = min(md, n) + largest
    band, det = detrend_oracle.balanced_band(cool, ci, keep)
    prepared, _ = detrend_oracle.prepare_band(band, det)
    miss = ~det
    out = []
    for kern in kernels:
        if n <= max(kern.shape):
            out.append(np.zeros((0, 3)))
            continue
        out_w = min(md, n - 1) + 1
        corr, _ = c_oracle.normxcorr2_band(prepared, n, 0, prepared.shape[1], kern, 0, n, 0, out_w, max_dist=md,
                                           miss_row=miss, miss_col=miss,
                                           missing_tol=cfg["max_perc_undetected"] / 100)
        out.append(foci_oracle.detect_table_band(prepared, 0, corr, 0, n, miss, kern.shape, cfg["pearson"],
                                                 cfg["max_perc_zero"] / 100, cfg["max_perc_undetected"] / 100,
                                                 diag_only=cfg["max_dist"] == 0))
    return out


def compare_genome(total_bins, max_dist_bins, seed=2, replay_steps=0):
    """replay_steps > 0: parallel.genome_step that many times on the same genome -- the first step the usual way, every later
    one the recorded call list replayed natively (chromosight_amd/plan.py, what bench.py's `sharded_genome` leg times) -- and
    the LAST step's records against the same oracle tables (not against detect_patterns)."""
    binsize = 2000
    oracle_tables.clear()
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, planted = make_cool(total_bins, max_dist_bins, binsize, seed=seed, template=template)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = max_dist_bins * binsize
    borders = copy.deepcopy(ck.borders)
    # ... and both patterns side by side (parallel.detect_patterns)
    both = parallel.detect_patterns(dcool, [loops, borders])
    for cfg, rec in zip((loops, borders), both):
        ref = parallel.detect_genome(dcool, cfg)
        assert rec.shape == ref.shape and np.array_equal(rec[:, [0, 1, 2, 5, 6]], ref[:, [0, 1, 2, 5, 6]])
        assert np.abs(rec[:, 3] - ref[:, 3]).max() < 1e-12
    n_found = {}
    # what the bench does: every block staged once at the loops' keep distance, the borders templates on band views
    shared = parallel.stage_genome(dcool, [loops, borders])
    for name, cfg in (("loops", loops), ("borders", borders)):
        rec = parallel.detect_genome(dcool, cfg)
        rec_shared = parallel.detect_genome(dcool, cfg, staged=shared)
        # same foci in the same order; scores to the last bits (a law is a float64 sum whose order varies from run to run)
        assert rec_shared.shape == rec.shape and np.array_equal(rec_shared[:, [0, 1, 2, 5, 6]], rec[:, [0, 1, 2, 5, 6]]), name
        assert np.abs(rec_shared[:, 3] - rec[:, 3]).max() < 1e-12, name
        md = max(cfg["max_dist"] // binsize, 1)
        kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
        total = 0
        for ci in range(dcool.n_chrom):
            want = oracle_block_tables(cool, ci, cfg, md, kernels, binsize)
            oracle_tables[(name, ci)] = want
            total += check_against_oracle(rec, want, ci, name)
        n_found[name] = total
    if replay_steps:
        dcool.__dict__.pop("_step_plans", None)
        steps = [parallel.genome_step(dcool, [loops, borders]) for _ in range(replay_steps)]
        plans = dcool.__dict__["_step_plans"]
        assert len(plans) == 1 and all(p.ok for p in plans.values()), [p.why for p in plans.values()]
        for name, rec in zip(("loops", "borders"), steps[-1]):
            total = sum(check_against_oracle(rec, oracle_tables[(name, ci)], ci, name + " (replayed step)") for ci in range(dcool.n_chrom))
            assert total == n_found[name]
    return n_found, planted


oracle_tables = {}


def check_against_oracle(rec, want, ci, name):
    """The records of chromosome ci (detect_genome's layout) against the oracle's per-template tables: same foci in the same
    order, scores to 1e-9.  Returns the number of patterns."""
    total = 0
    for ki, tab in enumerate(want):
        got = rec[(rec[:, 0] == ci) & (rec[:, 5] == ki)]
        assert got.shape[0] == tab.shape[0], (name, ci, ki, got.shape[0], tab.shape[0])
        if tab.shape[0]:
            assert np.array_equal(got[:, 1:3], tab[:, :2]), (name, ci, ki)      # same foci, same order
            assert np.abs(got[:, 3] - tab[:, 2]).max() < 1e-9, (name, ci, ki)
        total += tab.shape[0]
    return total


# CS_GENOME_SEEDS="3,4,5": more seeded genomes for an occasional long run (seeds 3 .. 14 were run at the end of round 3)
@pytest.mark.parametrize("seed", [2] + [int(x) for x in os.environ.get("CS_GENOME_SEEDS", "").split(",") if x])
def test_synthetic_genome_small_vs_oracle_pipeline(seed):
    found, planted = compare_genome(30_000, 300, seed=seed)
    print(f"30k-bin genome: {found} patterns, {len(planted)} planted loops")
    assert found["loops"] > 100 and found["borders"] >= 0


@pytest.mark.parametrize("sizes,md,seed", [([12, 36, 400, 400], 400, 101), ([900, 60, 400, 36, 36, 400, 19, 60], 40, 121),
                                           ([2500, 19, 400, 12, 150], 1, 123), ([36, 36, 900, 150], 150, 122),
                                           # a draft assembly: 240 contigs of 20 .. 160 bins (240 blocks in every batch entry)
                                           ([20 + (37 * i * i + 11 * i) % 141 for i in range(240)], 60, 131)])
def test_genomes_with_very_short_chromosomes_vs_oracle_pipeline(sizes, md, seed):
    """Chromosomes of a few dozen bins next to long ones (the long form: tools/fuzz_genomes.py): blocks not larger than the
    template are skipped, blocks short enough to be staged dense take the per-block calls -- where, in float32 mode, a 1-D
    pattern once kept the candidates of EVERY diagonal of a dense block (the scanned range was reset for dense layouts) and
    reported a focus at the wrong bin.  loops + borders + hairpins side by side == the CPU pipeline of the oracles."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(sum(sizes), md, 2000, seed=seed, template=template, chrom_sizes=sizes)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = md * 2000
    cfgs = [loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)]
    recs = parallel.detect_patterns(dcool, cfgs, staged=parallel.stage_genome(dcool, cfgs))
    total = 0
    for cfg, rec in zip(cfgs, recs):
        mdc = max(cfg["max_dist"] // 2000, 1)
        kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
        for ci in range(dcool.n_chrom):
            for ki, tab in enumerate(oracle_block_tables(cool, ci, cfg, mdc, kernels, 2000)):
                got = rec[(rec[:, 0] == ci) & (rec[:, 5] == ki)]
                assert got.shape[0] == tab.shape[0], (cfg["name"], ci, ki, got.shape[0], tab.shape[0])
                if tab.shape[0]:
                    assert np.array_equal(got[:, 1:3], tab[:, :2]), (cfg["name"], ci, ki)
                    assert np.abs(got[:, 3] - tab[:, 2]).max() < 1e-9, (cfg["name"], ci, ki)
                total += tab.shape[0]
    assert total > 50


@pytest.mark.parametrize("win", [9, 13, 23, 33])
def test_win_size_templates_vs_oracle_pipeline(win):
    """--win-size (cli/chromosight.py:689-695): every template resized (order-1 spline, pipeline.with_win_size) before the
    scan; templates smaller and larger than the built-in ones through the same chains == the CPU pipeline of the oracles.
    (Templates above 17 x 17 leave the masked tile kernel: the many-blocks entry once built the matrix-core weight image for
    them anyway and wrote past its host buffer -- a crash at --win-size >= 19 with a 2-D pattern; now it answers
    "unsupported" and the blocks take the per-block calls.)"""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    sizes = [3000, 1200, 500, 30]
    cool, _ = make_cool(sum(sizes), 200, 2000, seed=310 + win, template=template, chrom_sizes=sizes)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    cfgs = [pipeline.with_win_size(c, win) for c in (loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins))]
    assert all(np.shape(k) == (win, win) for c in cfgs for k in c["kernels"])
    recs = parallel.detect_patterns(dcool, cfgs, staged=parallel.stage_genome(dcool, cfgs))
    total = 0
    for cfg, rec in zip(cfgs, recs):
        mdc = max(cfg["max_dist"] // 2000, 1)
        kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
        for ci in range(dcool.n_chrom):
            for ki, tab in enumerate(oracle_block_tables(cool, ci, cfg, mdc, kernels, 2000)):
                got = rec[(rec[:, 0] == ci) & (rec[:, 5] == ki)]
                assert got.shape[0] == tab.shape[0], (cfg["name"], ci, ki, got.shape[0], tab.shape[0])
                if tab.shape[0]:
                    assert np.array_equal(got[:, 1:3], tab[:, :2]), (cfg["name"], ci, ki)
                    assert np.abs(got[:, 3] - tab[:, 2]).max() < 1e-9, (cfg["name"], ci, ki)
                total += tab.shape[0]
    assert total > 20


def test_c4_genome_200k_vs_oracle_pipeline():
    """C4 of BASELINE.md at full size: 200 000 bins, 23 blocks, max_dist 1000, loops + 3 borders
    templates; every pattern table equal to the CPU oracle pipeline's (coordinates bit-exact and in
    order, scores <= 1e-9)."""
    found, planted = compare_genome(200_000, 1000, replay_steps=3)       # (... and the replayed genome step bench.py times)
    print(f"C4 genome: {found} patterns, {len(planted)} planted loops")
    assert found["loops"] > 500 and found["borders"] > 10000


# ------------------------------------------------------------------------------------------------
# two ranks, real detector: detect_genome over gloo with both ranks on this GPU
# ------------------------------------------------------------------------------------------------
WORKER = r"""
import os, sys, copy, numpy as np
sys.path.insert(0, os.environ["CS_ROOT"])
import torch.distributed as dist
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(12_000, 200, 2000, seed=5, template=template)
dcool = pipeline.DeviceCool(cool)
cfg = copy.deepcopy(ck.loops); cfg["max_dist"] = 200 * 2000; cfg["max_iterations"] = 2
rec = parallel.detect_genome(dcool, cfg)
# ... and two patterns side by side on blocks staged once (what bench.py's sharded genome does on every rank)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 200 * 2000
rec_l, rec_b = parallel.detect_patterns(dcool, [loops, copy.deepcopy(ck.borders)])
if dist.get_rank() == 0:
    np.save(os.environ["CS_OUT"], rec)
    np.save(os.environ["CS_OUT"] + ".loops.npy", rec_l)
    np.save(os.environ["CS_OUT"] + ".borders.npy", rec_b)
dist.destroy_process_group()
"""


def test_two_ranks_equal_single_process(tmp_path):
    """parallel.detect_genome with the real device detector on 2 ranks (gloo rendezvous, both ranks on
    this GPU) == the single-process tables, including the second iteration whose template is the
    all-reduced pileup of the first."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(12_000, 200, 2000, seed=5, template=template)
    cfg = copy.deepcopy(ck.loops)
    cfg["max_dist"] = 200 * 2000
    cfg["max_iterations"] = 2
    single = parallel.detect_genome(pipeline.DeviceCool(cool), cfg)
    assert single.shape[0] > 50 and (single[:, 6] == 1).any()
    out = tmp_path / "rec.npy"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CS_ROOT=root, CS_OUT=str(out), CHROMOSIGHT_HIP_DEVICE="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    both = np.load(out)
    assert both.shape == single.shape
    assert np.array_equal(both[:, [0, 1, 2, 5, 6]], single[:, [0, 1, 2, 5, 6]])
    # scores agree to rounding: the distance law is a float64 atomic reduction (order-dependent at the
    # 1e-16 level) and the second template is a pileup summed in another order
    assert np.abs(both[:, 3] - single[:, 3]).max() < 1e-9
    assert np.allclose(both[:, 4], single[:, 4], rtol=1e-6, atol=1e-300)
    # detect_patterns on two ranks (blocks staged once per rank, the three borders templates in one chain, one record
    # exchange per pattern) == the single process: records of every template in the single-process order
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    want_l, want_b = parallel.detect_patterns(pipeline.DeviceCool(cool), [loops, copy.deepcopy(ck.borders)])
    for name, want in (("loops", want_l), ("borders", want_b)):
        got = np.load(str(out) + f".{name}.npy")
        assert got.shape == want.shape and want.shape[0] > 30, name
        assert np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]]), name
        assert np.abs(got[:, 3] - want[:, 3]).max() < 1e-9, name
    assert set(np.unique(want_b[:, 5])) == {0.0, 1.0, 2.0}


def test_genome_step_as_one_native_call_equals_the_two_calls():
    """parallel.genome_step: the first step of a layout runs stage_genome + detect_patterns and records the library calls it
    made; every later step is ONE cs_run_calls on those arguments (chromosight_amd/plan.py) -- staging, both patterns' chains
    (the 1-D chain behind cs_stream_wait_tiles: it starts when the 2-D chain's tile workgroups are resident) and the acceptance
    rules recomputed natively -- and must give the records of the two calls, every time; a second genome (other data, same
    code path) gets its own plan."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    borders = copy.deepcopy(ck.borders)
    for seed in (5, 6):
        # (chromosomes long enough to be staged as bands for both patterns: a short one staged dense for the loops needs a
        # second staging call for the borders, and such a step is not planned)
        cool, _ = make_cool(12_000, 200, 2000, seed=seed, template=template, chrom_sizes=[4000, 3500, 2500, 2000])
        dcool = pipeline.DeviceCool(cool)
        staged = parallel.stage_genome(dcool, [loops, borders])
        want = parallel.detect_patterns(dcool, [loops, borders], staged=staged)
        first = parallel.genome_step(dcool, [loops, borders])
        plans = dcool.__dict__["_step_plans"]
        assert len(plans) == 1 and all(p.ok for p in plans.values()), [p.why for p in plans.values()]
        for step in [first] + [parallel.genome_step(dcool, [loops, borders]) for _ in range(4)]:
            for got, ref in zip(step, want):
                assert got.shape == ref.shape and ref.shape[0] > 30
                assert np.array_equal(got[:, [0, 1, 2, 5, 6]], ref[:, [0, 1, 2, 5, 6]])
                assert np.abs(got[:, 3] - ref[:, 3]).max() < 1e-9
                assert np.allclose(got[:, 4], ref[:, 4], rtol=1e-9, atol=1e-300)


def test_2d_chain_retry_and_fallback_paths(monkeypatch):
    """The 2-D chain behind the tile kernels is enqueued before the candidate counts are known (cs_detect_foci_blocks: segments
    formed on the device, launches sized for a bound).  Its three ways out, forced here, must give the default run's records:
    a block's list outgrows its room (the call goes round again with more: CHROMOSIGHT_HIP_TEST_CAND_CAP), more candidates than
    the launches were sized for (the host-paced chain on the same lists: CHROMOSIGHT_HIP_TEST_DEFER_BOUND), a block with more
    candidates than the labelling workgroup's LDS arrays hold (a low threshold: the sorted route, then remembered for the layout)
    -- as a plain call sequence and as a replayed genome step (prepare form + full form)."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(12_000, 200, 2000, seed=5, template=template, chrom_sizes=[4000, 3500, 2500, 2000])
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    borders = copy.deepcopy(ck.borders)

    def same(got, want, what):
        assert got.shape == want.shape and want.shape[0] > 30, what
        assert np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]]), what
        assert np.abs(got[:, 3] - want[:, 3]).max() < 1e-9, what

    # (0.11: ~ 10^4 candidates per block in float32 arithmetic; 0.08: below 0.1 the 2-D pattern is evaluated in float64, block by
    # block on worker threads -- not a planned step, but the same records)
    for cfg2 in (loops, dict(loops, pearson=0.11), dict(loops, pearson=0.08)):
        monkeypatch.setenv("CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN", "1")
        want = parallel.detect_genome(pipeline.DeviceCool(cool), cfg2)
        monkeypatch.delenv("CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN")
        for env in ({}, {"CHROMOSIGHT_HIP_TEST_CAND_CAP": "100"}, {"CHROMOSIGHT_HIP_TEST_DEFER_BOUND": "50"},
                    {"CHROMOSIGHT_HIP_TEST_CAND_CAP": "100", "CHROMOSIGHT_HIP_TEST_DEFER_BOUND": "50"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            dcool = pipeline.DeviceCool(cool)
            for rep in range(3):
                same(parallel.detect_genome(dcool, cfg2), want, (cfg2["pearson"], env, rep))
            dcool = pipeline.DeviceCool(cool)
            want_b = None
            for rep in range(4):                               # (steps 2 .. 4: the replayed list)
                got, got_b = parallel.genome_step(dcool, [cfg2, borders])
                same(got, want, (cfg2["pearson"], env, "step", rep))
                want_b = got_b if want_b is None else want_b
                same(got_b, want_b, "borders beside it")
            assert cfg2["pearson"] < 0.1 or all(p.ok for p in dcool.__dict__["_step_plans"].values())
            for k in env:
                monkeypatch.delenv(k)


def test_single_pattern_steps_are_planned_and_feed_pipeline_detect():
    """A 2-D pattern alone (loops) and a 1-D pattern alone (borders) are steps a StepPlan covers too: replayed steps equal
    detect_genome's records; pipeline.detect -- the CLI counterpart -- runs the same orchestration (parallel.genome_step: its
    second call on a DeviceCool is the replayed list) and returns the same table every time; so is hairpins (ONE 1-D template:
    the joint chain serves 1 to 4 templates); two iterations are not planned and take the usual calls."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(12_000, 200, 2000, seed=7, template=template, chrom_sizes=[4000, 3500, 2500, 2000])
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    for cfg in (loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)):
        dcool = pipeline.DeviceCool(cool)
        want = parallel.detect_genome(dcool, cfg)
        steps = [parallel.genome_step(dcool, [cfg]) for _ in range(4)]
        plans = dcool.__dict__["_step_plans"]
        assert len(plans) == 1 and all(p.ok for p in plans.values()), [p.why for p in plans.values()]
        for step in steps:
            got = step[0]
            assert got.shape == want.shape and want.shape[0] > (30 if cfg["max_dist"] or len(cfg["kernels"]) > 1 else 3)
            assert np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]])
            assert np.abs(got[:, 3] - want[:, 3]).max() < 1e-9
        dcool = pipeline.DeviceCool(cool)
        tables = [pipeline.detect(dcool, cfg) for _ in range(3)]
        assert all(p.ok for p in dcool.__dict__["_step_plans"].values())
        assert len(tables[0]) > (20 if cfg["max_dist"] or len(cfg["kernels"]) > 1 else 2)
        for t in tables[1:]:
            assert t[["bin1", "bin2", "kernel_id"]].equals(tables[0][["bin1", "bin2", "kernel_id"]])
            assert np.abs(t["score"].to_numpy() - tables[0]["score"].to_numpy()).max() < 1e-9
    for cfg in (dict(loops, max_iterations=2),):
        dcool = pipeline.DeviceCool(cool)
        a, b = pipeline.detect(dcool, cfg), pipeline.detect(dcool, cfg)
        assert not dcool.__dict__.get("_step_plans") and a[["bin1", "bin2"]].equals(b[["bin1", "bin2"]])


def test_lazy_float64_bands_equal_the_stored_ones(monkeypatch):
    """stage_genome's default: the float64 band of a block is stored for its first diagonals only and the float64 kernels
    recompute every other pixel they read from the pixel table (cs_stage_block.d_lazy).
    (1) On ONE staging that stores every diagonal and builds the descriptors (lazy64="all"), the same blocks read through
    copies of their descriptors with near_w cut to 0 / 8 / 64 diagonals give the records, scores and windows of the stored
    bands BIT FOR BIT, for the 2-D pattern and for the 1-D pattern's templates.
    (2) The default staging == the stored-band staging (CHROMOSIGHT_HIP_F64_TWIN=1) up to the arrival order of the distance
    law's additions (two stagings differ in the last bit), and a lazy block handed to a per-block path is staged again."""
    from chromosight_amd._lib import CS_F64, LAYOUT_BAND, LAYOUT_BAND_LAZY, CsMatrix
    from chromosight_amd.utils import detection as cud
    from chromosight_amd import engine
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    borders = copy.deepcopy(ck.borders)
    cool, _ = make_cool(12_000, 200, 2000, seed=7, template=template, chrom_sizes=[4000, 3500, 2500, 2000])
    dcool = pipeline.DeviceCool(cool)
    dev = dcool.dev

    # ---- (1) same staging, stored against recomputed
    blocks = dcool.stage_blocks([0, 1, 2, 3], 200, 17, lazy64="all")
    dev.sync()
    assert all(b.sig.layout == LAYOUT_BAND_LAZY for b in blocks)
    border_blocks = [dcool.view_for(b, 1, 17) for b in blocks]
    assert all(v is not None for v in border_blocks)

    def with_sig(blks, make):
        out = []
        for b in blks:
            c = copy.copy(b)
            c.restage, c._full = None, None
            c.buffer = c.pool = None                     # (the copies own nothing)
            c.sig = make(b)
            out.append(c)
        return out

    def scan(loop_blocks, bord_blocks):
        res_l = pipeline.detect_blocks(dcool, loop_blocks, loops, template, want_windows=True)
        kspecs = [engine.KernelSpec(np.asarray(k, dtype=np.float64), None) for k in borders["kernels"]]
        res_b = cud.detect_many_on_device(dev, bord_blocks, kspecs, borders, want_windows=True, raw=True)
        return res_l, res_b

    stored = lambda parents: (lambda b: CsMatrix(parents[b.name].buffer.ptr, CS_F64, LAYOUT_BAND, b.sig.ld, 0, b.sig.band_w, 0))
    by_name = {b.name: b for b in blocks}
    want_l, want_b = scan(with_sig(blocks, stored(by_name)), with_sig(border_blocks, stored(by_name)))
    keep = []
    for near_w in (0, 8, 64):
        def patched(b):
            raw = np.empty(128, dtype=np.uint8)
            dev._check(dev.lib.cs_memcpy_d2h(dev.ctx, raw.ctypes.data, b.sig.d_ptr, 128, None))
            raw[80:84] = np.frombuffer(np.int32(near_w).tobytes(), dtype=np.uint8)        # LazyBand::near_w
            buf = dev.to_device(raw, np.uint8)
            keep.append(buf)
            return CsMatrix(buf.ptr, CS_F64, LAYOUT_BAND_LAZY, b.sig.ld, 0, b.sig.band_w, 0)
        got_l, got_b = scan(with_sig(blocks, patched), with_sig(border_blocks, patched))
        n_rec = 0
        for (gt, gw), (wt, ww) in zip(got_l, want_l):
            assert (gt is None) == (wt is None)
            if gt is not None:
                assert np.array_equal(gt, wt) and np.array_equal(gw, ww, equal_nan=True), near_w
                n_rec += len(gt)
        assert n_rec > 30
        n_rec = 0
        for got_t, want_t in zip(got_b, want_b):
            for (gt, gw), (wt, ww) in zip(got_t, want_t):
                assert (gt is None) == (wt is None)
                if gt is not None:
                    assert np.array_equal(gt, wt) and np.array_equal(gw, ww, equal_nan=True), near_w
                    n_rec += len(gt)
        assert n_rec > 100

    # ---- (2) the default staging against the stored-band staging
    def run():
        staged = parallel.stage_genome(dcool, [loops, borders])
        return staged, parallel.detect_patterns(dcool, [loops, borders], staged=staged)

    monkeypatch.setenv("CHROMOSIGHT_HIP_F64_TWIN", "1")
    staged_t, want = run()
    assert all(b.sig.layout == LAYOUT_BAND for b in staged_t.values())
    monkeypatch.delenv("CHROMOSIGHT_HIP_F64_TWIN")
    staged_l, got = run()
    assert all(b.sig.layout == LAYOUT_BAND_LAZY for b in staged_l.values())
    for g, w in zip(got, want):
        assert w.shape[0] > 30 and g.shape == w.shape
        assert np.array_equal(g[:, [0, 1, 2, 5, 6]], w[:, [0, 1, 2, 5, 6]])
        assert np.abs(g[:, 3] - w[:, 3]).max() < 1e-12 and np.allclose(g[:, 4], w[:, 4], rtol=1e-9, atol=1e-300)
    # the per-block entry reads the band itself: the lazy block is staged once more, with its band
    ci = sorted(staged_l)[1]
    table_l, win_l = pipeline.detect_block(dcool, staged_l[ci], loops, template, raw=True)
    table_t, win_t = pipeline.detect_block(dcool, staged_t[ci], loops, template, raw=True)
    assert len(table_t) > 5 and np.array_equal(table_l[:, :2], table_t[:, :2]) and np.abs(table_l[:, 2] - table_t[:, 2]).max() < 1e-12
    assert np.allclose(win_l, win_t, rtol=1e-12, atol=1e-300, equal_nan=True)
    assert staged_l[ci]._full is not None and staged_l[ci]._full.sig.layout == LAYOUT_BAND


def test_switchable_chains_equal_the_default_run(monkeypatch):
    """The switchable routes of the two chains == the default run of the same patterns."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(12_000, 200, 2000, seed=5, template=template)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 200 * 2000
    want_l, want_b = parallel.detect_patterns(pipeline.DeviceCool(cool), [loops, copy.deepcopy(ck.borders)])
    # the 2-D chain paced by the host's read of the candidate counts instead of enqueued behind the tile kernels with the counts
    # on the device (CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN=1), the 1-D foci through the labelling workgroup instead of the run kernels
    # (CHROMOSIGHT_HIP_NO_PATH_FOCI=1), the runs of a 1-D pattern scored by the general lane walk instead of rescore_run17
    # (CHROMOSIGHT_HIP_NO_RUN17=1), the labelling workgroups on global arrays instead of LDS (CHROMOSIGHT_HIP_NO_LDS_FOCI=1: the
    # sorted route for every list), the mask tables behind the staging instead of beside it, one candidate list sorted on the
    # device instead of the blocks' own segments.  (Every switch three times: the second and third repetition run on a context that has
    # seen the layout -- the deferred chain's launches sized from the previous call.)
    for switch in ("CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN", "CHROMOSIGHT_HIP_NO_PATH_FOCI", "CHROMOSIGHT_HIP_NO_RUN17",
                   "CHROMOSIGHT_HIP_NO_LDS_FOCI", "CHROMOSIGHT_HIP_NO_EARLY_TABLES", "CHROMOSIGHT_HIP_NO_SEGMENTED",
                   "CHROMOSIGHT_HIP_NO_COUNTS_BAND",       # (... the detrended bands of the tiler pass instead of the bands of raw counts,
                   "CHROMOSIGHT_HIP_HOST_PVALUES",         #  the records' p-values computed by cs_accept_records instead of copied,
                   "CHROMOSIGHT_HIP_TEMPLATE_FUSION"):     #  ONE pass of the run kernel for the three templates of borders -- what long lists take)
        monkeypatch.setenv(switch, "1")
        dcool_sw = pipeline.DeviceCool(cool)
        for rep in range(3):
            got_l, got_b = parallel.detect_patterns(dcool_sw if rep else pipeline.DeviceCool(cool), [loops, copy.deepcopy(ck.borders)])
            for got, want in ((got_l, want_l), (got_b, want_b)):
                assert got.shape == want.shape and want.shape[0] > 30, switch
                assert np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]]), switch
                assert np.abs(got[:, 3] - want[:, 3]).max() < 1e-9, switch
        monkeypatch.delenv(switch)
    # the wave-per-window kernels with the general functions instead of the compile-time-size ones (cs_launch_aux.h
    # rescore_pixel_sq / lazy_gather_window_sq / SigReader: the same sums in the same order): records bit for bit on ONE staging
    # (the distance law of a staging is summed with LDS atomics: two stagings differ in the last bits), on lazily evaluated
    # and on stored float64 bands
    for twin in (False, True):
        if twin:
            monkeypatch.setenv("CHROMOSIGHT_HIP_F64_TWIN", "1")
        dcool = pipeline.DeviceCool(cool)
        cfgs = [loops, copy.deepcopy(ck.borders)]
        staged = parallel.stage_genome(dcool, cfgs)
        fast_l, fast_b = parallel.detect_patterns(dcool, cfgs, staged=staged)
        monkeypatch.setenv("CHROMOSIGHT_HIP_NO_FAST_WINDOWS", "1")
        slow_l, slow_b = parallel.detect_patterns(dcool, cfgs, staged=staged)
        monkeypatch.delenv("CHROMOSIGHT_HIP_NO_FAST_WINDOWS")
        for fast, slow, want in ((fast_l, slow_l, want_l), (fast_b, slow_b, want_b)):
            assert fast.shape == want.shape and np.array_equal(fast, slow), twin
            assert np.array_equal(fast[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]]) and np.abs(fast[:, 3] - want[:, 3]).max() < 1e-12


# ------------------------------------------------------------------------------------------------
# device foci under load: large and tangled foci, ties, low thresholds
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,max_dist,pearson,seed", [(900, 120, 0.02, 1), (1500, 400, 0.05, 2), (700, 699, 0.0, 3),
                                                     (2500, 60, 0.08, 4), (600, 40, -0.05, 5)])
def test_device_foci_tangled_components(n, max_dist, pearson, seed):
    """Thresholds near zero make 10-50 % of the band candidate pixels: foci of thousands of pixels with
    holes and long arms, where the union-find (concurrent hooks, path halving) has real contention.
    Tables must equal the foci oracle run on the float64 oracle map: same foci, same order."""
    from chromosight_amd.utils import detection as cud
    rng = np.random.default_rng(seed)
    kern = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    keep = min(max_dist, n) + 17
    ii, jj = np.indices((n, n))
    # smooth-ish signal so that the coefficient map has large connected regions above a low threshold
    base = rng.gamma(4, 0.25, size=(n // 4 + 2, n // 4 + 2))
    a = np.kron(base, np.ones((4, 4)))[:n, :n] * rng.gamma(20, 0.05, size=(n, n))
    a = np.triu(a)
    a[(jj - ii > keep)] = 0
    miss = rng.random(n) < 0.03
    a[miss, :] = 0
    a[:, miss] = 0
    valid = np.flatnonzero(~miss)

    class Map:
        pass
    cmap = Map()
    cmap.matrix, cmap.detectable_bins, cmap.max_dist, cmap.inter = sp.csr_matrix(a), (valid, valid.copy()), max_dist, False
    cfg = dict(pearson=pearson, max_perc_undetected=50.0, max_perc_zero=100.0, max_dist=5 * max_dist)
    tab, wins = cud.pattern_detector(cmap, cfg, kern, full=True)
    want, _ = c_oracle.normxcorr2(a, kern, max_dist=max_dist, sym_upper=True, full=True, miss_row=miss, miss_col=miss,
                                  missing_tol=0.5)
    trimmed = np.where((jj - ii >= 0) & (jj - ii <= max_dist), want, 0.0)
    n_cand = int(((trimmed >= pearson) & (trimmed != 0)).sum())
    ref = foci_oracle.detect_table(a, trimmed, miss, miss, kern.shape, pearson=pearson, zero_tol=1.0, missing_tol=0.5)
    got = np.zeros((0, 3)) if tab is None else tab[["bin1", "bin2", "score"]].to_numpy(dtype=np.float64)
    print(f"n={n} max_dist={max_dist} pearson={pearson}: {n_cand} candidate pixels, {ref.shape[0]} validated foci")
    assert n_cand > 0.05 * n * min(max_dist, n)
    assert got.shape == ref.shape
    assert np.array_equal(got[:, :2], ref[:, :2])
    assert np.abs(got[:, 2] - ref[:, 2]).max() < 1e-9


def test_batched_1d_patterns_equal_per_block_calls():
    """cs_detect_foci_batch (all sub-matrices of a 1-D pattern in one launch chain, results written into the
    page-locked buffers by the last kernel) == one cs_detect_foci per sub-matrix: same records in the same
    order, same windows; also without windows (the last iteration of a genome run does not fetch them)."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(30_000, 300, 2000, seed=7, template=template, chrom_sizes=[9000, 10, 6000, 15000, 60])
    dcool = pipeline.DeviceCool(cool)
    for name in ("borders", "hairpins"):
        cfg = copy.deepcopy(getattr(ck, name))
        max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
        largest = max(np.shape(k)[0] for k in cfg["kernels"])
        blocks = [dcool.stage_intra(ci, max_dist, largest, resident=True) for ci in range(dcool.n_chrom)]
        for kernel in cfg["kernels"]:
            kernel = np.asarray(kernel, dtype=np.float64)
            one = pipeline.detect_blocks(dcool, blocks, cfg, kernel, raw=True, batch=False)
            many = pipeline.detect_blocks(dcool, blocks, cfg, kernel, raw=True, batch=True)
            bare = pipeline.detect_blocks(dcool, blocks, cfg, kernel, raw=True, batch=True, want_windows=False)
            assert one[1][0] is None and many[1][0] is None      # the 10-bin block is skipped; the 60-bin one is dense
            total = 0
            for a, b, c in zip(one, many, bare):
                if a[0] is None:
                    assert b[0] is None and c[0] is None
                    continue
                total += len(a[0])
                assert np.array_equal(a[0][:, :2], b[0][:, :2]) and np.array_equal(a[0][:, :2], c[0][:, :2])
                assert np.allclose(a[0][:, 2:], b[0][:, 2:], rtol=0, atol=1e-12, equal_nan=True)
                assert np.allclose(a[0][:, 2:], c[0][:, 2:], rtol=0, atol=1e-12, equal_nan=True)
                assert np.allclose(a[1], b[1], rtol=0, atol=0, equal_nan=True) and c[1] is None
            assert total > 50, (name, total)


def test_templates_of_a_1d_pattern_in_one_chain_equal_one_chain_each():
    """cs_detect_foci_batch_templates (the three borders templates as virtual blocks of one launch chain, with and without
    windows) == cs_detect_foci_batch template by template: same records in the same order, same windows."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(30_000, 300, 2000, seed=13, template=template, chrom_sizes=[9000, 700, 6000, 14300])
    dcool = pipeline.DeviceCool(cool)
    cfg = copy.deepcopy(ck.borders)
    kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
    blocks = [dcool.stage_intra(ci, 1, 17, resident=True) for ci in range(dcool.n_chrom)]
    for want in (True, False):
        joint = pipeline.detect_blocks_templates(dcool, blocks, cfg, kernels, want_windows=want)
        assert joint is not None
        joint = joint()
        assert len(joint) == len(kernels)
        total = 0
        for kernel, (table, kept, windows) in zip(kernels, joint):
            one = pipeline.detect_blocks(dcool, blocks, cfg, kernel, raw=True, want_windows=want, merged=True)
            assert isinstance(one, tuple)
            assert np.array_equal(kept, one[1]) and np.array_equal(table[:, :2], one[0][:, :2])
            assert np.allclose(table[:, 2:], one[0][:, 2:], rtol=0, atol=1e-12, equal_nan=True)
            if want:
                assert np.allclose(windows, one[2], rtol=0, atol=0, equal_nan=True)
            else:
                assert windows is None and one[2] is None
            total += len(table)
        assert total > 100
    # templates of different sizes, or a 2-D pattern: the entry does not apply
    assert pipeline.detect_blocks_templates(dcool, blocks, cfg, [kernels[0], kernels[1][:15, :15]]) is None
    assert pipeline.detect_blocks_templates(dcool, blocks, ck.loops, kernels) is None


def test_templates_scanned_side_by_side_keep_to_their_own_context(monkeypatch):
    """Truncated-SVD templates do not take the batched entries: the three borders templates are then scanned by three host
    threads, each on its own context and stream (they once shared the genome's context on that path: wrong tables).  Same
    records as one template after the other."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(30_000, 300, 2000, seed=17, template=template, chrom_sizes=[9000, 700, 6000, 14300])
    dcool = pipeline.DeviceCool(cool)
    cfg = copy.deepcopy(ck.borders)
    for _ in range(3):
        side_by_side = parallel.detect_genome(dcool, cfg, tsvd=0.999)
    monkeypatch.setenv("CHROMOSIGHT_HIP_NO_TEMPLATE_OVERLAP", "1")
    in_turn = parallel.detect_genome(dcool, cfg, tsvd=0.999)
    assert side_by_side.shape == in_turn.shape and side_by_side.shape[0] > 100
    assert np.array_equal(side_by_side[:, [0, 1, 2, 5, 6]], in_turn[:, [0, 1, 2, 5, 6]])
    assert np.abs(side_by_side[:, 3] - in_turn[:, 3]).max() < 1e-12


def test_three_patterns_side_by_side_repeat_exactly():
    """stage_genome + detect_patterns with THREE patterns (loops on the calling thread; borders -- three templates, one chain --
    and hairpins on pool threads, each on its own context; short chromosomes staged dense for loops are staged again, banded,
    by the threads of the 1-D patterns): every repetition gives the records of the first.  Found by this test's long form
    (tools/stress_genome_repeat.py): a single-template pattern on a pool thread used the genome's own context beside the
    calling thread, and two threads staging at once handed each other's block tables to the kernels."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(30_000, 300, 2000, seed=2, template=template)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops)
    loops["max_dist"] = 300 * 2000
    cfgs = [loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)]
    first = None
    for it in range(80):
        staged = parallel.stage_genome(dcool, cfgs)
        recs = parallel.detect_patterns(dcool, cfgs, staged=staged)
        if first is None:
            first = recs
            # ... and the first equals one pattern after the other
            for cfg, rec in zip(cfgs, recs):
                alone = parallel.detect_genome(dcool, cfg)
                assert rec.shape == alone.shape and np.array_equal(rec[:, [0, 1, 2, 5, 6]], alone[:, [0, 1, 2, 5, 6]]), cfg["name"]
            assert all(r.shape[0] > 100 for r in recs)
            continue
        for cfg, a, b in zip(cfgs, first, recs):
            assert a.shape == b.shape and np.array_equal(a[:, [0, 1, 2, 5, 6]], b[:, [0, 1, 2, 5, 6]]), (it, cfg["name"])
            assert np.abs(a[:, 3] - b[:, 3]).max() < 1e-12, (it, cfg["name"])


def test_run_scoring_of_1d_patterns_equals_wave_per_pixel(monkeypatch):
    """The float64 scoring of the enumerated diagonals one lane per pixel from an LDS tile
    (rescore_run_batch_kernel; the direct route for workgroups that straddle two sub-matrices) against the
    wave-per-pixel kernel (CHROMOSIGHT_HIP_NO_RUN_RESCORE=1) and against the direct lane route
    (CHROMOSIGHT_HIP_RUN_NO_LDS=1): same foci, scores within float64 summation-order noise."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(30_000, 300, 2000, seed=11, template=template, chrom_sizes=[9000, 700, 6000, 14300])
    dcool = pipeline.DeviceCool(cool)
    for name in ("borders", "hairpins"):
        cfg = copy.deepcopy(getattr(ck, name))
        max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
        largest = max(np.shape(k)[0] for k in cfg["kernels"])
        blocks = [dcool.stage_intra(ci, max_dist, largest, resident=True) for ci in range(dcool.n_chrom)]
        kernel = np.asarray(cfg["kernels"][0], dtype=np.float64)
        got = pipeline.detect_blocks(dcool, blocks, cfg, kernel, raw=True, batch=True, want_windows=False)
        results = {}
        for switch in ("CHROMOSIGHT_HIP_NO_RUN_RESCORE", "CHROMOSIGHT_HIP_RUN_NO_LDS"):
            monkeypatch.setenv(switch, "1")
            results[switch] = pipeline.detect_blocks(dcool, blocks, cfg, kernel, raw=True, batch=True, want_windows=False)
            monkeypatch.delenv(switch)
        total = 0
        for k, a in enumerate(got):
            for other in results.values():
                b = other[k]
                assert (a[0] is None) == (b[0] is None)
                if a[0] is None:
                    continue
                assert np.array_equal(a[0][:, :2], b[0][:, :2])
                assert np.allclose(a[0][:, 2:], b[0][:, 2:], rtol=0, atol=1e-12, equal_nan=True)
            total += 0 if a[0] is None else len(a[0])
        assert total > 50, (name, total)


def test_native_rccl_exchange_single_rank():
    """csrc/cs_comm.cpp on the GPU: librccl loads, a communicator comes up, and the two exchanges of the sharded path
    (count + padded all-gather of records, all-reduce of a float64 vector) round-trip.  One rank is all a one-GPU box
    can run -- RCCL refuses two ranks on one device -- so this pins loading, staging and the call sequence; the gloo
    tests (tests/test_parallel.py) pin what detect_genome does with several ranks."""
    import time
    from chromosight_amd.parallel import NativeComm
    comm = None
    for attempt in range(3):
        try:
            comm = NativeComm(0, 0, 1, NativeComm.unique_id())
            break
        except RuntimeError as exc:
            # (seen once in ~ 40 runs of this suite: ncclCommInitRank itself fails with "unhandled cuda error" on a box whose
            # other GPUs are busy -- RCCL's start-up, before any call of this library's; the product path falls back to the
            # process group's collectives then, parallel.exchange_self_check)
            if "ncclCommInitRank" not in str(exc):
                raise
            last = str(exc)
            time.sleep(1.0)
    if comm is None:
        # not a skip: a transport that does not come up is a finding -- reported as an expected failure with RCCL's
        # own error text, so that it shows in the summary without turning a box's start-up trouble into a red suite
        print(f"[rccl] ncclCommInitRank failed three times: {last}")
        pytest.xfail(f"RCCL does not start on this box (three attempts): {last}")
    rng = np.random.default_rng(4)
    rows = rng.random((1234, 7))
    got, counts = comm.allgather_rows(rows)
    assert counts.tolist() == [1234] and np.array_equal(got, rows)
    got, counts = comm.allgather_rows(np.zeros((0, 7)))
    assert counts.tolist() == [0] and got.shape == (0, 7)
    big = rng.random((50_000, 7))
    got, counts = comm.allgather_rows(big)
    assert np.array_equal(got, big)
    # the retry: a first capacity that is too small comes back as CS_ERR_OVERFLOW (on every rank: the capacities travel
    # with the counts), the second call has room for the sum of the counts
    import os
    os.environ["CHROMOSIGHT_HIP_GATHER_CAP"] = "100"
    try:
        got, counts = comm.allgather_rows(big[:5000])
    finally:
        del os.environ["CHROMOSIGHT_HIP_GATHER_CAP"]
    assert counts.tolist() == [5000] and np.array_equal(got, big[:5000])
    assert "room for 100" in comm.lib.cs_comm_last_error(comm.handle).decode()
    vec = rng.random(2 * 289 + 1)
    assert np.array_equal(comm.allreduce_sum(vec), vec)
    # the one-collective form (cs_comm_allgather_rows_once: what a replayed step of a sharded run uses): the first exchange of a
    # width learns the slot from the two-collective form, the later ones send a slot of that size; a list that outgrows it
    # comes back as CS_ERR_OVERFLOW on every rank and is sent again
    wide = rng.random((3000, 8))
    for n in (3000, 2900, 0, 3100, 9000, 10):
        got, counts = comm.allgather_rows_once(wide[:n] if n <= 3000 else np.tile(wide, (3, 1))[:n])
        want = wide[:n] if n <= 3000 else np.tile(wide, (3, 1))[:n]
        assert counts.tolist() == [n] and np.array_equal(got, want), n
    assert comm._slots[8] >= 256
    comm.close()


def test_records_carry_the_p_values_the_host_would_compute():
    """cs_focus.pval: the kernel that writes a record forms its p-value (reference detection.py:332-336 + stats.py:43-81:
    Fisher z, two-sided normal tail) and cs_accept_records copies it (flags bit 1) instead of computing it -- against the
    host arithmetic of the same call without the flag, record by record, for a 2-D and a 1-D pattern of a small genome, and
    against scipy's own expression."""
    import scipy.stats as ss
    from chromosight_amd import engine
    from chromosight_amd.utils import detection as cid
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(9_000, 150, 2000, seed=11, template=template)
    dcool = pipeline.DeviceCool(cool)
    for cfg in (copy.deepcopy(ck.loops), copy.deepcopy(ck.borders)):
        if cfg["name"] == "loops":
            cfg["max_dist"] = 150 * 2000
        staged = parallel.stage_genome(dcool, [cfg])
        blocks = [staged[ci] for ci in sorted(staged)]
        kernel = np.asarray(cfg["kernels"][0], dtype=np.float64)
        seen = 0
        for block in blocks:
            table, _ = pipeline.detect_block(dcool, block, cfg, kernel, raw=True)          # (device records, accepted with their own p-values)
            if table is None or len(table) == 0:
                continue
            seen += len(table)
            score, pval = table[:, 2], table[:, 3]
            nz = score != 0
            assert np.all(pval[~nz] == 1.0)
            assert np.all((pval > 0) & (pval <= 1))
        assert seen > 20, cfg["name"]
    # record by record: the flag on and off over the same device records
    cfg = copy.deepcopy(ck.loops)
    cfg["max_dist"] = 150 * 2000
    staged = parallel.stage_genome(dcool, [cfg])
    block = staged[sorted(staged)[0]].full()
    kspec = engine.KernelSpec(template)
    rec, _ = engine.run_detect_foci(dcool.dev, block.sig, block.shape, kspec, pearson=cfg["pearson"], lo_diag=0,
                                    hi_diag=min(block.max_dist, block.shape[0] - 1), inter=False, diag_only=False, full=True, sym_upper=True,
                                    max_dist=block.max_dist, mask_mode=1, miss_row=block.miss_row, miss_col=block.miss_col,
                                    missing_tol=cfg["max_perc_undetected"] / 100, want_windows=False)
    assert len(rec) > 5
    counts = np.array([len(rec)], dtype=np.int64)
    t_dev, ok_dev, _ = cid.accept_native(rec, counts, [block.shape], [block.max_dist], kspec, cfg, inter=False, full=True, compact=False, pvals=True)
    t_host, ok_host, _ = cid.accept_native(rec, counts, [block.shape], [block.max_dist], kspec, cfg, inter=False, full=True, compact=False, pvals=False)
    assert np.array_equal(ok_dev, ok_host) and np.array_equal(t_dev[:, :3], t_host[:, :3], equal_nan=True)
    assert np.allclose(t_dev[:, 3], t_host[:, 3], rtol=1e-13, atol=0.0)
    # ... and scipy's expression of stats.py:43-81 on the records' own fields
    n_obs = np.where(rec["n_obs"] == 0, 289.0, rec["n_obs"])
    z = np.arctanh(rec["score"]) * np.sqrt(n_obs - 3.0)
    want = np.where(rec["score"] != 0, 2.0 * ss.norm.cdf(-np.abs(z)), 1.0)
    assert np.allclose(rec["pval"], want, rtol=1e-12, atol=0.0)
