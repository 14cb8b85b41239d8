"""World-size-2 tests of the multi-GPU layer on CPU (gloo): block assignment, gather of
variable-length records, and detect_blocks with a stand-in detector (the HIP detector itself is
covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chromosight_amd import parallel


def test_assign_blocks_lpt():
    costs = [100, 90, 10, 10, 10, 80, 5]
    owned = parallel.assign_blocks(costs, 3)
    assert sorted(sum(owned, [])) == list(range(len(costs)))
    loads = [sum(costs[i] for i in o) for o in owned]
    assert max(loads) <= 105
    assert parallel.assign_blocks(costs, 3) == owned          # deterministic
    assert parallel.assign_blocks(costs, 1) == [list(range(len(costs)))]
    assert parallel.assign_blocks([], 2) == [[], []]


def test_block_cost():
    assert parallel.block_cost((100, 100), 9, False) == 1000
    assert parallel.block_cost((100, 100), None, False) == 10000
    assert parallel.block_cost((30, 70), None, True) == 2100
    assert parallel.block_cost((50, 50), 1000, False) == 2500


class _Block:
    def __init__(self, n, seed):
        self.shape = (n, n)
        self.max_dist = 10
        self.inter = False
        self.seed = seed


def _fake_detector(cmap, cfg, kernel, coords=None, full=True, tsvd=None):
    """Deterministic stand-in: the table depends only on the block."""
    import pandas as pd
    rng = np.random.default_rng(cmap.seed)
    n = int(rng.integers(0, 6))
    if n == 0:
        return None, None
    return pd.DataFrame({
        "bin1": rng.integers(0, cmap.shape[0], n), "bin2": rng.integers(0, cmap.shape[0], n),
        "score": rng.random(n), "pvalue": rng.random(n)}), np.zeros((n, 3, 3))


def _expected(blocks):
    rows = []
    for i, b in enumerate(blocks):
        t, _ = _fake_detector(b, None, None)
        if t is None:
            continue
        rec = np.column_stack([np.full(len(t), i, float)] + [t[c].to_numpy(float) for c in ("bin1", "bin2", "score", "pvalue")])
        rows.append(rec)
    return np.concatenate(rows)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        blocks = [_Block(40 + 7 * i, seed=i) for i in range(9)]
        got = parallel.detect_blocks(blocks, {}, None, detector=_fake_detector)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), got)
        # ragged gather, including an empty contribution
        local = np.zeros((0, 5)) if rank == 0 else np.arange(15, dtype=float).reshape(3, 5)
        g = parallel.gather_records(local)
        np.save(os.path.join(out_dir, f"gather{rank}.npy"), g)
        # CPU ranks exchange over torch / gloo; the start-up self-check of bench.py has nothing native to check there
        assert parallel.transport() == "torch_gloo" and parallel.exchange_self_check() == "torch_gloo"
        # a local run under an initialised process group (genome_step(local=True): pipeline.detect): one rank alone may make it --
        # rank 1 only here -- and it sees a world of one: no sharding, and the record exchange is the identity (no collective)
        assert parallel._world()[2] == 2
        if rank == 1:
            parallel._LOCAL_DEPTH[0] += 1
            try:
                assert parallel._world() == (None, 0, 1)
                rec = np.arange(14, dtype=float).reshape(2, 7)
                assert parallel._exchange_records(rec, 1, 1) is rec
                assert parallel.assign_blocks([5, 3, 2], parallel._world()[2]) == [[0, 1, 2]]
            finally:
                parallel._LOCAL_DEPTH[0] -= 1
        assert parallel._world()[2] == 2
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_detect_blocks_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    blocks = [_Block(40 + 7 * i, seed=i) for i in range(9)]
    want = _expected(blocks)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(got, want)
        g = np.load(tmp_path / f"gather{r}.npy")
        assert np.array_equal(g, np.arange(15, dtype=float).reshape(3, 5))


def test_single_process_path():
    blocks = [_Block(40 + 7 * i, seed=i) for i in range(5)]
    got = parallel.detect_blocks(blocks, {}, None, detector=_fake_detector)
    assert np.array_equal(got, _expected(blocks))


# ------------------------------------------------------------------------------------------------
# detect_genome: templates x iterations, gathered tables, all-reduced pileup
# ------------------------------------------------------------------------------------------------
class _Genome:
    binsize = 1000
    n_chrom = 7

    def chrom_size(self, ci):
        return 50 + 13 * ci


def _fake_stage(genome, ci, max_dist, largest):
    return {"ci": ci, "n": genome.chrom_size(ci), "max_dist": max_dist, "largest": largest}


def _fake_detect(genome, block, cfg, kernel, tsvd):
    """The table depends on the block AND on the template (so that a wrong pileup shows)."""
    import pandas as pd
    rng = np.random.default_rng(block["ci"] * 1000 + int(round(float(np.nansum(kernel)) * 1e6)) % 997)
    n = int(rng.integers(0, 5))
    if n == 0:
        return None, None
    wins = rng.random((n, 3, 3))
    wins[rng.random((n, 3, 3)) < 0.2] = np.nan
    return pd.DataFrame({"bin1": rng.integers(0, block["n"], n), "bin2": rng.integers(0, block["n"], n),
                         "score": rng.random(n), "pvalue": rng.random(n)}), wins


_CFG = {"max_dist": 20_000, "max_iterations": 3, "kernels": [np.full((3, 3), 0.5), np.arange(9.0).reshape(3, 3) / 7]}


# single iteration, three templates: every template's records travel in ONE exchange at the end of the call
_CFG_ONE = {"max_dist": 20_000, "max_iterations": 1,
            "kernels": [np.full((3, 3), 0.5), np.arange(9.0).reshape(3, 3) / 7, np.eye(3) + 0.25]}


def _genome_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        got = parallel.detect_genome(_Genome(), _CFG, stage=_fake_stage, detect=_fake_detect)
        np.save(os.path.join(out_dir, f"genome{rank}.npy"), got)
        got = parallel.detect_genome(_Genome(), _CFG_ONE, stage=_fake_stage, detect=_fake_detect)
        np.save(os.path.join(out_dir, f"genome_one{rank}.npy"), got)
    finally:
        dist.destroy_process_group()


def test_detect_genome_world2_equals_single(tmp_path):
    single = parallel.detect_genome(_Genome(), _CFG, stage=_fake_stage, detect=_fake_detect)
    assert single.shape[1] == len(parallel.GENOME_FIELDS) and single.shape[0] > 10
    assert set(np.unique(single[:, 6])) >= {0.0, 1.0}         # the refined templates found patterns too
    world = 2
    mp.spawn(_genome_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        got = np.load(tmp_path / f"genome{r}.npy")
        # the pileup of the sharded run is a sum over ranks: same template up to rounding, and the
        # stand-in detector keys on it only through a rounded checksum
        assert got.shape == single.shape
        assert np.array_equal(got[:, [0, 1, 2, 5, 6]], single[:, [0, 1, 2, 5, 6]])
        assert np.allclose(got[:, 3:5], single[:, 3:5], rtol=0, atol=1e-12)
    single_one = parallel.detect_genome(_Genome(), _CFG_ONE, stage=_fake_stage, detect=_fake_detect)
    assert len(np.unique(single_one[:, 5])) == 3
    for r in range(world):
        assert np.array_equal(np.load(tmp_path / f"genome_one{r}.npy"), single_one)      # same rows, same order


# ------------------------------------------------------------------------------------------------
# quantify: sub-matrices dealt to the ranks, one exchange of scores and windows (QuantifyShard)
# ------------------------------------------------------------------------------------------------
class _QGenome:
    def chrom_size(self, ci):
        return 60 + 11 * ci


def _quantify_todo():
    """The work list pipeline.quantify builds: (ca, cb, indices of the positions, block-local coordinates)."""
    rng = np.random.default_rng(7)
    pairs = [(0, 0), (0, 2), (1, 1), (1, 3), (2, 2), (3, 3), (2, 3)]
    todo, start = [], 0
    for ca, cb in pairs:
        n = int(rng.integers(1, 9))
        todo.append((ca, cb, np.arange(start, start + n), rng.integers(0, 50, size=(n, 2))))
        start += n
    return todo, start


def _fake_scores(todo_part, n_pos, n_k=3, kk=(3, 3)):
    """What a rank would compute for its sub-matrices: values keyed on (position, template) only."""
    score = [np.full(n_pos, np.nan) for _ in range(n_k)]
    pval = [np.full(n_pos, np.nan) for _ in range(n_k)]
    win = [np.full((n_pos,) + kk, np.nan) for _ in range(n_k)]
    for _, _, sel, _ in todo_part:
        for k in range(n_k):
            score[k][sel] = np.sin(sel * 1.5 + k)
            pval[k][sel] = np.cos(sel * 0.5 + k) ** 2
            win[k][sel] = (sel[:, None, None] + np.arange(9.0).reshape(kk)) * (k + 1)
            win[k][sel[::3], 0, 0] = np.nan                    # windows carry NaN (missing bins): they must survive the exchange
    return score, pval, win


def _quantify_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        todo, n_pos = _quantify_todo()
        shard = parallel.QuantifyShard()
        mine = shard.select(todo, _QGenome(), 30)
        score, pval, win = _fake_scores(mine, n_pos)
        score, pval, win = shard.merge(score, pval, win, [sel for _, _, sel, _ in mine])
        np.savez(os.path.join(out_dir, f"q{rank}.npz"), score=np.array(score), pval=np.array(pval), win=np.array(win),
                 mine=np.array([list(t[:2]) for t in mine]).reshape(-1, 2))
    finally:
        dist.destroy_process_group()


def test_quantify_shard_world2(tmp_path):
    """Every rank ends with the scores, p-values and windows of ALL positions, whichever rank scored them; the shares are
    disjoint, cover the work list and are balanced by staged pixels + positions."""
    todo, n_pos = _quantify_todo()
    want = _fake_scores(todo, n_pos)
    world = 2
    mp.spawn(_quantify_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    shares = []
    for r in range(world):
        got = np.load(tmp_path / f"q{r}.npz")
        for name, ref in zip(("score", "pval", "win"), want):
            assert np.array_equal(got[name], np.array(ref), equal_nan=True), (r, name)
        shares.append({tuple(x) for x in got["mine"].tolist()})
    assert shares[0].isdisjoint(shares[1]) and shares[0] | shares[1] == {(ca, cb) for ca, cb, _, _ in todo}
    assert all(len(s) >= 2 for s in shares)
    # a single process keeps everything and exchanges nothing
    alone = parallel.QuantifyShard()
    assert alone.select(todo, _QGenome(), 30) == todo


# ------------------------------------------------------------------------------------------------
# one sub-matrix row-split over the ranks, map level (bench.py's `north_star_c4p_split` leg at N > 1)
# ------------------------------------------------------------------------------------------------
def _split_case(n=101, w=7):
    """A deterministic 'band' of n rows x w diagonals and its 'coefficient map' (a stand-in: the correlation itself is the
    device's business, tests/test_gpu_row_window.py)."""
    rng = np.random.default_rng(7)
    band = rng.poisson(3.0, size=(n, w)).astype(np.float64)
    corr = rng.random((n, w))
    return band, corr


def _split_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        band, corr = _split_case()
        n, w = band.shape
        a, b = parallel.split_rows(n, world)[rank]
        own = band[a:b]
        law_part = np.concatenate([own.sum(axis=0), (own > 0).sum(axis=0).astype(np.float64)])
        calls = []

        def correlate():
            calls.append("correlate")

        def candidates():
            calls.append("candidates")
            r, d = np.nonzero(corr[a:b] >= 0.8)
            return np.column_stack([r + a, r + a + d, corr[a:b][r, d]])

        scan = parallel.SplitBlockScan(n, law_part, correlate, candidates)
        assert scan.rows == (a, b)
        for _ in range(2):
            law, merged = scan.step()
        assert calls == ["correlate", "candidates"] * 2 and scan.step_ms >= scan.exchange_ms >= 0.0
        np.save(os.path.join(out_dir, f"law{rank}.npy"), law)
        np.save(os.path.join(out_dir, f"cand{rank}.npy"), merged)
    finally:
        dist.destroy_process_group()


def test_split_block_scan_world2(tmp_path):
    """parallel.SplitBlockScan on 2 ranks (gloo): the all-reduced (sum, count) of the law and the all-gathered candidates are
    those of the whole block, on every rank, in row order; one rank: the step is the correlation call alone."""
    world = 2
    mp.spawn(_split_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    band, corr = _split_case()
    want_law = np.concatenate([band.sum(axis=0), (band > 0).sum(axis=0).astype(np.float64)])
    r, d = np.nonzero(corr >= 0.8)
    want_cand = np.column_stack([r, r + d, corr[r, d]])
    for rank in range(world):
        assert np.allclose(np.load(tmp_path / f"law{rank}.npy"), want_law, rtol=1e-14, atol=0)
        assert np.array_equal(np.load(tmp_path / f"cand{rank}.npy"), want_cand)
    done = []
    scan = parallel.SplitBlockScan(band.shape[0], want_law, lambda: done.append(1), lambda: 1 / 0)
    law, merged = scan.step()
    assert done == [1] and merged is None and scan.rows == (0, band.shape[0]) and np.array_equal(law, want_law)
