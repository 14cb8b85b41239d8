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
