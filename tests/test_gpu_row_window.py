"""Row windows of the correlation map (cs_normxcorr2_params.row_begin / row_end, cs_matrix.row0) and
one sub-matrix split over several GPUs (cs_candidates + cs_label_foci, parallel.detect_split_block;
SURVEY.md 8(e)): every window of a map must equal the same rows of the whole map, and a split block
must give the tables of the one-GPU run."""
import copy
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import chromosight_amd.kernels as ck
from chromosight_amd import engine, parallel, pipeline
from chromosight_amd._lib import (LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE, CsMatrix, get_device, np_dtype_code)
from tools.synthetic_genome import make_cool

pytestmark = pytest.mark.gpu


def _windows(ms, cuts):
    edges = [0] + sorted(cuts) + [ms]
    return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a]


def _run(dev, sig, shape, kspec, out, window=None, **kw):
    params = engine._corr_params(shape, kspec, True, kw.get("sym_upper", False), kw.get("max_dist"), kw.get("mask_mode", MASK_NONE),
                                 kw.get("miss_row"), kw.get("miss_col"), None, 0.75, engine.compute_code(kw.get("precision")),
                                 window)
    dev._check(dev.lib.cs_normxcorr2(dev.ctx, None, C.byref(sig), C.byref(kspec.struct), C.byref(params), C.byref(out), None))


@pytest.mark.parametrize("n,cols,ksize,precision", [(700, 900, 17, "f32"), (513, 300, 7, "f32"), (400, 400, 17, "f64"),
                                                    (300, 500, (5, 9), "f32")])
def test_dense_row_windows_equal_full_map(n, cols, ksize, precision):
    """Dense map, no mask: windows (with and without a slab-only input buffer) == the whole map."""
    dev = get_device()
    rng = np.random.default_rng(n + cols)
    km, kn = (ksize, ksize) if isinstance(ksize, int) else ksize
    sig_h = rng.gamma(2.0, 1.0, size=(n, cols)).astype(np.float32)
    kern = rng.normal(size=(km, kn))
    if km == 17:
        kern = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    kspec = engine.KernelSpec(kern, None)
    dt = np.float32 if precision == "f32" else np.float64
    d_sig = dev.to_device(sig_h.astype(dt))
    sig = CsMatrix(d_sig.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0)
    d_full = dev.zeros(n * cols, dt)
    _run(dev, sig, (n, cols), kspec, CsMatrix(d_full.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0), precision=precision)
    full = d_full.download().reshape(n, cols)
    assert np.abs(full).max() > 0.05
    kh = (km - 1) // 2
    tol = 2e-6 if precision == "f32" else 1e-12
    for a, b in _windows(n, [37, 200, 201, n - 5]):
        # (1) whole input buffer, window of the output written into a window-sized buffer
        d_out = dev.zeros((b - a) * cols, dt)
        _run(dev, sig, (n, cols), kspec, CsMatrix(d_out.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0, a), (a, b),
             precision=precision)
        got = d_out.download().reshape(b - a, cols)
        assert np.abs(got - full[a:b]).max() <= tol, (a, b)
        # (2) slab input: only the rows the window needs
        ra, rb = max(0, a - kh), min(n, b + kh)
        d_slab = dev.to_device(np.ascontiguousarray(sig_h[ra:rb]).astype(dt))
        slab = CsMatrix(d_slab.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0, ra)
        d_out2 = dev.zeros((b - a) * cols, dt)
        _run(dev, slab, (n, cols), kspec, CsMatrix(d_out2.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0, a), (a, b),
             precision=precision)
        got2 = d_out2.download().reshape(b - a, cols)
        assert np.array_equal(got2, got), (a, b)


def _band_of(dense, lo, w, ld):
    n = dense.shape[0]
    band = np.zeros((n, ld), dtype=dense.dtype)
    for d in range(w):
        idx = np.arange(max(0, -(lo + d)), min(n, dense.shape[1] - (lo + d)))
        band[idx, d] = dense[idx, idx + lo + d]
    return band


@pytest.mark.parametrize("n,max_dist,ksize,precision", [(3000, 150, 17, "f32"), (1800, 60, 7, "f32"), (1200, 100, 17, "f64"),
                                                        (2600, 700, 17, "f32")])      # (the band leaves the matrix over 11 tile rows)
def test_band_row_windows_with_bin_masks(n, max_dist, ksize, precision):
    """Band in, scanned diagonals out, per-bin missing flags (the detect configuration): windows of the
    map from slab inputs == the whole map, including the rows next to the matrix edges and windows
    whose halo crosses masked bins."""
    dev = get_device()
    rng = np.random.default_rng(n)
    keep = max_dist + ksize
    in_w, out_w = keep + 1, max_dist + 1
    ld_in, ld_out = (in_w + 63) // 64 * 64, (out_w + 63) // 64 * 64
    dt = np.float32 if precision == "f32" else np.float64
    band = np.zeros((n, ld_in), dtype=dt)
    band[:, :in_w] = rng.gamma(2.0, 0.5, size=(n, in_w))
    for i in range(n):                      # nothing stored beyond the last column
        band[i, max(0, n - i):] = 0
    miss = (rng.random(n) < 0.04).astype(np.uint8)
    miss[[0, 1, n - 1, 500, 501, 502]] = 1
    zero = miss.astype(bool)
    for i in np.flatnonzero(zero):
        band[i, :] = 0
    for d in range(in_w):
        rows = np.arange(0, n - d)
        band[rows[zero[rows + d]], d] = 0
    kern = np.asarray(ck.loops["kernels"][0], dtype=np.float64) if ksize == 17 else rng.normal(size=(ksize, ksize))
    kspec = engine.KernelSpec(kern, None)
    d_miss = dev.to_device(miss)
    d_band = dev.to_device(band)
    sig = CsMatrix(d_band.ptr, np_dtype_code(dt), LAYOUT_BAND, ld_in, 0, in_w)
    kw = dict(sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss, miss_col=d_miss, precision=precision)
    d_full = dev.zeros(n * ld_out, dt)
    _run(dev, sig, (n, n), kspec, CsMatrix(d_full.ptr, np_dtype_code(dt), LAYOUT_BAND, ld_out, 0, out_w), **kw)
    full = d_full.download().reshape(n, ld_out)[:, :out_w]
    assert np.abs(full).max() > 0.05
    kh = (ksize - 1) // 2
    tol = 2e-6 if precision == "f32" else 1e-12
    for a, b in _windows(n, [3, 490, 505, 1100, n - 2]):
        ra, rb = max(0, a - kh), min(n, b + kh)
        d_slab = dev.to_device(np.ascontiguousarray(band[ra:rb]))
        slab = CsMatrix(d_slab.ptr, np_dtype_code(dt), LAYOUT_BAND, ld_in, 0, in_w, ra)
        d_out = dev.zeros((b - a) * ld_out, dt)
        _run(dev, slab, (n, n), kspec, CsMatrix(d_out.ptr, np_dtype_code(dt), LAYOUT_BAND, ld_out, 0, out_w, a), (a, b), **kw)
        got = d_out.download().reshape(b - a, ld_out)[:, :out_w]
        assert np.abs(got - full[a:b]).max() <= tol, (a, b)


@pytest.mark.parametrize("seed", range(10))
def test_random_band_row_windows(seed, monkeypatch):
    """Random band geometries under the mirrored 17 x 17 template (the masked matrix-core tile kernel): bands that leave the matrix on
    some, most or all rows (its strips end at the last column: MfmaDenseArgs::by_cut), random row windows whose mask tables cover
    the window only (MaskPrepArgs::r_lo ..) == the same rows of the whole map, and == the window with the tables of every bin
    (CHROMOSIGHT_HIP_FULL_MASK_TABLES=1: bit for bit)."""
    dev = get_device()
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(400, 3200))
    max_dist = int(rng.choice([n // 30 + 20, n // 4, n - 5, n + 40, 1000]))
    ksize = 17
    keep = max_dist + ksize
    in_w, out_w = keep + 1, max_dist + 1
    ld_in, ld_out = (in_w + 63) // 64 * 64, (out_w + 63) // 64 * 64
    band = np.zeros((n, ld_in), dtype=np.float32)
    band[:, :in_w] = rng.gamma(2.0, 0.5, size=(n, in_w))
    for i in range(n):
        band[i, max(0, n - i):] = 0
    miss = (rng.random(n) < 0.03).astype(np.uint8)
    miss[[0, n - 1, n // 2]] = 1
    zero = miss.astype(bool)
    band[zero, :] = 0
    for d in range(in_w):
        rows = np.arange(0, max(0, n - d))
        band[rows[zero[rows + d]], d] = 0
    kspec = engine.KernelSpec(np.asarray(ck.loops["kernels"][0], dtype=np.float64), None)
    d_miss, d_band = dev.to_device(miss), dev.to_device(band)
    f32 = np_dtype_code(np.float32)
    sig = CsMatrix(d_band.ptr, f32, LAYOUT_BAND, ld_in, 0, in_w)
    kw = dict(sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss, miss_col=d_miss, precision="f32")
    d_full = dev.zeros(n * ld_out, np.float32)
    _run(dev, sig, (n, n), kspec, CsMatrix(d_full.ptr, f32, LAYOUT_BAND, ld_out, 0, out_w), **kw)
    assert dev.lib.cs_last_kernel(dev.ctx) == 5            # the masked matrix-core tile kernel
    full = d_full.download().reshape(n, ld_out)[:, :out_w]
    assert np.abs(full).max() > 0.05
    cuts = sorted(set(int(c) for c in rng.integers(1, n - 1, size=3)) | {n - int(rng.integers(2, 70))})
    for a, b in _windows(n, cuts):
        ra, rb = max(0, a - 8), min(n, b + 8)
        d_slab = dev.to_device(np.ascontiguousarray(band[ra:rb]))
        slab = CsMatrix(d_slab.ptr, f32, LAYOUT_BAND, ld_in, 0, in_w, ra)
        got = []
        for full_tables in (False, True):
            if full_tables:
                monkeypatch.setenv("CHROMOSIGHT_HIP_FULL_MASK_TABLES", "1")
            d_out = dev.zeros((b - a) * ld_out, np.float32)
            _run(dev, slab, (n, n), kspec, CsMatrix(d_out.ptr, f32, LAYOUT_BAND, ld_out, 0, out_w, a), (a, b), **kw)
            got.append(d_out.download().reshape(b - a, ld_out)[:, :out_w])
            monkeypatch.delenv("CHROMOSIGHT_HIP_FULL_MASK_TABLES", raising=False)
        assert np.array_equal(got[0], got[1]), (n, max_dist, a, b)
        assert np.abs(got[0] - full[a:b]).max() <= 2e-6, (n, max_dist, a, b)


def test_row_window_argument_errors():
    dev = get_device()
    kspec = engine.KernelSpec(np.asarray(ck.loops["kernels"][0], dtype=np.float64), None)
    d = dev.zeros(100 * 100, np.float32)
    m = CsMatrix(d.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, 100, 0, 0)
    for window in [(-1, 10), (50, 101)]:
        params = engine._corr_params((100, 100), kspec, True, False, None, MASK_NONE, None, None, None, 0.75,
                                     engine.compute_code("f32"), window)
        assert dev.lib.cs_normxcorr2(dev.ctx, None, C.byref(m), C.byref(kspec.struct), C.byref(params), C.byref(m), None) == -1


@pytest.mark.parametrize("precision", ["f32", "f64"])
def test_valid_mode_row_windows(precision):
    """full = False (zero margins of (k-1)/2, same geometry): windows == the whole map."""
    dev = get_device()
    rng = np.random.default_rng(12)
    n, cols = 300, 420
    dt = np.float32 if precision == "f32" else np.float64
    sig_h = rng.gamma(2.0, 1.0, size=(n, cols)).astype(dt)
    kspec = engine.KernelSpec(np.asarray(ck.loops["kernels"][0], dtype=np.float64), None)
    d_sig = dev.to_device(sig_h)
    sig = CsMatrix(d_sig.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0)

    def run(out, window=None):
        params = engine._corr_params((n, cols), kspec, False, False, None, MASK_NONE, None, None, None, 0.75,
                                     engine.compute_code(precision), window)
        dev._check(dev.lib.cs_normxcorr2(dev.ctx, None, C.byref(sig), C.byref(kspec.struct), C.byref(params), C.byref(out), None))

    d_full = dev.zeros(n * cols, dt)
    run(CsMatrix(d_full.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0))
    full = d_full.download().reshape(n, cols)
    assert np.all(full[:8] == 0) and np.all(full[-8:] == 0) and np.abs(full[8:-8, 8:-8]).max() > 0.05
    for a, b in _windows(n, [5, 8, 150, n - 8]):
        d_out = dev.zeros((b - a) * cols, dt)
        run(CsMatrix(d_out.ptr, np_dtype_code(dt), LAYOUT_DENSE, cols, 0, 0, a), (a, b))
        got = d_out.download().reshape(b - a, cols)
        assert np.abs(got - full[a:b]).max() <= (2e-6 if precision == "f32" else 1e-12), (a, b)


# ------------------------------------------------------------------------------------------------
# candidates per window + joint labelling == cs_detect_foci on the whole block
# ------------------------------------------------------------------------------------------------
def _genome(n=9000, seed=11):
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(n, 300, 2000, seed=seed, template=template, chrom_sizes=[n])
    return cool


class _ThreadGather:
    """all_gather between the threads that stand for the ranks."""

    def __init__(self, n):
        import threading
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n

    def rank(self, r):
        def all_gather(arr):
            self.slots[r] = arr
            self.barrier.wait()
            out = np.concatenate(self.slots, axis=0)
            self.barrier.wait()
            return out
        return all_gather


@pytest.mark.parametrize("pattern", ["loops", "borders"])
@pytest.mark.parametrize("parts", [2, 3, 7])
def test_split_block_in_process(pattern, parts):
    """The split-block flow with one thread (own context and stream) per part and a barrier for the
    collectives: every part stages its row window (+ halo) with the summed distance law, finds its
    candidates; the merged candidates are labelled by every part; every part scores its foci.
    Records == detect_block on the whole block, in order, on every part."""
    import concurrent.futures
    from chromosight_amd._lib import Device
    from chromosight_amd.utils import detection as cid
    dcool = pipeline.DeviceCool(_genome())
    cfg = copy.deepcopy(getattr(ck, pattern))
    if pattern == "loops":
        cfg["max_dist"] = 300 * 2000
    kernel = np.asarray(cfg["kernels"][0], dtype=np.float64)
    max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
    largest = max(np.shape(k)[0] for k in cfg["kernels"])
    n = dcool.chrom_size(0)
    whole = dcool.stage_intra(0, max_dist, largest, resident=True)
    assert whole.sig.layout == LAYOUT_BAND
    want, want_win = pipeline.detect_block(dcool, whole, cfg, kernel, raw=True)
    assert want is not None and len(want) > (20 if pattern == "loops" else 100)
    rows = parallel.split_rows(n, parts)
    partial = []
    for w in rows:                                   # pass 1: every part's (sum, count) per diagonal
        dcool.stage_intra(0, max_dist, largest, rows=w, reduce=lambda x: (partial.append(x.copy()), x)[1])
    total = np.sum(partial, axis=0)
    blocks = [dcool.stage_intra(0, max_dist, largest, rows=w, reduce=lambda x: total, resident=True) for w in rows]
    dcool.dev.sync()
    kspec = engine.KernelSpec(kernel, None)
    gather = _ThreadGather(parts)

    def one(r):
        dev = Device(dcool.dev.index)
        blk = blocks[r]
        return cid.detect_split_on_device(dev, blk.sig, blk.shape, blk.row_window, kspec, cfg, blk.miss_row, blk.miss_col,
                                          max_dist=blk.max_dist, all_gather=gather.rank(r), raw=True)

    with concurrent.futures.ThreadPoolExecutor(max_workers=parts) as pool:
        results = list(pool.map(one, range(parts)))
    for got, got_win in results:
        assert got.shape == want.shape
        assert np.array_equal(got[:, :2], want[:, :2])
        assert np.abs(got[:, 2] - want[:, 2]).max() < 1e-9
        assert np.allclose(got[:, 3], want[:, 3], rtol=1e-6, atol=1e-300)
        assert np.allclose(got_win, want_win, rtol=0, atol=1e-9, equal_nan=True)


WORKER = r"""
import os, sys, copy, numpy as np
sys.path.insert(0, os.environ["CS_ROOT"])
import torch.distributed as dist
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(9000, 300, 2000, seed=11, template=template, chrom_sizes=[9000])
dcool = pipeline.DeviceCool(cool)
out = []
for name in ("loops", "borders"):
    cfg = copy.deepcopy(getattr(ck, name))
    if name == "loops":
        cfg["max_dist"] = 300 * 2000
    blk = parallel.stage_split(dcool, 0, cfg)
    for kernel in cfg["kernels"]:
        tab, win = parallel.detect_split_block(dcool, 0, cfg, kernel, raw=True, staged=blk)
        out.append(tab)
        out.append(win.reshape(len(win), -1))
if dist.get_rank() == 1:
    np.savez(os.environ["CS_OUT"], *out)
dist.destroy_process_group()
"""


def test_split_block_three_ranks(tmp_path):
    """parallel.detect_split_block on 3 ranks (gloo rendezvous, all on this GPU): all-reduced distance
    law, all-gathered candidates, gathered records == one GPU, for loops and the borders templates."""
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(9000, 300, 2000, seed=11, template=template, chrom_sizes=[9000])
    dcool = pipeline.DeviceCool(cool)
    want = []
    for name in ("loops", "borders"):
        cfg = copy.deepcopy(getattr(ck, name))
        if name == "loops":
            cfg["max_dist"] = 300 * 2000
        max_dist = max(cfg["max_dist"] // dcool.binsize, 1)
        largest = max(np.shape(k)[0] for k in cfg["kernels"])
        blk = dcool.stage_intra(0, max_dist, largest, resident=True)
        for kernel in cfg["kernels"]:
            tab, win = pipeline.detect_block(dcool, blk, cfg, kernel, raw=True)
            want.append(tab)
            want.append(win.reshape(len(win), -1))
    out = tmp_path / "split.npz"
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CS_ROOT=root, CS_OUT=str(out), CHROMOSIGHT_HIP_DEVICE="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29541", WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(3)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.load(out)
    got = [got[f"arr_{k}"] for k in range(len(want))]
    for k in range(0, len(want), 2):
        assert got[k].shape == want[k].shape and len(want[k]) > 20
        assert np.array_equal(got[k][:, :2], want[k][:, :2])
        assert np.abs(got[k][:, 2] - want[k][:, 2]).max() < 1e-9
        assert np.allclose(got[k + 1], want[k + 1], rtol=0, atol=1e-9, equal_nan=True)


SPLIT_WORKER = r"""
import os, sys, numpy as np
sys.path.insert(0, os.environ["CS_ROOT"])
import torch.distributed as dist
import bench
from chromosight_amd._lib import get_device
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
dev = get_device(0)
w = bench.SplitC4P(dev, rank, world, "f32", n=int(os.environ["CS_N"]))
for _ in range(2):
    law, merged = w.scan.step()
np.savez(os.environ["CS_OUT"] + f".{rank}.npz", rows=np.asarray(w.rows), out=w.out_buf.download()[:, :w.out_w], law=law, merged=merged)
# ... and the timed form of the same leg (what bench.py --gpus N prints), two steps
def reduce_max(x):
    import torch
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
leg = bench.time_c4p_split(dev, rank, 0, world, dist, "f32", dev.sync, reduce_max, steps=2, warmup=1, n=int(os.environ["CS_N"]))
assert leg["n_gpus"] == world and len(leg["per_rank"]) == world and leg["kernel_id"] == 5 and leg["candidates"] == len(merged), leg
dist.destroy_process_group()
"""


def test_split_map_two_ranks_equal_the_one_rank_map(tmp_path, monkeypatch):
    """bench.py's `north_star_c4p_split` leg (parallel.SplitBlockScan, N > 1) on 2 ranks (gloo rendezvous, both on this GPU)
    with a 6000-bin block of the C4' generator, in its map form (CS_BENCH_SPLIT_MAP=1: coefficient map of the rows + compaction):
    the two ranks' row windows together ARE the one-rank coefficient map (to 2e-6: same kernel, other tile boundaries), the merged
    candidates are the pixels >= 0.3 of the ranks' rows in row-major order on both ranks, the all-reduced law sums are those of
    the whole band.  The default form (cs_candidates: candidate epilogue, no map): test_split_candidates_two_ranks_... below."""
    import bench
    from tools.synthetic_genome import band_workload
    monkeypatch.setenv("CS_BENCH_SPLIT_MAP", "1")
    n = 6000
    dev = get_device()
    one = bench.SplitC4P(dev, 0, 1, "f32", n=n)
    one.scan.step()
    dev.sync()
    full = one.out_buf.download()[:, :one.out_w]
    assert dev.lib.cs_last_kernel(dev.ctx) == 5
    band, band_w, _miss, _n, max_dist = band_workload("c4p", 0, n=n)
    out = tmp_path / "split"
    script = tmp_path / "worker.py"
    script.write_text(SPLIT_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CS_ROOT=root, CS_OUT=str(out), CS_N=str(n), CHROMOSIGHT_HIP_DEVICE="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29547", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    parts = [np.load(f"{out}.{r}.npz") for r in range(2)]
    assert [tuple(p["rows"]) for p in parts] == parallel.split_rows(n, 2)
    got = np.concatenate([p["out"] for p in parts], axis=0)
    # (a window that does not start on a tile boundary of the whole map is cut into other tiles: other power-of-two scales per
    # tile, last-bit differences -- the tolerance of the row-window tests above)
    assert got.shape == full.shape and np.abs(got - full).max() <= 2e-6
    ii, dd = np.nonzero(got >= 0.3)
    want = np.column_stack([ii, ii + dd, got[ii, dd].astype(np.float64)])
    assert abs(len(want) - int((full >= 0.3).sum())) <= 2
    assert len(want) > 20
    own = band[:, :band_w]
    law = np.concatenate([own.sum(axis=0, dtype=np.float64), (own > 0).sum(axis=0).astype(np.float64)])
    for p in parts:
        assert np.array_equal(p["merged"], want)
        assert np.allclose(p["law"], law, rtol=1e-12, atol=0)


SPLIT_WORKER_FUSED = r"""
import os, sys, numpy as np
sys.path.insert(0, os.environ["CS_ROOT"])
import torch.distributed as dist
import bench
from chromosight_amd._lib import get_device
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
dev = get_device(0)
w = bench.SplitC4P(dev, rank, world, "f32", n=int(os.environ["CS_N"]))
assert w.fused
for _ in range(2):
    law, merged = w.scan.step()
np.savez(os.environ["CS_OUT"] + f".{rank}.npz", rows=np.asarray(w.rows), merged=merged, mine=w.cand)
def reduce_max(x):
    import torch
    t = torch.tensor([x], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
leg = bench.time_c4p_split(dev, rank, 0, world, dist, "f32", dev.sync, reduce_max, steps=2, warmup=1, n=int(os.environ["CS_N"]))
assert leg["n_gpus"] == world and len(leg["per_rank"]) == world and leg["candidates"] == len(merged), leg
dist.destroy_process_group()
"""


def test_split_candidates_two_ranks_equal_the_one_rank_candidates(tmp_path, monkeypatch):
    """The default form of the `north_star_c4p_split` leg: every rank runs cs_candidates on its row window -- the
    masked tile kernel's candidate epilogue, no coefficient map, float64 re-scoring -- and the candidates are all-gathered.  On 2
    ranks (gloo rendezvous, both on this GPU): each rank's list lies in its own rows, the merged list is the same on both ranks and
    IS the one-rank list (coordinates exactly, float64 scores to 1e-9: the re-scoring does not depend on how the rows were tiled),
    and it is the set of pixels >= 0.3 of the map form up to the pixels within float32 rounding of the threshold."""
    import bench
    monkeypatch.delenv("CS_BENCH_SPLIT_MAP", raising=False)
    n = 6000
    dev = get_device()
    one = bench.SplitC4P(dev, 0, 1, "f32", n=n)
    assert one.fused
    one.scan.step()                                   # (one rank: the step is the chain alone)
    want = one.cand
    assert len(want) > 20 and np.all(want[:, 2] >= 0.3)
    out = tmp_path / "split"
    script = tmp_path / "worker.py"
    script.write_text(SPLIT_WORKER_FUSED)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CS_ROOT=root, CS_OUT=str(out), CS_N=str(n), CHROMOSIGHT_HIP_DEVICE="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29549", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    parts = [np.load(f"{out}.{r}.npz") for r in range(2)]
    for r, p in enumerate(parts):
        a, b = p["rows"]
        assert np.all((p["mine"][:, 0] >= a) & (p["mine"][:, 0] < b))
        assert np.array_equal(p["merged"], parts[0]["merged"])
    merged = parts[0]["merged"]
    assert merged.shape == want.shape and np.array_equal(merged[:, :2], want[:, :2])
    assert np.abs(merged[:, 2] - want[:, 2]).max() < 1e-9
    # against the map form on one rank: same pixels but for those within float32 rounding of the threshold
    monkeypatch.setenv("CS_BENCH_SPLIT_MAP", "1")
    ref = bench.SplitC4P(dev, 0, 1, "f32", n=n)
    ref.scan.correlate()
    cmap = ref.scan.candidates()
    key = lambda t: set(map(tuple, t[:, :2].astype(np.int64)))      # noqa: E731
    assert len(key(cmap) ^ key(want)) <= 2
