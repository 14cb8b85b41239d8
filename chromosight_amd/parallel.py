"""Multi-GPU driver of the hot path: one process per GPU, sub-matrices sharded across ranks,
candidate records gathered at the end.

Mirrors the only parallelism of the reference -- `multiprocessing.Pool(threads).imap` over
independent sub-matrices, one task = one sub-matrix x one kernel
(reference chromosight/cli/chromosight.py:406-410, 748-752) -- with `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).  Sub-matrices are
independent (own distance law, mask and foci), so the data path has no collective; the only
exchange is the final gather of the per-block pattern tables (a few MB at most).

    tables = detect_blocks(blocks, kernel_config, kernel_matrix, detector=...)

`blocks` is a list of objects with the attributes pattern_detector reads (`matrix`,
`detectable_bins`, `max_dist`, `inter`); every rank passes the same list (or builds the same
list lazily through `loader`), processes its own share and receives the gathered result.
"""
import os
import sys

import numpy as np

RECORD_FIELDS = ("block", "bin1", "bin2", "score", "pvalue")


def block_cost(shape, max_dist, inter):
    """Pixels the correlation scans for one sub-matrix (SURVEY.md 8(d) unit of work)."""
    rows, cols = shape
    if inter or max_dist is None:
        return int(rows) * int(cols)
    return int(rows) * int(min(max_dist + 1, cols))


def assign_blocks(costs, world_size):
    """Longest-processing-time-first assignment of blocks to ranks.  Returns a list with, for
    every rank, the (sorted) indices of its blocks; deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0] * world_size
    owned = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owned[r].append(i)
        load[r] += costs[i]
    return [sorted(o) for o in owned]


_COMMS = {}


class NativeComm:
    """The library's RCCL communicator (csrc/cs_comm.cpp: ncclAllGather / ncclAllReduce behind the C ABI, host arrays in
    and out, no torch tensors in the exchange)."""

    def __init__(self, device, rank, world, unique_id):
        import ctypes as C
        from ._lib import load_library
        self.C, self.lib = C, load_library()
        self.rank, self.world = rank, world
        self.handle = C.c_void_p()
        rc = self.lib.cs_comm_create(int(device), int(rank), int(world), unique_id, C.byref(self.handle))
        if rc != 0:
            raise RuntimeError(f"cs_comm_create failed ({rc}): {self.lib.cs_comm_last_error(None).decode()}")

    @staticmethod
    def unique_id():
        import ctypes as C
        from ._lib import load_library
        buf = (C.c_char * 128)()
        rc = load_library().cs_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"cs_comm_unique_id failed ({rc})")
        return bytes(buf)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"RCCL exchange failed ({rc}): {self.lib.cs_comm_last_error(self.handle).decode()}")

    def allgather_rows(self, rows, first_cap=0):
        """Rank-order concatenation of every rank's (n_i, w) float64 array, and the per-rank counts.  first_cap: this rank's
        first capacity in rows (0: a guess from its own count) -- the self-check hands a small one so that the retry runs."""
        C = self.C
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        width = int(rows.shape[1])
        counts = np.zeros(self.world, dtype=np.int64)
        # first guess; CHROMOSIGHT_HIP_GATHER_CAP forces a small one (tests of the retry).  The library exchanges the
        # capacities with the counts and answers "overflow" on EVERY rank when any rank lacks room, so all ranks retry
        # together with the exact total (the collectives stay matched whatever the per-rank guesses were).
        cap = int(first_cap) or int(os.environ.get("CHROMOSIGHT_HIP_GATHER_CAP", 0)) or max(4 * rows.shape[0] * self.world, 4096)
        while True:
            out = np.empty((cap, width))
            rc = self.lib.cs_comm_allgather_rows(self.handle, rows.ctypes.data, rows.shape[0], width, out.ctypes.data, cap,
                                                 counts.ctypes.data)
            if rc == -4:                      # CS_ERR_OVERFLOW: the counts say how much room is needed
                cap = int(counts.sum())
                continue
            self._check(rc)
            return out[:int(counts.sum())], counts

    def allgather_rows_once(self, rows):
        """allgather_rows in ONE collective (cs_comm_allgather_rows_once): the slot is what the previous exchange of this width
        needed, with a quarter to spare -- a step of a sharded run finds about as many records as the step before; a list that
        outgrows it makes every rank call again with the longest list as the slot."""
        C = self.C
        rows = np.ascontiguousarray(rows, dtype=np.float64)
        width = int(rows.shape[1])
        counts = np.zeros(self.world, dtype=np.int64)
        slots = self.__dict__.setdefault("_slots", {})
        slot = slots.get(width) or int(os.environ.get("CHROMOSIGHT_HIP_GATHER_CAP", 0))
        if not slot:
            # the first exchange of this width: every rank must pass the SAME slot (the blocks of a collective have one size), and
            # only the counts of all ranks can say which -- the two-collective form learns them
            out, counts = self.allgather_rows(rows)
            longest = int(counts.max())
            slots[width] = max(longest + longest // 4 + 16, 256)
            return out, counts
        while True:
            out = np.empty((slot * self.world, width))
            rc = self.lib.cs_comm_allgather_rows_once(self.handle, rows.ctypes.data, rows.shape[0], width, slot, out.ctypes.data,
                                                      slot * self.world, counts.ctypes.data)
            longest = int(counts.max()) if self.world else 0
            if rc == -4:                      # CS_ERR_OVERFLOW, on every rank alike: the counts say what slot is needed
                slot = longest + longest // 4 + 16
                continue
            self._check(rc)
            slots[width] = max(longest + longest // 4 + 16, 256)
            return out[:int(counts.sum())], counts

    def allreduce_sum(self, array):
        out = np.ascontiguousarray(array, dtype=np.float64).copy()
        self._check(self.lib.cs_comm_allreduce_f64(self.handle, out.ctypes.data, out.size))
        return out

    def close(self):
        if self.handle:
            self.lib.cs_comm_destroy(self.handle)
            self.handle = None


def native_comm():
    """The RCCL communicator of this process group when the ranks are GPU ranks (torch backend "nccl"), else None (CPU
    tests over gloo keep the torch path).  The 128-byte id is created on rank 0 and broadcast through torch.distributed
    once; every exchange after that goes through the library."""
    dist, rank, world = _world()
    if dist is None or world == 1 or dist.get_backend() != "nccl" or os.environ.get("CHROMOSIGHT_HIP_TORCH_EXCHANGE"):
        return None
    key = (rank, world)
    if key not in _COMMS:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())

        def all_ok(ok):
            # every rank takes the same branch: a rank that could not load or create its communicator must not leave the
            # others inside a collective of the library
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        # 1. local: the library and the RCCL of its HIP runtime load on this rank; rank 0 makes the id
        try:
            from ._lib import load_library
            if load_library().cs_comm_available() != 0:
                raise RuntimeError("librccl of the library's HIP runtime does not load")
            my_id, err = (NativeComm.unique_id() if rank == 0 else None), None
        except Exception as exc:                              # noqa: BLE001 -- whatever it is, the torch path remains
            my_id, err = None, exc
        if not all_ok(err is None):
            _COMMS[key] = None
            if err is not None:
                sys.stderr.write(f"[chromosight_amd] rank {rank}: native RCCL exchange unavailable ({err}); torch.distributed instead\n")
            return None
        box = [my_id if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        # 2. collective: ncclCommInitRank on every rank
        try:
            comm, err = NativeComm(torch.cuda.current_device(), rank, world, box[0]), None
        except Exception as exc:                              # noqa: BLE001
            comm, err = None, exc
        if not all_ok(err is None):
            if comm is not None:
                comm.close()
            comm = None
            if err is not None:
                sys.stderr.write(f"[chromosight_amd] rank {rank}: RCCL communicator not created ({err}); torch.distributed instead\n")
        _COMMS[key] = comm
    return _COMMS[key]


# > 0 while a caller that asked for a purely local run is inside the genome drivers (genome_step(local=True): pipeline.detect):
# an initialised process group is then none of their business -- no sharding of the blocks, no collective
_LOCAL_DEPTH = [0]


def _world():
    if _LOCAL_DEPTH[0] > 0:
        return None, 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist, dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return None, 0, 1


def gather_records(records, device=None):
    """All-gather variable-length float64 record arrays (n_i x n_fields) from every rank.
    Count exchange followed by one padded all_gather (fixed-size records)."""
    dist, rank, world = _world()
    records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, len(RECORD_FIELDS))
    if dist is None or world == 1:
        return records
    comm = native_comm()
    if comm is not None:
        merged, _ = comm.allgather_rows(records)
        return merged[np.argsort(merged[:, 0], kind="stable")]
    import torch
    dev = device if device is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    count = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    width = max(max(counts), 1)
    padded = torch.zeros((width, len(RECORD_FIELDS)), dtype=torch.float64, device=dev)
    if records.shape[0]:
        padded[:records.shape[0]] = torch.from_numpy(records).to(dev)
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    out = [p[:c].cpu().numpy() for p, c in zip(parts, counts)]
    merged = np.concatenate(out, axis=0) if out else records
    # block order, then the per-block order of pattern_detector (stable)
    return merged[np.argsort(merged[:, 0], kind="stable")]


def detect_blocks(blocks, kernel_config, kernel_matrix, detector=None, full=True, tsvd=None,
                  loader=None, coords=None):
    """Run pattern_detector on this rank's share of `blocks` and gather all tables.

    blocks : list of contact-map-like objects, or of lightweight descriptors when `loader`
        (a callable descriptor -> contact map) is given; every descriptor needs `.shape`,
        `.max_dist`, `.inter` (the contact maps themselves are only built by the owning rank).
    coords : optional list (one entry per block, or None) of coordinate arrays -> quantify mode.
    Returns a float64 array of records (block, bin1, bin2, score, pvalue), identical on all ranks.
    """
    if detector is None:
        from .utils.detection import pattern_detector as detector
    dist, rank, world = _world()
    costs = []
    for b in blocks:
        shape = b.shape if hasattr(b, "shape") and not hasattr(b, "matrix") else b.matrix.shape
        costs.append(block_cost(shape, b.max_dist, b.inter))
    mine = assign_blocks(costs, world)[rank]
    rows = []
    for idx in mine:
        cmap = loader(blocks[idx]) if loader is not None else blocks[idx]
        block_coords = None if coords is None else coords[idx]
        if coords is not None and block_coords is None:
            continue
        table, _ = detector(cmap, kernel_config, kernel_matrix, coords=block_coords, full=full, tsvd=tsvd)
        if table is None or len(table) == 0:
            continue
        rec = np.empty((len(table), len(RECORD_FIELDS)))
        rec[:, 0] = idx
        rec[:, 1] = table["bin1"].to_numpy(dtype=np.float64)
        rec[:, 2] = table["bin2"].to_numpy(dtype=np.float64)
        rec[:, 3] = table["score"].to_numpy(dtype=np.float64)
        rec[:, 4] = table["pvalue"].to_numpy(dtype=np.float64)
        rows.append(rec)
    local = np.concatenate(rows, axis=0) if rows else np.zeros((0, len(RECORD_FIELDS)))
    return gather_records(local)


# ================================================================================================
# whole-genome detect, blocks sharded over the ranks
# ================================================================================================
GENOME_FIELDS = ("block", "bin1", "bin2", "score", "pvalue", "kernel_id", "iteration")


def _allreduce_sum(array):
    dist, rank, world = _world()
    if dist is None or world == 1:
        return array
    comm = native_comm()
    if comm is not None:
        return comm.allreduce_sum(array).reshape(np.shape(array))
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.from_numpy(np.array(array, dtype=np.float64, order="C")).to(dev)     # (a copy: on CPU ranks `.to` shares the caller's memory)
    dist.all_reduce(t)
    return t.cpu().numpy()


def all_gather_rows(array):
    """Rank-order concatenation of every rank's float64 (n_i, w) array, on every rank (count exchange +
    one padded all_gather).  Rows keep their order within a rank."""
    dist, rank, world = _world()
    array = np.ascontiguousarray(array, dtype=np.float64)
    if array.ndim != 2:
        raise ValueError("all_gather_rows needs a 2-D array")
    if dist is None or world == 1:
        return array
    comm = native_comm()
    if comm is not None:
        return comm.allgather_rows(array)[0]
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    count = torch.tensor([array.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    padded = torch.zeros((max(max(counts), 1), array.shape[1]), dtype=torch.float64, device=dev)
    if array.shape[0]:
        padded[:array.shape[0]] = torch.from_numpy(array).to(dev)
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


def split_rows(n, world):
    """Row windows of one n-row block over `world` ranks (equal rows: the band has the same number of
    scanned pixels on every row)."""
    return [(n * r // world, n * (r + 1) // world) for r in range(world)]


def detect_split_block(genome, ci, kernel_config, kernel, tsvd=None, smooth=False, band_dtype=np.float64,
                       want_windows=True, raw=False, staged=None):
    """One intra-chromosomal block over ALL ranks (the single-block case, e.g. one
    200 000-bin chromosome): every rank stages its row window of the block and the template's halo from
    its copy of the pixel table -- no halo exchange --, the per-diagonal (sum, count) of the distance
    law are all-reduced so every part is detrended by the same law, the thresholded pixels are
    all-gathered and labelled identically everywhere, each rank scores the foci of its rows, and the
    records are gathered.  Same (table, windows) as pipeline.detect_block on one GPU, on every rank.
    `staged`: the block returned by an earlier call's stage_split (reused across templates)."""
    from . import pipeline
    block = staged if staged is not None else stage_split(genome, ci, kernel_config, smooth=smooth, band_dtype=band_dtype)
    return pipeline.detect_block(genome, block, kernel_config, kernel, tsvd=tsvd, want_windows=want_windows, raw=raw,
                                 all_gather=all_gather_rows)


def stage_split(genome, ci, kernel_config, smooth=False, band_dtype=np.float64, resident=True):
    """This rank's row window of block `ci`, detrended by the all-reduced distance law."""
    dist, rank, world = _world()
    max_dist = max(kernel_config["max_dist"] // genome.binsize, 1)
    largest = max(np.shape(k)[0] for k in kernel_config["kernels"])
    rows = split_rows(genome.chrom_size(ci), world)[rank]
    if world == 1:
        return genome.stage_intra(ci, max_dist, largest, smooth=smooth, band_dtype=band_dtype, resident=resident)
    return genome.stage_intra(ci, max_dist, largest, smooth=smooth, band_dtype=band_dtype, resident=resident, rows=rows,
                              reduce=_allreduce_sum)


class SplitBlockScan:
    """The correlation pass of ONE sub-matrix row-split over all ranks (the single 200 000-bin block of the
    north star), at map level: rank r owns the output rows split_rows(n, world)[r] and holds only those rows of the band plus
    the template's halo (cs_matrix.row0: no halo exchange).  One step is what the split costs a rank beside its share of
    the tiles:

      1. the per-diagonal (sum, count) of the distance law of its rows all-reduced -- every part must be detrended by the
         same law (stage_split does this once per staging; here it is part of every step so that the collective is timed);
      2. `correlate()`: the coefficient map of its rows (engine.run_normxcorr2(..., row_window=rows));
      3. `candidates()`: the (row, col, value) triples of its rows at or above the threshold, all-gathered so that every
         rank can label the same merged list (detect_split_block continues from there).

    correlate / candidates are the caller's closures (device calls in bench.py and the GPU tests, numpy stand-ins in the CPU
    test); with one rank the step is the correlation call alone -- the exchanges fall away, which makes the 1-rank figure
    the plain kernel figure.  Times of the last step: `exchange_ms` (both collectives), `step_ms`."""

    def __init__(self, n, law_part, correlate, candidates):
        dist, self.rank, self.world = _world()
        self.n = int(n)
        self.rows = split_rows(self.n, self.world)[self.rank]
        self.law_part = np.ascontiguousarray(law_part, dtype=np.float64)
        self.correlate, self.candidates = correlate, candidates
        self.exchange_ms = self.step_ms = 0.0

    def step(self):
        import time
        t0 = time.perf_counter()
        if self.world == 1:
            self.correlate()
            self.step_ms = (time.perf_counter() - t0) * 1e3
            return self.law_part, None
        law = _allreduce_sum(self.law_part)
        t1 = time.perf_counter()
        self.correlate()
        cand = np.ascontiguousarray(self.candidates(), dtype=np.float64).reshape(-1, 3)
        t2 = time.perf_counter()
        merged = all_gather_rows(cand)
        t3 = time.perf_counter()
        self.exchange_ms = ((t1 - t0) + (t3 - t2)) * 1e3
        self.step_ms = (t3 - t0) * 1e3
        return law, merged


def _gather(records, n_fields, once=False):
    """gather_records for an arbitrary record width.  once: the one-collective form of the native exchange (repeating steps)."""
    dist, rank, world = _world()
    records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, n_fields)
    if dist is None or world == 1:
        return records
    comm = native_comm()
    if comm is not None:
        merged, _ = comm.allgather_rows_once(records) if once else comm.allgather_rows(records)
        return merged[np.argsort(merged[:, 0], kind="stable")]
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    count = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = torch.cat(counts).cpu().tolist()                 # one device -> host copy, not one per rank
    width = max(max(counts), 1)
    padded = torch.zeros((width, n_fields), dtype=torch.float64, device=dev)
    if records.shape[0]:
        padded[:records.shape[0]] = torch.from_numpy(records).to(dev)
    parts = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    parts = torch.stack(parts).cpu().numpy()                   # likewise
    merged = np.concatenate([parts[r, :c] for r, c in enumerate(counts)], axis=0)
    return merged[np.argsort(merged[:, 0], kind="stable")]


def stage_genome(genome, kernel_configs, owned=None, smooth=False, band_dtype=np.float64, lazy64=None):
    """Stage this rank's blocks for several pattern configurations (detect_genome(..., staged=...)): ONCE at the longest
    keep distance any of them needs, the others scan band views of the same blocks (DeviceCool.view_for).
    lazy64 (default: on for templates of up to 17 x 17 in float32 arithmetic, CHROMOSIGHT_HIP_F64_TWIN=1 turns it off): the
    float64 bands are not stored beyond their first diagonals -- the exact evaluation of the candidates and the windows of
    the records recompute the pixels they read from the pixel table (pipeline.DeviceCool._stage_fast).
    Returns a StagedSet {chromosome: StagedBlock} (for_config(i): the set configuration i scans)."""
    dist, rank, world = _world()
    if lazy64 is None:
        from .engine import get_precision
        lazy64 = (not smooth and not os.environ.get("CHROMOSIGHT_HIP_F64_TWIN") and get_precision() == "f32"
                  and np.dtype(band_dtype) == np.float64
                  and all(max(np.shape(k)) <= 17 for cfg in kernel_configs for k in cfg["kernels"]))
    dists = [max(cfg["max_dist"] // genome.binsize, 1) for cfg in kernel_configs]
    tallest = [max(np.shape(k)[0] for k in cfg["kernels"]) for cfg in kernel_configs]
    max_dist = max(dists)
    if owned is None:
        costs = [block_cost((genome.chrom_size(ci),) * 2, max_dist, False) for ci in range(genome.n_chrom)]
        owned = assign_blocks(costs, world)[rank]
    owned = list(owned)
    events = hasattr(genome, "dev") and hasattr(genome.dev, "new_event")

    def stage(which, slot):
        md = max(dists[i] for i in which)
        extra = dict(lazy64=True) if lazy64 and hasattr(genome, "_stage_fast") else {}
        out = StagedSet(zip(owned, genome.stage_blocks(owned, md, max(tallest[i] for i in which), smooth=smooth, band_dtype=band_dtype,
                                                       **extra)))
        if events:
            # staging is asynchronous on the genome's stream: streams of other contexts wait for this event, not the host
            name = f"_ready_event{slot}"
            if getattr(genome, name, None) is None:
                setattr(genome, name, genome.dev.new_event())
            genome.dev.record(getattr(genome, name))
            out.ready = getattr(genome, name)
        return out

    wide = list(range(len(kernel_configs)))
    staged = stage(wide, 0)
    staged.by_config = {}
    # Blocks a configuration cannot take as a band view of the shared staging -- a short chromosome staged dense for the
    # widest pattern is a band for a narrower one -- are staged for it here, on the genome's stream, instead of by the
    # host thread that will scan that configuration (detect_patterns: the thread would have to use the genome's context
    # beside the calling thread, and the templates of a 1-D pattern could not share one launch chain).
    if hasattr(genome, "view_for") and not smooth:
        extra_any = False
        for i in wide:
            need = []
            for ci in owned:
                blk = staged[ci]
                same = blk.max_dist == dists[i] and blk.keep == min(dists[i], genome.chrom_size(ci)) + tallest[i]
                if not same and genome.view_for(blk, dists[i], tallest[i]) is None:
                    need.append(ci)
            if need:
                own = StagedSet(staged)
                own.update(zip(need, genome.stage_blocks(need, dists[i], tallest[i], smooth=smooth, band_dtype=band_dtype)))
                staged.by_config[i] = own
                extra_any = True
        if extra_any and events:
            genome.dev.record(staged.ready)                  # (one event: after everything staged here)
            for own in staged.by_config.values():
                if own is not None:
                    own.ready = staged.ready
    return staged


class StagedSet(dict):
    """{chromosome: StagedBlock} of stage_genome + the event that fires when their staging is complete."""
    ready = None
    by_config = None

    def for_config(self, i):
        return (self.by_config or {}).get(i) or self


# wall time this process has spent inside record exchanges (bench.py --gpus N reports it per rank)
TIMERS = {"exchange_ms": 0.0, "exchanges": 0}


def _exchange_records(local, n_kernels, n_iterations):
    """The ONE record exchange of a detect call: count + padded all-gather (RCCL: csrc/cs_comm.cpp); per (template,
    iteration) the rows come in the order separate gathers would give -- ranks concatenated, stable by block."""
    if _world()[2] == 1:
        return local                                         # already (template, iteration, block)-ordered
    import time
    t0 = time.perf_counter()
    merged = _gather(local, len(GENOME_FIELDS))              # ranks concatenated, stably sorted by block
    TIMERS["exchange_ms"] += (time.perf_counter() - t0) * 1e3
    TIMERS["exchanges"] += 1
    return merged[np.argsort(merged[:, 5] * n_iterations + merged[:, 6], kind="stable")]


def _exchange_records_many(locals_, kernel_configs):
    """The record exchanges of SEVERAL configurations of one step as ONE exchange (and, on the native transport, one collective:
    NativeComm.allgather_rows_once): the lists travel side by side with the configuration's index in an eighth column and come
    apart again in the order _exchange_records gives each of them."""
    if _world()[2] == 1:
        return list(locals_)
    import time
    t0 = time.perf_counter()
    w = len(GENOME_FIELDS)
    parts = []
    for i, rec in enumerate(locals_):
        rec = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, w)
        parts.append(np.concatenate([rec, np.full((rec.shape[0], 1), float(i))], axis=1))
    merged = _gather(np.concatenate(parts, axis=0) if parts else np.zeros((0, w + 1)), w + 1, once=True)
    TIMERS["exchange_ms"] += (time.perf_counter() - t0) * 1e3
    TIMERS["exchanges"] += 1
    out = []
    for i, cfg in enumerate(kernel_configs):
        mine = merged[merged[:, w] == i][:, :w]               # (ranks concatenated, stably sorted by block: _gather)
        out.append(mine[np.argsort(mine[:, 5] * cfg["max_iterations"] + mine[:, 6], kind="stable")])
    return out


def transport():
    """What carries this process group's exchanges: "native_rccl" (csrc/cs_comm.cpp), "torch_nccl", "torch_gloo" or "none"."""
    dist, rank, world = _world()
    if dist is None or world == 1:
        return "none"
    if native_comm() is not None:
        return "native_rccl"
    return "torch_" + str(dist.get_backend())


def exchange_self_check():
    """Before a timed multi-GPU region: one padded all-gather of rows with UNEVEN counts (rank r sends 3 r + 1 rows; rank 0
    with a first capacity that is too small, so the retry runs), three exchanges of the one-collective form (slot learnt, slot
    kept, slot outgrown) and one all-reduce through the library's communicator,
    compared with torch.distributed's collectives on the same data.  On a mismatch or an error on ANY rank every rank
    drops to the torch transport, loudly.  Returns the transport in use afterwards."""
    dist, rank, world = _world()
    if dist is None or world == 1:
        return "none"
    comm = native_comm()
    if comm is None:
        return transport()
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    # torch's reference collective first and OUTSIDE the try: every rank issues the same torch collectives whatever happens to
    # the native ones on it (a rank that skipped this all-reduce after a native error would leave the others inside it)
    vec = np.linspace(0.0, 1.0, 97) * (rank + 1)
    t = torch.from_numpy(vec.copy()).to(dev)
    dist.all_reduce(t)
    want_sum = t.cpu().numpy()
    ok, why = True, ""
    try:
        mine = (np.arange((3 * rank + 1) * 5, dtype=np.float64).reshape(-1, 5) + 1000.0 * rank)
        got, counts = comm.allgather_rows(mine, first_cap=2 if rank == 0 else 0)      # (rank 0 too small: the retry runs)
        want = np.concatenate([np.arange((3 * r + 1) * 5, dtype=np.float64).reshape(-1, 5) + 1000.0 * r for r in range(world)])
        if counts.tolist() != [3 * r + 1 for r in range(world)] or not np.array_equal(got, want):
            ok, why = False, "all-gather of rows differs from the expected concatenation"
        # the one-collective form a replayed step uses (cs_comm_allgather_rows_once), on a width of its own: the first exchange learns
        # the slot (two collectives), the second fits it, the third outgrows it on the last rank (every rank sends again)
        for k, grow in enumerate((1, 1, 40)):
            n_r = lambda r: (2 * r + 3) * (grow if r == world - 1 else 1)      # noqa: E731
            mine9 = np.arange(n_r(rank) * 9, dtype=np.float64).reshape(-1, 9) + 1e6 * rank + k
            got9, counts9 = comm.allgather_rows_once(mine9)
            want9 = np.concatenate([np.arange(n_r(r) * 9, dtype=np.float64).reshape(-1, 9) + 1e6 * r + k for r in range(world)])
            if ok and (counts9.tolist() != [n_r(r) for r in range(world)] or not np.array_equal(got9, want9)):
                ok, why = False, f"one-collective all-gather of rows (exchange {k}) differs from the expected concatenation"
        comm.__dict__.get("_slots", {}).pop(9, None)
        got_sum = comm.allreduce_sum(vec)              # (always issued: the native collectives stay matched across the ranks)
        if ok and not np.allclose(got_sum, want_sum, rtol=1e-13, atol=0):
            ok, why = False, "all-reduce differs from torch.distributed's"
    except Exception as exc:                                  # noqa: BLE001 -- whatever it is, the torch path remains
        ok, why = False, repr(exc)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if not bool(flag.item()):
        if not ok:
            sys.stderr.write(f"[chromosight_amd] rank {rank}: native RCCL exchange failed its self-check ({why}); torch.distributed instead\n")
        _COMMS[(rank, world)] = None
    return transport()


_PATTERN_THREADS = None


def detect_patterns(genome, kernel_configs, owned=None, staged=None, tsvd=None, exchange=True):
    """detect_genome for SEVERAL pattern configurations on blocks staged once (stage_genome), concurrently: a genome
    scan is a few bandwidth / matrix-core-bound launches (a 2-D pattern's tile kernels) next to many latency-bound ones
    (1-D patterns: a few thousand pixels per block and template; labelling, statistics, synchronisations) -- run one after
    the other the latter leave the chip idle, run side by side they disappear under the former.  One host thread per
    configuration whose templates are not iterated; the record exchanges then follow in configuration order on this
    thread (collectives must be issued in the same order on every rank), and so do iterated configurations as a whole.
    Returns the list of detect_genome results."""
    global _PATTERN_THREADS
    if staged is None:
        staged = stage_genome(genome, kernel_configs, owned=owned)
    dist, rank, world = _world()
    if owned is None:
        max_dist = max(max(cfg["max_dist"] // genome.binsize, 1) for cfg in kernel_configs)
        costs = [block_cost((genome.chrom_size(ci),) * 2, max_dist, False) for ci in range(genome.n_chrom)]
        owned = assign_blocks(costs, world)[rank]
    owned = list(owned)
    complete = all(ci in staged for ci in owned)
    side = [i for i, cfg in enumerate(kernel_configs) if cfg["max_iterations"] == 1] if complete and len(kernel_configs) > 1 else []
    if _PATTERN_THREADS is None and side:
        import concurrent.futures
        _PATTERN_THREADS = concurrent.futures.ThreadPoolExecutor(max_workers=4, thread_name_prefix="chromosight-pattern")
    ready = getattr(staged, "ready", None)
    if side and ready is None:
        genome.dev.sync()                                    # staging is complete before several threads and streams read it
    pick = lambda i: staged.for_config(i) if isinstance(staged, StagedSet) else staged
    # the configuration that scans the widest band stays on this thread (a hand-over to a pool thread costs 50-100 us, and
    # its pass is the longest); the others go to the pool first
    here = max(side, key=lambda i: kernel_configs[i]["max_dist"]) if side else None
    futures = {i: _PATTERN_THREADS.submit(detect_genome, genome, kernel_configs[i], tsvd=tsvd, owned=owned, staged=pick(i),
                                          exchange=False, exclusive=len(side) < 2, own_context=True)
               for i in side if i != here}
    local = {}
    if here is not None:
        local[here] = detect_genome(genome, kernel_configs[here], tsvd=tsvd, owned=owned, staged=pick(here), exchange=False,
                                    exclusive=len(side) < 2)
    results = [None] * len(kernel_configs)
    for i, cfg in enumerate(kernel_configs):
        if i not in side:
            results[i] = detect_genome(genome, cfg, tsvd=tsvd, owned=owned, staged=pick(i), exclusive=not side, exchange=exchange)
    for i in side:                                           # (collectives in the same order on every rank)
        mine = local[i] if i in local else futures[i].result()
        # (exchange=False: this rank's records -- genome_step exchanges the configurations of a step together)
        results[i] = _exchange_records(mine, len(kernel_configs[i]["kernels"]), 1) if exchange else mine
    return results


def genome_step(genome, kernel_configs, owned=None, tsvd=None, local=False):
    """One detect step of a sharded genome: stage_genome + detect_patterns, the results of detect_patterns.
    local=True: this process alone, whatever process group is initialised -- every block, no collective (pipeline.detect: a
    call one rank may make on its own, as the reference's detect is).  For the usual
    pair of configurations (a 2-D pattern and a 1-D pattern with several templates, single iterations) every step after the
    first is ONE native call on the arguments the first step used (chromosight_amd/plan.py: cs_run_calls) plus one record
    exchange per configuration; anything else -- and any step whose call list reports an error -- takes the two calls."""
    if local:
        _LOCAL_DEPTH[0] += 1
        try:
            return genome_step(genome, kernel_configs, owned=owned, tsvd=tsvd)
        finally:
            _LOCAL_DEPTH[0] -= 1
    from . import plan as _plan
    dist, rank, world = _world()
    if owned is None:
        max_dist = max(max(cfg["max_dist"] // genome.binsize, 1) for cfg in kernel_configs)
        costs = [block_cost((genome.chrom_size(ci),) * 2, max_dist, False) for ci in range(genome.n_chrom)]
        owned = assign_blocks(costs, world)[rank]
    owned = list(owned)
    if not owned or not _plan.plannable(genome, kernel_configs, tsvd):
        staged = stage_genome(genome, kernel_configs, owned=owned)
        return detect_patterns(genome, kernel_configs, owned=owned, staged=staged, tsvd=tsvd)
    key = (tuple(owned), tuple((cfg["max_dist"], cfg["pearson"], cfg["max_perc_undetected"], cfg["max_perc_zero"],
                                tuple(np.asarray(k, dtype=np.float64).tobytes() for k in cfg["kernels"])) for cfg in kernel_configs))
    plans = genome.__dict__.setdefault("_step_plans", {})
    plan = plans.get(key)
    if plan is not None and plan.ok:
        local = plan.run()
        if local is not None:
            return _exchange_records_many(local, kernel_configs)     # (one exchange for the step's configurations)
        plans.pop(key, None)                                  # (a result list outgrew its capacity: the usual path, a new plan)
    from . import _lib
    _lib.CAPTURE = captured = []
    try:
        staged = stage_genome(genome, kernel_configs, owned=owned)
        # (this rank's records, then the ONE exchange a replayed step makes too: whichever way a rank takes through a step --
        # a plan that outgrew its buffers falls back here -- the ranks issue the same collectives)
        results = detect_patterns(genome, kernel_configs, owned=owned, staged=staged, tsvd=tsvd, exchange=False)
    finally:
        _lib.CAPTURE = None
    results = _exchange_records_many(results, kernel_configs)
    if len(plans) > 8:
        plans.clear()
    plans[key] = _plan.StepPlan(genome, kernel_configs, owned, captured, staged)
    return results


def detect_genome(genome, kernel_config, tsvd=None, smooth=False, band_dtype=np.float64, stage=None, detect=None,
                  owned=None, staged=None, exchange=True, exclusive=True, own_context=False):
    """`chromosight detect` over all intra-chromosomal blocks of a DeviceCool, sharded over the ranks
    like the reference's Pool.imap over sub-matrices (cli/chromosight.py:738-755): every rank stages
    and scans its own blocks (LPT assignment by band pixels), the per-block tables are all-gathered
    (count + padded all_gather: RCCL over xGMI with the "nccl" backend), and when the template is
    refined over iterations the pileup is formed from an all-reduce of the per-rank window sums and
    counts, so every rank continues with the same template (cli/chromosight.py:791).

    `genome` needs n_chrom, chrom_size(ci), binsize; `stage(genome, ci, max_dist, largest)` and
    `detect(genome, block, cfg, kernel, tsvd)` default to the device pipeline (injected in CPU tests).
    `owned`: this rank's block indices when the caller fixed the assignment (e.g. a rank that only
    holds the pixels of its own chromosomes); default: LPT by scanned pixels.
    `staged`: blocks of stage_genome (staged once for several patterns); a block that cannot serve this configuration
    through a band view is staged here as usual.  exchange=False: this rank's records only (detect_patterns).
    exclusive=False: other configurations are being scanned on this GPU at the same time (detect_patterns).
    own_context=True: this call runs on a host thread beside another one that uses the genome's own context and stream
    (detect_patterns): every device call goes to a context and stream of this thread (a context serves one call in flight).
    Returns float64 records (block, bin1, bin2, score, pvalue, kernel_id, iteration), block-local
    bins, identical on all ranks, in the single-process order."""
    batch = None
    stage_default = stage is None
    if stage is None and detect is None:
        from . import pipeline
        batch = lambda g, blks, cfg, k, t, w=True, defer=False, dev=None, stream=None: pipeline.detect_blocks(
            g, blks, cfg, k, tsvd=t, raw=True, want_windows=w, defer=defer, dev=dev, stream=stream, merged=True,
            exclusive=exclusive and len(kernel_config["kernels"]) == 1)
    if stage is None or detect is None:
        from . import pipeline
        stage = stage or (lambda g, ci, md, lk: g.stage_intra(ci, md, lk, smooth=smooth, band_dtype=band_dtype,
                                                              resident=True))
        detect = detect or (lambda g, blk, cfg, k, t: pipeline.detect_block(g, blk, cfg, k, tsvd=t, raw=True))
    dist, rank, world = _world()
    max_dist = max(kernel_config["max_dist"] // genome.binsize, 1)
    largest = max(np.shape(k)[0] for k in kernel_config["kernels"])
    sizes = [genome.chrom_size(ci) for ci in range(genome.n_chrom)]
    costs = [block_cost((n, n), max_dist, False) for n in sizes]
    mine = list(owned) if owned is not None else assign_blocks(costs, world)[rank]
    have = {}
    staged_in = staged
    if staged is not None and not smooth:
        for ci in mine:
            blk = staged.get(ci)
            if blk is None:
                continue
            view = blk if (blk.max_dist == max_dist and blk.keep == min(max_dist, genome.chrom_size(ci)) + largest) \
                else genome.view_for(blk, max_dist, largest)
            if view is not None:
                have[ci] = view
    todo = [ci for ci in mine if ci not in have]
    if batch is not None and stage_default:
        # the device pipeline stages its blocks with one native call (pipeline.DeviceCool.stage_blocks)
        fresh = dict(zip(todo, genome.stage_blocks(todo, max_dist, largest, smooth=smooth, band_dtype=band_dtype))) if todo else {}
    else:
        fresh = {ci: stage(genome, ci, max_dist, largest) for ci in todo}
    staged = {**have, **fresh}
    out = []
    pending = []
    # Templates that do not depend on each other (a single iteration each) are scanned concurrently: one host thread per
    # template, each with its own context and stream on this GPU (pipeline._Workers), so the latency-bound stages of one
    # template's chain (labelling, statistics, the synchronisations) run under another's, and this thread applies the
    # acceptance rules (numpy, a third of a template's wall time on the 23-block genome) while the others are on the device.
    kernels = [np.asarray(k, dtype=np.float64) for k in kernel_config["kernels"]]
    overlap = (batch is not None and stage_default and kernel_config["max_iterations"] == 1 and len(kernels) > 1 and bool(mine)
               and hasattr(genome, "workers") and not os.environ.get("CHROMOSIGHT_HIP_NO_TEMPLATE_OVERLAP"))
    futures = []
    joint = None
    # (a 1-D pattern with ONE template -- hairpins -- takes the same joint chain: the entry serves 1 to 4 templates, and it is the
    # chain a StepPlan replays, chromosight_amd/plan.py)
    one_template_1d = (batch is not None and stage_default and kernel_config["max_iterations"] == 1 and len(kernels) == 1 and bool(mine)
                       and hasattr(genome, "workers") and not os.environ.get("CHROMOSIGHT_HIP_NO_TEMPLATE_OVERLAP"))
    if (overlap or one_template_1d) and kernel_config["max_dist"] == 0 and tsvd is None and not todo:
        # a 1-D pattern's templates share one launch chain (cs_detect_foci_batch_templates): on a worker context, so that the
        # chain runs beside whatever this genome's own stream is doing (another pattern's tile kernels)
        pool = genome.workers(1)
        ready = getattr(staged_in, "ready", None)
        if ready is None:
            genome.dev.sync()
        dev, stream = pool.device()
        if ready is not None:
            dev.wait_event(ready, stream)
        joint = pipeline.detect_blocks_templates(genome, [staged[ci] for ci in mine], kernel_config, kernels, dev=dev, stream=stream)
        if joint is not None:
            joint = joint()                                  # the acceptance rules, here
            overlap = False
    if overlap:
        pool = genome.workers(min(len(kernels), 3))
        ready = getattr(staged_in, "ready", None) if not todo else None
        if ready is None:
            genome.dev.sync()                                # the staged blocks are complete before other streams read them
        blocks_mine = [staged[ci] for ci in mine]

        def scan(k):
            dev, stream = pool.device()
            if ready is not None:
                dev.wait_event(ready, stream)                # device-side: the staging of these blocks has finished
            return batch(genome, blocks_mine, kernel_config, kernels[k], tsvd, False, True, dev, stream)

        futures = [pool.pool.submit(scan, k) for k in range(len(kernels))]
    for kernel_id, kernel in enumerate(kernels):
        for it in range(kernel_config["max_iterations"]):
            rows, wins = [], []
            # the windows only feed the pileup of the next iteration: the last one does not fetch them
            need_windows = it + 1 < kernel_config["max_iterations"]
            if joint is not None:
                results = joint[kernel_id]
            elif overlap:
                # the acceptance rules (many small numpy calls) here, one template after the other: spread over the worker
                # threads they fight for the interpreter lock and each takes four times as long
                results = futures[kernel_id].result()()
            elif batch and own_context and hasattr(genome, "workers"):
                # a single template on a pool thread: not on the genome's context, which the calling thread is using
                dev_t, stream_t = genome.workers(1).device()
                ready_t = getattr(staged_in, "ready", None) if not todo else None
                if ready_t is not None:
                    dev_t.wait_event(ready_t, stream_t)
                else:
                    genome.dev.sync()
                results = batch(genome, [staged[ci] for ci in mine], kernel_config, kernel, tsvd, need_windows, False, dev_t, stream_t)
            else:
                results = batch(genome, [staged[ci] for ci in mine], kernel_config, kernel, tsvd, need_windows) if batch else None
            if isinstance(results, tuple):
                # one native call covered every block: its table is already the concatenation of the blocks' tables
                table, kept, windows = results
                rec = np.empty((len(table), len(GENOME_FIELDS)))
                rec[:, 0] = np.repeat(np.asarray(mine, dtype=np.float64), kept)
                rec[:, 1:5] = table
                rec[:, 5] = kernel_id
                rec[:, 6] = it
                rows.append(rec)
                if windows is not None:
                    wins.append(windows)
            for pos, ci in enumerate(mine if not isinstance(results, tuple) else ()):
                table, windows = results[pos] if results is not None else detect(genome, staged[ci], kernel_config,
                                                                                kernel, tsvd)
                if table is None or len(table) == 0:
                    continue
                rec = np.empty((len(table), len(GENOME_FIELDS)))
                rec[:, 0] = ci
                if isinstance(table, np.ndarray):              # raw records (bin1, bin2, score, pvalue)
                    rec[:, 1:5] = table
                else:
                    rec[:, 1] = table["bin1"].to_numpy(dtype=np.float64)
                    rec[:, 2] = table["bin2"].to_numpy(dtype=np.float64)
                    rec[:, 3] = table["score"].to_numpy(dtype=np.float64)
                    rec[:, 4] = table["pvalue"].to_numpy(dtype=np.float64)
                rec[:, 5] = kernel_id
                rec[:, 6] = it
                rows.append(rec)
                if windows is not None:
                    wins.append(windows)
            local = np.concatenate(rows, axis=0) if rows else np.zeros((0, len(GENOME_FIELDS)))
            pending.append(local)                            # gathered once, at the end of the call
            if kernel_config["max_iterations"] == 1:
                continue
            # iterated template: ONE all-reduce per iteration carries the pileup sums, their counts and the number of
            # patterns (no pattern on any rank: next template, cli/chromosight.py:786-789); the records wait for the end
            if it + 1 == kernel_config["max_iterations"]:
                continue                                     # nothing depends on the last iteration's patterns
            stack = np.concatenate(wins, axis=0) if wins else np.zeros((0,) + kernel.shape)
            if world == 1:
                if local.shape[0] == 0:
                    break
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    kernel = np.nanmean(stack, axis=0)           # detection.py:158-174
                continue
            both = np.concatenate([np.nansum(stack, axis=0).ravel(), np.sum(~np.isnan(stack), axis=0).astype(np.float64).ravel(),
                                   [float(local.shape[0])]])
            both = _allreduce_sum(both)
            if both[-1] == 0:
                break
            kk = kernel.size
            with np.errstate(all="ignore"):
                kernel = (both[:kk] / both[kk:2 * kk]).reshape(kernel.shape)
    local = np.concatenate(pending, axis=0) if pending else np.zeros((0, len(GENOME_FIELDS)))
    if not exchange:
        return local                                         # detect_patterns: the caller exchanges (in a fixed order)
    return _exchange_records(local, len(kernels), kernel_config["max_iterations"])


# ================================================================================================
# `quantify`, sub-matrices sharded over the ranks
# ================================================================================================
class QuantifyShard:
    """How pipeline.quantify shares its sub-matrices among the ranks: `select` keeps this rank's share of the (ca, cb,
    positions, coordinates) work list -- longest-processing-time-first by the pixels a sub-matrix stages plus the windows its
    positions read, the same list on every rank --, `merge` exchanges the scores once (one padded all-gather of
    (position, template, score, p-value, window) rows: RCCL with the "nccl" backend, gloo in the CPU tests)."""

    def __init__(self):
        self.dist, self.rank, self.world = _world()

    def select(self, todo, dcool, max_dist):
        costs = []
        for ca, cb, sel, _ in todo:
            n_r, n_c = dcool.chrom_size(ca), dcool.chrom_size(cb)
            staged = n_r * min(max_dist + 1, n_c) if ca == cb else n_r * n_c
            costs.append(int(staged) + 512 * len(sel))
        owned = assign_blocks(costs, self.world)[self.rank]
        return [todo[i] for i in owned]

    def merge(self, score_out, pval_out, win_out, sels):
        n_k = len(score_out)
        kk = int(np.prod(win_out[0].shape[1:]))
        where = np.concatenate(sels) if sels else np.zeros(0, dtype=np.int64)
        rows = np.empty((n_k * where.size, 4 + kk))
        for k in range(n_k):
            part = rows[k * where.size:(k + 1) * where.size]
            part[:, 0] = where
            part[:, 1] = k
            part[:, 2] = score_out[k][where]
            part[:, 3] = pval_out[k][where]
            part[:, 4:] = win_out[k][where].reshape(where.size, kk)
        merged = all_gather_rows(rows)
        pos = merged[:, 0].astype(np.int64)
        for k in range(n_k):
            m = merged[:, 1] == k
            score_out[k][pos[m]] = merged[m, 2]
            pval_out[k][pos[m]] = merged[m, 3]
            win_out[k][pos[m]] = merged[m, 4:].reshape((-1,) + win_out[k].shape[1:])
        return score_out, pval_out, win_out


def quantify_genome(cool, positions, kernel_config, **options):
    """`chromosight quantify` with the sub-matrices sharded over the ranks like the reference's process pool
    (cli/chromosight.py:396-410: one task per sub-matrix that holds a position): every rank stages and scores its own
    sub-matrices (pipeline.quantify: one native call per template over all of them), ONE exchange carries the scores, and
    every rank returns the same (table, windows) a single process returns."""
    from . import pipeline
    return pipeline.quantify(cool, positions, kernel_config, shard=QuantifyShard(), **options)
