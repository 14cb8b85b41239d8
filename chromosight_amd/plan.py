"""A genome step as ONE native call (reference: the per-sub-matrix tasks of cli/chromosight.py:738-755).

`parallel.genome_step(genome, [cfg_2d, cfg_1d])` = stage_genome + detect_patterns.  On a rank's share of a genome the
interpreter between the native calls of a step costs as much as their kernels (~ 350 us of a 1.1 ms step), and
every step of a run on the same genome layout -- the steps of an iterated template, the patterns of a run, the steps of a
benchmark -- makes the SAME calls on the same buffers.  So the first step runs the usual way with the library's entries
recording their arguments (_lib.CAPTURE), a StepPlan turns them into a cs_run_calls list -- staging, the ready event, the
2-D pattern's chain and its acceptance rules on the calling thread; the event wait, the 1-D pattern's chain and its
acceptance rules on a worker thread of the library -- and every later step is that one call plus the slicing of its
result tables.  Everything is recomputed on the device each time (nothing is cached but arguments); a step whose call list
reports an error (a result list outgrew its capacity) falls back to the usual path and the plan is rebuilt.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (CALL_ACCEPT_RECORDS, CALL_DETECT_FOCI_BATCH_FINISH, CALL_DETECT_FOCI_BATCH_TEMPLATES, CALL_DETECT_FOCI_BLOCKS,
                   CALL_EVENT_RECORD, CALL_STAGE_BLOCKS, CALL_STREAM_WAIT_EVENT, CALL_STREAM_WAIT_TILES, FOCUS_DTYPE,
                   CsCall, raw_arg)

def _device_pvalues():
    from .utils.detection import device_pvalues
    return device_pvalues()


# argument slots of every entry: 'p' pointer, 'i' integer, 'd' double, in the order of the C prototype
_SLOTS = {
    "cs_stage_blocks": (CALL_STAGE_BLOCKS, "ppppid"),
    "cs_event_record": (CALL_EVENT_RECORD, "ppp"),
    "cs_stream_wait_event": (CALL_STREAM_WAIT_EVENT, "ppp"),
    "cs_detect_foci_blocks": (CALL_DETECT_FOCI_BLOCKS, "ppippppppipp"),
    "cs_detect_foci_batch_templates": (CALL_DETECT_FOCI_BATCH_TEMPLATES, "ppipippppipp"),
}


def _fill(call, fn, kinds, values, lane, after=-1):
    call.fn, call.lane, call.after, call.rc = fn, lane, after, 0
    np_, ni, nd = 0, 0, 0
    for kind, v in zip(kinds, values):
        v = raw_arg(v)
        if kind == "p":
            call.p[np_] = int(v) or None
            np_ += 1
        elif kind == "i":
            call.i[ni] = int(v)
            ni += 1
        else:
            call.d[nd] = float(v)
            nd += 1


_EPOCH = [0]


def _next_epoch():
    """Tile epochs (cs_stream_wait_tiles): 1 .. 2^23 - 1, growing, process-wide."""
    _EPOCH[0] = _EPOCH[0] % ((1 << 23) - 1) + 1
    return _EPOCH[0]


def _split(kernel_configs):
    """(2-D configuration or None, 1-D configuration or None) of a list of one or two configurations: a 2-D pattern scans a band
    (max_dist > 0), a 1-D pattern the first diagonals (max_dist == 0)."""
    cfg2 = [c for c in kernel_configs if c["max_dist"] > 0]
    cfg1 = [c for c in kernel_configs if c["max_dist"] == 0]
    return (cfg2[0] if cfg2 else None), (cfg1[0] if cfg1 else None)


class StepPlan:
    """The native call list of stage_genome + detect_patterns for (genome, configurations, owned): a 2-D pattern next to a 1-D
    pattern (the usual pair: loops + borders), or either of them alone (`chromosight detect` of one pattern: pipeline.detect)."""

    def __init__(self, genome, kernel_configs, owned, captured, staged):
        self.ok = False
        self.why = "not built"
        self.order = ["2" if c["max_dist"] > 0 else "1" for c in kernel_configs]
        cfg2, cfg1 = _split(kernel_configs)
        by_name = {}
        for name, args, _thread in captured:
            by_name.setdefault(name, []).append(args)
        stage, blocks, batch = (by_name.get(k, []) for k in ("cs_stage_blocks", "cs_detect_foci_blocks", "cs_detect_foci_batch_templates"))
        want = (1, 1 if cfg2 is not None else 0, 1 if cfg1 is not None else 0)
        if (len(stage), len(blocks), len(batch)) != want:
            # (extra stagings -- short chromosomes staged dense for the wider pattern --, retries, block-by-block fall-backs)
            self.why = f"{len(stage)} staging calls, {len(blocks)} 2-D chains, {len(batch)} 1-D chains in the step ({want} wanted)"
            return
        ctx_a = raw_arg(stage[0][0])
        self.genome, self.owned, self.staged = genome, list(owned), staged
        self.keep = (captured, kernel_configs)               # every argument array stays alive with the plan
        dev = genome.dev
        locks = [genome.dev.lock]
        sizes = [genome.chrom_size(ci) for ci in self.owned]

        def accept_io(cap, n_virtual, max_dist, kernel):
            geo = np.empty((3, n_virtual), dtype=np.int32)
            reps = n_virtual // len(sizes)
            geo[0] = geo[1] = np.tile(np.asarray(sizes, dtype=np.int32), reps)
            geo[2] = np.minimum(max_dist, 2 ** 31 - 1)
            return dict(geo=geo, table=np.empty((cap, 4)), ok=np.empty(cap, dtype=np.uint8), kept=np.zeros(n_virtual, dtype=np.int64),
                        k=np.shape(kernel))

        self.n_templates = 0
        if cfg2 is not None:
            if raw_arg(blocks[0][0]) != ctx_a or blocks[0][11] is not None or int(blocks[0][2]) != len(self.owned):
                self.why = "the 2-D chain: another context, windows asked for, or not every owned block"
                return
            n2 = int(np.frombuffer(blocks[0][10], dtype=np.int64).sum())
            # result buffers of the plan's own (page-locked: the chains write into them from the device; the devices' shared
            # result pools may be re-allocated by other calls)
            self.cap2 = max(4 * n2, 4096)
            self.rec2 = dev.pinned_empty(self.cap2, FOCUS_DTYPE)
            self.counts2 = blocks[0][10]                      # ctypes int64 array of the recorded call
            self.acc2 = accept_io(self.cap2, len(self.owned), max(cfg2["max_dist"] // genome.binsize, 1), cfg2["kernels"][0])
        if cfg1 is not None:
            ready = getattr(staged, "ready", None)
            if ready is None:
                self.why = "the staging has no ready event"
                return
            ctx_b, stream_b = raw_arg(batch[0][0]), raw_arg(batch[0][1])
            rec_ev = [a for a in by_name.get("cs_event_record", []) if raw_arg(a[0]) == ctx_a and raw_arg(a[1]) == raw_arg(ready)]
            waits = [a for a in by_name.get("cs_stream_wait_event", []) if raw_arg(a[0]) == ctx_b and raw_arg(a[1]) == stream_b
                     and raw_arg(a[2]) == raw_arg(ready)]
            if not rec_ev or not waits or batch[0][11] is not None or int(batch[0][2]) != len(self.owned):
                self.why = (f"the 1-D chain: event records {len(rec_ev)}, event waits {len(waits)}, windows asked for "
                            f"{batch[0][11] is not None}, blocks {int(batch[0][2])} of {len(self.owned)}")
                return
            # the worker context of the 1-D chain (a pattern thread's, pipeline._Workers.device) as the object whose lock
            # serialises its use
            dev_b = _lib.device_of(ctx_b)
            if dev_b is None:
                self.why = "the 1-D chain ran on a context this process does not manage"
                return
            if dev_b is not genome.dev:
                locks.append(dev_b.lock)
            self.dev_b = dev_b                                # (the worker context stays alive with the plan)
            self.n_templates = int(batch[0][4])
            n1 = int(np.frombuffer(batch[0][10], dtype=np.int64).sum())
            self.cap1 = max(2 * n1, 4096)
            self.rec1 = dev.pinned_empty(self.cap1, FOCUS_DTYPE)
            self.counts1 = batch[0][10]
            self.acc1 = accept_io(self.cap1, len(self.owned) * self.n_templates, max(cfg1["max_dist"] // genome.binsize, 1), cfg1["kernels"][0])
        self.locks = tuple(locks)
        # The list.  Lane 0 (the calling thread): staging, the 2-D chain, its acceptance rules.  Lane 1: the 1-D chain in the
        # entry's asynchronous form on its worker context -- enqueued as soon as the staging is, finished (cs_detect_foci_batch_
        # finish) and accepted afterwards --, so the two chains share the device the way the two pattern threads of
        # detect_patterns make them, minus the interpreter.  Lane 2: the 2-D chain's prepare form.
        # The two chains wait for the same event (the staging) and would race for the workgroup slots: when the 1-D chain's run
        # kernel is resident first, the persistent tile workgroups start late on the CUs it holds (three of its workgroups fill
        # a CU's LDS) and -- their tile ranges are static -- finish late: a share's step took 0.75 instead of 0.6 ms, step by
        # step at random (profiles/r04b_step_modes.txt).  So the 1-D chain DEPENDS on the tile launch: cs_stream_wait_tiles at
        # its head lets it go when the last tile workgroup has started (round 4 held it back with a sleeping wave of a fixed
        # 45 us, a constant tuned on one genome and one box); it then runs in what the tile workgroups leave and still ends
        # before the 2-D chain's tail.
        calls = (CsCall * 10)()
        k = 0

        def nxt():
            nonlocal k
            k += 1
            return calls[k - 1]
        self.i_wait = self.keep_prep = self.params2 = None
        i_prep = i_stage = -1
        if cfg2 is not None:
            a = list(blocks[0])
            a[8], a[9] = self.rec2.ctypes.data, self.cap2
            if True:
                # ONE persistent launch for the tiles of all blocks (cs_foci_params.exclusive): with the 1-D chain held back
                # behind it the persistent workgroups are not displaced, and one launch beats 23 over three streams -- the
                # 23-block genome 3.25 -> 3.02 ms, a share of 2: 1.89 -> 1.69, of 4: 1.11 -> 0.96 (profiles/r04b_step_modes.txt).
                # Without the dependency the same launch was the slower choice (round 3: 5.6 against 4.4 ms).
                a[7][0].exclusive = 1
            # The 2-D chain's PREPARE form first (cs_foci_params.reserved & 2 on a copy of the parameter table): mask tables,
            # zeroed counters and both argument tables go to a side lane before the staging is even enqueued, so the tile launch
            # behind the staging waits for a lane that finished long ago instead of for events that fire after it reaches them.
            prep = list(a)
            prep_params = type(a[7])()
            C.memmove(prep_params, a[7], C.sizeof(a[7]))
            prep_params[0].reserved |= 2
            prep[7] = prep_params
            self.keep_prep, self.params2 = prep_params, a[7]
            # (on a lane of its own: its ~ 25 us of host work run beside lane 0's staging call instead of in front of it; the two
            # calls touch disjoint parts of the genome's context: staging scratch and tables on one side, the foci pool, the mask
            # tables and the template's weights on the other.  The full call waits for it: `after`)
            i_prep = k
            _fill(nxt(), *_SLOTS["cs_detect_foci_blocks"], prep, 2)
        i_stage = k
        _fill(nxt(), *_SLOTS["cs_stage_blocks"], stage[0], 0)
        if cfg1 is not None:
            i_ready = k
            _fill(nxt(), *_SLOTS["cs_event_record"], rec_ev[-1], 0)
            _fill(nxt(), *_SLOTS["cs_stream_wait_event"], waits[-1], 1, after=i_ready)
            if cfg2 is not None:
                self.i_wait = k
                _fill(nxt(), CALL_STREAM_WAIT_TILES, "pppii", (ctx_b, stream_b, ctx_a, 1, 1000), 1)     # (the epoch is set per run)
            b = list(batch[0])
            b[8], b[9] = self.rec1.ctypes.data, self.cap1
            b[7][0].reserved = 1                              # asynchronous form: return once the chain is enqueued
            _fill(nxt(), *_SLOTS["cs_detect_foci_batch_templates"], b, 1)
        if cfg2 is not None:
            _fill(nxt(), *_SLOTS["cs_detect_foci_blocks"], a, 0, after=i_prep)
            self._accept(nxt(), self.rec2, self.counts2, self.acc2, cfg2, 0)
        if cfg1 is not None:
            _fill(nxt(), CALL_DETECT_FOCI_BATCH_FINISH, "ppp", (ctx_b, stream_b, self.counts1), 1)
            self._accept(nxt(), self.rec1, self.counts1, self.acc1, cfg1, 1)
        self.calls, self.n_calls = calls, k
        self.lib = dev.lib
        self.ok = True
        self.why = ""

    @staticmethod
    def _accept(call, rec, counts, io, cfg, lane):
        km, kn = io["k"]
        values = (rec.ctypes.data, io["geo"].shape[1], counts, io["geo"][0].ctypes.data, io["geo"][1].ctypes.data,
                  io["geo"][2].ctypes.data, 0, int(km), int(kn), cfg["max_perc_undetected"] / 100, cfg["max_perc_zero"] / 100, 1,
                  3 if _device_pvalues() else 1,            # (flags: compact, records carry their p-values)
                  io["table"].ctypes.data, io["ok"].ctypes.data, io["kept"].ctypes.data)
        assert len(values) == 16
        _fill(call, CALL_ACCEPT_RECORDS, "pippppiiiddiippp", values, lane)

    def run(self):
        """One step: the records of every configuration of this rank, in the configurations' order and detect_genome's layout
        (block, bin1, bin2, score, pvalue, kernel_id, iteration) -- or None when a call reported an error."""
        import contextlib
        # one call in flight per context (engine._one_call_per_context): the list runs on the genome's context (lanes 0, 2) and
        # on the 1-D chain's worker context (lane 1); both are held for the call, always in the same order
        with contextlib.ExitStack() as held:
            for lock in self.locks:
                held.enter_context(lock)
            if self.params2 is not None:
                # this step's tile epoch: carried by the 2-D chain's parameter tables (both forms), awaited by the 1-D chain
                epoch = _next_epoch()
                if self.i_wait is not None:
                    self.calls[self.i_wait].i[0] = epoch
                for params in (self.keep_prep, self.params2):
                    params[0].reserved = (params[0].reserved & 0xff) | (epoch << 8)
            rc = self.lib.cs_run_calls(self.calls, self.n_calls)
        if rc != 0:
            if os.environ.get("CHROMOSIGHT_HIP_DEBUG"):
                import sys
                sys.stderr.write(f"[chromosight_amd] step plan failed: rc {[c.rc for c in self.calls]}\n")
            return None
        out = []
        cols = self.__dict__.get("_id_cols")
        if cols is None:
            # (block and template of every (template, block) slot of a result list: the same arrays every step)
            owned = np.asarray(self.owned, dtype=np.float64)
            cols = self._id_cols = {which: (np.tile(owned, n_t), np.repeat(np.arange(n_t, dtype=np.float64), len(self.owned)))
                                    for which, n_t in (("2", 1), ("1", self.n_templates))}
        for which in self.order:
            io = self.acc2 if which == "2" else self.acc1
            kept = io["kept"]
            total = int(kept.sum())
            rec = np.zeros((total, 7))
            blk, ker = cols[which]
            rec[:, 0] = np.repeat(blk, kept)
            rec[:, 1:5] = io["table"][:total]
            rec[:, 5] = np.repeat(ker, kept)
            out.append(rec)
        return out


def plannable(genome, kernel_configs, tsvd):
    """The configurations a StepPlan covers: a 2-D pattern with one square template (loops), a 1-D pattern with 1-4 templates of
    one square size (borders, hairpins), or one of each; single iterations, no truncated SVD, the device pipeline."""
    if tsvd is not None or not 1 <= len(kernel_configs) <= 2:
        return False
    if not (hasattr(genome, "view_for") and hasattr(genome, "dev") and hasattr(genome.dev, "pinned_empty")):
        return False
    cfg2, cfg1 = _split(kernel_configs)
    if (cfg2 is not None) + (cfg1 is not None) != len(kernel_configs):
        return False                                          # (two patterns of the same kind)
    for cfg in kernel_configs:
        if cfg["max_iterations"] != 1 or cfg["max_dist"] < 0:
            return False
        shapes = {np.shape(k) for k in cfg["kernels"]}
        if len(shapes) != 1 or next(iter(shapes))[0] != next(iter(shapes))[1]:
            return False
    if cfg2 is not None and len(cfg2["kernels"]) != 1:
        return False
    if cfg1 is not None and not 1 <= len(cfg1["kernels"]) <= 4:
        return False
    return True
