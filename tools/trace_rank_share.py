#!/usr/bin/env python3
"""Timeline of one rank's step (tools/time_rank_share.py) across its host threads: when each native call and each numpy
stage starts and ends, relative to the start of the step.  python tools/trace_rank_share.py [n_gpus] [rank]"""
import copy, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import engine, parallel, pipeline
from chromosight_amd.utils import detection as cid
from tools.synthetic_genome import genome_sizes, make_cool

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sizes = genome_sizes(200_000)
costs = [parallel.block_cost((int(n), int(n)), 1000, False) for n in sizes]
mine = parallel.assign_blocks(costs, world)[rank]
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template, only=mine)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 2_000_000
borders = copy.deepcopy(ck.borders)
trace, T0 = [], [0.0]

def timed(mod, name, label=None):
    fn = getattr(mod, name)
    def wrap(*a, **k):
        t0 = time.perf_counter(); out = fn(*a, **k); t1 = time.perf_counter()
        trace.append((threading.current_thread().name[-12:], label or name, (t0 - T0[0]) * 1e3, (t1 - T0[0]) * 1e3)); return out
    setattr(mod, name, wrap)

timed(engine, "run_detect_foci_blocks"); timed(engine, "run_detect_foci_batch"); timed(cid, "accept_many")
timed(parallel, "stage_genome"); timed(parallel, "detect_genome"); timed(parallel, "_exchange_records")
timed(pipeline, "detect_blocks")
for it in range(8):
    dcool.dev.sync(); trace.clear(); T0[0] = time.perf_counter()
    staged = parallel.stage_genome(dcool, [loops, borders], owned=mine)
    rec = parallel.detect_patterns(dcool, [loops, borders], owned=mine, staged=staged)
    dcool.dev.sync(); total = (time.perf_counter() - T0[0]) * 1e3
print(f"step {total:.3f} ms")
for th, name, a, b in sorted(trace, key=lambda r: r[2]):
    print(f"  {th:>12s} {name:26s} {a:7.3f} -> {b:7.3f}  ({b - a:.3f})")
