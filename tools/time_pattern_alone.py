#!/usr/bin/env python3
"""ONE pattern of the C4 genome step alone on the device (no second chain beside it): the share of rank `rank` of `world`
is staged for loops + borders as the step does, the device is drained, then only pattern `which` (0 loops, 1 borders) is
detected -- its launch chain uncontended.
    python tools/time_pattern_alone.py [world] [rank] [which]"""
import copy, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import genome_sizes, make_cool

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 1
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sizes = genome_sizes(200_000)
costs = [parallel.block_cost((int(n), int(n)), 1000, False) for n in sizes]
mine = parallel.assign_blocks(costs, world)[rank]
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template, only=mine)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 2_000_000
borders = copy.deepcopy(ck.borders)
cfgs = [loops, borders]
ts = []
for it in range(14):
    staged = parallel.stage_genome(dcool, cfgs, owned=mine)
    dcool.dev.sync(); t0 = time.perf_counter()
    rec = parallel.detect_genome(dcool, cfgs[which], owned=mine, staged=staged.for_config(which), exchange=False, exclusive=False,
                                 own_context=which != 0)
    dcool.dev.sync(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{world} GPUs, rank {rank}, pattern {cfgs[which]['name']} alone: {np.mean(ts[4:]):.3f} ms (min {min(ts):.3f}); {len(rec)} records")
