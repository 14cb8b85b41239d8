#!/usr/bin/env python3
"""cProfile of the host side of the C4 genome step (main thread): python tools/prof_c4_host.py [n_gpus rank]"""
import copy, cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import genome_sizes, make_cool

world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sizes = genome_sizes(200_000)
costs = [parallel.block_cost((int(n), int(n)), 1000, False) for n in sizes]
mine = parallel.assign_blocks(costs, world)[rank]
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template, only=mine)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 2_000_000
borders = copy.deepcopy(ck.borders)

def step():
    staged = parallel.stage_genome(dcool, [loops, borders], owned=mine)
    return parallel.detect_patterns(dcool, [loops, borders], owned=mine, staged=staged)

for _ in range(5):
    step()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
