import sys, copy, time, cProfile, pstats, os
sys.path.insert(0,'.')
import numpy as np
import chromosight_amd.kernels as ck
from chromosight_amd import pipeline
from tools.synthetic_genome import make_cool
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool,_ = make_cool(200_000, 1000, 2000, seed=2, template=template)
dcool = pipeline.DeviceCool(cool)
cfg = copy.deepcopy(ck.loops); cfg["max_dist"]=2_000_000
for _ in range(3): pipeline.detect(dcool, cfg, win_size=21)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): t = pipeline.detect(dcool, cfg, win_size=21)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
