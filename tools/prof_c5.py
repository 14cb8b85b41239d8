import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, pandas as pd
from chromosight_amd import pipeline
here = os.getcwd()
cool = dict(np.load(os.path.join(here, "tests", "golden", "yeast_cool.npz"), allow_pickle=True))
g = dict(np.load(os.path.join(here, "tests", "golden", "yeast_quantify.npz"), allow_pickle=True))
names = [str(n) for n in cool["chrom_names"]]
binsize = int(cool["binsize"])
rows = []
for bi in range(int(g["n_blocks"])):
    ca, cb = (int(x) for x in g[f"b{bi}_chroms"])
    for r, c in g[f"b{bi}_coords"]:
        rows.append((names[ca], int(r) * binsize, (int(r) + 1) * binsize, names[cb], int(c) * binsize, (int(c) + 1) * binsize))
positions = pd.DataFrame(rows, columns=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
           kernels=[g[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
md = int(g["cfg_max_dist_bp"])
cool = pipeline.DeviceCool(cool)          # resident pixel table: the upload is not part of a step
for _ in range(3):
    pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=md)
t0 = time.perf_counter()
for _ in range(10):
    pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=md)
print("ms per quantify", (time.perf_counter() - t0) * 100)
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    pipeline.quantify(cool, positions, cfg, inter=True, max_dist_bp=md)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
