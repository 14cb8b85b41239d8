#!/usr/bin/env python3
"""Where a C5 step (pipeline.quantify from a resident pixel table) spends its time: wall time of the functions it calls,
summed over a step.  python tools/time_c5_phases.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, pandas as pd
from chromosight_amd import engine, pipeline
from chromosight_amd.utils import detection as cid

cool = dict(np.load(os.path.join(ROOT, "tests", "golden", "yeast_cool.npz"), allow_pickle=True))
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "yeast_quantify.npz"), allow_pickle=True))
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
dcool = pipeline.DeviceCool(cool)
acc = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        t0 = time.perf_counter()
        out = fn(*a, **k)
        acc[label or name] = acc.get(label or name, 0.0) + (time.perf_counter() - t0) * 1e3
        return out
    setattr(obj, name, wrap)


timed(pipeline.DeviceCool, "stage_blocks"); timed(pipeline.DeviceCool, "stage_inter"); timed(pipeline.DeviceCool, "bins_of")
timed(cid, "quantify_many_on_device"); timed(engine, "run_quantify_blocks"); timed(cid, "_accept_records")
timed(pipeline, "fdr_correction"); timed(pipeline, "sub_matrices")
for _ in range(3):
    pipeline.quantify(dcool, positions, cfg, inter=True, max_dist_bp=md)
acc.clear()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    pipeline.quantify(dcool, positions, cfg, inter=True, max_dist_bp=md)
total = (time.perf_counter() - t0) * 1e3 / n
print(f"step {total:.3f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {v / n:7.3f} ms")
print(f"  {'everything else':28s} {total - sum(v for k, v in acc.items() if k not in ('run_quantify_blocks', '_accept_records')) / n:7.3f} ms")
