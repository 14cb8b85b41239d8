import copy, cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import genome_sizes, make_cool
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template)
dcool = pipeline.DeviceCool(cool)
chroms = list(range(dcool.n_chrom))
for _ in range(5):
    b = dcool.stage_blocks(chroms, 1000, 17); dcool.dev.sync(); del b
ts = []
for _ in range(20):
    t0 = time.perf_counter(); b = dcool.stage_blocks(chroms, 1000, 17); ts.append(time.perf_counter() - t0); dcool.dev.sync(); del b
print("host time of stage_blocks: %.1f us (min %.1f)" % (np.mean(ts) * 1e6, min(ts) * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    b = dcool.stage_blocks(chroms, 1000, 17); dcool.dev.sync(); del b
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
