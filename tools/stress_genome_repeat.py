#!/usr/bin/env python3
"""Race hunt: the sharded genome step (stage_genome + detect_patterns: host threads, lanes, priority streams) repeated;
every repetition must give the records of the first (coordinates exactly, scores to 1e-12: the laws are order-dependent
float64 sums).  python tools/stress_genome_repeat.py [repetitions] [total_bins] [max_dist_bins]"""
import copy, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
total = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
md = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(total, md, 2000, seed=2, template=template)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = md * 2000
cfgs = [loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)][:int(os.environ.get("CS_STRESS_CONFIGS", "3"))]
first = None
for it in range(reps):
    staged = parallel.stage_genome(dcool, cfgs)
    recs = parallel.detect_patterns(dcool, cfgs, staged=staged)
    if first is None:
        first = recs
        print("patterns:", [r.shape[0] for r in recs])
        continue
    for name, a, b in zip(("loops", "borders", "hairpins"), first, recs):
        if a.shape != b.shape:
            ka = {tuple(r) for r in a[:, [0, 1, 2, 5]].tolist()}
            kb = {tuple(r) for r in b[:, [0, 1, 2, 5]].tolist()}
            print("repetition", it, name, a.shape, b.shape, "only in first:", sorted(ka - kb)[:12], "only now:", sorted(kb - ka)[:12])
            again = parallel.detect_patterns(dcool, cfgs, staged=staged)
            print("same staged blocks scanned again:", [r.shape[0] for r in again])
        assert a.shape == b.shape, (it, name, a.shape, b.shape)
        assert np.array_equal(a[:, [0, 1, 2, 5, 6]], b[:, [0, 1, 2, 5, 6]]), (it, name)
        assert np.abs(a[:, 3] - b[:, 3]).max() < 1e-12, (it, name)
print(f"{reps} repetitions identical")
