#!/usr/bin/env python3
"""Race hunt, second form: different sets of patterns scanned side by side (parallel.detect_patterns), each repetition
checked against one pattern after the other (parallel.detect_genome).  python tools/stress_pattern_sets.py [repetitions]"""
import copy, itertools, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(30_000, 300, 2000, seed=3, template=template)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 300 * 2000
near = copy.deepcopy(ck.loops); near["max_dist"] = 60 * 2000; near["name"] = "near loops"
pool = {"loops": loops, "near": near, "borders": copy.deepcopy(ck.borders), "hairpins": copy.deepcopy(ck.hairpins)}
truth = {k: parallel.detect_genome(dcool, v) for k, v in pool.items()}
print({k: v.shape[0] for k, v in truth.items()})
sets = [c for n in (2, 3, 4) for c in itertools.permutations(pool, n)][:: max(1, int(os.environ.get("CS_STRESS_STRIDE", "3")))]
for it in range(reps):
    names = sets[it % len(sets)]
    cfgs = [pool[k] for k in names]
    staged = parallel.stage_genome(dcool, cfgs)
    recs = parallel.detect_patterns(dcool, cfgs, staged=staged)
    for k, got in zip(names, recs):
        want = truth[k]
        assert got.shape == want.shape, (it, names, k, got.shape, want.shape)
        assert np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]]), (it, names, k)
        assert np.abs(got[:, 3] - want[:, 3]).max() < 1e-12, (it, names, k)
print(f"{reps} repetitions over {len(sets)} pattern sets equal one pattern after the other")
