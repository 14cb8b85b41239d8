#!/usr/bin/env python3
"""pipeline.detect from several threads at once (each its own decoded .cool: example and yeast fixtures; the process-wide
Device, the worker pools and the staging scratch are shared): tables must equal the single-threaded ones.
python tools/stress_pipeline_threads.py [rounds]"""
import copy, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import pipeline

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cools = [dict(np.load(os.path.join(ROOT, "tests", "golden", f), allow_pickle=True)) for f in ("example_cool.npz", "yeast_cool.npz")]
jobs = [(cools[0], ck.loops), (cools[1], ck.borders), (cools[0], ck.hairpins), (cools[1], ck.loops)]
cols = ["bin1", "bin2", "score", "pvalue"]
want = [pipeline.detect(c, copy.deepcopy(cfg))[cols].to_numpy(dtype=np.float64) for c, cfg in jobs]
bad = []
def thread(k):
    for _ in range(rounds):
        got = pipeline.detect(jobs[k][0], copy.deepcopy(jobs[k][1]))[cols].to_numpy(dtype=np.float64)
        if got.shape != want[k].shape or not np.array_equal(got[:, :2], want[k][:, :2]) or np.abs(got[:, 2] - want[k][:, 2]).max() > 1e-9:
            bad.append(k)
            return
ths = [threading.Thread(target=thread, args=(k,)) for k in range(len(jobs))]
[t.start() for t in ths]; [t.join() for t in ths]
print("threads whose tables differed:", sorted(set(bad)) or "none", [w.shape[0] for w in want])
sys.exit(1 if bad else 0)
