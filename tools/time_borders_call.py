#!/usr/bin/env python3
"""Where one borders template on the 23-block genome spends its time: the native batch call against the Python around it."""
import copy, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import pipeline, engine
from chromosight_amd.utils import detection as cid
from tools.synthetic_genome import make_cool

template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template)
dcool = pipeline.DeviceCool(cool)
dev = dcool.dev
borders = copy.deepcopy(ck.borders)
blocks = dcool.stage_blocks(list(range(dcool.n_chrom)), 1, 17)
acc = {}
def timed(mod, name):
    fn = getattr(mod, name)
    def wrap(*a, **k):
        t0 = time.perf_counter(); out = fn(*a, **k); acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return out
    setattr(mod, name, wrap)
timed(engine, "run_detect_foci_batch"); timed(cid, "accept_many"); timed(cid, "_accept_records")
lib_fn = dev.lib.cs_detect_foci_batch
def native(*a):
    t0 = time.perf_counter(); rc = lib_fn(*a); acc.setdefault("native cs_detect_foci_batch", []).append((time.perf_counter() - t0) * 1e3); return rc
class L:  # proxy so that the timed native call is used
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, n): return native if n == "cs_detect_foci_batch" else getattr(self._lib, n)
dev.lib = L(dev.lib)
tot = []
for it in range(12):
    for kern in borders["kernels"]:
        dev.sync(); t0 = time.perf_counter()
        pipeline.detect_blocks(dcool, blocks, borders, kern, raw=True, want_windows=False)
        dev.sync(); tot.append((time.perf_counter() - t0) * 1e3)
print(f"detect_blocks (one borders template) {np.mean(tot[6:]):.3f} ms")
for k, v in acc.items():
    print(f"  {k:32s} {np.mean(v[6:]):.3f} ms x {len(v) // len(tot)}")
