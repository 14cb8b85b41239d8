#!/usr/bin/env python3
"""Leak hunt: the genome step repeated; host RSS and free device memory every 500 steps must level off.
python tools/stress_leaks.py [steps]"""
import copy, os, resource, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(30_000, 300, 2000, seed=2, template=template)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 300 * 2000
cfgs = [loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)]

def vram_used():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True, timeout=20).stdout
        return int(out.strip().splitlines()[-1].split(",")[2]) >> 20
    except Exception:
        return -1

for it in range(steps + 1):
    if it % 500 == 0:
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10
        print(f"step {it:5d}: max RSS {rss} MiB, VRAM used {vram_used()} MiB", flush=True)
    staged = parallel.stage_genome(dcool, cfgs)
    recs = parallel.detect_patterns(dcool, cfgs, staged=staged)
    if it % 7 == 0:
        pipeline.detect(dcool, ck.hairpins)
