#!/usr/bin/env python3
"""Every step of a rocprofv3 --kernel-trace run of tools/time_rank_share.py: span, and start / duration of the kernels named on
the command line -- to see what differs between a fast and a slow step.
    python tools/kernel_step_modes.py <dir with *_kernel_trace.csv> [kernel substring ...]"""
import csv, glob, os, sys

d = sys.argv[1]
names = sys.argv[2:] or ["corr_mfma_blocks", "cs_wait_tiles", "rescore_run_batch", "stage_tile_kernel", "segments_from_counts"]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "stage_law_kernel" in r[2]]
for a, b in zip(starts[:-1], starts[1:]):
    step = rows[a:b]
    t0 = step[0][0]
    out = [f"{len(step):3d} kernels {(max(r[1] for r in step) - t0) / 1e3:8.1f} us |"]
    for n in names:
        hit = [r for r in step if n in r[2]]
        out.append(f"{n[:14]} " + (f"@{(hit[0][0] - t0) / 1e3:7.1f} +{(hit[0][1] - hit[0][0]) / 1e3:7.1f}" if hit else "   -   ") + " |")
    print(" ".join(out))
