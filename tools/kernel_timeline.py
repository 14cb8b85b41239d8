#!/usr/bin/env python3
"""Device timeline of the LAST step of a rocprofv3 --kernel-trace run: every kernel with its start relative to the first
kernel of the step, its duration, queue and the idle gap before it.
    python tools/kernel_timeline.py <dir with *_kernel_trace.csv> [marker kernel substring that starts a step]"""
import csv, glob, os, sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "stage_law_kernel"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
if len(starts) < 2:
    sys.exit("marker kernel not found twice")
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0][0]
busy_end = t0
print(f"{len(step)} kernels in the step, {(step[-1][1] - t0) / 1e3:.1f} us from the first start to the last end")
for s, e, name, q in step:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0] if not short.startswith("(") else short
    short = short.replace("cs::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim::")[:70]
    gap = (s - busy_end) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q:>3s}  gap {gap:7.1f}  {short}")
    busy_end = max(busy_end, e)
