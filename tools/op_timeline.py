#!/usr/bin/env python3
"""Device timeline of the LAST step of a rocprofv3 --kernel-trace --memory-copy-trace run: kernels AND copies (direction, bytes),
start relative to the step's first kernel, duration, the idle gap before each.
    python tools/op_timeline.py <dir> [marker kernel substring that starts a step]"""
import csv, glob, os, sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "stage_law_kernel"
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], "q" + r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
                     "COPY " + r.get("Direction", "").replace("MEMORY_COPY_", "") + " " + r.get("Bytes", r.get("Size", "?")) + " B", "dma"))
rows.sort()
starts = [i for i, r in enumerate(rows) if marker in r[2]]
if len(starts) < 2:
    sys.exit("marker kernel not found twice")
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0][0]
busy_end = t0
print(f"{len(step)} operations in the step, {(step[-1][1] - t0) / 1e3:.1f} us from the first start to the last end")
for s, e, name, q in step:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "")
    short = short.split("(")[0] if not short.startswith(("(", "COPY")) else short
    short = short.replace("cs::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim::")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {q:>4s}  gap {(s - busy_end) / 1e3:7.1f}  {short}")
    busy_end = max(busy_end, e)
