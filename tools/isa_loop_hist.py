#!/usr/bin/env python3
"""Histogram of the instructions inside the innermost hot loop of a gfx950 .s listing
(the loop that contains the most v_pk_fma_f32).  Usage: isa_loop_hist.py file.s"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
best = None
for i, l in enumerate(lines):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        lo, hi = labels[m.group(1)], i
        n = sum("v_pk_fma_f32" in x for x in lines[lo:hi])
        if best is None or n > best[0]:
            best = (n, lo, hi)
n, lo, hi = best
print(f"loop lines {lo}-{hi}, pk_fma={n}")
hist = collections.Counter()
for l in lines[lo:hi]:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if m and not l.strip().startswith(";"):
        hist[m.group(1)] += 1
cls = collections.Counter()
for k, v in hist.items():
    c = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_")) else "other"
    if k in ("s_nop", "s_waitcnt"): c = k
    if k.startswith("s_load") or k.startswith("s_buffer"): c = "smem"
    if k.startswith("s_cbranch") or k == "s_branch": c = "branch"
    cls[c] += v
print(dict(cls))
for k, v in hist.most_common(45):
    print(f"{v:5d} {k}")
