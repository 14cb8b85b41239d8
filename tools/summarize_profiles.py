"""Copy the summaries of a tools/collect_profiles.sh run from gpurun_out/<tag>/ into profiles/
(tracked): bench lines, per-kernel statistics, and the per-dispatch HBM counters of the C2 kernel
with the FETCH_SIZE calibration applied.    python tools/summarize_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


for name in ("c2", "c3", "c4p", "dense16384", "c2_f64", "c2_nosym"):
    f = os.path.join(src, f"bench_{name}.json")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_bench_{name}.json"))
for w in ("c2", "c3", "c4p"):
    f = find(f"trace_{w}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"))


def per_dispatch(path, kernel_substr):
    acc = collections.defaultdict(list)
    if not path:
        return {}
    for r in csv.DictReader(open(path)):
        if kernel_substr in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


summary = {"note": "per-dispatch means from rocprofv3 --pmc passes (one counter group per pass) on "
                   "`python bench.py --workload c2 --steps 5 --warmup 2`; FETCH_SIZE / WRITE_SIZE in KiB"}
calib = per_dispatch(find("calib/**/*counter_collection.csv"), "read4x4")
factor = None
if calib.get("FETCH_SIZE"):
    factor = 2097152.0 / calib["FETCH_SIZE"]
    summary["fetch_calibration"] = {
        "pattern": "4 x 4-byte global loads per lane over 2 GiB read once (tools/ubench/fetch_calib.hip)",
        "reported_KiB": calib["FETCH_SIZE"], "true_KiB": 2097152.0, "true_over_reported": factor}
c2 = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    c2.update(per_dispatch(find(f"pmc_{c}/**/*counter_collection.csv"), "corr_stream_kernel"))
c2.update(per_dispatch(find("pmc_sq/**/*counter_collection.csv"), "corr_stream_kernel"))
if c2:
    if "FETCH_SIZE" in c2 and "WRITE_SIZE" in c2:
        f = factor if factor else 2.0
        c2["hbm_bytes_per_dispatch"] = (c2["FETCH_SIZE"] * f + c2["WRITE_SIZE"]) * 1024.0
        c2["hbm_bytes_note"] = (f"FETCH_SIZE x {f:.3f} (calibration) + WRITE_SIZE (equals the output size exactly, "
                                "factor 1), KiB -> bytes")
    summary["c2_4096_f32"] = c2
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:1500])
