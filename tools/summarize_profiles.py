"""Copy the summaries of a tools/collect_profiles.sh run from gpurun_out/<tag>/ into profiles/
(tracked): bench lines, per-kernel statistics, micro-benchmark outputs, and the per-dispatch counters of
the dominant kernel of the dense (c2) and banded (c3k, c4p) workloads with the FETCH_SIZE calibration
applied, plus achieved GB/s of the helper kernels of the CSR path.    python tools/summarize_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def find(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


for name in ("c2", "c3", "c3k", "c4p", "c4", "c5", "dense16384", "c2_f64", "c2_nosym", "c2_stream", "c3k_stream", "c4p_stream"):
    f = os.path.join(src, f"bench_{name}.json")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_bench_{name}.json"))
for w in ("c2", "c3", "c3k", "c4p", "c4"):
    f = find(f"trace_{w}/**/*kernel_stats.csv")
    if f:
        # kernel names of the sort / scan library are hundreds of characters long: keep 160
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, f"{tag}_{w}_kernel_stats.csv"), "w", newline="") as out:
            wr = csv.writer(out)
            for r in rows:
                wr.writerow([r[0][:160]] + r[1:])
f = os.path.join(src, "template_kernels.txt")
if os.path.exists(f) and os.path.getsize(f):
    shutil.copy(f, os.path.join(dst, f"{tag}_template_kernels.txt"))
for u in ("coexec", "fma_rate", "write_rate", "mfma_rate"):
    f = os.path.join(src, f"ubench_{u}.txt")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_ubench_{u}.txt"))


def per_dispatch(path, kernel_substr):
    """Mean per dispatch of every counter of the kernels whose name contains (one of) kernel_substr."""
    subs = (kernel_substr,) if isinstance(kernel_substr, str) else tuple(kernel_substr)
    acc = collections.defaultdict(list)
    if not path:
        return {}
    for r in csv.DictReader(open(path)):
        if any(k in r["Kernel_Name"] for k in subs):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


CORR_KERNELS = ("corr_stream_kernel", "corr_mfma_dense_kernel")      # the dominant kernel of a correlation call


summary = {"note": "per-dispatch means from rocprofv3 --pmc passes (one counter group per pass) on "
                   "`python bench.py --workload <w> --steps 5 --warmup 2`; FETCH_SIZE / WRITE_SIZE in KiB"}
calib = per_dispatch(find("calib/**/*counter_collection.csv"), "read4x4")
factor = None
if calib.get("FETCH_SIZE"):
    factor = 2097152.0 / calib["FETCH_SIZE"]
    summary["fetch_calibration"] = {
        "pattern": "4 x 4-byte global loads per lane over 2 GiB read once (tools/ubench/fetch_calib.hip)",
        "reported_KiB": calib["FETCH_SIZE"], "true_KiB": 2097152.0, "true_over_reported": factor}
f = factor if factor else 2.0
for w, key in (("c2", "c2_4096_f32"), ("c3k", "c3k_band_50000x234"), ("c4p", "c4p_band_200000x1001")):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rec.update(per_dispatch(find(f"pmc_{w}_{c}/**/*counter_collection.csv"), CORR_KERNELS))
    rec.update(per_dispatch(find(f"pmc_{w}_sq/**/*counter_collection.csv"), CORR_KERNELS))
    if w in ("c2", "c4p"):
        rec.update(per_dispatch(find(f"pmc_{w}_mfma/**/*counter_collection.csv"), CORR_KERNELS))
        if rec.get("SQ_VALU_MFMA_BUSY_CYCLES") and rec.get("GRBM_GUI_ACTIVE"):
            # busy cycles are summed over the 1024 SIMDs, GUI_ACTIVE over the 8 XCDs
            rec["mfma_pipe_busy_frac"] = (rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (rec["GRBM_GUI_ACTIVE"] / 8.0)
    rec["kernel"] = "corr_mfma_dense_kernel (<VEC4, REG, RSYM> instance: see the kernel trace)"
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes_per_dispatch"] = (rec["FETCH_SIZE"] * f + rec["WRITE_SIZE"]) * 1024.0
        rec["hbm_bytes_note"] = (f"FETCH_SIZE x {f:.3f} (calibration) + WRITE_SIZE, KiB -> bytes")
    if rec:
        summary[key] = rec

# helper kernels of the CSR path (c3): bytes from the counters, time from the kernel trace
helpers = {}
stats = find("trace_c3/**/*kernel_stats.csv")
avg_ns = {}
if stats:
    for r in csv.DictReader(open(stats)):
        avg_ns[r["Name"]] = float(r["AverageNs"])
# (corr_mfma_dense_kernel: the tile kernel of the SAME profiled C3 step -- on a band of raw counts since round 5 -- so that the
# step's bytes come from one run instead of borrowing the c3k record)
for kname in ("stage_law_kernel", "stage_tile_kernel", "stage_finish_kernel", "distance_law_kernel", "csr_to_band_rows_kernel",
              "csr_band_extent_kernel", "law_finish_kernel", "mask_prep_kernel", "corr_mfma_dense_kernel"):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rec.update(per_dispatch(find(f"pmc_c3_{c}/**/*counter_collection.csv"), kname))
    t = [v for k, v in avg_ns.items() if kname in k]
    if t:
        rec["avg_us"] = t[0] / 1e3
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec and t:
        rec["hbm_bytes"] = (rec["FETCH_SIZE"] * f + rec["WRITE_SIZE"]) * 1024.0
        rec["achieved_GBps"] = rec["hbm_bytes"] / (t[0] * 1e-9) / 1e9
        rec["frac_of_8TBps"] = rec["achieved_GBps"] / 8000.0
    if rec:
        helpers[kname] = rec
if helpers:
    summary["c3_helper_kernels"] = helpers
# the candidate instances of the pipeline (C4 genome; a rank's share of 8)
for run, key, kernels in (("c4", "c4_genome_cand_instances", ("corr_mfma_dense_kernel",)),
                          ("share", "rank_share_8_blocks_kernel", ("corr_mfma_blocks_kernel",))):
    rec = per_dispatch(find(f"pmc_{run}_mfma/**/*counter_collection.csv"), kernels)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rec.update(per_dispatch(find(f"pmc_{run}_{c}/**/*counter_collection.csv"), kernels))
    if rec.get("SQ_VALU_MFMA_BUSY_CYCLES") and rec.get("GRBM_GUI_ACTIVE"):
        rec["mfma_pipe_busy_frac"] = (rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (rec["GRBM_GUI_ACTIVE"] / 8.0)
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes_per_dispatch"] = (rec["FETCH_SIZE"] * f + rec["WRITE_SIZE"]) * 1024.0
    if rec:
        rec["kernel"] = " / ".join(kernels) + " (candidate instances: per-dispatch means over the launches of the run)"
        summary[key] = rec
# the two-pass kernel of the templates with a side of 18 .. 33 (cs_corr_wide.hip), 21 x 21 on the dense 4096^2 map and on C4'
for w, key in (("dense", "wide_21x21_dense_4096"), ("c4p", "wide_21x21_c4p_band_200000x1001")):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "mfma"):
        rec.update(per_dispatch(find(f"pmc_wide_{w}_{c}/**/*counter_collection.csv"), "corr_mfma_wide_kernel"))
    if rec.get("SQ_VALU_MFMA_BUSY_CYCLES") and rec.get("GRBM_GUI_ACTIVE"):
        rec["mfma_pipe_busy_frac"] = (rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (rec["GRBM_GUI_ACTIVE"] / 8.0)
    if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        rec["hbm_bytes_per_dispatch"] = (rec["FETCH_SIZE"] * f + rec["WRITE_SIZE"]) * 1024.0
    fs = find(f"trace_wide_{w}/**/*kernel_stats.csv")
    if fs:
        rows = list(csv.reader(open(fs)))
        with open(os.path.join(dst, f"{tag}_wide_{w}_kernel_stats.csv"), "w", newline="") as out:
            wr = csv.writer(out)
            for r in rows:
                wr.writerow([r[0][:160]] + r[1:])
        for r in csv.DictReader(open(fs)):
            if "corr_mfma_wide_kernel" in r["Name"]:
                rec["avg_us_per_dispatch"] = float(r["AverageNs"]) / 1e3
                rec["dispatches"] = int(r["Calls"])
    log = os.path.join(src, f"trace_wide_{w}.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                try:
                    rec["profiled_run"] = json.loads(line)
                except ValueError:
                    pass
    if rec:
        rec["kernel"] = ("corr_mfma_wide_kernel<MASKED, TWO>; per-dispatch means -- a masked call on a wide band is TWO dispatches "
                         "(inner tiles, then the rim): a call moves twice the per-dispatch bytes")
        summary[key] = rec
# the shader clocks under load of every profiled bench command (a recomputed fraction can be compared with the
# driver's box)
clocks = {}
for w in ("c2", "c3", "c3k", "c4p", "c4"):
    log = os.path.join(src, f"trace_{w}.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                try:
                    d = json.loads(line)
                except ValueError:
                    continue
                if "gpu_state" in d:
                    clocks[w] = {"gpu_state": d["gpu_state"], "ms_per_step": d.get("ms_per_step"), "kernel_ms": d.get("kernel_ms")}
if clocks:
    summary["profiled_runs_clocks"] = clocks
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_counters.json"), "w"), indent=1)
for name in ("rank_share.txt", "rank_share_host_trace.txt", "rank_share_timeline.txt", "rank_share_native_laps.txt", "c4_phases.txt",
             "c5_phases.txt", "tile_kernel_sections.txt", "c4_modes.txt", "genome_timeline.txt", "genome_native_laps.txt",
             "borders_alone_timeline.txt", "wide_kernel_sections.txt"):
    f2 = os.path.join(src, name)
    if os.path.exists(f2) and os.path.getsize(f2):
        shutil.copy(f2, os.path.join(dst, f"{tag}_{name}"))
print(json.dumps(summary, indent=1)[:3000])
