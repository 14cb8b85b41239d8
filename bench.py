#!/usr/bin/env python3
"""Benchmark of the hot path: Mpixels/s of normxcorr2 with the 17x17 loops template
(BASELINE.json metric).  One "step" = one pass of the device normxcorr2 over one synthetic
contact map already resident in HBM.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workloads:
  c2   (default) dense 4096x4096 float32, gamma(4, 0.25), seed 0; normxcorr2(full=False, no mask)
  c3   C3 as BASELINE.md defines it: N=50000 pixel table (CSR) resident in HBM, raw poisson(200/(d+1))
       counts on diagonals 0..250, 2 % unbalanced bins, seed 1; one step = band extents + distance law +
       fused detrend / CSR->band tiler + normxcorr2(full=True, sym_upper, mask, max_dist=233)
  c3k  the correlation call of c3 alone on a pre-tiled float32 band (kernel + mask tables)
  c4p  N=200000 single block, band to max_dist=1000 (+17), seed 2; same call as c3k
  c4   end to end, not kernel-only: 200000 bins in 23 blocks (hg38 proportions), max_dist=1000,
       planted loops; one step = block preparation (balance, distance law, detrend) + pattern_detector
       (correlation, thresholding, foci, validation) of every block, blocks sharded over the GPUs,
       records gathered with torch.distributed (strong scaling: the genome is fixed)

Multi-GPU: the path shards over independent sub-matrices (reference cli/chromosight.py:748-752),
so every rank processes its own map (weak scaling) with no data-path collective; the job
throughput is N * pixels * steps / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

# (before anything initialises HIP -- torch.distributed with N > 1, the library with N = 1: chromosight_amd/__init__.py says why)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PIXEL_17 = 714.0     # 2*k^2 + separable box sums + epilogue at 17x17
FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 vector = FP32 matrix (dense) peak
FP64_PEAK_TFLOPS = 78.6
F16_MFMA_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense float16 / bfloat16 MFMA peak
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)       # a C2 step is 0.11 ms: 300 steps = 33 ms of kernel time
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c3k", "c4p", "c4", "c5"])
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--size", type=int, default=4096, help="c2 only: side of the dense map (default 4096)")
    ap.add_argument("--plan", action="store_true",
                    help="no GPU work: print the block -> rank assignment of the sharded C4 genome for --gpus N and exit")
    return ap.parse_args()


def loops_kernel():
    import chromosight_amd.kernels as ck
    return ck.loops["kernels"][0]


class Workload:
    """Device-resident inputs/outputs of one rank plus the launch closure."""

    def __init__(self, name, dev, rank, precision, size=4096):
        from chromosight_amd import engine
        from chromosight_amd._lib import (CsMatrix, LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE,
                                          np_dtype_code)
        self.name = name
        self.dev = dev
        self.engine = engine
        self.kspec = engine.KernelSpec(loops_kernel())
        out_dtype = np.float64 if precision == "f64" else np.float32
        self.precision = precision
        if name == "c2":
            n = size
            rng = np.random.default_rng(0 + rank)
            self.host_sig = rng.gamma(4.0, 0.25, size=(n, n)).astype(np.float32)
            self.shape = (n, n)
            # device-resident maps with the engine's row pitch (never a power of two: HBM channel spread, engine.map_pitch)
            self.sig_buf, ld_in = engine.to_device_map(dev, self.host_sig)
            ld_out = engine.map_pitch(n, np.dtype(out_dtype).itemsize)
            self.sig = CsMatrix(self.sig_buf.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_in, 0, 0)
            self.out_buf = dev.empty((n, ld_out), out_dtype)
            self.out = CsMatrix(self.out_buf.ptr, np_dtype_code(out_dtype), LAYOUT_DENSE, ld_out, 0, 0)
            self.kwargs = dict(full=False, sym_upper=False, max_dist=None, mask_mode=MASK_NONE)
            self.pixels = n * n
            self.bytes_per_pixel = 4 + np.dtype(out_dtype).itemsize
            self.desc = (f"C2: dense {n}x{n} float32 gamma(4,0.25) seed 0, 17x17 loops template, "
                         f"normxcorr2(full=False, no mask); device-resident, row pitch {ld_in} / {ld_out} elements")
        elif name == "c3":
            # CSR in, detrend included (two passes over nnz * 8 B + 4 B per pixel written)
            from chromosight_amd import pipeline
            from tools.synthetic_genome import make_cool
            n, max_dist = 50_000, 233
            cool, _ = make_cool(n, max_dist, 2000, seed=1 + rank, loops_per_10k=0, chrom_sizes=[n])
            self.dcool = pipeline.DeviceCool(cool, dev)
            self.max_dist = max_dist
            self.shape = (n, n)
            out_w = max_dist + 1
            ld_out = (out_w + 63) // 64 * 64
            self.out_buf = dev.zeros((n, ld_out), out_dtype)
            self.out = CsMatrix(self.out_buf.ptr, np_dtype_code(out_dtype), LAYOUT_BAND, ld_out, 0, out_w)
            self.kwargs = dict(full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, missing_tol=0.5)
            self.pixels = n * out_w
            self.nnz = self.dcool.nnz
            # (SURVEY 8d prices the path with two passes over the pixel table; since round 5 the step makes one and moves a
            # float32 band of counts instead of the second -- the figure is kept: it is what BASELINE's C3 is priced against)
            self.bytes_per_pixel = (2 * self.nnz * 8 + 4 * (n + 1) * 2) / self.pixels + np.dtype(out_dtype).itemsize
            self.desc = (f"C3: N={n} pixel table (CSR, {self.nnz} stored pixels on diagonals 0..{max_dist + 17}, raw "
                         "poisson counts, 2% unbalanced bins) resident in HBM; per step: band extents + distance law "
                         "+ CSR->band of raw counts in the same pass + normxcorr2 with the detrend fused into the tile read (full=True, sym_upper, mask, "
                         f"max_dist={max_dist}, missing_tol=0.5), 17x17 loops template")
            self.host_sig = None
        else:
            from tools.synthetic_genome import band_workload
            band, band_w_in, miss, n, max_dist = band_workload("c3" if name == "c3k" else name, rank)
            keep = max_dist + 17
            ld_in = band.shape[1]
            self.shape = (n, n)
            self.sig_buf = dev.to_device(band)
            # (band_workload's rows are zero behind their stored diagonals and beyond the matrix: CS_LAYOUT_BAND_PADDED, what the
            # library's own staging pass hands the same call)
            from chromosight_amd._lib import LAYOUT_BAND_PADDED
            padded = ld_in >= band_w_in + 4
            self.sig = CsMatrix(self.sig_buf.ptr, np_dtype_code(np.float32), LAYOUT_BAND_PADDED if padded else LAYOUT_BAND, ld_in, 0, band_w_in)
            out_w = max_dist + 1
            ld_out = (out_w + 63) // 64 * 64
            self.out_buf = dev.zeros((n, ld_out), out_dtype)
            self.out = CsMatrix(self.out_buf.ptr, np_dtype_code(out_dtype), LAYOUT_BAND, ld_out, 0, out_w)
            self.miss = dev.to_device(miss)
            self.kwargs = dict(full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS,
                               miss_row=self.miss, miss_col=self.miss, missing_tol=0.5)
            self.pixels = n * out_w
            self.bytes_per_pixel = 4.0 * band_w_in / out_w + np.dtype(out_dtype).itemsize
            self.desc = (f"{name.upper()}: N={n} upper band (diagonals 0..{keep}) float32, 2% missing bins, "
                         f"17x17 loops template, normxcorr2(full=True, sym_upper, mask, max_dist={max_dist}, "
                         "missing_tol=0.5), band layout in HBM")
            self.host_sig = None

    def step(self):
        if self.name == "c3":
            # (the band of raw counts, detrended by the tile kernel as it splits a landed tile; CS_BENCH_C3_DETRENDED=1: the
            # detrended band of the tiler pass, for A/B runs)
            block = self.dcool.stage_blocks([0], self.max_dist, 17, band_dtype=np.float32,
                                            counts=not (os.environ.get("CS_BENCH_C3_DETRENDED") or os.environ.get("CHROMOSIGHT_HIP_NO_COUNTS_BAND")))[0]
            self.engine.run_normxcorr2(self.dev, block.sig, self.shape, self.kspec, self.out, precision=self.precision,
                                       miss_row=block.miss_row, miss_col=block.miss_col, **self.kwargs)
            return
        self.engine.run_normxcorr2(self.dev, self.sig, self.shape, self.kspec, self.out,
                                   precision=self.precision, **self.kwargs)


# cs_last_kernel() ids (include/chromosight_hip.h) -> what served the launch
KERNELS = {
    0: ("none", ""),
    1: ("corr_generic_kernel", "runtime-size LDS-tiled kernel, float32 / float64 FMA"),
    2: ("corr_stream_kernel", "packed FP32 FMA streaming kernel (v_pk_fma_f32); the loops template is vertically symmetric "
                              "and shares each row product between two template rows (169 instead of 289 packed products "
                              "per column pair)"),
    3: ("corr_mfma_kernel", "first general matrix-core kernel (masks as a 0/1 plane), float16 head/tail operands"),
    4: ("corr_mfma_dense_kernel<VEC4,false,false>",
        "v_mfma_f32_16x16x32_f16: signal and template split into float16 head + tail (3 products per term, float32 "
        "accumulation = float32-equivalent coefficients, parity 1e-5), template rows as 32x16 Toeplitz operands (17 of 32 k "
        "useful), box sums by two separable MFMA passes: 252 MFMAs per 16x64 output pixels = 4032 executed flop/pixel for "
        "714 algorithmic"),
    5: ("corr_mfma_dense_kernel<true,true,RSYM>",
        "masked matrix-core tile kernel: the same float16-pair contraction, per-bin missing masks factorised into per-row / "
        "per-column template sums + correction records (mask_prep_kernel), cross term on the matrix cores"),
    6: ("corr_sep_kernel", "separable evaluation of exactly rank-1 templates"),
    7: ("corr_mfma_wide_kernel<MASKED,TWO>",
        "two-pass matrix-core kernel for templates with a side of 18 .. 33 (cs_corr_wide.hip): two k = 32 Toeplitz passes per template "
        "row, weight fragments from L2, per-bin masks factorised per tile (1-D tables + a V table for flagged row x flagged column)"),
}
MFMA_KERNELS = (3, 4, 5)


def prewarm(step, sync, seconds=0.1):
    """Untimed: run the step until the clocks have ramped (a 0.1 ms step timed right after an idle period runs
    5-10 % slow: at 20 timed steps the ramp would be inside the timed region).  Short on purpose: the ramp takes tens of
    milliseconds, and the same 20 steps after 2 s of this run 2-3 % slower than after 0.05 s (0.1059 / 0.1043 / 0.1021 ms
    after 2.0 / 0.4 / 0.05 s on one box; 300 timed steps: 0.1022)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(16):
            step()
        sync()


def time_steps(dev, step, sync, steps, warmup, dist=None, local_rank=0):
    """W untimed warm-up steps, then exactly `steps` steps between two HIP events on the launch stream and two full
    synchronisations (+ barriers with N > 1).  Returns (wall seconds, kernel ms per step)."""
    for _ in range(warmup):
        step()
    sync()
    ev0, ev1 = dev.new_event(), dev.new_event()
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    sync()
    t0 = time.perf_counter()
    dev.record(ev0)
    for _ in range(steps):
        step()
    dev.record(ev1)
    sync()
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
    elapsed = time.perf_counter() - t0
    return elapsed, dev.elapsed_ms(ev0, ev1) / steps


def wide_template(k):
    """A full-rank k x k template (a blob on a gradient plus seeded noise): what `--win-size k` makes of the loops template in
    spirit -- the same generator as tools/time_wide.py and tests/test_gpu_wide.py."""
    rng = np.random.default_rng(1000 * k)
    i, j = np.indices((k, k))
    c = (k - 1) / 2
    return 0.4 + np.exp(-((i - c) ** 2 + (j - c) ** 2) / (0.08 * k * k + 1)) + 0.02 * (i - j) + 0.15 * rng.normal(size=(k, k))


def time_wide_templates(dev, wl, sync, sizes=(21, 33), steps=10):
    """The resident workload `wl` under templates with a side above 17 (`--win-size`, reference cli/chromosight.py:365-370,
    preprocessing.py:731-807): kernel time by HIP events, the kernel that served it, and the fraction of the FP32 roof at the
    template's own 2 k^2 + 8 k flop per pixel (SURVEY 8d's count at k)."""
    out = {}
    keep = wl.kspec
    try:
        for k in sizes:
            wl.kspec = wl.engine.KernelSpec(wide_template(k))
            _, ms = time_steps(dev, wl.step, sync, steps, 3)
            kid = int(dev.lib.cs_last_kernel(dev.ctx))
            flop = 2 * k * k + 8 * k
            tf = flop * wl.pixels / (ms * 1e-3) / 1e12
            out[f"{k}x{k}"] = {"kernel_ms": round(ms, 4), "gpixel_per_s": round(wl.pixels / ms / 1e6, 1), "kernel": KERNELS.get(kid, ("?", ""))[0],
                               "kernel_id": kid, "flop_per_pixel": flop, "achieved_tflops": round(tf, 1),
                               "frac_fp32_roof": round(tf / FP32_PEAK_TFLOPS, 4)}
    finally:
        wl.kspec = keep
    return out


def gpu_clocks():
    """Shader clocks (MHz) of the node's GPUs from sysfs, in file order; [] where sysfs does not show them."""
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")):
        try:
            out.append(int(open(f).read()) // 1000000)
        except (OSError, ValueError):
            out.append(-1)
    return out


def gpu_state(step, sync, idle0=None, seconds=0.25):
    """What the box-to-box spread of the headline step correlates with: the shader clock of every GPU of the node (sysfs
    hwmon freq1_input) sampled idle and then during `seconds` of back-to-back steps, AFTER the timed region.  The GPU whose
    clock rises is this process's; the others that sit near their top clock are other tenants' work on the same node
    (shared power and cooling).  None where sysfs does not show the clocks."""
    import threading
    files = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
    if not files:
        return None

    def read_all():
        out = []
        for f in files:
            try:
                out.append(int(open(f).read()) // 1000000)
            except (OSError, ValueError):
                out.append(-1)
        return out

    try:
        sync()
        # (idle0: the clocks when the process started, before it touched its GPU -- a GPU that was busy then is another tenant's)
        idle = idle0 if idle0 and len(idle0) == len(files) else read_all()
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                samples.append(read_all())
                time.sleep(0.01)

        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(16):
                step()
            sync()
            n += 16
        dt = time.perf_counter() - t0
        stop.set()
        th.join(1.0)
        if not samples:
            return None
        busy = [sorted(s[k] for s in samples)[len(samples) // 2] for k in range(len(files))]
        mine = max(range(len(files)), key=lambda k: busy[k] - idle[k])
        return {"sclk_mhz_under_load": busy[mine], "sclk_mhz_min_under_load": min(s[mine] for s in samples), "sclk_mhz_idle": idle[mine],
                "gpus_on_node": len(files), "other_gpus_busy": sum(1 for k in range(len(files)) if k != mine and idle[k] > 1000),
                "ms_per_step_during_probe": round(dt / n * 1e3, 4),
                "note": "sampled after the timed region during %.2f s of back-to-back steps (host-paced: every 16 steps "
                        "synchronised); idle = when this process started; other_gpus_busy = GPUs of the node that were above "
                        "1 GHz then (other tenants: the node's power and cooling are shared)" % seconds}
    except Exception as e:                                   # (never part of a measurement)
        return {"error": repr(e)}


def roofline_of(wl, kernel_ms, kernel_id, precision, traffic=None, traffic_note=None):
    """`roofline` object of one workload: algorithmic flops (714 per pixel at 17 x 17) per launch over the
    launch's average duration, against the FP32 peak -- the roof of a float32-exact evaluation, vector FMA and f32-input
    MFMA alike -- plus, when a matrix-core kernel served it, the flops it EXECUTES on float16 operands against the dense
    float16 MFMA peak, and the bytes view."""
    name, how = KERNELS.get(kernel_id, ("?", ""))
    flop = FLOP_PER_PIXEL_17 * wl.pixels
    achieved_tf = flop / (kernel_ms * 1e-3) / 1e12
    peak_tf = FP64_PEAK_TFLOPS if precision == "f64" else FP32_PEAK_TFLOPS
    achieved_gbs = wl.bytes_per_pixel * wl.pixels / (kernel_ms * 1e-3) / 1e9
    roof = {
        "bound": "mfma", "achieved": round(achieved_tf, 2), "peak": peak_tf, "unit": "TFLOP/s",
        "frac": round(achieved_tf / peak_tf, 4),
        # PMC counters cannot be read from inside the timed process: `traffic` (HBM bytes of THIS run's launches) stays null, and the
        # figure of the committed rocprofv3 --pmc passes of the same command on the builder's box sits beside it under its own name
        "traffic": None, "traffic_profiled": traffic, "traffic_note": traffic_note,
        "kernel": name, "kernel_id": kernel_id,
        "peak_kind": ("FP32: dense f32 MFMA = vector FMA peak, the roof of a float32-exact evaluation of 714 flop/pixel; "
                      "the flops actually executed are priced under `executed`" if precision == "f32" else "FP64 vector peak"),
        "note": f"714 flop/pixel (SURVEY 8d) x {wl.pixels} pixels per launch / {kernel_ms:.4f} ms (HIP events on the launch "
                f"stream); the arithmetic roof binds before HBM at 17x17 (89 flop/B).  Served by {name}: {how}",
        "hbm": {"achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved_gbs / HBM_PEAK_GBS, 4), "bytes_per_pixel": round(wl.bytes_per_pixel, 2)},
    }
    if kernel_id in MFMA_KERNELS and precision == "f32":
        executed_tf = achieved_tf * 4032.0 / FLOP_PER_PIXEL_17
        roof["executed"] = {"pipe": "mfma_f16", "achieved": round(executed_tf, 1), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(executed_tf / F16_MFMA_PEAK_TFLOPS, 4), "flop_per_pixel": 4032}
    return roof


def cpu_baseline_band(budget_s=8.0, rows=8192):
    """CPU baseline of the north-star workload (C4': banded, masked, full mode): oracle/oracle.c's band entry point on a
    block of `rows` bins from the same generator (same band, same 2 % missing bins), all host cores, and one pass of a
    smaller window on ONE core -- the reference parallelises over sub-matrices only (cli/chromosight.py:748-752), so on
    a single 200 000-bin block it has one core whatever --threads says."""
    from oracle import c_oracle
    from tools.synthetic_genome import band_workload
    band, band_w, miss, n, max_dist = band_workload("c4p", 0, n=rows)
    band = band.astype(np.float64)
    out_w = max_dist + 1
    threads = c_oracle.max_threads()
    kern = loops_kernel()
    args = dict(max_dist=max_dist, sym_upper=True, full=True, miss_row=miss, miss_col=miss, missing_tol=0.5)
    c_oracle.normxcorr2_band(band, n, 0, band_w, kern, 0, 64, 0, out_w, n_threads=threads, **args)
    passes, t0 = 0, time.perf_counter()
    while True:
        c_oracle.normxcorr2_band(band, n, 0, band_w, kern, 0, n, 0, out_w, n_threads=threads, **args)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 100:
            break
    t1 = time.perf_counter()
    one_rows = 256
    c_oracle.normxcorr2_band(band, n, 0, band_w, kern, 1024, 1024 + one_rows, 0, out_w, n_threads=1, **args)
    dt1 = time.perf_counter() - t1
    return {
        "value": round(passes * n * out_w / dt / 1e6, 3), "unit": "Mpixel/s", "cores": threads, "kind": "port",
        "single_core": {"value": round(one_rows * out_w / dt1 / 1e6, 3), "unit": "Mpixel/s", "cores": 1,
                        "note": "what a single sub-matrix can use in the reference (one process per sub-matrix)"},
        "reference_python": {"value": 0.27, "unit": "Mpixel/s", "cores": 1,
                             "source": "BASELINE.md section 3: the reference's banded full-mode normxcorr2 on the survey "
                                       "host (0.25-0.29 Mpixel/s), not re-run here: the reference cannot travel"},
        "sample": f"{passes} passes over a {n}-bin block of the C4' generator (band 0..{max_dist + 17}, 2 % missing bins, "
                  f"{n * out_w / 1e6:.1f} Mpixel each), float64 C restatement oracle/oracle.c (band entry), OpenMP x{threads}, "
                  f"{dt:.1f} s; single core: {one_rows} rows, {dt1:.1f} s",
    }


def cpu_baseline(workload, budget_s=12.0):
    """The C restatement of the oracle ("port", oracle/oracle.c with OpenMP on all host cores)
    timed on the same map: whole passes over the first rows until ~budget_s of wall time."""
    from oracle import c_oracle
    if workload.host_sig is None:
        return None
    rows = min(workload.host_sig.shape[0], 2048 + 16)
    sample = workload.host_sig[:rows].astype(np.float64)
    threads = c_oracle.max_threads()
    c_oracle.normxcorr2(sample[:64], loops_kernel(), n_threads=threads)  # load + warm the library
    passes, t0 = 0, time.perf_counter()
    while True:
        c_oracle.normxcorr2(sample, loops_kernel(), n_threads=threads)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 200:
            break
    return {
        "value": round(passes * sample.size / dt / 1e6, 3), "unit": "Mpixel/s", "cores": threads, "kind": "port",
        "reference_python": {"value": 1.94, "unit": "Mpixel/s", "cores": 1,
                             "source": "BASELINE.md section 3: the reference's own normxcorr2 (scipy), dense 4096^2, "
                                       "17x17 loops template, survey host (Xeon 2.1 GHz), not re-run here: the "
                                       "reference cannot travel to the GPU box"},
        "sample": f"{passes} passes over the first {rows} rows of the same map ({sample.size / 1e6:.2f} Mpixel "
                  f"each), float64 C restatement oracle/oracle.c, OpenMP x{threads}, {dt:.1f} s",
    }


def api_call_ms(workload):
    """PCIe-inclusive cost of the reference-shaped Python call on the same map (host ndarray in, host
    float64 ndarray out): never `value`, reported next to it.  The source is a plain (PAGEABLE) numpy array -- what a
    caller of chromosight.utils.detection.normxcorr2 holds; the result lands in the library's page-locked pool."""
    if workload.host_sig is None:
        return None
    from chromosight_amd.utils import detection as cud
    k = loops_kernel()
    cud.normxcorr2(workload.host_sig, k)
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        out, _ = cud.normxcorr2(workload.host_sig, k)
        best = min(best, time.perf_counter() - t0)
        del out
    return round(best * 1e3, 3)


def detect_wallclock():
    """Second half of the BASELINE.json metric: `detect` wall-clock (loops defaults) on the
    reference's test map, data_test/example.cool (decoded fixture tests/golden/example_cool.npz),
    whole pipeline: block assembly, detrend, correlation, foci, validation, post-filters."""
    import copy
    import chromosight_amd.kernels as ck
    from chromosight_amd import pipeline
    path = os.path.join(ROOT, "tests", "golden", "example_cool.npz")
    if not os.path.exists(path):
        return None
    cool = dict(np.load(path))
    pipeline.detect(cool, copy.deepcopy(ck.loops))          # warm-up (module imports, weight upload)
    t0 = time.perf_counter()
    table = pipeline.detect(cool, copy.deepcopy(ck.loops))
    dt = time.perf_counter() - t0
    return {"seconds": round(dt, 4), "patterns": int(len(table)), "expected_patterns": 89,
            "input": "data_test/example.cool (720 bins, 3 chromosomes), loops defaults, 1 GPU"}


def pmc_traffic(args, wl):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_counters.json, written by tools/summarize_profiles.py: FETCH_SIZE with its
    measured calibration + WRITE_SIZE).  Counters cannot be read from inside the timed process, so
    this is the figure of the profiled run of the same command; None when no such record exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.json")))
    wl = wl if isinstance(wl, str) else args.workload           # (callers pass the workload's name or the Workload itself)
    if wl == "c3" and args.precision == "f32":
        # the whole step: law pass + finish + tiler + mask tables (c3_helper_kernels) + the tile kernel (its c3k record)
        for path in reversed(files):
            try:
                rec = json.load(open(path))
            except (OSError, ValueError):
                continue
            helpers, tile = rec.get("c3_helper_kernels", {}), rec.get("c3k_band_50000x234", {})
            own = helpers.get("corr_mfma_dense_kernel", {}).get("hbm_bytes")
            if own is not None and "stage_tile_kernel" not in helpers:
                # round 5: the law pass writes the band of raw counts, no tiler; every kernel of the profiled C3 step itself
                parts = [helpers.get(k, {}).get("hbm_bytes") for k in ("stage_law_kernel", "stage_finish_kernel", "mask_prep_kernel")] + [own]
                if all(p is not None for p in parts):
                    return int(sum(parts)), (f"{os.path.basename(path)}: sum over the kernels of the profiled C3 step (law pass writing the band of "
                                             "counts, finish, mask tables, tile kernel) of FETCH_SIZE x calibration + WRITE_SIZE")
            parts = [helpers.get(k, {}).get("hbm_bytes") for k in ("stage_law_kernel", "stage_finish_kernel", "stage_tile_kernel",
                                                                   "mask_prep_kernel")] + [tile.get("hbm_bytes_per_dispatch")]
            if all(p is not None for p in parts):
                return int(sum(parts)), (f"{os.path.basename(path)}: sum over the step's kernels (law pass, finish, tiler, mask tables: "
                                         "c3_helper_kernels; tile kernel: c3k record) of FETCH_SIZE x calibration + WRITE_SIZE")
        return None, "no PMC record"
    key = {"c2": "c2_4096_f32", "c3k": "c3k_band_50000x234", "c4p": "c4p_band_200000x1001"}.get(wl)
    if key is None or (wl == "c2" and args.size not in (None, 4096)) or args.precision != "f32":
        return None, "no PMC record for this workload"
    for path in reversed(files):
        try:
            rec = json.load(open(path)).get(key, {})
        except (OSError, ValueError):
            continue
        if "hbm_bytes_per_dispatch" in rec:
            return int(rec["hbm_bytes_per_dispatch"]), f"{os.path.basename(path)}: {rec.get('hbm_bytes_note', '')}"
    return None, "no PMC record"


class GenomeC4:
    """C4 of BASELINE.md: 200 000 bins in 23 blocks (hg38 proportions), max_dist 1000 bins, planted
    loops, pixel table resident in HBM (DeviceCool: the rank's own chromosomes only).  One step =
    for loops (1 template, band to 1000 bins) and borders (3 templates, 1-D): stage every owned block
    (band extents, distance law, detrend + tiler), correlation, device foci / validation statistics,
    all-gather of the records -- everything `chromosight detect` does between reading the .cool and
    the final table."""

    def __init__(self, args, rank, local_rank, world, total_bins=200_000):
        import copy
        import chromosight_amd
        import chromosight_amd.kernels as ck
        from chromosight_amd import parallel, pipeline
        from chromosight_amd._lib import get_device
        from tools.synthetic_genome import genome_sizes, make_cool
        chromosight_amd.set_precision(args.precision)
        self.parallel = parallel
        self.binsize, self.max_dist = 2000, 1000
        template = np.asarray(loops_kernel(), dtype=np.float64)
        sizes = genome_sizes(total_bins)
        costs = [parallel.block_cost((int(n), int(n)), self.max_dist, False) for n in sizes]
        self.mine = parallel.assign_blocks(costs, world)[rank]
        t0 = time.perf_counter()
        cool, self.planted = make_cool(total_bins, self.max_dist, self.binsize, seed=2, template=template, only=self.mine)
        self.gen_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.dcool = pipeline.DeviceCool(cool, get_device(local_rank))
        self.dcool.dev.sync()
        self.upload_s = time.perf_counter() - t0
        self.stored_pixels = int(cool["count"].size)
        self.loops = copy.deepcopy(ck.loops)
        self.loops["max_dist"] = self.max_dist * self.binsize
        self.borders = copy.deepcopy(ck.borders)
        self.sizes = sizes
        self.n_chrom = len(sizes)
        self.loop_pixels = int(sum(int(n) * (min(self.max_dist, int(n) - 1) + 1) for n in sizes))
        self.border_pixels = int(sum(3 * int(n) * 2 for n in sizes))

    def step(self):
        # every owned block is staged ONCE per step, at the loops' keep distance; the borders templates scan band views of
        # the same blocks (the law of a diagonal does not depend on how many diagonals are kept: identical values)
        # and the two patterns are scanned side by side (parallel.detect_patterns: the 1-D templates' latency-bound chains
        # run under the loops template's tile kernels)
        # (parallel.genome_step = stage_genome + detect_patterns; after the first step of a layout: one native call that
        # replays the step's calls on the same buffers, chromosight_amd/plan.py)
        rec_l, rec_b = self.parallel.genome_step(self.dcool, [self.loops, self.borders], owned=self.mine)
        return rec_l, rec_b


def time_genome(args, rank, local_rank, world, dist, torch, steps, warmup):
    """Time `steps` passes of the sharded C4 genome; returns the summary dict (same on every rank)."""
    g = GenomeC4(args, rank, local_rank, world)

    def sync():
        g.dcool.dev.sync()
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
            torch.cuda.synchronize(local_rank)

    # which transport carries the exchanges, checked against torch.distributed before anything is timed (the first N > 1
    # run of the library's own RCCL communicator happens on the driver's node: it must be diagnosable, not fragile)
    transport = g.parallel.exchange_self_check() if dist is not None else "none"
    rec = None
    for _ in range(max(warmup, 1)):
        rec = g.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        rec = g.step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the same steps one by one (each bracketed by the synchronisation of the timed region), outside the timed region: median, minimum
    # and the list beside the mean -- a step that ends on host threads has modes (profiles/r05_genome_step_modes.txt), and the mean of
    # eight steps does not say which one a box was in
    singles = []
    for _ in range(min(max(steps, 4), 16)):
        sync()
        t1 = time.perf_counter()
        rec = g.step()
        sync()
        singles.append((time.perf_counter() - t1) * 1e3)
    if dist is not None:
        t = torch.tensor(singles, dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        singles = [float(x) for x in t.tolist()]
    # per-rank phases, outside the timed region (a device synchronisation between staging and detection would cost the
    # timed steps their overlap): staging until its kernels are done, detection without the exchanges, the exchanges
    phases = []
    for _ in range(3):
        g.dcool.dev.sync()
        g.parallel.TIMERS.update(exchange_ms=0.0, exchanges=0)
        ta = time.perf_counter()
        staged = g.parallel.stage_genome(g.dcool, [g.loops, g.borders], owned=g.mine)
        g.dcool.dev.sync()
        tb = time.perf_counter()
        g.parallel.detect_patterns(g.dcool, [g.loops, g.borders], owned=g.mine, staged=staged)
        g.dcool.dev.sync()
        tc = time.perf_counter()
        ex = g.parallel.TIMERS["exchange_ms"]
        phases.append(((tb - ta) * 1e3, (tc - tb) * 1e3 - ex, ex))
    mine = {"rank": rank, "blocks": len(g.mine), "stage_ms": round(min(p[0] for p in phases), 3),
            "detect_ms": round(min(p[1] for p in phases), 3), "exchange_ms": round(min(p[2] for p in phases), 3)}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    pixels = g.loop_pixels + g.border_pixels
    detect_c4 = None
    if world == 1:
        # `chromosight detect` END TO END at genome scale (BASELINE.json's "detect wall-clock"), one pattern per run like the CLI:
        # resident pixel table -> final table (staging, correlation, foci, acceptance rules, neighbour removal, min_dist,
        # Benjamini-Hochberg q-values: pipeline.detect, the counterpart of cli/chromosight.py:601-878 without the file I/O)
        from chromosight_amd import pipeline
        detect_c4 = {}
        import copy as _copy
        import chromosight_amd.kernels as _ck
        for name, cfg in (("loops", g.loops), ("borders", g.borders), ("hairpins", _copy.deepcopy(_ck.hairpins))):
            for _ in range(2):
                table = pipeline.detect(g.dcool, cfg)
            times = []
            for _ in range(7):
                g.dcool.dev.sync()
                t0 = time.perf_counter()
                table = pipeline.detect(g.dcool, cfg)
                times.append((time.perf_counter() - t0) * 1e3)
            detect_c4[name] = {"ms": round(float(np.median(times)), 3), "min_ms": round(min(times), 3), "rows": int(len(table)),
                               "templates": len(cfg["kernels"])}
        detect_c4["input"] = ("the C4 genome (200 000 bins, 23 chromosomes), pixel table resident in HBM; median of 7 runs of "
                              "pipeline.detect per pattern, final table included")
    return {
        "detect_wallclock_c4": detect_c4,
        "transport": transport, "per_rank": per_rank,
        "value": round(pixels * steps / elapsed / 1e6, 1), "unit": "Mpixel/s", "ms_per_genome": round(elapsed / steps * 1e3, 2),
        "single_steps_ms": {"median": round(float(np.median(singles)), 3), "min": round(min(singles), 3), "max": round(max(singles), 3),
                            "steps": [round(x, 3) for x in singles],
                            "note": "the same step timed one at a time (synchronised on both sides, max over ranks), outside the timed region"},
        "n_gpus": world, "steps": steps, "scaling": "strong",
        "workload": "C4: 200000 bins in 23 blocks (hg38 proportions), 2 kb bins, max_dist 1000 bins, 2 % unbalanced bins, "
                    "planted loops, pixel table resident in HBM; per step and per pattern (loops, borders x3): band "
                    "extents + distance law + detrend/tiler of every block, correlation, device foci + validation "
                    "statistics, all-gather of the records (RCCL); blocks sharded over the GPUs (LPT)",
        "correlation_pixels_per_step": pixels, "loop_pixels": g.loop_pixels, "border_pixels": g.border_pixels,
        "stored_pixels_rank0": g.stored_pixels, "blocks": g.n_chrom,
        "patterns": {"loops": int(rec[0].shape[0]), "borders": int(rec[1].shape[0])},
        "setup": {"generate_s": round(g.gen_s, 2), "upload_s": round(g.upload_s, 3),
                  "upload_bytes_rank0": int(g.dcool.upload_bytes)},
    }


def run_c4(args, rank, local_rank, world, dist, torch):
    """C4 of BASELINE.md end to end, blocks sharded over the ranks (strong scaling)."""
    r = time_genome(args, rank, local_rank, world, dist, torch, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({
            "metric": "Mpixels/s detect end to end (C4: 23 blocks, loops + 3 borders templates)",
            "value": r["value"], "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": r["ms_per_genome"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": r["workload"], "correlation_pixels_per_step": r["correlation_pixels_per_step"],
                       "loop_pixels": r["loop_pixels"], "border_pixels": r["border_pixels"],
                       "stored_pixels_rank0": r["stored_pixels_rank0"],
                       "parallelism": f"{world} rank(s), {r['blocks']} blocks"},
            "patterns": r["patterns"], "setup": r["setup"],
        }))


def time_c5(local_rank, world, dist, torch, steps, warmup):
    """C5 of BASELINE.md: `quantify --inter` with the three 11 x 11 borders templates on the committed 17-chromosome
    yeast map (tests/golden/yeast_cool.npz, the decoded .cool; positions = the fixture's: cohesin-peak pairs on the
    intra blocks, seeded positions on the inter blocks).  The pixel table is uploaded once (DeviceCool, reported under
    `setup`); one step = pipeline.quantify from the resident table: staging of every sub-matrix that holds a position
    (detrend / median scaling), one native call per template over all of them, best-of-templates selection.  N > 1: the
    sub-matrices are dealt to the ranks (parallel.quantify_genome: the reference's pool over sub-matrices), one exchange of
    scores and windows, every rank assembles the table -- strong scaling of one fixed job.  Returns the leg's dictionary."""
    import pandas as pd
    from chromosight_amd import parallel, pipeline
    from chromosight_amd._lib import get_device
    here = os.path.dirname(os.path.abspath(__file__))
    cool = dict(np.load(os.path.join(here, "tests", "golden", "yeast_cool.npz"), allow_pickle=True))
    g = dict(np.load(os.path.join(here, "tests", "golden", "yeast_quantify.npz"), allow_pickle=True))
    names = [str(n) for n in cool["chrom_names"]]
    binsize = int(cool["binsize"])
    rows, n_inter = [], 0
    for bi in range(int(g["n_blocks"])):
        ca, cb = (int(x) for x in g[f"b{bi}_chroms"])
        for r, c in g[f"b{bi}_coords"]:
            rows.append((names[ca], int(r) * binsize, (int(r) + 1) * binsize, names[cb], int(c) * binsize, (int(c) + 1) * binsize))
            n_inter += ca != cb
    positions = pd.DataFrame(rows, columns=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
    cfg = dict(pearson=0.15, max_perc_undetected=75.0, max_perc_zero=10.0, max_dist=0, min_dist=0,
               kernels=[g[f"kernel{ki}"] for ki in range(3)], max_iterations=1, min_separation=5000)
    md = int(g["cfg_max_dist_bp"])
    dev = get_device(local_rank)
    t0 = time.perf_counter()
    dcool = pipeline.DeviceCool(cool, dev)
    dev.sync()
    upload_ms = (time.perf_counter() - t0) * 1e3
    run = (lambda: parallel.quantify_genome(dcool, positions, cfg, inter=True, max_dist_bp=md)) if world > 1 else \
        (lambda: pipeline.quantify(dcool, positions, cfg, inter=True, max_dist_bp=md))

    def sync():
        dev.sync()
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    table = None
    for _ in range(max(warmup, 1)):
        table, _w = run()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        table, _w = run()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed / steps * 1e3
    n_fixture = int(sum(len(g[f"b{bi}_coords"]) for bi in range(int(g["n_blocks"]))))
    return {
        "value": round(len(positions) * 3 / (ms * 1e-3), 1), "unit": "scored (position, template) pairs/s",
        "n_gpus": world, "steps": steps, "warmup": max(warmup, 1), "ms_per_step": round(ms, 3), "scaling": "strong", "dtype": "f64",
        "data": "tests/golden/yeast_cool.npz (real map, committed fixture)",
        "config": {"workload": "C5: 17-chromosome yeast map (6074 bins), quantify --inter, win-size 11, "
                               f"{len(positions)} positions ({n_inter} on inter-chromosomal blocks), pixel table resident in HBM",
                   "positions": len(positions), "positions_in_fixture": n_fixture, "templates": 3, "rows_out": int(len(table)),
                   "parallelism": f"{world} rank(s), sub-matrices dealt longest-first, one exchange of scores + windows"},
        "setup": {"upload_ms": round(upload_ms, 3), "upload_bytes": int(dcool.upload_bytes)},
    }


def run_c5(args, rank, local_rank, world, dist, torch):
    r = time_c5(local_rank, world, dist, torch, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(dict({"metric": "positions/s quantify end to end (C5: yeast, --inter, 3 x 11x11 borders templates)",
                               "higher_is_better": True, "vs_baseline": None}, **r)))


class SplitC4P:
    """The north-star block at N > 1: ONE 200 000-bin block (the C4' band of rank 0's seed, the same on every rank) row-split
    over the ranks -- strong scaling of the `north_star_c4p` leg, whose N = 1 form is the plain correlation call.  Rank r holds
    rows split_rows(n, N)[r] of the band plus the template's halo and per step (parallel.SplitBlockScan): all-reduces the
    per-diagonal (sum, count) of its rows (the distance law of a split block), runs the same masked tile kernel on its row
    window with the candidate epilogue (cs_candidates: coordinates of the candidate pixels, re-scored in float64, no map) and
    all-gathers the candidates of its rows."""

    def __init__(self, dev, rank, world, precision, n=None):
        from chromosight_amd import engine, parallel
        from chromosight_amd._lib import CsMatrix, LAYOUT_BAND, LAYOUT_BAND_PADDED, MASK_BINS, np_dtype_code
        from tools.synthetic_genome import band_workload
        band, band_w, miss, n, max_dist = band_workload("c4p", 0, n=n)
        self.n, self.max_dist, self.dev = n, max_dist, dev
        a, b = self.rows = parallel.split_rows(n, world)[rank]
        kh = 8
        ra, rb = max(0, a - kh), min(n, b + kh)
        own = band[a:b, :band_w]
        law_part = np.concatenate([own.sum(axis=0, dtype=np.float64), (own > 0).sum(axis=0).astype(np.float64)])
        slab = np.ascontiguousarray(band[ra:rb])
        ld_in = band.shape[1]
        del band, own
        self.out_w = out_w = max_dist + 1
        ld_out = (out_w + 63) // 64 * 64
        f32 = np_dtype_code(np.float32)
        self.sig_buf, self.out_buf, self.miss_buf = dev.to_device(slab), dev.zeros((b - a, ld_out), np.float32), dev.to_device(miss)
        sig = CsMatrix(self.sig_buf.ptr, f32, LAYOUT_BAND_PADDED if ld_in >= band_w + 4 else LAYOUT_BAND, ld_in, 0, band_w, ra)
        out = CsMatrix(self.out_buf.ptr, f32, LAYOUT_BAND, ld_out, 0, out_w, a)
        # the rank's output rows as a matrix of their own (rows and columns counted from `a`: same diagonals, same storage)
        out_local = CsMatrix(self.out_buf.ptr, f32, LAYOUT_BAND, ld_out, 0, out_w, 0)
        kspec = engine.KernelSpec(loops_kernel())
        ev0, ev1 = dev.new_event(), dev.new_event()
        self.kernel_ms = []

        def correlate():
            dev.record(ev0)
            engine.run_normxcorr2(dev, sig, (n, n), kspec, out, full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS,
                                  miss_row=self.miss_buf, miss_col=self.miss_buf, missing_tol=0.5, precision=precision,
                                  row_window=(a, b))
            dev.record(ev1)

        def candidates():
            rows, cols, vals = engine.run_compact(dev, out_local, (b - a, n - a), 0.3, 0, max_dist)
            self.kernel_ms.append(dev.elapsed_ms(ev0, ev1))       # (run_compact's download has synchronised the stream)
            return np.column_stack([rows.astype(np.float64) + a, cols.astype(np.float64) + a, vals])

        # The pipeline's own form (default): ONE call per step, cs_candidates on the rank's row window -- the masked
        # tile kernel's candidate epilogue appends the coordinates of its candidate pixels (no coefficient map is written, nothing
        # is compacted), the float64 re-scoring keeps those at or above the threshold -- which is what utils.detection.
        # detect_split_on_device does before the exchange.  CS_BENCH_SPLIT_MAP=1: the map + compaction form above.
        self.fused = not os.environ.get("CS_BENCH_SPLIT_MAP")

        def correlate_fused():
            dev.record(ev0)
            rows, cols, vals = engine.run_candidates(dev, sig, (n, n), kspec, (a, b), pearson=0.3, lo_diag=0, hi_diag=max_dist, inter=False,
                                                     full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=self.miss_buf,
                                                     miss_col=self.miss_buf, missing_tol=0.5, precision=precision)
            dev.record(ev1)
            dev.sync()
            self.kernel_ms.append(dev.elapsed_ms(ev0, ev1))       # (the chain: tile kernel with the candidate epilogue + float64 re-scoring)
            self.cand = np.column_stack([rows.astype(np.float64), cols.astype(np.float64), vals])

        def candidates_fused():
            return self.cand

        if self.fused:
            self.scan = parallel.SplitBlockScan(n, law_part, correlate_fused, candidates_fused)
        else:
            self.scan = parallel.SplitBlockScan(n, law_part, correlate, candidates)


def _barrier(dist, local_rank):
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[local_rank])
    else:
        dist.barrier()


def time_c4p_split(dev, rank, local_rank, world, dist, precision, full_sync, reduce_max, steps=20, warmup=3, n=None):
    """Times SplitC4P: `value` = the block's pixels x steps / max-over-ranks wall time (barriers and device synchronisation on
    both sides of the timed steps)."""
    from chromosight_amd import parallel
    w = SplitC4P(dev, rank, world, precision, n=n)
    scan, merged = w.scan, None
    for _ in range(warmup):
        _law, merged = scan.step()
    full_sync()
    _barrier(dist, local_rank)
    w.kernel_ms.clear()
    t0 = time.perf_counter()
    ex = 0.0
    for _ in range(steps):
        _law, merged = scan.step()
        ex += scan.exchange_ms
    full_sync()
    _barrier(dist, local_rank)
    elapsed = reduce_max(time.perf_counter() - t0)
    mine = {"rank": rank, "rows": [int(w.rows[0]), int(w.rows[1])], "kernel_ms": round(float(np.mean(w.kernel_ms)), 4),
            "exchange_ms": round(ex / steps, 4)}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    kid = int(dev.lib.cs_last_kernel(dev.ctx))
    n, out_w = w.n, w.out_w
    return {"value": round(n * out_w * steps / elapsed / 1e6, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4), "scaling": "strong", "transport": parallel.transport(),
            "per_rank": per_rank, "kernel_id": kid, "pixels_per_step": n * out_w,
            "candidates": int(len(merged)) if merged is not None else 0,
            "workload": f"C4P split: ONE N={n} block (band 0..{w.max_dist + 17}, 2% missing bins, rank 0's seed on every rank) row-split "
                        f"over {world} ranks: per step the distance law's (sum, count) all-reduced, normxcorr2(full=True, sym_upper, "
                        f"mask, max_dist={w.max_dist}) on the rank's row window (slab + halo resident) "
                        + ("through cs_candidates (the tile kernel's candidate epilogue: no coefficient map; float64 re-scoring), the "
                           "candidates >= 0.3 all-gathered" if w.fused else "coefficients >= 0.3 compacted from the map and all-gathered")
                        + "; N = 1 is the `north_star_c4p` leg (the correlation call alone)",
            "form": "cs_candidates" if w.fused else "map + compaction"}


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script with
    torch.distributed.run on 127.0.0.1 (one process per GPU) and relay their output."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


EXTRAS_TIMEOUT_S = 240


CLOCKS_AT_START = gpu_clocks()          # (module import: before this process has touched a GPU)


def main():
    args = parse()
    if args.plan:
        return print_plan(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        try:
            import torch
            if not torch.cuda.is_available():
                torch = None
        except Exception:
            torch = None

    if args.workload == "c5":
        run_c5(args, rank, local_rank, world, dist, torch)
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.workload == "c4":
        run_c4(args, rank, local_rank, world, dist, torch)
        if dist is not None:
            dist.destroy_process_group()
        return

    import chromosight_amd
    from chromosight_amd._lib import get_device
    chromosight_amd.set_precision(args.precision)
    dev = get_device(local_rank)
    wl = Workload(args.workload, dev, rank, args.precision, args.size)

    def full_sync():
        if torch is not None:
            torch.cuda.synchronize(local_rank)
        dev.sync()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    prewarm(wl.step, full_sync)
    elapsed, kernel_ms = time_steps(dev, wl.step, full_sync, args.steps, args.warmup, dist, local_rank)
    kernel_id = int(dev.lib.cs_last_kernel(dev.ctx))
    elapsed = reduce_max(elapsed)

    # The headline line is complete at this point.  The legs below are reported extras that use more of the stack (the
    # RCCL exchange of the sharded genome has only ever run where several GPUs exist): if one of them hangs, every rank
    # leaves through this timer and rank 0 still prints the line, with a note instead of the leg.
    head = {}
    if rank == 0:
        total_pixels = wl.pixels * args.steps * world
        traffic, traffic_note = pmc_traffic(args, wl)
        roof = roofline_of(wl, kernel_ms, kernel_id, args.precision, traffic, traffic_note)
        head = {
            "metric": "Mpixels/s normxcorr2 (17x17 loops kernel)",
            "value": round(total_pixels / elapsed / 1e6, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": wl.desc, "pixels_per_step_per_gpu": wl.pixels,
                       "parallelism": f"{world} independent sub-matrices, one per GPU"},
            "roofline": roof,
            "roofline_hbm": dict(roof["hbm"], bound="hbm"),
            "kernel_ms": round(kernel_ms, 4),
        }
        if world == 1:
            head["gpu_state"] = gpu_state(wl.step, full_sync, CLOCKS_AT_START)
    extras = {}

    def bail_out(why=None):
        if rank == 0:
            line = dict(head)
            line.update(extras)
            line["extras_note"] = why or f"a reported extra did not finish within {EXTRAS_TIMEOUT_S} s; the line was printed without it"
            sys.stdout.write(json.dumps(line) + "\n")
            sys.stdout.flush()
        os._exit(0)

    watchdog = None
    if world > 1 and not args.no_cpu_baseline:
        import signal
        import threading
        watchdog = threading.Timer(EXTRAS_TIMEOUT_S, bail_out)
        watchdog.daemon = True
        watchdog.start()
        # the launcher ends the other ranks with SIGTERM when one of them dies inside an extra: the measured headline
        # line still leaves rank 0
        signal.signal(signal.SIGTERM, lambda *_: bail_out("a rank ended inside a reported extra (SIGTERM from the launcher); "
                                                          "the line was printed without it"))

    # the north-star configuration next to the headline one, kernel only, at every N (each rank its own 200 000-bin
    # block: weak scaling like `value`): C4' = one 200 000-bin block, band to 1000 bins, masks, full mode
    c4p = None
    if args.workload == "c2" and args.size == 4096 and not args.no_cpu_baseline:
        try:
            del wl.out_buf
            w4 = Workload("c4p", dev, rank, args.precision)
            prewarm(w4.step, full_sync, 0.1)
            steps4 = 20
            el4, ms4 = time_steps(dev, w4.step, full_sync, steps4, 3, dist, local_rank)
            id4 = int(dev.lib.cs_last_kernel(dev.ctx))
            el4 = reduce_max(el4)
            t4, t4_note = pmc_traffic(argparse.Namespace(workload="c4p", size=None, precision=args.precision), w4)
            c4p = {"value": round(w4.pixels * steps4 * world / el4 / 1e6, 1), "unit": "Mpixel/s", "n_gpus": world, "steps": steps4,
                   "warmup": 3, "ms_per_step": round(el4 / steps4 * 1e3, 4), "kernel_ms": round(ms4, 4), "scaling": "weak",
                   "workload": w4.desc, "pixels_per_step_per_gpu": w4.pixels,
                   "roofline": roofline_of(w4, ms4, id4, args.precision, t4, t4_note)}
            if world == 1:
                # templates above 17 x 17 on the same resident band (north_star: "small (<= ~21 x 21) pattern kernel"; `--win-size`)
                try:
                    extras["wide_templates"] = {"c4p_masked_band": time_wide_templates(dev, w4, full_sync),
                                                "note": "cs_corr_wide.hip: two k = 32 Toeplitz passes per template row on the matrix cores; "
                                                        "kernel ms by HIP events over 10 launches; FP32 roof 157.3 TFLOP/s at 2 k^2 + 8 k flop/pixel"}
                except Exception as exc:
                    extras["wide_templates"] = {"error": repr(exc)}
            del w4
        except Exception as exc:
            c4p = {"error": repr(exc)}
        extras["north_star_c4p"] = c4p
        # ... and the north-star sentence itself at N > 1: the ONE 200 000-bin block split over the ranks (strong scaling)
        if world > 1:
            try:
                extras["north_star_c4p_split"] = time_c4p_split(dev, rank, local_rank, world, dist, args.precision, full_sync, reduce_max)
            except Exception as exc:
                extras["north_star_c4p_split"] = {"error": repr(exc)}

    # C3 as BASELINE.md defines it (pixel table in, detrend included) next to it: the one single-GPU BASELINE configuration
    # whose step is more than the correlation call -- staging kernels (law pass writing the band of counts, finish), mask tables,
    # tile kernel
    c3 = None
    if args.workload == "c2" and args.size == 4096 and not args.no_cpu_baseline and world == 1:
        try:
            w3 = Workload("c3", dev, rank, args.precision)
            prewarm(w3.step, full_sync, 0.1)
            steps3 = 20
            el3, ms3 = time_steps(dev, w3.step, full_sync, steps3, 3, dist, local_rank)
            id3 = int(dev.lib.cs_last_kernel(dev.ctx))
            t3, t3_note = pmc_traffic(argparse.Namespace(workload="c3", size=None, precision=args.precision), "c3")
            roof3 = roofline_of(w3, ms3, id3, args.precision, t3, t3_note)
            roof3["note"] = ("whole step (HIP events around the law pass that also writes the band of raw counts + its finish + the mask "
                             "tables + the tile kernel, which detrends the tiles it fetches: 4 launches): 714 "
                             f"flop/pixel x {w3.pixels} pixels / {ms3:.4f} ms against the FP32 peak; bytes view (SURVEY 8d): 2 passes over "
                             f"the stored pixels (8 B each) + row pointers + 4 B per pixel written = {w3.bytes_per_pixel:.1f} B/pixel -- "
                             "the step itself makes ONE pass over the pixel table and writes / reads a 4 B band of counts in place of "
                             "the second")
            c3 = {"value": round(w3.pixels * steps3 / el3 / 1e6, 1), "unit": "Mpixel/s", "n_gpus": 1, "steps": steps3, "warmup": 3,
                  "ms_per_step": round(el3 / steps3 * 1e3, 4), "kernel_ms": round(ms3, 4), "workload": w3.desc,
                  "pixels_per_step": w3.pixels, "stored_pixels": int(w3.nnz), "roofline": roof3,
                  "roofline_hbm": dict(roof3["hbm"], bound="hbm")}
            del w3
        except Exception as exc:
            c3 = {"error": repr(exc)}
        extras["c3_from_csr"] = c3

    genome = None
    if not args.no_cpu_baseline and args.workload == "c2":
        # the sharded end-to-end genome next to the kernel figure, at every N (strong scaling, records
        # gathered over RCCL): all ranks take part
        try:
            genome = time_genome(args, rank, local_rank, world, dist, torch, steps=8, warmup=3)
        except Exception as exc:
            genome = {"error": repr(exc)}
    if watchdog is not None:
        watchdog.cancel()
        signal.signal(signal.SIGTERM, signal.SIG_DFL)

    if rank == 0:
        out = dict(head)
        if c4p is not None:
            out["north_star_c4p"] = c4p
        if "north_star_c4p_split" in extras:
            out["north_star_c4p_split"] = extras["north_star_c4p_split"]
        if c3 is not None:
            out["c3_from_csr"] = c3
        if "wide_templates" in extras:
            out["wide_templates"] = extras["wide_templates"]
        if genome is not None:
            out["sharded_genome"] = genome
        if not args.no_cpu_baseline and world == 1:     # reported extras: rank 0 at N = 1 only
            try:
                out["detect_wallclock"] = detect_wallclock()
            except Exception as exc:
                out["detect_wallclock"] = {"error": repr(exc)}
            try:
                # C5 of BASELINE.md (quantify --inter on the committed yeast map) next to it: ~ 5 ms per step
                q = time_c5(local_rank, 1, None, torch, steps=10, warmup=2)
                q["positions_match_fixture"] = bool(q["config"]["rows_out"] == q["config"]["positions"] == q["config"]["positions_in_fixture"])
                out["quantify_c5"] = q
            except Exception as exc:
                out["quantify_c5"] = {"error": repr(exc)}
            try:
                # C2 in float64 arithmetic (the reference's own type) and under templates above 17 x 17, on fresh resident maps
                w64 = Workload("c2", dev, rank, "f64", args.size)
                _, ms64 = time_steps(dev, w64.step, full_sync, 5, 2)
                out["c2_f64"] = {"kernel_ms": round(ms64, 4), "gpixel_per_s": round(w64.pixels / ms64 / 1e6, 1),
                                 "kernel": KERNELS.get(int(dev.lib.cs_last_kernel(dev.ctx)), ("?", ""))[0],
                                 "frac_fp64_roof": round(FLOP_PER_PIXEL_17 * w64.pixels / (ms64 * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                                 "note": "the same map and template in float64 arithmetic and containers (packed-FMA streaming kernel)"}
                del w64
                w32 = Workload("c2", dev, rank, "f32", args.size)
                if "wide_templates" in out and isinstance(out["wide_templates"], dict):
                    out["wide_templates"]["c2_dense_map"] = time_wide_templates(dev, w32, full_sync)
                del w32
            except Exception as exc:
                out["c2_f64"] = {"error": repr(exc)}
            try:
                out["api_call_ms"] = api_call_ms(wl)
                out["api_call_note"] = ("host float32 ndarray in (pageable source, uploaded in 12 row slabs), float64 ndarray out "
                                        "(page-locked pool); PCIe-inclusive, never `value`; box to box 3.5 - 6 ms with the host's "
                                        "memory placement")
            except Exception as exc:
                out["api_call_ms"] = {"error": repr(exc)}
            try:
                out["cpu_baseline"] = cpu_baseline(wl)
            except Exception as exc:  # the baseline is a reported extra, never the measured path
                out["cpu_baseline"] = {"error": repr(exc)}
            if isinstance(c4p, dict) and "error" not in c4p:
                try:
                    c4p["cpu_baseline"] = cpu_baseline_band()
                    cb = c4p["cpu_baseline"]
                    c4p["vs_cpu"] = {"all_cores": round(c4p["value"] / cb["value"], 1),
                                     "one_core": round(c4p["value"] / cb["single_core"]["value"], 1),
                                     "note": "north star: >= 50x the reference's --threads=<cores> CPU throughput on this block"}
                except Exception as exc:
                    c4p["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def print_plan(args):
    """--plan: what `--gpus N` will do with the sharded C4 genome, without touching a GPU: the LPT assignment
    (parallel.assign_blocks) and the scanned pixels per rank, so that the first multi-GPU run can be checked from its log."""
    from chromosight_amd import parallel
    from tools.synthetic_genome import genome_sizes
    sizes = [int(n) for n in genome_sizes(200_000)]
    max_dist = 1000
    costs = [parallel.block_cost((n, n), max_dist, False) for n in sizes]
    owned = parallel.assign_blocks(costs, args.gpus)
    per_rank = [int(sum(costs[i] for i in o)) for o in owned]
    print(json.dumps({"plan": "C4 genome, 23 blocks, LPT by scanned pixels", "n_gpus": args.gpus, "blocks": sizes,
                      "owned": owned, "loop_pixels_per_rank": per_rank, "total": int(sum(costs)),
                      "balance": round(max(per_rank) / (sum(per_rank) / len(per_rank)), 3),
                      "speedup_bound": round(sum(per_rank) / max(per_rank), 2),
                      "exchange": "one record exchange per detect_genome call (count + padded all-gather, RCCL)"}))


if __name__ == "__main__":
    main()
