#!/usr/bin/env python3
"""Templates with a side of 18 .. 33 (what `--win-size` makes): which kernel serves them and how long a call takes, on
the banded workloads C3 / C4' (per-bin masks) and on the dense 4096^2 map of C2 (no mask).  Algorithmic work per pixel of
a k x k template: 2 k^2 + 8 k flop -> fraction of the FP32 roof (157.3 TFLOP/s) next to the rate.
    python tools/time_wide.py [c3|c4p|dense] ...   ->  profiles/<tag>_template_kernels.txt (tools/collect_profiles.sh)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chromosight_amd.kernels as ck  # noqa: E402
from chromosight_amd import engine  # noqa: E402
from chromosight_amd._lib import LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE, CsMatrix, get_device, np_dtype_code  # noqa: E402
from tools.synthetic_genome import band_workload  # noqa: E402

KERNELS = {1: "runtime-size", 2: "streaming", 3: "matrix cores (general)", 4: "matrix cores (dense tile)",
           5: "matrix cores (masked tile)", 6: "separable", 7: "matrix cores (two-pass)"}
FP32_PEAK = 157.3e12


def template(k, seed=0):
    if isinstance(k, str):
        return np.asarray(getattr(ck, k)["kernels"][0], dtype=np.float64)
    rng = np.random.default_rng(1000 * k + seed)
    i, j = np.indices((k, k))
    c = (k - 1) / 2
    return 0.4 + np.exp(-((i - c) ** 2 + (j - c) ** 2) / (0.08 * k * k + 1)) + 0.02 * (i - j) + 0.15 * rng.normal(size=(k, k))


def timed(dev, call, reps):
    for _ in range(3):
        call()
    dev.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dev.sync()
    return (time.perf_counter() - t0) / reps * 1e3


def report(label, env, dev, ms, pixels, k):
    served = KERNELS.get(dev.lib.cs_last_kernel(dev.ctx), "?")
    switches = " ".join(f"{a}={b}" for a, b in env.items()) or "default"
    flop = 2 * k * k + 8 * k
    print(f"{label:26s} {switches:30s} kernel: {served:26s} {ms:8.3f} ms/call {pixels / ms / 1e6:7.1f} Gpixel/s"
          f"  {flop:5d} flop/px -> {pixels * flop / (ms * 1e-3) / FP32_PEAK:5.3f} of the FP32 roof", flush=True)


def band_cases(dev, workload):
    band, band_w, miss, n, max_dist = band_workload(workload)
    print(f"# band workload {workload}: {n} bins, {max_dist + 1} diagonals, per-bin masks (2 % of the bins), float32")
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    d_sig, d_out = dev.to_device(band), dev.zeros((n, ld_out), np.float32)
    d_miss = dev.to_device(miss)
    slow = workload == "c3"
    cases = [("loops 17x17", "loops", {}), ("loops 17x17", "loops", {"CHROMOSIGHT_HIP_WIDE_ALL": "1"})]
    for k in (19, 21, 25, 33):
        cases.append((f"full rank {k}x{k}", k, {}))
        cases.append((f"full rank {k}x{k}", k, {"CHROMOSIGHT_HIP_WIDE_PLANE": "1"}))
        if slow or k == 21:
            cases.append((f"full rank {k}x{k}", k, {"CHROMOSIGHT_HIP_NO_WIDE": "1"}))
    cases += [("stripes_left 31x31 (rank 1)", "stripes_left", {}),
              ("stripes_left 31x31 (rank 1)", "stripes_left", {"CHROMOSIGHT_HIP_NO_SEPARABLE": "1"})]
    for label, name, env in cases:
        k = template(name)
        spec = engine.KernelSpec(k)
        os.environ.update(env)

        def call():
            engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_BAND, band.shape[1], 0, band_w),
                                  (n, n), spec, CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w),
                                  full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss,
                                  miss_col=d_miss, missing_tol=0.5, precision="f32")
        ms = timed(dev, call, 5 if env.get("CHROMOSIGHT_HIP_NO_WIDE") or env.get("CHROMOSIGHT_HIP_NO_SEPARABLE") else 20)
        report(label, env, dev, ms, n * out_w, k.shape[0])
        for a in env:
            del os.environ[a]


def dense_cases(dev):
    n = 4096
    print(f"# dense workload: {n} x {n} float32 gamma(4, 0.25) (C2's map), no mask, full=False")
    sig = np.random.default_rng(0).gamma(4.0, 0.25, size=(n, n)).astype(np.float32)
    d_sig, ld_in = engine.to_device_map(dev, sig)
    ld_out = engine.map_pitch(n, 4)
    d_out = dev.empty((n, ld_out), np.float32)
    cases = [("loops 17x17", "loops", {}), ("loops 17x17", "loops", {"CHROMOSIGHT_HIP_WIDE_ALL": "1"})]
    for k in (19, 21, 25, 33):
        cases.append((f"full rank {k}x{k}", k, {}))
    cases.append(("full rank 21x21", 21, {"CHROMOSIGHT_HIP_NO_WIDE": "1"}))
    for label, name, env in cases:
        k = template(name)
        spec = engine.KernelSpec(k)
        os.environ.update(env)

        def call():
            engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_in, 0, 0), (n, n), spec,
                                  CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_out, 0, 0),
                                  full=False, sym_upper=False, max_dist=None, mask_mode=MASK_NONE, precision="f32")
        ms = timed(dev, call, 5 if env else 20)
        report(label, env, dev, ms, n * n, k.shape[0])
        for a in env:
            del os.environ[a]


def main():
    dev = get_device()
    for what in (sys.argv[1:] or ["c3", "c4p", "dense"]):
        if what == "dense":
            dense_cases(dev)
        else:
            band_cases(dev, what)


if __name__ == "__main__":
    main()
