#!/usr/bin/env python3
"""Correlation call of the C3 band (50 000 bins, 234 diagonals, per-bin masks) with templates that do not take the
benched kernels: which kernel serves them (cs_last_kernel) and how long a call takes.  Used by
tools/collect_profiles.sh -> profiles/<tag>_template_kernels.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import chromosight_amd.kernels as ck  # noqa: E402
from chromosight_amd import engine  # noqa: E402
from chromosight_amd._lib import LAYOUT_BAND, MASK_BINS, CsMatrix, get_device, np_dtype_code  # noqa: E402
from tools.synthetic_genome import band_workload  # noqa: E402

KERNELS = {1: "runtime-size", 2: "streaming", 3: "matrix cores (general)", 4: "matrix cores (dense tile)",
           5: "matrix cores (masked tile)", 6: "separable"}


def main():
    dev = get_device()
    workload = sys.argv[1] if len(sys.argv) > 1 else "c3"          # "c3" (50 000 x 234) or "c4p" (200 000 x 1001)
    band, band_w, miss, n, max_dist = band_workload(workload)
    print(f"# band workload {workload}: {n} bins, {max_dist + 1} diagonals, per-bin masks, float32")
    out_w = max_dist + 1
    ld_out = (out_w + 63) // 64 * 64
    d_sig, d_out = dev.to_device(band), dev.zeros((n, ld_out), np.float32)
    d_miss = dev.to_device(miss)
    cases = [("loops 17x17", "loops", {}), ("loops 17x17", "loops", {"CHROMOSIGHT_HIP_MFMA_REG": "0"}),
             ("loops_small 7x7", "loops_small", {}), ("loops_small 7x7", "loops_small", {"CHROMOSIGHT_HIP_MFMA_REG": "1"}),
             ("borders 17x17 (rows not mirrored)", "borders", {}),
             ("borders 17x17 (rows not mirrored)", "borders", {"CHROMOSIGHT_HIP_MFMA_REG": "0"}),
             ("hairpins 15x15", "hairpins", {}), ("hairpins 15x15", "hairpins", {"CHROMOSIGHT_HIP_MFMA_REG": "0"}),
             ("random 13x13", 13, {"CHROMOSIGHT_HIP_MFMA_REG": "0"}), ("random 13x13", 13, {"CHROMOSIGHT_HIP_MFMA_REG": "1"}),
             ("random 11x11", 11, {"CHROMOSIGHT_HIP_MFMA_REG": "0"}), ("random 11x11", 11, {"CHROMOSIGHT_HIP_MFMA_REG": "1"}),
             ("random 9x9", 9, {"CHROMOSIGHT_HIP_MFMA_REG": "0"}), ("random 9x9", 9, {"CHROMOSIGHT_HIP_MFMA_REG": "1"}),
             ("stripes_left 31x31 (rank 1)", "stripes_left", {}),
             ("stripes_left 31x31 (rank 1)", "stripes_left", {"CHROMOSIGHT_HIP_NO_SEPARABLE": "1"})]
    for label, name, env in cases:
        k = (np.random.default_rng(name).normal(size=(name, name)) + 0.3 if isinstance(name, int)
             else np.asarray(getattr(ck, name)["kernels"][0], dtype=np.float64))
        spec = engine.KernelSpec(k)
        os.environ.update(env)

        def call():
            engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_BAND, band.shape[1], 0, band_w),
                                  (n, n), spec, CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w),
                                  full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss,
                                  miss_col=d_miss, missing_tol=0.5, precision="f32")
        for _ in range(3):
            call()
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        dev.sync()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        served = KERNELS.get(dev.lib.cs_last_kernel(dev.ctx), "?")
        switches = " ".join(f"{a}={b}" for a, b in env.items()) or "default"
        print(f"{label:34s} {switches:34s} kernel: {served:28s} {ms:7.3f} ms/call  {n * out_w / ms / 1e6:6.1f} Gpixel/s")
        for a in env:
            del os.environ[a]


if __name__ == "__main__":
    main()
