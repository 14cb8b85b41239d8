#!/usr/bin/env python3
"""Per-phase cycle shares of the two-pass matrix-core kernel (cs_corr_wide.hip, CS_WD_PROFILE build:
`make -C chromosight_amd/csrc prof`).

    python tools/prof_wide_sections.py [dense c4p c3] [k ...]

Lane 0 of every wave adds the cycles between two stamps to a device counter: where a wave's time goes per tile (waits
at barriers and for loads are part of the phase that ends with them)."""
import ctypes as C
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import chromosight_amd._lib as L                                    # noqa: E402

L._LIB_PATH = ROOT / "chromosight_amd" / "csrc" / "build" / "libchromosight_hip_prof.so"
import numpy as np                                                  # noqa: E402

NAMES = ["addresses, loads issued", "barrier, mask bits", "wait for the loads, maximum", "barrier, split, plane writes, tables",
         "barrier", "box sums", "cross term", "mask sums", "epilogue"]


def main():
    import chromosight_amd
    from chromosight_amd import engine
    from chromosight_amd._lib import LAYOUT_BAND, LAYOUT_DENSE, MASK_BINS, MASK_NONE, CsMatrix, get_device, np_dtype_code
    from tools.synthetic_genome import band_workload
    from tools.time_wide import template
    chromosight_amd.set_precision("f32")
    dev = get_device(0)
    lib = dev.lib
    lib.cs_debug_wide_profile.restype = C.c_int
    lib.cs_debug_wide_profile.argtypes = [C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 16)()
    args = sys.argv[1:] or ["dense", "c4p"]
    sizes = [int(a) for a in args if a.isdigit()] or [21, 33]
    for name in [a for a in args if not a.isdigit()]:
        if name == "dense":
            n = 4096
            sig = np.random.default_rng(0).gamma(4.0, 0.25, size=(n, n)).astype(np.float32)
            d_sig, ld_in = engine.to_device_map(dev, sig)
            ld_out = engine.map_pitch(n, 4)
            d_out = dev.empty((n, ld_out), np.float32)

            def call(spec):
                engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_in, 0, 0), (n, n), spec,
                                      CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_DENSE, ld_out, 0, 0),
                                      full=False, sym_upper=False, max_dist=None, mask_mode=MASK_NONE, precision="f32")
        else:
            band, band_w, miss, n, max_dist = band_workload(name)
            out_w = max_dist + 1
            ld_out = (out_w + 63) // 64 * 64
            d_sig, d_out = dev.to_device(band), dev.zeros((n, ld_out), np.float32)
            d_miss = dev.to_device(miss)

            def call(spec):
                engine.run_normxcorr2(dev, CsMatrix(d_sig.ptr, np_dtype_code(np.float32), LAYOUT_BAND, band.shape[1], 0, band_w),
                                      (n, n), spec, CsMatrix(d_out.ptr, np_dtype_code(np.float32), LAYOUT_BAND, ld_out, 0, out_w),
                                      full=True, sym_upper=True, max_dist=max_dist, mask_mode=MASK_BINS, miss_row=d_miss,
                                      miss_col=d_miss, missing_tol=0.5, precision="f32")
        for k in sizes:
            spec = engine.KernelSpec(template(k))
            for _ in range(3):
                call(spec)
            dev.sync()
            lib.cs_debug_wide_profile(buf)
            steps = 10
            for _ in range(steps):
                call(spec)
            dev.sync()
            lib.cs_debug_wide_profile(buf)
            total = sum(buf[i] for i in range(15))
            waves = max(buf[15], 1)
            print(f"{name} {k}x{k}: {waves // steps // 4} tiles per launch, {total / waves:.0f} cycles per wave and tile")
            for i, what in enumerate(NAMES):
                print(f"   {buf[i] / waves:9.0f} cycles  {100.0 * buf[i] / max(total, 1):5.1f} %   {what}")


if __name__ == "__main__":
    main()
