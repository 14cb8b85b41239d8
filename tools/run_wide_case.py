#!/usr/bin/env python3
"""One workload under one template with a side of 18 .. 33, `steps` launches: what rocprofv3 profiles for the two-pass
matrix-core kernel (tools/collect_profiles.sh).    python tools/run_wide_case.py <dense|c3|c4p> <k> [steps]
Prints the kernel time by HIP events, the kernel that served the call and the shader clocks under load."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
from chromosight_amd._lib import get_device  # noqa: E402


def main():
    what, k = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = get_device()
    wl = bench.Workload("c2" if what == "dense" else what, dev, 0, "f32")
    wl.kspec = wl.engine.KernelSpec(bench.wide_template(k))
    _, ms = bench.time_steps(dev, wl.step, dev.sync, steps, 3)
    kid = int(dev.lib.cs_last_kernel(dev.ctx))
    flop = 2 * k * k + 8 * k
    print(json.dumps({"workload": what, "template": f"{k}x{k}", "steps": steps, "kernel_ms": round(ms, 4),
                      "gpixel_per_s": round(wl.pixels / ms / 1e6, 1), "kernel_id": kid, "kernel": bench.KERNELS.get(kid, ("?", ""))[0],
                      "flop_per_pixel": flop, "frac_fp32_roof": round(flop * wl.pixels / (ms * 1e-3) / 1e12 / bench.FP32_PEAK_TFLOPS, 4),
                      "gpu_state": bench.gpu_state(wl.step, dev.sync)}))


if __name__ == "__main__":
    main()
