#!/usr/bin/env python3
"""Per-step durations of the C2 call right after a synchronisation (HIP events around every step): is the average of the
first 20 steps -- what `bench.py --steps 20` times -- the steady state?

    python tools/step_transient.py [steps] [idle_ms]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import chromosight_amd  # noqa: E402
from chromosight_amd._lib import get_device  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    idle_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    chromosight_amd.set_precision("f32")
    dev = get_device(0)
    wl = bench.Workload("c2", dev, 0, "f32")
    bench.prewarm(wl.step, dev.sync)
    for rep in range(4):
        for _ in range(5):
            wl.step()
        dev.sync()
        if idle_ms:
            time.sleep(idle_ms * 1e-3)
        evs = [dev.new_event() for _ in range(steps + 1)]
        dev.record(evs[0])
        for k in range(steps):
            wl.step()
            dev.record(evs[k + 1])
        dev.sync()
        us = [dev.elapsed_ms(evs[k], evs[k + 1]) * 1e3 for k in range(steps)]
        print(f"rep {rep}: first 20: {sum(us[:20]) / 20:.1f} us  last 20: {sum(us[-20:]) / 20:.1f} us   " +
              " ".join(f"{u:.0f}" for u in us))


if __name__ == "__main__":
    main()
