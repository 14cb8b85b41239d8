#!/usr/bin/env python3
"""Per-phase cycle shares of the matrix-core tile kernels (CS_MF_PROFILE build: `make -C chromosight_amd/csrc prof`).

    python tools/prof_mfma_sections.py [c2 c3k c4p]

Thread 0 of every workgroup adds the cycles between two stamps of the tile loop to a device counter; the shares say where
a wave's time goes (waits at barriers and for the DMA included in the phase that ends with them)."""
import ctypes as C
import os
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import chromosight_amd._lib as L                                    # noqa: E402

L._LIB_PATH = ROOT / "chromosight_amd" / "csrc" / "build" / "libchromosight_hip_prof.so"
import bench                                                        # noqa: E402

NAMES = {0: "wait DMA + barrier + read landed tile + max", 1: "barrier + split + plane writes", 6: "emit previous tile",
         2: "barrier + box sums", 3: "barrier + DMA issue", 4: "cross term", 5: "-",
         7: "  emit: statistics block, U vectors", 8: "  emit: mask sums (tables, cross term)", 9: "  emit: correction records",
         10: "  emit: coefficients", 11: "  emit: stores"}


def main():
    import chromosight_amd
    from chromosight_amd._lib import get_device
    chromosight_amd.set_precision("f32")
    dev = get_device(0)
    lib = dev.lib
    lib.cs_debug_mfma_profile.restype = C.c_int
    lib.cs_debug_mfma_profile.argtypes = [C.POINTER(C.c_ulonglong)]
    buf = (C.c_ulonglong * 16)()
    for name in (sys.argv[1:] or ["c2", "c3k", "c4p"]):
        wl = bench.Workload(name, dev, 0, "f32")
        for _ in range(5):
            wl.step()
        dev.sync()
        lib.cs_debug_mfma_profile(buf)
        steps = 20
        for _ in range(steps):
            wl.step()
        dev.sync()
        lib.cs_debug_mfma_profile(buf)
        total = sum(buf[k] for k in range(15))
        tiles = buf[15]
        print(f"{name}: {tiles // steps} tiles per launch, {total / max(tiles, 1):.0f} cycles per tile (thread 0 of each workgroup)")
        for k in (0, 1, 7, 8, 9, 10, 11, 6, 2, 3, 4):
            print(f"   {buf[k] / max(tiles, 1):9.0f} cycles  {100.0 * buf[k] / max(total, 1):5.1f} %   {NAMES[k]}")
        del wl


if __name__ == "__main__":
    main()
