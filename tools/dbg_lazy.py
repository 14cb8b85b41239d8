import copy, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline, engine
from chromosight_amd._lib import CS_F64, LAYOUT_BAND, LAYOUT_BAND_LAZY, CsMatrix
from tools.synthetic_genome import make_cool
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 200 * 2000
cool, _ = make_cool(12_000, 200, 2000, seed=7, template=template, chrom_sizes=[4000, 3500, 2500, 2000])
dcool = pipeline.DeviceCool(cool)
dev = dcool.dev
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
blocks = dcool.stage_blocks([0, 1, 2, 3], 200, 17, lazy64=(mode if mode != "none" else False))
dev.sync()
print("staged", [b.sig.layout for b in blocks], flush=True)
for b in blocks:
    if b.sig.layout == LAYOUT_BAND_LAZY:
        raw = np.empty(128, dtype=np.uint8)
        dev._check(dev.lib.cs_memcpy_d2h(dev.ctx, raw.ctypes.data, b.sig.d_ptr, 128, None))
        print(raw[:48].view(np.uint64), raw[48:72].view(np.int64), raw[64:72].view(np.float64), raw[72:88].view(np.int32), flush=True)
res = pipeline.detect_blocks(dcool, blocks, loops, template, want_windows=True)
print("detect ok", sum(len(t) for t, w in res if t is not None), flush=True)
