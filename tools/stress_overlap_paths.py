import copy, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(30_000, 300, 2000, seed=4, template=template)
dcool = pipeline.DeviceCool(cool)
cfg = copy.deepcopy(ck.borders)
os.environ["CHROMOSIGHT_HIP_NO_TEMPLATE_OVERLAP"] = "1"
want = parallel.detect_genome(dcool, cfg, tsvd=0.999)
del os.environ["CHROMOSIGHT_HIP_NO_TEMPLATE_OVERLAP"]
for it in range(300):
    got = parallel.detect_genome(dcool, cfg, tsvd=0.999)
    assert got.shape == want.shape and np.array_equal(got[:, [0, 1, 2, 5, 6]], want[:, [0, 1, 2, 5, 6]]), it
print("tsvd overlap path: 300 repetitions equal", want.shape)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 300 * 2000; loops["max_iterations"] = 3
w2 = parallel.detect_genome(dcool, loops)
for it in range(100):
    g2 = parallel.detect_genome(dcool, loops)
    assert g2.shape == w2.shape and np.array_equal(g2[:, [0, 1, 2, 5, 6]], w2[:, [0, 1, 2, 5, 6]]), it
print("iterated loops: 100 repetitions equal", w2.shape)
