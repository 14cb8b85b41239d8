"""Run tests/test_gpu_random_sweep.py::test_random_configuration over a range of extra seeds on the
GPU box:  python tools/stress_random_sweep.py 24 424"""
import sys, os, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib.util, pathlib
spec = importlib.util.spec_from_file_location("sweep", "tests/test_gpu_random_sweep.py")
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    try:
        mod.test_random_configuration.__wrapped__(seed) if hasattr(mod.test_random_configuration, "__wrapped__") else mod.test_random_configuration(seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:200]))
        print("FAIL", seed, str(e)[:200], flush=True)
    except Exception as e:
        bad.append((seed, repr(e)[:200]))
        print("ERROR", seed, repr(e)[:300], flush=True)
print("done", lo, hi, "failures:", len(bad), bad[:10])
