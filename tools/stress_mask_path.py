"""Random band / mask / template / precision configurations through the checks of
tests/test_gpu_regular_mask.py (factorised vs general mask path vs C oracle), each also with a
forced strip height:  python tools/stress_mask_path.py 0 150"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location("rm", "tests/test_gpu_regular_mask.py")
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(120, 1400))
    k = int(rng.choice([7, 9, 11, 13, 15, 17]))
    md = int(rng.choice([rng.integers(1, 12), rng.integers(12, 80), rng.integers(80, 400), n + 30]))
    frac = float(rng.choice([0.0, 0.01, 0.05, 0.25]))
    prec = "f64" if seed % 3 == 0 else "f32"
    flat = [False, True, 2][seed % 3 if seed % 5 else 2]
    case = (n, k, md, frac, prec, flat)
    for env in ({}, {"CHROMOSIGHT_HIP_STRIP_H": str(int(rng.choice([8, 14, 32, 70])))}):
        os.environ.update(env)
        try:
            mod.test_band_regular_vs_general_and_oracle(case)
        except AssertionError as e:
            bad.append((seed, case, env, str(e)[:150])); print("FAIL", seed, case, env, str(e)[:150], flush=True)
        except Exception as e:
            bad.append((seed, case, env, repr(e)[:150])); print("ERROR", seed, case, env, repr(e)[:200], flush=True)
        for kk in env: os.environ.pop(kk, None)
print("done", lo, hi, "failures:", len(bad))
