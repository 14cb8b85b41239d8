#!/usr/bin/env python3
"""Two ranks (gloo rendezvous, both on this GPU) repeating the sharded genome step: every repetition must give the records of
the first on both ranks, and both ranks the same records.  python tools/stress_two_ranks.py [repetitions]   (spawns itself)"""
import copy, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def worker(reps):
    import torch.distributed as dist
    import chromosight_amd.kernels as ck
    from chromosight_amd import parallel, pipeline
    from tools.synthetic_genome import make_cool
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
    cool, _ = make_cool(30_000, 300, 2000, seed=6, template=template)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops); loops["max_dist"] = 300 * 2000
    cfgs = [loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)]
    first = None
    for it in range(reps):
        staged = parallel.stage_genome(dcool, cfgs)
        recs = parallel.detect_patterns(dcool, cfgs, staged=staged)
        if first is None:
            first = recs
            continue
        for a, b in zip(first, recs):
            assert a.shape == b.shape and np.array_equal(a[:, [0, 1, 2, 5, 6]], b[:, [0, 1, 2, 5, 6]]), (dist.get_rank(), it)
    np.save(os.environ["CS_OUT"] + f".{dist.get_rank()}.npy", np.concatenate(first))
    dist.destroy_process_group()

if __name__ == "__main__":
    if "RANK" in os.environ:
        worker(int(sys.argv[1]))
    else:
        reps = sys.argv[1] if len(sys.argv) > 1 else "100"
        out = "/tmp/cs_two_ranks"
        env = dict(os.environ, CS_OUT=out, CHROMOSIGHT_HIP_DEVICE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), reps], env=dict(env, RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
        codes = [p.wait(timeout=900) for p in procs]
        assert codes == [0, 0], codes
        a, b = np.load(out + ".0.npy"), np.load(out + ".1.npy")
        assert a.shape == b.shape and np.array_equal(a, b)
        print(f"2 ranks x {reps} repetitions identical, both ranks hold the same {a.shape[0]} records")
