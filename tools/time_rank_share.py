#!/usr/bin/env python3
"""What ONE rank of an N-GPU run of the C4 genome does, timed on one GPU: its LPT share of the 23 blocks, and -- VERDICT r5 1(d) -- the
record exchange of the step on a single-rank RCCL communicator (csrc/cs_comm.cpp: the step's lists in one collective; device
staging, the collective launch and the download are all there, only the peers are missing).  CS_NO_EXCHANGE=1: without it.
    python tools/time_rank_share.py [n_gpus] [rank]"""
import copy, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import genome_sizes, make_cool

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sizes = genome_sizes(200_000)
costs = [parallel.block_cost((int(n), int(n)), 1000, False) for n in sizes]
mine = parallel.assign_blocks(costs, world)[rank]
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template, only=mine)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 2_000_000
borders = copy.deepcopy(ck.borders)
comm = None
if not os.environ.get("CS_NO_EXCHANGE"):
    try:
        comm = parallel.NativeComm(0, 0, 1, parallel.NativeComm.unique_id())
    except RuntimeError as exc:
        print(f"# no single-rank RCCL communicator on this box ({exc}): steps timed without the exchange")
ts, ex = [], []
for it in range(int(os.environ.get("CS_STEPS", "24"))):
    dcool.dev.sync(); t0 = time.perf_counter()
    rec = parallel.genome_step(dcool, [loops, borders], owned=mine)
    t1 = time.perf_counter()
    if comm is not None and not os.environ.get("CS_COMM_IDLE"):
        # what parallel._exchange_records_many does with N > 1 ranks: the step's lists side by side, the configuration's index in an
        # eighth column, ONE collective (cs_comm_allgather_rows_once; CS_TWO_PHASE=1: count exchange + padded all-gather per list)
        if os.environ.get("CS_TWO_PHASE"):
            for r in rec:
                got, counts = comm.allgather_rows(np.ascontiguousarray(r, dtype=np.float64))
                assert int(counts[0]) == len(r)
        else:
            both = np.concatenate([np.concatenate([r, np.full((len(r), 1), float(i))], axis=1) for i, r in enumerate(rec)], axis=0)
            got, counts = comm.allgather_rows_once(both)
            assert int(counts[0]) == len(both)
    dcool.dev.sync(); t2 = time.perf_counter()
    ts.append((t2 - t0) * 1e3); ex.append((t2 - t1) * 1e3)
px = sum(costs[i] for i in mine)
plans = dcool.__dict__.get("_step_plans", {})
if not all(p.ok for p in plans.values()) or not plans:
    print("no step plan:", [p.why for p in plans.values()])
line = (f"{world} GPUs, rank {rank}: blocks {mine} ({px / 1e6:.1f} Mpixel of {sum(costs) / 1e6:.1f}), step {np.mean(ts[4:]):.3f} ms (median {np.median(ts[4:]):.3f}, min {min(ts):.3f}); "
      f"patterns {[len(r) for r in rec]}")
line += f"; exchange (single-rank RCCL, inside the step) median {np.median(ex[4:]):.3f} ms" if comm is not None else "; no exchange in the step"
slow = sorted(t for t in ts[4:] if t > 1.2 * np.median(ts[4:]))
if slow:
    line += f"; {len(slow)} of {len(ts) - 4} steps above 1.2 x the median: " + " ".join(f"{t:.3f}" for t in slow[-8:])
print(line, flush=True)
if comm is not None:
    comm.close()
if os.environ.get("STEPS"):
    print("steps (ms):", " ".join(f"{t:.3f}" for t in ts))
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for it in range(20):
        staged = parallel.stage_genome(dcool, [loops, borders], owned=mine)
        rec = parallel.detect_patterns(dcool, [loops, borders], owned=mine, staged=staged)
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
