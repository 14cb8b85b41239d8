"""What a live RCCL communicator costs the steps of a rank's share beside it (tools/time_rank_share.py found 0.3 ms per step): the same
steps before it exists, while it idles, after one exchange, after it is destroyed; CPU affinity and thread count each time."""
import os, sys, time, copy
sys.path.insert(0, os.getcwd())
import numpy as np
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import genome_sizes, make_cool
sizes = genome_sizes(200_000)
costs = [parallel.block_cost((int(n), int(n)), 1000, False) for n in sizes]
mine = parallel.assign_blocks(costs, 8)[1]
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
cool, _ = make_cool(200_000, 1000, 2000, seed=2, template=template, only=mine)
dcool = pipeline.DeviceCool(cool)
loops = copy.deepcopy(ck.loops); loops["max_dist"] = 2_000_000
borders = copy.deepcopy(ck.borders)
def steps(tag, n=24):
    ts = []
    for it in range(n):
        dcool.dev.sync(); t0 = time.perf_counter()
        parallel.genome_step(dcool, [loops, borders], owned=mine)
        dcool.dev.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag}: median {np.median(ts[4:]):.3f} ms, min {min(ts):.3f}; affinity {len(os.sched_getaffinity(0))} cpus; threads {len(os.listdir('/proc/self/task'))}", flush=True)
if os.environ.get("PROBE_BASE"):
    steps("nothing of RCCL touched")
if os.environ.get("PROBE_LOAD"):
    from chromosight_amd._lib import load_library
    print("cs_comm_available:", load_library().cs_comm_available(), flush=True)
    steps("librccl loaded (cs_comm_available)")
if os.environ.get("PROBE_UID"):
    uid = parallel.NativeComm.unique_id()
    steps("ncclGetUniqueId called")
comm = None
for attempt in range(4):
    try:
        comm = parallel.NativeComm(0, 0, 1, parallel.NativeComm.unique_id())
        break
    except RuntimeError as exc:
        print("communicator:", exc, flush=True)
        time.sleep(1.0)
if comm is None:
    sys.exit(0)
steps("communicator alive, idle")
comm.allgather_rows(np.zeros((10, 8)))
steps("after one exchange, idle")
comm.close()
steps("communicator destroyed")
steps("communicator destroyed, again")
for k in ("HSA_", "NCCL_", "RCCL_", "HIP_", "GPU_"):
    print({a: b for a, b in os.environ.items() if a.startswith(k)})
