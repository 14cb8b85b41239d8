#!/usr/bin/env python3
"""The Python call surface from several threads at once (the reference's functions are plain numpy / scipy and can be called
that way): four threads, each its own maps, templates and masks, through the process-wide Device -- results must equal the
single-threaded ones.  python tools/stress_api_threads.py [rounds]"""
import os, sys, threading
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import chromosight_amd.kernels as ck
from chromosight_amd.utils import detection as cud
from chromosight_amd.utils import preprocessing as cup

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
jobs = []
for t in range(4):
    rng = np.random.default_rng(100 + t)
    n = 600 + 100 * t
    a = np.triu(rng.gamma(4, 0.25, size=(n, n)))
    kern = np.asarray([ck.loops, ck.borders, ck.hairpins, ck.loops][t]["kernels"][0], dtype=np.float64)
    valid = np.flatnonzero(rng.random(n) > 0.03)
    mask = cup.make_missing_mask((n, n), valid, valid, max_dist=200, sym_upper=True)
    miss = np.ones(n, bool); miss[valid] = False
    a[miss, :] = 0; a[:, miss] = 0
    jobs.append((sp.csr_matrix(a), kern, mask, rng.gamma(4, 0.25, size=(500 + 50 * t, 640)).astype(np.float64)))

def work(job):
    s, kern, mask, dense = job
    c1, p1 = cud.normxcorr2(s, kern, max_dist=200, sym_upper=True, full=True, missing_mask=mask, missing_tol=0.6, pval=True)
    c2, _ = cud.normxcorr2(dense, kern, full=False)
    return c1.toarray(), p1.toarray(), c2

want = [work(j) for j in jobs]
bad = []
def thread(k):
    for _ in range(rounds):
        got = work(jobs[k])
        for g, w in zip(got, want[k]):
            if not np.array_equal(g, w, equal_nan=True):
                bad.append(k)
                return
ths = [threading.Thread(target=thread, args=(k,)) for k in range(4)]
[t.start() for t in ths]; [t.join() for t in ths]
print("threads whose results differed from the single-threaded ones:", sorted(set(bad)) or "none", f"({rounds} rounds x 4 threads)")
sys.exit(1 if bad else 0)
