import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosight_amd._lib import FOCUS_DTYPE
from chromosight_amd.utils import detection as cid
import chromosight_amd.kernels as ck
for n,nb in ((1000,23),(2000,23),(2047,23),(2048,23),(4000,23),(8000,23),(8000,1)):
    rng=np.random.default_rng(0)
    rec=np.zeros(n,FOCUS_DTYPE)
    rec["bin1"]=rng.integers(0,8000,n); rec["bin2"]=rec["bin1"]+rng.integers(0,900,n)
    rec["inside"]=1; rec["n_zero"]=rng.integers(0,20,n); rec["n_missing"]=rng.integers(0,200,n)
    rec["score"]=rng.uniform(0.3,0.9,n); rec["n_obs"]=289-rec["n_missing"]
    counts=np.full(nb,n//nb); counts[-1]+=n-counts.sum()
    blocks=[types.SimpleNamespace(shape=(8700,8700),max_dist=1000) for _ in range(nb)]
    kspec=types.SimpleNamespace(km=17,kn=17)
    for _ in range(5): cid.accept_many(blocks,rec,None,counts,kspec,ck.loops,merged=True)
    t0=time.perf_counter()
    for _ in range(100): out=cid.accept_many(blocks,rec,None,counts,kspec,ck.loops,merged=True)
    print(n,nb,(time.perf_counter()-t0)*10,"ms")
