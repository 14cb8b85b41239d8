#!/usr/bin/env python3
"""Random genomes (chromosome sizes from a few bins to thousands, random scanning distance) through stage_genome +
detect_patterns (loops, borders, hairpins) against the CPU pipeline of the oracles: tables bit-exact in coordinates and order,
scores <= 1e-9.  python tools/fuzz_genomes.py [genomes]"""
import copy, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import chromosight_amd.kernels as ck
from chromosight_amd import parallel, pipeline
from tools.synthetic_genome import make_cool
import test_gpu_device_pipeline as T

genomes = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(os.environ.get("CS_FUZZ_SEED", "21")))
template = np.asarray(ck.loops["kernels"][0], dtype=np.float64)
total_patterns = 0
for g in range(genomes):
    n_chrom = int(rng.integers(2, 9))
    sizes = [int(x) for x in rng.choice([12, 19, 36, 60, 150, 400, 900, 2500, 4000], size=n_chrom)]
    md = int(rng.choice([1, 8, 40, 150, 400]))
    cool, _ = make_cool(sum(sizes), md, 2000, seed=100 + g, template=template, chrom_sizes=sizes)
    dcool = pipeline.DeviceCool(cool)
    loops = copy.deepcopy(ck.loops); loops["max_dist"] = md * 2000
    cfgs = [loops, copy.deepcopy(ck.borders), copy.deepcopy(ck.hairpins)]
    staged = parallel.stage_genome(dcool, cfgs)
    recs = parallel.detect_patterns(dcool, cfgs, staged=staged)
    for cfg, rec in zip(cfgs, recs):
        mdc = max(cfg["max_dist"] // 2000, 1)
        kernels = [np.asarray(k, dtype=np.float64) for k in cfg["kernels"]]
        for ci in range(dcool.n_chrom):
            want = T.oracle_block_tables(cool, ci, cfg, mdc, kernels, 2000)
            for ki, tab in enumerate(want):
                got = rec[(rec[:, 0] == ci) & (rec[:, 5] == ki)]
                assert got.shape[0] == tab.shape[0], (g, sizes, md, cfg["name"], ci, ki, got.shape[0], tab.shape[0])
                if tab.shape[0]:
                    assert np.array_equal(got[:, 1:3], tab[:, :2]), (g, sizes, md, cfg["name"], ci, ki)
                    assert np.abs(got[:, 3] - tab[:, 2]).max() < 1e-9, (g, sizes, md, cfg["name"], ci, ki)
                total_patterns += tab.shape[0]
    print(f"genome {g}: sizes {sizes}, max_dist {md}: ok", flush=True)
print(f"{genomes} genomes, {total_patterns} patterns equal the oracle pipeline")
