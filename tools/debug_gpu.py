import sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np, scipy.sparse as sp
import chromosight_amd
from chromosight_amd.utils import detection as cud, preprocessing as cup
from oracle import pearson_oracle as orc
G = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"
def coo(g, prefix):
    return sp.coo_matrix((g[f"{prefix}_val"], (g[f"{prefix}_row"], g[f"{prefix}_col"])), shape=tuple(g[f"{prefix}_shape"]))
def report(name, got, ref, k=5):
    err = np.abs(got - ref)
    idx = np.argsort(err.ravel())[::-1][:k]
    print(name, "max", err.max())
    for t in idx:
        i, j = np.unravel_index(t, err.shape)
        if err[i, j] > 0:
            print("   ", (i, j), "got", got[i, j], "ref", ref[i, j])
for prec in ("f64", "f32"):
    chromosight_amd.set_precision(prec)
    print("=====", prec)
    g = np.load(G / "xcorr2.npz")
    for c in range(3):
        d = cud.xcorr2(g[f"sig{c}"], g["gauss_kernel"], threshold=1e-4)
        report(f"xcorr dense{c}", d, g[f"dense{c}"])
        k1 = np.ones((11, 11)) / 121
        cst = cud.xcorr2(sp.csr_matrix(g[f"sig{c}"]), k1).toarray()
        report(f"xcorr const{c}", cst, g[f"const{c}"])
    report("rect", cud.xcorr2(g["rand"], g["rect_kernel_5x9"]), g["rand_rect_5x9"])
    report("tsvd", cud.xcorr2(sp.csr_matrix(g["rand"]), chromosight_amd.kernels.loops["kernels"][0], tsvd=0.999).toarray(), g["rand_loops_tsvd999"])
    g = np.load(G / "normxcorr2_mask.npz")
    for i in range(int(g["n_intra"])):
        sig, k, valid = g[f"intra{i}_sig"], g[f"intra{i}_kernel"], g[f"intra{i}_valid"]
        md, mtol = int(g[f"intra{i}_max_dist"]), float(g[f"intra{i}_tol"])
        n = sig.shape[0]
        mask = cup.make_missing_mask((n, n), valid, valid, max_dist=md, sym_upper=True)
        c, p = cud.normxcorr2(sp.csr_matrix(sig), k, max_dist=md, sym_upper=True, full=True, missing_mask=mask, missing_tol=mtol, pval=True)
        report(f"intra{i} md={md} k={k.shape}", c.toarray(), g[f"intra{i}_corr"])
        report(f"intra{i} pval", p.toarray(), g[f"intra{i}_pval"], k=2)
    e = np.load(G / "example_blocks.npz")
    ci, pname, ki = 0, "borders", 1
    det = e[f"chr{ci}_det"]
    m = coo(e, f"chr{ci}_{pname}_prepared").tocsr()
    md = int(e[f"chr{ci}_{pname}_max_dist"])
    kern = chromosight_amd.kernels.borders["kernels"][ki]
    mask = cup.make_missing_mask(m.shape, det, det, max_dist=md, sym_upper=True)
    c, p = cud.normxcorr2(m, kern, max_dist=md, sym_upper=True, full=True, missing_mask=mask, pval=True, missing_tol=0.75)
    report("chr0 borders1", c.toarray(), coo(e, f"chr{ci}_{pname}{ki}_corr").toarray())
