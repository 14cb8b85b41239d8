"""Synthetic decoded-.cool dictionaries for scale tests and the C4 bench workload (BASELINE.md:
total bins split over 23 blocks sized like hg38 chr1-22,X; upper band of Poisson counts with a
1/(d+1) distance law, ICE-like weights with 2 % unbalanced bins, planted loop templates)."""
import numpy as np

HG38_MB = [248, 242, 198, 190, 181, 171, 159, 145, 138, 133, 135, 133, 114, 107, 102, 90, 83, 80, 58, 64, 46, 50, 156]


def genome_sizes(total_bins=200_000, chrom_sizes=None):
    if chrom_sizes is None:
        frac = np.asarray(HG38_MB, dtype=np.float64) / sum(HG38_MB)
        chrom_sizes = np.maximum((frac * total_bins).astype(np.int64), 64)
    return np.asarray(chrom_sizes, dtype=np.int64)


def make_cool(total_bins=200_000, max_dist_bins=1000, binsize=2000, seed=2, loops_per_10k=300, template=None,
              chrom_sizes=None, largest_kernel=17, only=None):
    """Decoded-.cool dictionary of a synthetic genome.  Every chromosome has its own random stream
    (seed, chromosome), so `only` (a list of chromosome indices) generates just the pixels of those
    chromosomes -- what one rank of a sharded run needs -- identical to the same chromosomes of the
    full genome.  Weights (2 % unbalanced bins) are always generated for the whole genome."""
    sizes = genome_sizes(total_bins, chrom_sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    keep = max_dist_bins + largest_kernel
    d = np.arange(keep + 1)
    lam = 200.0 / (d + 1.0)
    b1l, b2l, cl, planted = [], [], [], []
    for c, m in enumerate(sizes):
        if only is not None and c not in only:
            continue
        rng = np.random.default_rng([seed, c])
        m = int(m)
        w = min(keep + 1, m)
        planted_c = []
        if template is not None and loops_per_10k and m > 4 * template.shape[0]:
            k = template.shape[0]
            kh = k // 2
            n_loops = max(1, int(m * loops_per_10k / 10_000))
            li = rng.integers(kh + 1, m - kh - 1, size=n_loops)
            ld = rng.integers(min(k + 2, w - kh - 2), max(min(max_dist_bins, w - kh - 2), k + 3), size=n_loops)
            planted_c = [(int(i), int(i + dd)) for i, dd in zip(li, ld) if i + dd + kh < m]
        rows_l, cols_l, cnt_l = [], [], []
        chunk = 8192
        t3 = None if template is None else (template / template.max()) ** 3
        pl = np.array(planted_c, dtype=np.int64).reshape(-1, 2)
        for r0 in range(0, m, chunk):                      # chunked: a 200 000-bin block is 1.6 GB of float64
            r1 = min(m, r0 + chunk)
            boost = np.ones((r1 - r0, w))
            if pl.shape[0]:
                kh = template.shape[0] // 2
                near = pl[(pl[:, 0] + kh >= r0) & (pl[:, 0] - kh < r1)]
                for i, j in near:
                    rr = np.arange(i - kh, i + kh + 1)[:, None]
                    cc = np.arange(j - kh, j + kh + 1)[None, :]
                    diag = cc - rr
                    ok = (diag >= 0) & (diag < w) & (rr >= r0) & (rr < r1)
                    np.multiply.at(boost, (np.broadcast_to(rr - r0, diag.shape)[ok], diag[ok]), 1.0 + 3.0 * t3[ok])
            counts = rng.poisson(lam[None, :w] * boost)
            rows = np.repeat(np.arange(r0, r1), w)
            cols = rows + np.tile(np.arange(w), r1 - r0)
            flat = counts.ravel()
            ok = (cols < m) & (flat > 0)
            rows_l.append(rows[ok] + off[c])
            cols_l.append(cols[ok] + off[c])
            cnt_l.append(flat[ok].astype(np.int32))
        b1l += rows_l
        b2l += cols_l
        cl += cnt_l
        planted += [(int(off[c] + i), int(off[c] + j)) for i, j in planted_c]
    n = int(off[-1])
    wrng = np.random.default_rng([seed, 10_000])
    weight = wrng.normal(1.0, 0.05, n) * 0.07
    weight[wrng.choice(n, n // 50, replace=False)] = np.nan
    empty = np.zeros(0, dtype=np.int64)
    cool = {
        "binsize": binsize, "chrom_offset": off, "chrom_names": np.array([f"chr{c + 1}" for c in range(len(sizes))]),
        "bin1_id": np.concatenate(b1l) if b1l else empty, "bin2_id": np.concatenate(b2l) if b2l else empty,
        "count": np.concatenate(cl) if cl else empty.astype(np.int32),
        "weight": weight,
        "bin_start": np.concatenate([np.arange(s) * binsize for s in sizes]),
        "bin_end": np.concatenate([(np.arange(s) + 1) * binsize for s in sizes]),
    }
    return cool, planted


BAND_WORKLOADS = {"c3": (50_000, 233, 1), "c4p": (200_000, 1000, 2)}


def band_workload(name, rank=0, n=None):
    """The banded kernel workloads of BASELINE.md section 4 as a float32 diagonal band
    (band[i, d] = pixel (i, i + d), diagonals 0 .. max_dist + 17), already detrended-like:
    poisson(200 / (d + 1)) counts divided by their expectation, 2 % missing bins zeroed.
    Returns (band, band_w, miss uint8 flags, n, max_dist).  Shared by bench.py and the full-size
    parity tests so that both run the same map."""
    n0, max_dist, seed = BAND_WORKLOADS[name]
    n = n0 if n is None else int(n)
    keep = max_dist + 17
    rng = np.random.default_rng(seed + rank)
    band_w = keep + 1
    ld = (band_w + 63) // 64 * 64
    band = np.zeros((n, ld), dtype=np.float32)
    d = np.arange(band_w)
    lam = 200.0 / (d + 1.0)
    chunk = 4096
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        band[r0:r1, :band_w] = rng.poisson(lam, size=(r1 - r0, band_w)) / lam
    miss = np.zeros(n, dtype=np.uint8)
    miss[rng.choice(n, size=n // 50, replace=False)] = 1
    band[miss.astype(bool), :] = 0
    for r0 in range(0, n, chunk):          # chunked: the column index table of C4' would take 1.6 GB
        r1 = min(n, r0 + chunk)
        cols = np.arange(r0, r1)[:, None] + d[None, :]
        view = band[r0:r1, :band_w]
        view[cols >= n] = 0
        view[miss[np.minimum(cols, n - 1)].astype(bool)] = 0
    return band, band_w, miss, n, max_dist
