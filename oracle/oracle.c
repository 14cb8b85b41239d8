/*
 * oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Plain C float64 restatement of the per-pixel definition of chromosight's normxcorr2
 * (reference chromosight/utils/detection.py:917-1131 for the arithmetic,
 * chromosight/utils/preprocessing.py:404-498 and :535-633 for the framed missing predicate).
 * It mirrors oracle/pearson_oracle.py (which is pinned against the reference's outputs by
 * tests/test_oracle_golden.py) and is itself checked against that module by the same test
 * file.  Used for parity checks at sizes where numpy is too slow, and as the `cpu_baseline`
 * ("port") leg of bench.py.  Nothing under chromosight_amd/ links or loads it.
 *
 * Build: make -C oracle   ->  oracle/liboracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int ms, ns, km, kn;
    int full, sym_upper, max_dist; /* max_dist < 0 : None */
    int masked;                    /* 0: no mask; 1: missing bins given */
    const uint8_t* miss_row;
    const uint8_t* miss_col;
} geom_t;

/* framed missing predicate for matrix coordinates (p, q), frame included */
static int missing_pred(const geom_t* g, int p, int q)
{
    int in_r = p >= 0 && p < g->ms, in_c = q >= 0 && q < g->ns;
    int d = q - p, m = 0;
    if (!g->masked) return 0;
    if (in_r && in_c) {
        m = g->miss_row[p] || g->miss_col[q];
        if (g->sym_upper) {
            int md = g->max_dist >= 0 ? g->max_dist : (g->ms < g->ns ? g->ms : g->ns);
            m = m && d >= 0 && d <= md;
        }
        if (!g->full) return m;
    } else {
        if (!g->full) return 0;
        if (g->sym_upper && g->max_dist >= 0) {
            if (q >= g->ns) m = p >= g->ms - g->max_dist - 2;
            else if (p < 0) m = (q < 0) ? 1 : (q < g->max_dist + g->kn);
            else m = 0;
        } else {
            m = 1;
        }
    }
    if (g->sym_upper) {
        int off = d + (g->kn - g->km);
        int big_k = g->km > g->kn ? g->km : g->kn;
        if (off <= -1 && off >= -big_k) m = 1;
    }
    return m;
}

static double thr(double x) { return fabs(x) < 1e-4 ? 0.0 : x; }

/*
 * sig: ms x ns row-major float64.  kernel / kernel_conv / kernel_sq: km x kn (kernel_conv and
 * kernel_sq may be NULL).  out_corr, out_nobs: ms x ns.  Returns 0.
 */
int oracle_normxcorr2(const double* sig, int ms, int ns, const double* kernel,
                      const double* kernel_conv, const double* kernel_sq, int km, int kn, int full,
                      int sym_upper, int max_dist, int masked, const uint8_t* miss_row,
                      const uint8_t* miss_col, double missing_tol, double* out_corr, double* out_nobs,
                      int n_threads)
{
    geom_t g = {ms, ns, km, kn, full, sym_upper, max_dist, masked, miss_row, miss_col};
    const int kk = km * kn;
    const double n = (double)kk;
    const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
    double ksum = 0, k2sum = 0, kmean, k2mean, kstd = 0, kvar;
    const double* kc = kernel_conv ? kernel_conv : kernel;
    double* kc2 = (double*)malloc(sizeof(double) * kk);
    int t;
    for (t = 0; t < kk; ++t) {
        ksum += kernel[t];
        k2sum += kernel[t] * kernel[t];
        kc2[t] = kernel_sq ? kernel_sq[t] : kc[t] * kc[t];
    }
    kmean = ksum / n;
    k2mean = k2sum / n;
    kvar = k2mean - kmean * kmean;
    for (t = 0; t < kk; ++t) kstd += (kernel[t] - kmean) * (kernel[t] - kmean);
    kstd = sqrt(kstd / n);
    const int cut = (int)((1.0 - missing_tol) * n);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int i = 0; i < ms; ++i) {
        for (int j = 0; j < ns; ++j) {
            double r = 0.0, nobs = n;
            int zero = 0;
            if (!full) zero = (i < kh) || (i > ms - km + kh) || (j < kw) || (j > ns - kn + kw);
            if (sym_upper && (j - i) + (full ? (kn - km) : 0) < 0) zero = 1;
            if (!zero) {
                double s1 = 0, s2 = 0, c = 0, nm = 0, km_ = 0, k2m = 0;
                for (int a = 0; a < km; ++a) {
                    int p = i - kh + a;
                    for (int b = 0; b < kn; ++b) {
                        int q = j - kw + b;
                        double v = 0.0;
                        if (p >= 0 && p < ms && q >= 0 && q < ns) v = sig[(size_t)p * ns + q];
                        s1 += v * (1.0 / n);
                        s2 += v * v * (1.0 / n);
                        c += v * (kc[a * kn + b] / n);
                        if (masked && missing_pred(&g, p, q)) {
                            nm += 1.0;
                            km_ += kc[a * kn + b];
                            k2m += kc2[a * kn + b];
                        }
                    }
                }
                double m1 = thr(s1), m2 = thr(s2), cz = thr(c), num, den;
                if (!masked) {
                    den = sqrt(m2 - m1 * m1) * kstd;
                    num = cz - m1 * kmean;
                } else if (thr(nm) == 0.0) {
                    den = sqrt((m2 - m1 * m1) * kvar);
                    num = cz - m1 * kmean;
                } else {
                    double np_ = n - nm;
                    double kmw = (ksum - thr(km_)) / np_;
                    double k2mw = (k2sum - thr(k2m)) / np_;
                    double m1w = m1 * n / np_, m2w = m2 * n / np_;
                    double dd = (m2w - m1w * m1w) * kvar;
                    dd = dd / kvar * (k2mw - kmw * kmw);
                    den = sqrt(dd);
                    if (np_ < cut) den = 0.0;
                    double o = m1w * kmean;
                    o = o * kmw * np_ / (kmean * n);
                    num = (cz - o) * n / np_;
                    nobs = np_;
                }
                r = (fabs(den) < 1e-10) ? 0.0 : num / den;
                if (!isfinite(r)) r = 0.0;
                if (r < -1.0) r = -1.0;
                if (r > 1.0) r = 1.0;
            }
            out_corr[(size_t)i * ns + j] = r;
            if (out_nobs) out_nobs[(size_t)i * ns + j] = nobs;
        }
    }
    free(kc2);
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
