// cs_corr_generic.hip -- runtime-size (km x kn, any odd sizes) sliding-window kernel.
//
// Serves every template the unrolled fast kernels do not cover (rectangular, 31x31 stripes,
// 81x81 centromeres, resized templates) and plain xcorr2 (reference detection.py:595-804).
// One lane = one output column, RG rows per lane; the signal (and mask) tile is staged in LDS;
// weights are wave-uniform loads.  Straightforward on purpose: it is also the in-library
// cross-check of the fast kernels.
#include "cs_device.h"
#include "cs_launch.h"

namespace cs {

constexpr int GEN_TW = 64;   // output columns per block (= lanes per wave)
constexpr int GEN_RG = 4;    // output rows per lane
constexpr int GEN_NW = 4;    // waves per block
constexpr int GEN_TH = GEN_RG * GEN_NW;

template <typename TC>
__global__ __launch_bounds__(256) void corr_generic_kernel(const CorrArgs<TC> A)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int km = A.km, kn = A.kn;
    const int kh = (km - 1) / 2, kw = (kn - 1) / 2;
    const int LH = GEN_TH + km - 1;
    const int LW = GEN_TW + kn - 1;
    const int LWP = (LW + 3) & ~3;
    TC* sS = reinterpret_cast<TC*>(smem_raw);
    uint8_t* sM = reinterpret_cast<uint8_t*>(smem_raw + sizeof(TC) * (size_t)LH * LWP);

    int i0, j0;
    if (!tile_origin(A, blockIdx.x, blockIdx.y, &i0, &j0)) return;
    const int tid = threadIdx.x;
    const bool masked = A.mask_mode != 0;

    for (int idx = tid; idx < LH * LWP; idx += 256) {
        const int tr = idx / LWP;
        const int tc = idx - tr * LWP;
        const int p = i0 - kh + tr;
        const int q = j0 - kw + tc;
        sS[idx] = load_signal(A, p, q);
        if (masked) sM[idx] = (tc < LW && missing_pred(A, p, q)) ? 1 : 0;
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int tr0 = wv * GEN_RG;

    TC cs_[GEN_RG], s1[GEN_RG], s2[GEN_RG], nm[GEN_RG], ka[GEN_RG], kb[GEN_RG];
#pragma unroll
    for (int i = 0; i < GEN_RG; ++i) cs_[i] = s1[i] = s2[i] = nm[i] = ka[i] = kb[i] = TC(0);

    const int kk = km * kn;
    for (int ki = 0; ki < km; ++ki) {
        // two-level summation (per template row, then across rows) keeps the float32 rounding
        // of the box sums at ~sqrt(kn) + sqrt(km) ulps instead of sqrt(km * kn)
        TC rc[GEN_RG], r1[GEN_RG], r2[GEN_RG];
#pragma unroll
        for (int i = 0; i < GEN_RG; ++i) rc[i] = r1[i] = r2[i] = TC(0);
        for (int kj = 0; kj < kn; ++kj) {
            const TC wc = A.w[ki * kn + kj];
#pragma unroll
            for (int i = 0; i < GEN_RG; ++i) {
                const TC v = sS[(tr0 + i + ki) * LWP + lane + kj];
                rc[i] = cs_fma(v, wc, rc[i]);
                r1[i] += v;
                r2[i] = cs_fma(v, v, r2[i]);
            }
            if (masked) {
                const TC wa = A.w[kk + ki * kn + kj];
                const TC wb = A.w[2 * kk + ki * kn + kj];
#pragma unroll
                for (int i = 0; i < GEN_RG; ++i) {
                    const TC m = (TC)sM[(tr0 + i + ki) * LWP + lane + kj];
                    nm[i] += m;
                    ka[i] = cs_fma(m, wa, ka[i]);
                    kb[i] = cs_fma(m, wb, kb[i]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < GEN_RG; ++i) {
            cs_[i] += rc[i];
            s1[i] += r1[i];
            s2[i] += r2[i];
        }
    }

#pragma unroll
    for (int i = 0; i < GEN_RG; ++i) {
        const int oi = i0 + tr0 + i;
        const int oj = j0 + lane;
        if (oi >= A.row_end || oj >= A.ns) continue;
        const int d = oj - oi;
        if (d < A.out_lo || d > A.out_hi) continue;
        TC r, nobs = A.ks.n;
        if (pixel_forced_zero(A, oi, oj)) {
            r = TC(0);
        } else if (A.xcorr_only) {
            r = cs_[i];
            if (cs_abs(r) < A.ks.thr) r = TC(0);
        } else {
            r = pearson_from_sums<TC>(cs_[i], s1[i], s2[i], nm[i], ka[i], kb[i], A.ks, masked, &nobs);
            if constexpr (sizeof(TC) == 4) r = cand_upper_from_sums(r, cs_[i], s1[i], s2[i], nm[i], ka[i], kb[i], A.ks, masked);
        }
        store_pixel(A, oi, oj, r, nobs);
    }
}

template <typename TC>
static int launch_generic(const CorrArgs<TC>& A, hipStream_t stream)
{
    const int LH = GEN_TH + A.km - 1;
    const int LWP = (GEN_TW + A.kn - 1 + 3) & ~3;
    const size_t smem = (sizeof(TC) + 1) * (size_t)LH * LWP + 16;
    if (smem > 160 * 1024) return -3;
    auto kern = corr_generic_kernel<TC>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(A.tiles_x, A.tiles_y), block(256);
    hipLaunchKernelGGL(kern, grid, block, smem, stream, A);
    return (int)hipGetLastError();
}

int launch_corr_generic_f32(const CorrArgs<float>& A, hipStream_t s) { return launch_generic<float>(A, s); }
int launch_corr_generic_f64(const CorrArgs<double>& A, hipStream_t s) { return launch_generic<double>(A, s); }
void corr_generic_tile(int, int, int* tw, int* th)
{
    *tw = GEN_TW;
    *th = GEN_TH;
}

}  // namespace cs
