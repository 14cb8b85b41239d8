// cs_corr_tile.h -- LDS-tiled sliding-window Pearson correlation for a K x K template
// (K compile-time, odd), the hot kernel of the path.
//
// Replaces the six sparse cross-correlations + fancy-index passes of the reference's
// _normxcorr2_sparse / _normxcorr2_dense (detection.py:917-1131, 1134-1273) by one pass:
//
//   stage   : (TH+K-1) x (TW+K-1) signal tile -> LDS (dense or diagonal-band source, virtual
//             zero frame), optional missing-mask tile evaluated analytically (cs_device.h)
//   compute : every lane owns RW=2 adjacent output columns x RH rows.  For each tile row it
//             reads K+1 consecutive values with aligned ds_read_b64 (conflict free: lane l
//             reads bytes [8l, 8l+8) + const), and feeds them to the 2*RH*K FMAs of the rows
//             that contain it.  Template weights are wave-uniform -> scalar (SGPR) operands.
//             sum S and sum S^2 use shared horizontal partial sums (separable ones-kernel).
//   epilogue: Pearson formula with the reference's thresholds, coalesced stores.
//
// Work per output pixel at 17x17: 289 FMA (S*K) + ~70 VALU for the two box sums
// (714 flop algorithmic).  LDS traffic: 9 ds_read_b64 per 2*17*RH FMAs.
#pragma once
#include "cs_device.h"

namespace cs {

template <int K, int RH>
struct TileGeom {
    static constexpr int RW = 2;
    static constexpr int NWAVES = 4;
    static constexpr int TW = kWave * RW;          // 128 output columns
    static constexpr int TH = NWAVES * RH;         // output rows
    static constexpr int LH = TH + K - 1;          // staged rows
    static constexpr int LW = TW + K - 1;          // staged columns actually needed
    static constexpr int LWP = ((LW + 3) / 4) * 4; // padded to 16 B
    static constexpr int NTHREADS = NWAVES * kWave;
};

// accumulate one staged row into the per-lane accumulators of the rows that contain it.
//   v[0..K]   : K+1 consecutive tile values starting at this lane's first output column
//   W(ki,kj)  : uniform weight
template <typename TC, int K, int RH, int R>
__device__ __forceinline__ void row_update(const TC (&v)[K + 1], const TC* __restrict__ w,
                                           TC (&acc)[RH][2])
{
    constexpr int ILO = (R - (K - 1)) > 0 ? (R - (K - 1)) : 0;
    constexpr int IHI = R < (RH - 1) ? R : (RH - 1);
#pragma unroll
    for (int i = ILO; i <= IHI; ++i) {
        const int ki = R - i;
#pragma unroll
        for (int kj = 0; kj < K; ++kj) {
            const TC wk = w[ki * K + kj];
            acc[i][0] = cs_fma(v[kj], wk, acc[i][0]);
            acc[i][1] = cs_fma(v[kj + 1], wk, acc[i][1]);
        }
    }
}

// distribute a horizontal partial (h[0], h[1] for the two columns) of staged row R to the
// vertical sums of the output rows that contain it; rows shared by all outputs go to `core`.
template <typename TC, int K, int RH, int R>
__device__ __forceinline__ void vert_update(const TC (&h)[2], TC (&s)[RH][2], TC (&core)[2])
{
    constexpr int ILO = (R - (K - 1)) > 0 ? (R - (K - 1)) : 0;
    constexpr int IHI = R < (RH - 1) ? R : (RH - 1);
    if constexpr (ILO == 0 && IHI == RH - 1) {
        core[0] += h[0];
        core[1] += h[1];
    } else {
#pragma unroll
        for (int i = ILO; i <= IHI; ++i) {
            s[i][0] += h[0];
            s[i][1] += h[1];
        }
    }
}

template <typename TC, int K, int RH, int R, bool BOX>
struct RowLoop {
    using G = TileGeom<K, RH>;
    // signal rows: S*Wc accumulation + box sums
    __device__ __forceinline__ static void run(const TC* __restrict__ srow, const TC* __restrict__ w,
                                               TC (&acc)[RH][2], TC (&s1)[RH][2], TC (&s2)[RH][2],
                                               TC (&c1)[2], TC (&c2)[2])
    {
        TC v[K + 1];
        const TC* p = srow + R * G::LWP;
#pragma unroll
        for (int t = 0; t < K + 1; ++t) v[t] = p[t];
        if constexpr (BOX) {
            TC h1[2], h2[2];
            h1[0] = v[0];
            h2[0] = v[0] * v[0];
#pragma unroll
            for (int t = 1; t < K; ++t) {
                h1[0] += v[t];
                h2[0] = cs_fma(v[t], v[t], h2[0]);
            }
            h1[1] = (h1[0] - v[0]) + v[K];
            h2[1] = cs_fma(v[K], v[K], cs_fma(-v[0], v[0], h2[0]));
            vert_update<TC, K, RH, R>(h1, s1, c1);
            vert_update<TC, K, RH, R>(h2, s2, c2);
        }
        row_update<TC, K, RH, R>(v, w, acc);
        if constexpr (R + 1 < RH + K - 1)
            RowLoop<TC, K, RH, R + 1, BOX>::run(srow, w, acc, s1, s2, c1, c2);
    }
};

// mask rows: nm = sum M, ka = sum M*Wa, kb = sum M*Wb
template <typename TC, int K, int RH, int R>
struct MaskLoop {
    using G = TileGeom<K, RH>;
    __device__ __forceinline__ static void run(const uint8_t* __restrict__ mrow,
                                               const TC* __restrict__ wa, const TC* __restrict__ wb,
                                               TC (&nm)[RH][2], TC (&ka)[RH][2], TC (&kb)[RH][2])
    {
        TC v[K + 1];
        const uint8_t* p = mrow + R * G::LWP;
        unsigned any = 0;
#pragma unroll
        for (int t = 0; t < K + 1; ++t) {
            unsigned b = p[t];
            any |= b;
            v[t] = (TC)b;
        }
        // wave-uniform skip of rows without any missing pixel in this wave's span
        if (__any(any != 0)) {
            TC h[2];
            h[0] = v[0];
#pragma unroll
            for (int t = 1; t < K; ++t) h[0] += v[t];
            h[1] = (h[0] - v[0]) + v[K];
            constexpr int ILO = (R - (K - 1)) > 0 ? (R - (K - 1)) : 0;
            constexpr int IHI = R < (RH - 1) ? R : (RH - 1);
#pragma unroll
            for (int i = ILO; i <= IHI; ++i) {
                nm[i][0] += h[0];
                nm[i][1] += h[1];
            }
            row_update<TC, K, RH, R>(v, wa, ka);
            row_update<TC, K, RH, R>(v, wb, kb);
        }
        if constexpr (R + 1 < RH + K - 1)
            MaskLoop<TC, K, RH, R + 1>::run(mrow, wa, wb, nm, ka, kb);
    }
};

template <typename TC, int K, int RH, bool MASKED>
__global__ __launch_bounds__(256) void corr_tile_kernel(const CorrArgs<TC> A)
{
    using G = TileGeom<K, RH>;
    constexpr int KK = K * K;
    constexpr int KH = (K - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    TC* sS = reinterpret_cast<TC*>(smem_raw);
    uint8_t* sM = reinterpret_cast<uint8_t*>(smem_raw + sizeof(TC) * G::LH * G::LWP);
    // flag kept in the dynamic region: a static __shared__ object would shift the 16-byte
    // alignment of the tile (ds_read_b64/b128 replay when misaligned)
    int* s_flag = reinterpret_cast<int*>(smem_raw + (sizeof(TC) + 1) * G::LH * G::LWP);

    int i0, j0;
    if (!tile_origin(A, blockIdx.x, blockIdx.y, &i0, &j0)) return;
    const int tid = threadIdx.x;
    if (MASKED && tid == 0) *s_flag = 0;
    if (MASKED) __syncthreads();

    // ---- stage ------------------------------------------------------------------------
    int my_any = 0;
    for (int idx = tid; idx < G::LH * G::LWP; idx += G::NTHREADS) {
        const int tr = idx / G::LWP;
        const int tc = idx - tr * G::LWP;
        const int p = i0 - KH + tr;
        const int q = j0 - KH + tc;
        sS[idx] = load_signal(A, p, q);
        if constexpr (MASKED) {
            const bool m = (tc < G::LW) ? missing_pred(A, p, q) : false;
            sM[idx] = m ? 1 : 0;
            my_any |= m ? 1 : 0;
        }
    }
    if constexpr (MASKED) {
        if (__any(my_any) && (tid & 63) == 0) atomicOr(s_flag, 1);
    }
    __syncthreads();

    // ---- compute ----------------------------------------------------------------------
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int tc0 = lane * 2;      // first of this lane's two output columns (tile coords)
    const int tr0 = wv * RH;       // first output row of this wave (tile coords)

    TC acc[RH][2], s1[RH][2], s2[RH][2], c1[2] = {TC(0), TC(0)}, c2[2] = {TC(0), TC(0)};
#pragma unroll
    for (int i = 0; i < RH; ++i) {
        acc[i][0] = acc[i][1] = TC(0);
        s1[i][0] = s1[i][1] = TC(0);
        s2[i][0] = s2[i][1] = TC(0);
    }
    const TC* srow = sS + tr0 * G::LWP + tc0;
    RowLoop<TC, K, RH, 0, true>::run(srow, A.w, acc, s1, s2, c1, c2);

    TC nm[RH][2], ka[RH][2], kb[RH][2];
#pragma unroll
    for (int i = 0; i < RH; ++i) {
        nm[i][0] = nm[i][1] = TC(0);
        ka[i][0] = ka[i][1] = TC(0);
        kb[i][0] = kb[i][1] = TC(0);
    }
    if constexpr (MASKED) {
        if (*s_flag) {
            const uint8_t* mrow = sM + tr0 * G::LWP + tc0;
            MaskLoop<TC, K, RH, 0>::run(mrow, A.w + KK, A.w + 2 * KK, nm, ka, kb);
        }
    }

    // ---- epilogue ---------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < RH; ++i) {
        const int oi = i0 + tr0 + i;
        if (oi >= A.ms) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int oj = j0 + tc0 + c;
            if (oj >= A.ns) continue;
            const int d = oj - oi;
            if (d < A.out_lo || d > A.out_hi) continue;
            TC r, nobs = A.ks.n;
            if (pixel_forced_zero(A, oi, oj)) {
                r = TC(0);
            } else {
                r = pearson_from_sums<TC>(acc[i][c], s1[i][c] + c1[c], s2[i][c] + c2[c],
                                          nm[i][c], ka[i][c], kb[i][c], A.ks,
                                          A.mask_mode != 0, &nobs);
            }
            store_pixel(A, oi, oj, r, nobs);
        }
    }
}

}  // namespace cs

namespace cs {
template <int K, int RH, typename TC>
constexpr size_t corr_tile_smem_bytes()
{
    using G = TileGeom<K, RH>;
    return (sizeof(TC) + 1) * G::LH * G::LWP + 16;
}
}  // namespace cs
