// cs_corr_stream.h -- the hot kernel: streaming sliding-window Pearson correlation for a
// K x K template (K odd, compile-time), without mask (MODE 0), with any missing mask (MODE 1) or
// with the factorised per-bin mask (MODE 2).
//
// Replaces the reference's sparse/dense cross-correlations of normxcorr2 (detection.py:1000-1085,
// 1213-1220: signal*1, signal^2*1, signal*K and the three mask correlations) by one pass.
//
// Design (gfx950, wave64; DESIGN.md section 4.1 has the measurements behind each choice):
//   * One WAVE owns a strip of 128 output columns (2 adjacent columns per lane) and walks down
//     strip_h output rows.  No block-level barrier: each wave has a private LDS ring of staged
//     rows.  Lane l reads its K+1 consecutive values of a row with aligned ds_read_b64.
//   * Rotating accumulators: X[s] is the partial sum of an output row over template rows 0..s.
//     Each loop iteration consumes TWO staged rows (A = t, B = t+1) per template row s:
//         y_s = X[s-1] + A * W[s]      z_s = y_(s-1) + B * W[s]      X'[s] = z_s
//     so one scalar load of W[s] feeds 2*K packed FMAs in two independent dependency chains,
//     y_(K-1) and z_(K-1) complete two output rows, and every staged value is loaded once and
//     used K times per column from registers.
//   * Template weights are wave-uniform: one s_load of a template row (K scalars) feeds K packed
//     FMAs (v_pk_fma_f32 on the lane's (column, column+1) accumulator pair with an SGPR weight,
//     the fastest FP32 FMA form on this chip, tools/ubench/fma_rate.hip).  The row is held twice
//     in registers, as even-aligned pairs (v[2m], v[2m+1]) and odd-aligned pairs (v[2m+1], v[2m+2]),
//     so that no operand needs a move.  289 weights never fit the 102 SGPRs; they are re-streamed
//     from the scalar cache once per pair of staged rows.
//   * SYM: vertically symmetric templates share each horizontal row product between two template
//     rows (steps2_rec).
//   * Box sums (sum S, sum S^2 over the window) are separable: horizontal K-sums of the entering
//     row minus those of the leaving row (kept in an LDS cache) go into float64 running sums.
//
// Work per output pixel at K = 17: 289 FMA (169 packed ops per column pair when SYM) + box sums and
// epilogue => FP32-FMA bound; HBM traffic ~ (1 + 16/128)(1 + 16/strip_h) * 4 B
// in + 4 B out per pixel.
#pragma once
#include "cs_device.h"

namespace cs {

template <int K>
struct StreamGeom {
    static constexpr int RW = 2;                    // output columns per lane
    static constexpr int TW = kWave * RW;           // 128 output columns per wave
    static constexpr int LW = TW + K - 1;           // staged columns per row
    static constexpr int LWP = ((LW + 3) / 4) * 4;  // row pitch in elements (16-byte multiple)
    static constexpr int RING = K + 4;              // rows t-K .. t+3 are live (two rows per iteration)
    // Float32 kernels without the -0.0 mask encoding keep the horizontal box sums of the last K
    // staged rows (4 floats per lane and row) instead of the rows themselves: the row leaving
    // the window is then one 16-byte LDS read instead of a row reload and 40 VALU ops, and the
    // data ring shrinks to the 4 rows in flight.
    static constexpr int DATA_ROWS_HC = 4;
    static constexpr int HC_FLOATS = K * kWave * 4;
#ifndef CS_NWAVES
#define CS_NWAVES 4
#endif
    static constexpr int NWAVES = CS_NWAVES;        // independent waves per workgroup
    static constexpr int ROWS = RING + 1;           // LDS rows per wave: the ring + the column-flag row (MODE 2)
    // elements of LDS per wave
    static constexpr int wave_elems(bool hcache) { return hcache ? (DATA_ROWS_HC + 1) * LWP + HC_FLOATS : ROWS * LWP; }
};

// weights are read through the constant address space so that the (wave-uniform) loads are
// selected as scalar s_load even though the kernel also stores to global memory
template <typename TC>
using ConstPtr = const __attribute__((address_space(4))) TC*;

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Missing pixels are carried inside the staged signal as -0.0: the reference guarantees that a
// missing pixel holds no signal (check_missing_mask, preprocessing.py:501-532), -0.0 adds nothing
// to any window sum, and the mask is recovered from the bit pattern, so the masked kernel needs no
// second LDS ring.  Genuine zeros (either sign) are staged as +0.0.
template <typename TC>
__device__ __forceinline__ TC encode_pixel(TC v, bool missing)
{
    v = (v == TC(0)) ? TC(0) : v;
    return missing ? -TC(0) : v;
}
__device__ __forceinline__ float missing_flag(float v) { return (__float_as_uint(v) == 0x80000000u) ? 1.0f : 0.0f; }
__device__ __forceinline__ double missing_flag(double v)
{
    return ((unsigned long long)__double_as_longlong(v) == 0x8000000000000000ull) ? 1.0 : 0.0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x8 __attribute__((ext_vector_type(8)));

// One template row of weights in scalar registers, loaded by hand-placed s_load so that the
// load of the NEXT row is in flight while the current row's FMAs issue.  hipcc cannot express
// this by itself: it either hoists all 289 loop-invariant loads out of the row loop (and spills
// them) or waits for each load right where it is issued.  The loads over-read up to 15 dwords
// past a row; the weight buffer is allocated with that slack (cs_api.cpp upload_weights).
template <typename TC, int K>
struct WRow;

template <int K>
struct WRow<float, K> {
    f32x16 a;
    f32x2 b;   // weight 16 (+ one over-read dword)
    template <int BYTE_OFF>
    __device__ __forceinline__ void issue(unsigned long long base)
    {
        if constexpr (K > 16)
            asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4"
                         : "=&s"(a), "=&s"(b) : "s"(base), "n"(BYTE_OFF), "n"(BYTE_OFF + 64));
        else
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(a) : "s"(base), "n"(BYTE_OFF));
    }
    // same load with a run-time (wave-uniform) byte offset
    __device__ __forceinline__ void issue_at(unsigned long long base, unsigned off)
    {
        if constexpr (K > 16) {
            const unsigned off2 = off + 64;
            asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx2 %1, %2, %4"
                         : "=&s"(a), "=&s"(b) : "s"(base), "s"(off), "s"(off2));
        } else {
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(a) : "s"(base), "s"(off));
        }
    }
    __device__ __forceinline__ void wait()
    {
        if constexpr (K > 16) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    }
    // aligned scalar pair holding weight kj (in its low half for even kj, high half for odd kj)
    template <int KJ>
    __device__ __forceinline__ f32x2 pair() const
    {
        if constexpr (KJ >= 16) return b;
        else return __builtin_shufflevector(a, a, (KJ & ~1), (KJ & ~1) + 1);
    }
};

// acc (+)= x * w broadcast, w = low (even KJ) or high (odd KJ) half of a scalar register pair.
// Written as asm so that the half is selected with op_sel instead of an s_mov + s_nop per weight.
template <int KJ>
__device__ __forceinline__ f32x2 pk_fma_w(f32x2 x, f32x2 wpair, f32x2 acc)
{
    f32x2 out;
    if constexpr ((KJ & 1) == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(out) : "v"(x), "s"(wpair), "v"(acc));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(out) : "v"(x), "s"(wpair), "v"(acc));
    return out;
}

template <int KJ>
__device__ __forceinline__ f32x2 pk_mul_w(f32x2 x, f32x2 wpair)
{
    f32x2 out;
    if constexpr ((KJ & 1) == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(out) : "v"(x), "s"(wpair));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(out) : "v"(x), "s"(wpair));
    return out;
}

template <int K>
struct WRow<double, K> {
    f64x8 a, b;
    double c;
    template <int BYTE_OFF>
    __device__ __forceinline__ void issue(unsigned long long base)
    {
        if constexpr (K > 16)
            asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx2 %2, %3, %6"
                         : "=&s"(a), "=&s"(b), "=&s"(c)
                         : "s"(base), "n"(BYTE_OFF), "n"(BYTE_OFF + 64), "n"(BYTE_OFF + 128));
        else if constexpr (K > 8)
            asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"
                         : "=&s"(a), "=&s"(b) : "s"(base), "n"(BYTE_OFF), "n"(BYTE_OFF + 64));
        else
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(a) : "s"(base), "n"(BYTE_OFF));
    }
    __device__ __forceinline__ void issue_at(unsigned long long base, unsigned off)
    {
        const unsigned off2 = off + 64, off3 = off + 128;
        if constexpr (K > 16)
            asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx2 %2, %3, %6"
                         : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(base), "s"(off), "s"(off2), "s"(off3));
        else if constexpr (K > 8)
            asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"
                         : "=&s"(a), "=&s"(b) : "s"(base), "s"(off), "s"(off2));
        else
            asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(a) : "s"(base), "s"(off));
    }
    __device__ __forceinline__ void wait()
    {
        if constexpr (K > 16) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c));
        else if constexpr (K > 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a));
    }
    __device__ __forceinline__ double get(int kj) const
    {
        return kj < 8 ? a[kj < 8 ? kj : 0] : (kj < 16 ? b[(kj >= 8 && kj < 16) ? kj - 8 : 0] : c);
    }
};

// ---- blocked packed FMAs ---------------------------------------------------------------------
// hipcc puts an s_nop between any two dependent inline-asm statements (it cannot see that they are
// plain VALU ops), which cost one issue slot per two packed FMAs when every FMA was its own asm
// statement.  A block therefore covers up to 4 weight pairs (8 template columns) of BOTH chains:
//   y += Ae[i] * w[i].lo;  z += Be[i] * w[i].lo;  y += Ao[i] * w[i].hi;  z += Bo[i] * w[i].hi
// plus, at the end of a template row of odd length, the last (even) column.
#define CS_PK_EVEN(acc, x, w) "v_pk_fma_f32 " acc ", " x ", " w ", " acc " op_sel_hi:[1,0,1]\n\t"
#define CS_PK_ODD(acc, x, w) "v_pk_fma_f32 " acc ", " x ", " w ", " acc " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define CS_PK_MUL(acc, x, w) "v_pk_mul_f32 " acc ", " x ", " w " op_sel_hi:[1,0]\n\t"
#define CS_PAIR(i)                                                                       \
    CS_PK_EVEN("%[y]", "%[ae" #i "]", "%[w" #i "]") CS_PK_EVEN("%[z]", "%[be" #i "]", "%[w" #i "]") \
    CS_PK_ODD("%[y]", "%[ao" #i "]", "%[w" #i "]") CS_PK_ODD("%[z]", "%[bo" #i "]", "%[w" #i "]")
#define CS_PAIR_FIRST(i)                                                                 \
    CS_PK_MUL("%[y]", "%[ae" #i "]", "%[w" #i "]") CS_PK_MUL("%[z]", "%[be" #i "]", "%[w" #i "]")   \
    CS_PK_ODD("%[y]", "%[ao" #i "]", "%[w" #i "]") CS_PK_ODD("%[z]", "%[bo" #i "]", "%[w" #i "]")
#define CS_TAIL CS_PK_EVEN("%[y]", "%[aet]", "%[wt]") CS_PK_EVEN("%[z]", "%[bet]", "%[wt]")
#define CS_IN(i, m) [ae##i] "v"(A.e[m]), [be##i] "v"(B.e[m]), [ao##i] "v"(A.o[m]), [bo##i] "v"(B.o[m]), [w##i] "s"(wk.template pair<2 * (m)>())
#define CS_IN_TAIL(m) [aet] "v"(A.e[m]), [bet] "v"(B.e[m]), [wt] "s"(wk.template pair<2 * (m)>())

// pairs M0 .. M0+NPAIR-1 of a template row; TAIL: also the last (even) column K-1;
// FIRST: the accumulators start at zero (first template row), so the first op is a multiply
template <int M0, int NPAIR, bool TAIL, bool FIRST, typename ROW, typename W>
__device__ __forceinline__ void pk_block(const ROW& A, const ROW& B, const W& wk, f32x2& y, f32x2& z)
{
    constexpr int MT = M0 + NPAIR;   // pair index holding column K-1 when TAIL
    if constexpr (!FIRST) {
        if constexpr (NPAIR == 4 && !TAIL)
            asm(CS_PAIR(0) CS_PAIR(1) CS_PAIR(2) CS_PAIR(3) : [y] "+v"(y), [z] "+v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN(2, M0 + 2), CS_IN(3, M0 + 3));
        else if constexpr (NPAIR == 4 && TAIL)
            asm(CS_PAIR(0) CS_PAIR(1) CS_PAIR(2) CS_PAIR(3) CS_TAIL : [y] "+v"(y), [z] "+v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN(2, M0 + 2), CS_IN(3, M0 + 3), CS_IN_TAIL(MT));
        else if constexpr (NPAIR == 3 && TAIL)
            asm(CS_PAIR(0) CS_PAIR(1) CS_PAIR(2) CS_TAIL : [y] "+v"(y), [z] "+v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN(2, M0 + 2), CS_IN_TAIL(MT));
        else if constexpr (NPAIR == 2 && TAIL)
            asm(CS_PAIR(0) CS_PAIR(1) CS_TAIL : [y] "+v"(y), [z] "+v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN_TAIL(MT));
        else if constexpr (NPAIR == 1 && TAIL)
            asm(CS_PAIR(0) CS_TAIL : [y] "+v"(y), [z] "+v"(z) : CS_IN(0, M0), CS_IN_TAIL(MT));
        else
            static_assert(NPAIR == 4, "unsupported packed block shape");
    } else {
        if constexpr (NPAIR == 4 && !TAIL)
            asm(CS_PAIR_FIRST(0) CS_PAIR(1) CS_PAIR(2) CS_PAIR(3) : [y] "=&v"(y), [z] "=&v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN(2, M0 + 2), CS_IN(3, M0 + 3));
        else if constexpr (NPAIR == 4 && TAIL)
            asm(CS_PAIR_FIRST(0) CS_PAIR(1) CS_PAIR(2) CS_PAIR(3) CS_TAIL : [y] "=&v"(y), [z] "=&v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN(2, M0 + 2), CS_IN(3, M0 + 3), CS_IN_TAIL(MT));
        else if constexpr (NPAIR == 3 && TAIL)
            asm(CS_PAIR_FIRST(0) CS_PAIR(1) CS_PAIR(2) CS_TAIL : [y] "=&v"(y), [z] "=&v"(z)
                : CS_IN(0, M0), CS_IN(1, M0 + 1), CS_IN(2, M0 + 2), CS_IN_TAIL(MT));
        else
            static_assert(NPAIR == 4, "unsupported packed block shape");
    }
}

// one template row (K odd) for both chains, in blocks of at most 4 weight pairs
template <int K, bool FIRST, typename ROW, typename W>
__device__ __forceinline__ void pk_row(const ROW& A, const ROW& B, const W& wk, f32x2& y, f32x2& z)
{
    static_assert(K % 2 == 1 && K >= 7 && K <= 17, "packed row blocks cover odd K in 7..17");
    constexpr int NP = K / 2;
    if constexpr (NP <= 4) {
        pk_block<0, NP, true, FIRST>(A, B, wk, y, z);
    } else {
        pk_block<0, 4, false, FIRST>(A, B, wk, y, z);
        pk_block<4, NP - 4, true, false>(A, B, wk, y, z);
    }
}

// accumulator of the lane's two adjacent output columns
template <typename TC>
struct Pair2 {
    TC x, y;
};
template <typename TC>
struct AccSel {
    using type = Pair2<TC>;
};
template <>
struct AccSel<float> {
    using type = f32x2;
};
template <typename TC>
using acc_t = typename AccSel<TC>::type;

__device__ __forceinline__ void pin_acc(f32x2& a) { asm volatile("" : "+v"(a)); }
template <typename TC>
__device__ __forceinline__ void pin_acc(Pair2<TC>& a)
{
    asm volatile("" : "+v"(a.x), "+v"(a.y));
}

template <typename TC>
__device__ __forceinline__ acc_t<TC> acc_zero()
{
    acc_t<TC> a;
    a.x = TC(0);
    a.y = TC(0);
    return a;
}

// Staged row of one lane: K+1 consecutive values v[0..K] starting at its first output column.
template <typename TC, int K>
struct RowRegs {
    TC v[K + 1];
    __device__ __forceinline__ void load(const TC* __restrict__ p)
    {
#pragma unroll
        for (int t = 0; t < K + 1; ++t) v[t] = p[t];
    }
    // a += (v[KJ], v[KJ+1]) * w[KJ]
    template <int KJ>
    __device__ __forceinline__ void fma(const WRow<TC, K>& wk, acc_t<TC>& a) const
    {
        a.x = cs_fma(v[KJ], wk.get(KJ), a.x);
        a.y = cs_fma(v[KJ + 1], wk.get(KJ), a.y);
    }
    __device__ __forceinline__ TC at(int t) const { return v[t]; }
    __device__ __forceinline__ void to_missing_flags()
    {
#pragma unroll
        for (int t = 0; t < K + 1; ++t) v[t] = missing_flag(v[t]);
    }
};

// float32: the row is kept as even pairs (v[2m], v[2m+1]) and odd pairs (v[2m+1], v[2m+2]) so
// that every packed FMA reads an aligned VGPR pair
template <int K>
struct RowRegs<float, K> {
    static constexpr int NE = (K + 2) / 2;   // even pairs cover v[0 .. K]
    static constexpr int NO = K / 2;         // odd pairs cover v[1 .. K-1]
    f32x2 e[NE];
    f32x2 o[NO];
    __device__ __forceinline__ void load(const float* __restrict__ p)
    {
#pragma unroll
        for (int m = 0; m < NE; ++m) e[m] = *reinterpret_cast<const f32x2*>(p + 2 * m);
#pragma unroll
        for (int m = 0; m < NO; ++m) {
            o[m].x = p[2 * m + 1];
            o[m].y = p[2 * m + 2];
        }
    }
    // a += (v[KJ], v[KJ+1]) * w[KJ]: one v_pk_fma_f32 on an aligned pair
    template <int KJ>
    __device__ __forceinline__ void fma(const WRow<float, K>& wk, f32x2& a) const
    {
        if constexpr (KJ & 1) a = pk_fma_w<KJ>(o[KJ >> 1], wk.template pair<KJ>(), a);
        else a = pk_fma_w<KJ>(e[KJ >> 1], wk.template pair<KJ>(), a);
    }
    __device__ __forceinline__ float at(int t) const { return (t & 1) ? e[t >> 1].y : e[t >> 1].x; }
    __device__ __forceinline__ void to_missing_flags()
    {
#pragma unroll
        for (int m = 0; m < NE; ++m) {
            e[m].x = missing_flag(e[m].x);
            e[m].y = missing_flag(e[m].y);
        }
#pragma unroll
        for (int m = 0; m < NO; ++m) {
            o[m].x = e[m].y;
            o[m].y = e[m + 1].x;
        }
    }
};

// horizontal K-sums of one staged row for this lane's two columns.  Four interleaved partial
// sums keep the dependency chains short (an in-order wave with one or two co-resident waves
// cannot hide a 16-deep chain of dependent adds); the entering and the leaving row use the same
// association, so the sliding difference stays exact.
// float32 rows are held as aligned pairs: packed adds / FMAs over the (K+1)/2 pairs give the sums of
// v[0..K] split by parity; the two windows drop v[K] and v[0] respectively.  Half the VALU ops of
// the scalar form below.  (Any fixed association is fine: each row's sums are computed once.)
template <int K>
__device__ __forceinline__ void row_box_packed(const RowRegs<float, K>& r, float (&h1)[2], float (&h2)[2])
{
    static_assert(K % 2 == 1, "pairs cover v[0..K] exactly for odd K");
    constexpr int NE = (K + 1) / 2;
    f32x2 a = r.e[0], b = r.e[1];
    f32x2 qa = r.e[0] * r.e[0], qb = r.e[1] * r.e[1];
#pragma unroll
    for (int m = 2; m < NE; ++m) {
        if (m & 1) {
            b += r.e[m];
            qb = __builtin_elementwise_fma(r.e[m], r.e[m], qb);
        } else {
            a += r.e[m];
            qa = __builtin_elementwise_fma(r.e[m], r.e[m], qa);
        }
    }
    a += b;
    qa += qb;
    const float t1 = a.x + a.y, t2 = qa.x + qa.y;      // sums over v[0..K]
    const float v0 = r.e[0].x, vk = r.e[NE - 1].y;
    h1[0] = t1 - vk;
    h1[1] = t1 - v0;
    h2[0] = fmaf(-vk, vk, t2);
    h2[1] = fmaf(-v0, v0, t2);
}

template <typename TC, int K>
__device__ __forceinline__ void row_box(const RowRegs<TC, K>& r, TC (&h1)[2], TC (&h2)[2])
{
    TC p1[4], p2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        p1[q] = (q < K) ? r.at(q) : TC(0);
        p2[q] = (q < K) ? r.at(q) * r.at(q) : TC(0);
    }
#pragma unroll
    for (int t = 4; t < K; ++t) {
        p1[t & 3] += r.at(t);
        p2[t & 3] = cs_fma(r.at(t), r.at(t), p2[t & 3]);
    }
    h1[0] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
    h2[0] = (p2[0] + p2[1]) + (p2[2] + p2[3]);
    h1[1] = (h1[0] - r.at(0)) + r.at(K);
    h2[1] = cs_fma(r.at(K), r.at(K), cs_fma(-r.at(0), r.at(0), h2[0]));
}

// Two staged rows against template row S (ascending S): see the header comment.  Compile-time
// recursion: every s_load has an immediate offset and the two scalar weight buffers alternate
// statically.  The load of W[S+1] is issued right after the wait for W[S] and is in flight during
// the 2*K packed FMAs of this step.
template <int KJ, typename TC, int K>
__device__ __forceinline__ void two_chains(const RowRegs<TC, K>& A, const RowRegs<TC, K>& B, const WRow<TC, K>& wk,
                                           acc_t<TC>& y, acc_t<TC>& z)
{
    if constexpr (KJ < K) {
        A.template fma<KJ>(wk, y);
        B.template fma<KJ>(wk, z);
        two_chains<KJ + 1, TC, K>(A, B, wk, y, z);
    }
}

// single chain: a += sum_kj (v[kj], v[kj+1]) * w[kj]   (MODE 2 cross product, off the hot path)
template <int KJ, typename TC, int K>
__device__ __forceinline__ void one_chain(const RowRegs<TC, K>& R, const WRow<TC, K>& wk, acc_t<TC>& a)
{
    if constexpr (KJ < K) {
        R.template fma<KJ>(wk, a);
        one_chain<KJ + 1, TC, K>(R, wk, a);
    }
}

// SYM: the template is symmetric under a vertical flip (row s == row K-1-s; true for the loops and
// stripes templates of the reference).  The horizontal product of a staged row with template row
// s, Q_s = row (*) W[s], then serves the two slots s and K-1-s: it is computed once for
// s <= (K-1)/2 (K packed ops) and added to both accumulators (one packed add each), so a staged
// row costs (K+1)/2 * K + K-1 packed ops instead of K * K, and only (K+1)/2 weight rows are loaded.
// SKIP / q_need (wave-uniform): only Q_0 .. Q_(q_need-1) feed output rows of the strip in this
// iteration -- fewer than (K+1)/2 while the strip warms up (staged rows t, t+1 < K-1 reach slots
// <= t+1 only) and in its last iterations (slots below t - rows_out + 1 belong to rows under the
// strip).  The SKIP instance of the chain tests every product against q_need and leaves stale Q
// values behind, which only reach accumulators of rows outside the strip; the row loop uses it in
// exactly those iterations and the branch-free instance otherwise.  (The same skip on the unfolded
// chain does not pay: there every slot renames an accumulator, and the register moves at the merge
// points cost more than the skipped FMAs.)
template <int S, typename TC, int K, int W_OFF, bool SYM, bool SKIP>
__device__ __forceinline__ void steps2_rec(const RowRegs<TC, K>& A, const RowRegs<TC, K>& B, unsigned long long w_base,
                                           acc_t<TC> (&X)[K - 1], acc_t<TC>& outA, acc_t<TC>& outB,
                                           acc_t<TC> yprev, acc_t<TC> zprev, WRow<TC, K>& cur, WRow<TC, K>& nxt,
                                           acc_t<TC> (&QA)[(K + 1) / 2], acc_t<TC> (&QB)[(K + 1) / 2], const int q_need)
{
    constexpr bool FOLD = SYM && sizeof(TC) == 4;
    constexpr int KM = (K - 1) / 2;                 // middle template row
    acc_t<TC> y, z;
    if constexpr (FOLD) {
        if constexpr (S <= KM) {
            if (!SKIP || S < q_need) {
                cur.wait();
                if constexpr (S < KM) {
                    if (!SKIP || S + 1 < q_need) nxt.template issue<W_OFF + (S + 1) * K * (int)sizeof(TC)>(w_base);
                }
                __builtin_amdgcn_sched_barrier(0);
                pk_row<K, true>(A, B, cur, QA[S], QB[S]);
                if constexpr (SKIP) {
                    pin_acc(QA[S]);
                    pin_acc(QB[S]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int F = S <= KM ? S : K - 1 - S;
        if constexpr (S == 0) {
            y = QA[0];
            z = QB[0];
        } else {
            y = X[S - 1] + QA[F];
            z = yprev + QB[F];
        }
    } else {
        cur.wait();
        if constexpr (S < K - 1) nxt.template issue<W_OFF + (S + 1) * K * (int)sizeof(TC)>(w_base);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S > 0) {
            y = X[S - 1];
            z = yprev;
        } else if constexpr (sizeof(TC) != 4) {
            y = acc_zero<TC>();
            z = acc_zero<TC>();
        }
        if constexpr (sizeof(TC) == 4) pk_row<K, S == 0>(A, B, cur, y, z);
        else two_chains<0, TC, K>(A, B, cur, y, z);
    }
    if constexpr (S > 0) X[S - 1] = zprev;
    // pin this step's FMAs between the two hand-placed scalar loads: without a (volatile) use of
    // their results the optimiser sinks them below all K loads and spills the weights
    pin_acc(y);
    pin_acc(z);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S < K - 1) {
        steps2_rec<S + 1, TC, K, W_OFF, SYM, SKIP>(A, B, w_base, X, outA, outB, y, z, nxt, cur, QA, QB, q_need);
    } else {
        outA = y;
        outB = z;
    }
}

// W_OFF: byte offset of the weight set inside the weight buffer (0 = signal weights)
template <typename TC, int K, int W_OFF, bool SYM, bool SKIP>
__device__ __forceinline__ void steps2(const RowRegs<TC, K>& A, const RowRegs<TC, K>& B, unsigned long long w_base,
                                       acc_t<TC> (&X)[K - 1], acc_t<TC>& outA, acc_t<TC>& outB,
                                       acc_t<TC> (&QA)[(K + 1) / 2], acc_t<TC> (&QB)[(K + 1) / 2], const int q_need)
{
    WRow<TC, K> wa, wb;
    wa.template issue<W_OFF>(w_base);
    steps2_rec<0, TC, K, W_OFF, SYM, SKIP && SYM && sizeof(TC) == 4>(A, B, w_base, X, outA, outB, acc_zero<TC>(),
                                                                      acc_zero<TC>(), wa, wb, QA, QB, q_need);
}

// value of lane `idx` (wave-uniform index) as a wave-uniform scalar
__device__ __forceinline__ float lane_value(float v, int idx)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), idx));
}
__device__ __forceinline__ double lane_value(double v, int idx)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, idx);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), idx);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Optional section timing of the row loop (diagnostics build, -DCS_PROFILE): wave cycles between the
// loop top, the start and the end of the FMA block, summed over all waves into cs_prof[].  s_memtime
// waits for outstanding LDS / scalar loads, so the stamps sit where the loop waits anyway.
#ifdef CS_PROFILE
__device__ unsigned long long cs_prof[8];
#define CS_STAMP(n)                                                 \
    {                                                               \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        prof_[n] += now_ - tlast_;                                  \
        tlast_ = now_;                                              \
    }
#else
#define CS_STAMP(n)
#endif

// Lean view of the launch arguments (only what the row loop needs stays in registers).
template <typename TC>
struct StreamArgs {
    const void* sig;        // input map
    void* out;              // coefficient map
    float* nobs;            // optional: present pixels of every window, same geometry as `out`
    int xcorr_only;         // plain cross-correlation (xcorr2): thresholded sum S*w instead of the coefficient
    unsigned long long w;   // device address of the centred template weights (K*K)
    long long ld_in, ld_out;
    int sig_is_f64, out_is_f64;
    int band_in, lo_in, bw_in;     // input layout: band flag, first stored diagonal, stored diagonals
    int band_out, out_lo, out_hi;  // output layout / produced diagonal range
    int lo_out;                    // first stored diagonal of the output band
    int ms, ns;
    int full, sym_upper;
    int strip_h, strips_x, strips_y;
    int split_sy, strip_h2;        // row blocks >= split_sy have height strip_h2 (two-height tiling), 0 = uniform
    int xcd_remap;                 // 1: workgroup -> strip mapping that keeps neighbouring strips on one XCD
    int row_begin, row_end;        // output row window (CorrArgs)
    long long row0_in, row0_out;   // first matrix row held by the input / output buffers
    // missing mask (MASKED kernels only)
    int mask_mode;                 // 1: per-bin flags, 2: explicit uint8 map with the signal's geometry
    int max_dist;                  // -1: None
    const uint8_t* miss_row;
    const uint8_t* miss_col;
    const uint8_t* mask;
    // MODE 2 tables (cs_mask_prep.hip), see CorrArgs
    int fix_on, fix_hi_w, fix_hi_d0;
    const TC* rowtab;
    const TC* coltab;
    const TC* fix_lo;
    const TC* fix_hi;
    const TC* fix_rows;
    const TC* fix_cols;
    int fix_top, fix_bot0, fix_width, fix_xband, fix_xlo, fix_side;
    KernelStats<TC> ks;
};

constexpr int kRowTabStride = 4;    // elements per row of rowtab: nr, RA, RB, flags of the window rows

// framed missing predicate (same rules as cs_device.h missing_pred) on the lean argument block
template <typename TC, int K>
__device__ __forceinline__ bool stream_missing(const StreamArgs<TC>& A, int p, int q, bool cflag, bool rflag,
                                               long long sig_idx, bool stored)
{
    const bool in_r = (p >= 0) & (p < A.ms);
    const bool in_c = (q >= 0) & (q < A.ns);
    const int d = q - p;
    const bool have_md = A.max_dist >= 0;
    bool m;
    if (in_r & in_c) {
        if (A.mask_mode == 1) {
            m = rflag | cflag;
            if (A.sym_upper) {
                const int md = have_md ? A.max_dist : min(A.ms, A.ns);
                m = m & (d >= 0) & (d <= md);
            }
        } else {
            m = stored ? (A.mask[sig_idx] != 0) : false;
            if (A.full && A.sym_upper && have_md) m = m & (d >= 0) & (d <= A.max_dist + K);
        }
        if (!A.full) return m;
    } else {
        if (!A.full) return false;
        if (A.sym_upper && have_md) {
            if (q >= A.ns) m = p >= A.ms - A.max_dist - 2;
            else if (p < 0) m = (q < 0) ? true : (q < A.max_dist + K);
            else m = false;
        } else {
            m = true;
        }
    }
    if (A.sym_upper) m = m | ((d <= -1) & (d >= -K));
    return m;
}

// MODE 0: no missing mask.  MODE 1: any mask, carried as -0.0 in the staged signal, two extra sets
// of rotating accumulators.  MODE 2: per-bin mask (missing = row flag | column flag): the mask sums
// factorise,
//     sum_missing W = RA[i] + CA[j] - sum_kj c[j+kj] * U_i[kj],   U_i[kj] = sum_ki r[i+ki] W[ki][kj],
// into per-row / per-column tables (cs_mask_prep.hip) plus a K-term cross product that is only
// needed on output rows with a flagged row in reach; pixels whose window leaves the matrix or the
// diagonal range 0..max_dist get a precomputed correction.  The data path is that of MODE 0.
template <typename TC, int K, int MODE, bool SYM, bool SKIP = false>
__global__ __launch_bounds__(64 * CS_NWAVES, 8 / CS_NWAVES) void corr_stream_kernel(const StreamArgs<TC> A)
{
    using G = StreamGeom<K>;
    constexpr int KH = (K - 1) / 2;
    constexpr bool MASKED = MODE == 1;
    constexpr bool REG = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    constexpr bool HCACHE = sizeof(TC) == 4 && !MASKED;
    constexpr int DATA_ROWS = HCACHE ? G::DATA_ROWS_HC : G::RING;
    TC* ring = reinterpret_cast<TC*>(smem_raw) + (size_t)wv * G::wave_elems(HCACHE);

    // strip of this wave (uniform per wave); x fastest so that neighbouring waves share halos
    // XCD-aware order (band outputs): the hardware deals workgroups round-robin to the 8 XCDs, each
    // with its own L2.  Neighbouring row blocks share K-1 halo rows (36 staged rows for 20 emitted on a
    // 234-diagonal band) and the two or more strips of a row block share halo columns, so every XCD gets
    // a contiguous range of strips instead of every 8th one: the halo is then re-read from its L2.
    int bid = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb >> 3, rem = nb & 7;
        const int x = bid & 7, r = bid >> 3;
        bid = x * q + min(x, rem) + r;
    }
    const int wid = __builtin_amdgcn_readfirstlane(bid * G::NWAVES + wv);
    if (wid >= A.strips_x * A.strips_y) return;
    const int sy = wid / A.strips_x;
    const int sx = wid - sy * A.strips_x;
    // two-height tiling: the row blocks of the second half of a single-generation launch are
    // shorter (launch_fast in cs_corr_fast.hip): their waves start second on every SIMD
    const bool late = A.split_sy > 0 && sy >= A.split_sy;
    const int strip_h = late ? A.strip_h2 : A.strip_h;
    const int i0 = A.row_begin + (late ? A.split_sy * A.strip_h + (sy - A.split_sy) * A.strip_h2 : sy * A.strip_h);   // first output row
    int j0 = sx * G::TW;                                  // first output column
    if (A.band_out) {
        // band outputs: the strips of a row block start at the block's first in-band column
        // (not at a multiple of 128), which saves one strip per row block
        int jmin = i0 + A.out_lo;
        if (jmin < 0) jmin = 0;
        j0 += jmin;
        const int dmin = j0 - (i0 + strip_h - 1);
        if (j0 >= A.ns || dmin > A.out_hi) return;
    }
    if (i0 >= A.row_end) return;
    const int rows_out = min(strip_h, A.row_end - i0);
    const int n_staged = rows_out + K - 1;
    // MODE 2: can any pixel of this strip need a correction table?  (wave-uniform; most strips of a
    // wide band or of a large rectangular map cannot, and skip the per-pixel lookup logic)
    bool strip_needs_fix = false;
    if constexpr (REG) {
        const int d_min = j0 - (i0 + rows_out - 1), d_max = j0 + G::TW - 1 - i0;
        strip_needs_fix = (i0 < A.fix_top) | (i0 + rows_out > A.fix_bot0) |
                          (A.fix_cols != nullptr && ((j0 < A.fix_side) | (j0 + G::TW > A.ns - A.fix_side))) |
                          (A.fix_on && ((d_min < K - 1 && d_max >= 0) | (d_max >= A.fix_hi_d0 && d_min < A.fix_hi_d0 + A.fix_hi_w)));
    }

    // zero the ring: rows "older" than the strip are read (and subtracted) as zeros, which
    // makes the row loop branch free
    TC* hcache = ring + (DATA_ROWS + 1) * G::LWP;      // after the data rows and the column-flag row
    if constexpr (HCACHE) {
        for (int idx = lane; idx < G::HC_FLOATS; idx += kWave) hcache[idx] = TC(0);
    } else {
        for (int idx = lane; idx < G::RING * G::LWP; idx += kWave) ring[idx] = TC(0);
    }

    // ---- per-lane staging state: lanes 0..LWP/4-1 move 4 consecutive elements of each row ----
    // element (p, q) lives at p * ld + q - shift(p); dense: shift = 0, band: shift = p + lo
    const int q_lane = j0 - KH + lane * 4;
    const int p_first = i0 - KH;
    const int shift0 = A.band_in ? p_first + A.lo_in : 0;
    long long in_idx = ((long long)p_first - A.row0_in) * A.ld_in + (q_lane - shift0);   // element index of e = 0
    const long long in_step = A.band_in ? A.ld_in - 1 : A.ld_in;          // per-row increment
    int dd = q_lane - shift0;                                              // stored-diagonal index
    const int dd_step = A.band_in ? -1 : 0;
    const unsigned bw = A.band_in ? (unsigned)A.bw_in : 0x7fffffffu;
    bool col_ok[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) col_ok[e] = (lane * 4 + e < G::LW) & (q_lane + e >= 0) & (q_lane + e < A.ns);
    const bool stage_lane = lane * 4 < G::LWP;

    bool cflag[4] = {false, false, false, false};
    if constexpr (MASKED) {
        if (A.mask_mode == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = q_lane + e;
                cflag[e] = (q >= 0 && q < A.ns) ? (A.miss_col[q] != 0) : false;
            }
        }
    }
    auto fetch = [&](int t, TC (&x)[4]) {
        const int p = p_first + t;
        const bool row_in = (p >= 0) & (p < A.ms);
        const bool row_ok = row_in & (t < n_staged);
        bool rflag = false;
        if constexpr (MASKED) {
            if (A.mask_mode == 1) rflag = row_in ? (A.miss_row[p] != 0) : false;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = row_ok & col_ok[e] & (A.band_in ? ((unsigned)(dd + e) < bw) : true);
            const long long idx = ok ? in_idx + e : 0;        // always a valid address
            TC val;
            if (A.sig_is_f64) val = (TC)((const double*)A.sig)[idx];
            else val = (TC)((const float*)A.sig)[idx];
            val = ok ? val : TC(0);
            if constexpr (MASKED) {
                const bool miss = (lane * 4 + e < G::LW) &&
                                  stream_missing<TC, K>(A, p, q_lane + e, cflag[e], rflag, idx, ok);
                val = encode_pixel(val, miss);
            }
            x[e] = val;
        }
        in_idx += in_step;
        dd += dd_step;
    };
    auto commit = [&](int slot, const TC (&x)[4]) {
        if (stage_lane) {
            TC* dst = ring + slot * G::LWP + lane * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = x[e];
        }
    };

    // ---- per-lane output state -------------------------------------------------------------
    const int oj0 = j0 + lane * 2;
    const int lo_o = A.band_out ? A.lo_out : 0;
    long long out_idx = ((long long)i0 - A.row0_out) * A.ld_out + (oj0 - (A.band_out ? i0 + lo_o : 0));
    const long long out_step = A.band_out ? A.ld_out - 1 : A.ld_out;
    int d_out = oj0 - i0;                                  // diagonal of column 0 at the first output row
    bool ocol_ok[2], ocol_margin[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        ocol_ok[c] = (oj0 + c) < A.ns;
        ocol_margin[c] = !A.full && ((oj0 + c < KH) | (oj0 + c > A.ns - K + KH));
    }

    // MODE 2: column terms of the factorised mask sums (constant over the strip) and the strip's
    // column flags as a staged row for the cross product
    TC ncol[2] = {TC(0), TC(0)}, ca_col[2] = {TC(0), TC(0)}, cb_col[2] = {TC(0), TC(0)};
    TC* cfl = ring + DATA_ROWS * G::LWP;
    if constexpr (REG) {
        for (int idx = lane; idx < G::LWP; idx += kWave) {
            const int q = j0 - KH + idx;
            const bool in = (q >= 0) & (q < A.ns) & (idx < G::LW);
            cfl[idx] = (in && A.miss_col[in ? q : 0] != 0) ? TC(1) : TC(0);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = oj0 + c;
            if (j < A.ns) {
                ncol[c] = A.coltab[j];
                ca_col[c] = A.coltab[(size_t)A.ns + j];
                cb_col[c] = A.coltab[2 * (size_t)A.ns + j];
            }
        }
    }

    acc_t<TC> X[K - 1];
#pragma unroll
    for (int s = 0; s < K - 1; ++s) X[s] = acc_zero<TC>();
    // shared row products of the folded (SYM) chain; they persist across iterations so that a
    // skipped product leaves a defined (stale) value behind
    acc_t<TC> QA[(K + 1) / 2], QB[(K + 1) / 2];
#pragma unroll
    for (int s = 0; s < (K + 1) / 2; ++s) QA[s] = QB[s] = acc_zero<TC>();
    // running box sums (sum S, sum S^2 over the K staged rows of the window), updated by
    // (row entering) - (row leaving).  The difference of the two float32 horizontal partials is
    // accumulated in float64, so the rounding does not grow with the strip height.
    double b1[2] = {0.0, 0.0}, b2[2] = {0.0, 0.0};
    // masked kernels: rotating accumulators of sum_missing Wa and sum_missing Wb, running count
    acc_t<TC> XA[MASKED ? K - 1 : 1], XB[MASKED ? K - 1 : 1];
    TC nmiss[2] = {TC(0), TC(0)};
#pragma unroll
    for (int s = 0; s < (MASKED ? K - 1 : 1); ++s) XA[s] = XB[s] = acc_zero<TC>();

    {
        TC x[4];
        fetch(0, x);
        commit(0, x);
        fetch(1, x);
        commit(1, x);
    }
    auto ring_next = [](int slot, int by) {
        slot += by;
        return slot >= DATA_ROWS ? slot - DATA_ROWS : slot;
    };
    int slot_a = 0;                  // ring slot of row t (row t+1 is the next slot)
    int slot_old = HCACHE ? 0 : G::RING - K;      // ring slot of row t - K; HCACHE: slot of row t in the box-sum cache

    // horizontal box sums of one ring row (and the number of missing flags in it)
    auto row_sums = [&](int slot, TC (&h1)[2], TC (&h2)[2], TC (&cnt)[2]) {
        RowRegs<TC, K> r;
        r.load(ring + slot * G::LWP + lane * 2);
        row_box<TC, K>(r, h1, h2);
        if constexpr (MASKED) {
            r.to_missing_flags();
            TC c0 = r.at(0);
#pragma unroll
            for (int t = 1; t < K; ++t) c0 += r.at(t);
            cnt[0] = c0;
            cnt[1] = (c0 - r.at(0)) + r.at(K);
        }
    };

    // box-sum cache (HCACHE): slot (row mod K) holds the sums of the row that leaves the window when
    // the row with the same index mod K enters; the row loop reads them, then stores the entering row's

    auto row_flags = [&](const RowRegs<TC, K>& r, TC (&cnt)[2]) {
        TC c0 = missing_flag(r.at(0));
#pragma unroll
        for (int q = 1; q < K; ++q) c0 += missing_flag(r.at(q));
        cnt[0] = c0;
        cnt[1] = (c0 - missing_flag(r.at(0))) + missing_flag(r.at(K));
    };

    // MODE 2: correction of the factorised mask sums for the pixels whose window leaves the matrix
    // (frame tables) or the diagonal range 0..max_dist (edge tables).  The table reads are issued by
    // fix_fetch BEFORE the FMA block of the iteration and consumed by emit after it, so their latency
    // is covered (every strip of a narrow band is an edge strip).  fx[c] = {d nm, d ka, d kb}.
    auto fix_fetch = [&](int oi, int d_row, TC (&fx)[2][3]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fx[c][0] = fx[c][1] = fx[c][2] = TC(0);
            const int d = d_row + c;
            const bool in_range = ocol_ok[c] & (d >= A.out_lo) & (d <= A.out_hi) & (oi < A.row_end);
            if (in_range) {
                const TC* f = nullptr;
                const int j = oj0 + c;
                const int x = A.fix_xband ? d - A.fix_xlo : j;
                if (oi < A.fix_top) f = A.fix_rows + ((size_t)oi * A.fix_width + x) * 4;
                // with edge tables the frame table only holds the pixels whose window leaves the matrix
                else if (oi >= A.fix_bot0 && (!A.fix_on || (oi + KH >= A.ms) | (j + KH >= A.ns)))
                    f = A.fix_rows + ((size_t)(A.fix_top + oi - A.fix_bot0) * A.fix_width + x) * 4;
                else if (A.fix_cols && (j < A.fix_side || j >= A.ns - A.fix_side))
                    f = A.fix_cols + ((size_t)oi * 2 * A.fix_side + (j < A.fix_side ? j : j - (A.ns - 2 * A.fix_side))) * 4;
                else if (A.fix_on) {
                    if (d >= 0 && d < K - 1) f = A.fix_lo + ((size_t)oi * (K - 1) + d) * 4;
                    else if (d >= A.fix_hi_d0 && d - A.fix_hi_d0 < A.fix_hi_w)
                        f = A.fix_hi + ((size_t)oi * A.fix_hi_w + (d - A.fix_hi_d0)) * 4;
                }
                if (f) {
                    fx[c][0] = f[0];
                    fx[c][1] = f[1];
                    fx[c][2] = f[2];
                }
            }
        }
    };

    // the common output (float32 map, no n_obs, not a plain cross-correlation) has a straight-line
    // epilogue: both columns of the lane together, one divergent region for the rare pixels near a
    // zeroing threshold, one store region.  Run pixel by pixel through the general code below, the four
    // pixels of an iteration were a chain of ~40 scalar branches and took half of the iteration.
    const bool fast_epi = (sizeof(TC) == 4) && !A.out_is_f64 && !A.nobs && !A.xcorr_only;

    auto emit = [&](int oi, const acc_t<TC>& cs2, const TC (&s1)[2], const TC (&s2)[2],
                    const TC (&nm)[2], const acc_t<TC>& ka2, const acc_t<TC>& kb2, const TC (&fx)[2][3]) {
        const bool row_margin = !A.full && ((oi < KH) | (oi > A.ms - K + KH));
        const TC csv[2] = {cs2.x, cs2.y}, kav[2] = {ka2.x, ka2.y}, kbv[2] = {kb2.x, kb2.y};
        if constexpr (sizeof(TC) == 4) {
            if (fast_epi) {
                const bool in0 = ocol_ok[0] & (d_out >= A.out_lo) & (d_out <= A.out_hi) & (oi < A.row_end);
                const bool in1 = ocol_ok[1] & (d_out + 1 >= A.out_lo) & (d_out + 1 <= A.out_hi) & (oi < A.row_end);
                const bool z0 = row_margin | ocol_margin[0] | (A.sym_upper && d_out < 0);
                const bool z1 = row_margin | ocol_margin[1] | (A.sym_upper && d_out + 1 < 0);
                bool rare0, rare1;
                float r0, r1;
                if constexpr (REG || MASKED) {
                    float n0 = nm[0], n1 = nm[1], a0 = kav[0], a1 = kav[1], b0 = kbv[0], b1 = kbv[1];
                    if constexpr (REG) {
                        n0 += fx[0][0]; a0 += fx[0][1]; b0 += fx[0][2];
                        n1 += fx[1][0]; a1 += fx[1][1]; b1 += fx[1][2];
                    }
                    r0 = pearson_masked_core(csv[0], s1[0], s2[0], n0, a0, b0, A.ks, rare0);
                    r1 = pearson_masked_core(csv[1], s1[1], s2[1], n1, a1, b1, A.ks, rare1);
                    if (rare0 | rare1) {
                        r0 = pearson_masked_f32(csv[0], s1[0], s2[0], n0, a0, b0, A.ks);
                        r1 = pearson_masked_f32(csv[1], s1[1], s2[1], n1, a1, b1, A.ks);
                    }
                } else {
                    r0 = pearson_nomask_core(csv[0], s1[0], s2[0], A.ks, rare0);
                    r1 = pearson_nomask_core(csv[1], s1[1], s2[1], A.ks, rare1);
                    if (rare0 | rare1) {
                        r0 = pearson_nomask_f32(csv[0], s1[0], s2[0], A.ks);
                        r1 = pearson_nomask_f32(csv[1], s1[1], s2[1], A.ks);
                    }
                }
                r0 = z0 ? 0.0f : r0;
                r1 = z1 ? 0.0f : r1;
                float* o = (float*)A.out + out_idx;
                if (!A.band_out && in0 && in1) {           // dense: the pair is 8-byte aligned
                    f32x2 pr;
                    pr.x = r0;
                    pr.y = r1;
                    *reinterpret_cast<f32x2*>(o) = pr;
                } else {
                    if (in0) o[0] = r0;
                    if (in1) o[1] = r1;
                }
                out_idx += out_step;
                d_out -= 1;
                return;
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int d = d_out + c;
            const bool in_range = ocol_ok[c] & (d >= A.out_lo) & (d <= A.out_hi) & (oi < A.row_end);
            const bool zero = row_margin | ocol_margin[c] | (A.sym_upper && d < 0);
            TC r;
            if constexpr (REG) {
                const TC nmv = nm[c] + fx[c][0], ka = kav[c] + fx[c][1], kb = kbv[c] + fx[c][2];
                if constexpr (sizeof(TC) == 4) {
                    r = pearson_masked_lean(csv[c], s1[c], s2[c], nmv, ka, kb, A.ks);
                } else {
                    TC nobs;
                    r = pearson_from_sums<TC>(csv[c], (TC)s1[c], (TC)s2[c], nmv, ka, kb, A.ks, true, &nobs);
                }
            } else if constexpr (MASKED) {
                if constexpr (sizeof(TC) == 4) {
                    r = pearson_masked_lean(csv[c], s1[c], s2[c], nm[c], kav[c], kbv[c], A.ks);
                } else {
                    TC nobs;
                    r = pearson_from_sums<TC>(csv[c], (TC)s1[c], (TC)s2[c], nm[c], kav[c], kbv[c], A.ks, true, &nobs);
                }
            } else if constexpr (sizeof(TC) == 4) {
                r = pearson_nomask_lean(csv[c], s1[c], s2[c], A.ks);
            } else {
                TC nobs;
                r = pearson_from_sums<TC>(csv[c], (TC)s1[c], (TC)s2[c], TC(0), TC(0), TC(0), A.ks, false, &nobs);
            }
            if constexpr (!MASKED && !REG) {
                if (A.xcorr_only) r = (cs_abs(csv[c]) < A.ks.thr) ? TC(0) : csv[c];   // detection.py:716-722
            }
            r = zero ? TC(0) : r;
            if (in_range) {
                if (A.out_is_f64) ((double*)A.out)[out_idx + c] = (double)r;
                else ((float*)A.out)[out_idx + c] = (float)r;
                if (A.nobs) {
                    TC present = A.ks.n;
                    if constexpr (REG) present = A.ks.n - (nm[c] + fx[c][0]);
                    else if constexpr (MASKED) present = A.ks.n - nm[c];
                    A.nobs[out_idx + c] = (float)present;
                }
            }
        }
        out_idx += out_step;
        d_out -= 1;
    };

    // MODE 2: per-row terms of the factorised mask sums {nr, RA, RB, window row flags} of 64
    // consecutive output rows, one row per lane (cs_mask_prep.hip rowtab[row][0..3]); the row loop reads
    // them with v_readlane instead of scalar loads from a table that never stays in the scalar cache
    // (their full latency used to sit on the critical path of every iteration).
    TC hdr[4] = {TC(0), TC(0), TC(0), TC(0)};
    int hdr_base = i0;
    auto hdr_load = [&]() {
        if constexpr (REG) {
            const int row = min(hdr_base + lane, A.ms - 1);
            const TC* src = A.rowtab + (size_t)row * kRowTabStride;
#pragma unroll
            for (int e = 0; e < 4; ++e) hdr[e] = src[e];
        }
    };
    hdr_load();

    // two staged rows (A = t, B = t + 1) per iteration; a virtual zero row pads an odd count.
    // The row registers of the next iteration are loaded from LDS right after the FMAs of this one
    // (before its epilogue), so the LDS latency is hidden behind the epilogue.
    // (only the float32 unmasked / factorised kernels have the registers for that)
    constexpr bool PIPE = HCACHE;
    RowRegs<TC, K> ra, rb;
    if constexpr (PIPE) {
        ra.load(ring + slot_a * G::LWP + lane * 2);
        rb.load(ring + ring_next(slot_a, 1) * G::LWP + lane * 2);
    }
#ifdef CS_PROFILE
    unsigned long long prof_[4] = {0, 0, 0, 0};
    unsigned long long tlast_ = __builtin_readcyclecounter();
    const unsigned long long tstart_ = tlast_;
#endif
    for (int t = 0; t < n_staged; t += 2) {
        CS_STAMP(3)      // (loop back-edge, LDS row loads of a non-pipelined kernel)
        if constexpr (!PIPE) {
            ra.load(ring + slot_a * G::LWP + lane * 2);
            rb.load(ring + ring_next(slot_a, 1) * G::LWP + lane * 2);
        }
        // box sums of the two entering rows minus the two leaving rows
        TC sA1[2], sA2[2];
        TC nmA[2] = {TC(0), TC(0)};
        if constexpr (HCACHE) {
            // both cache cells are read before the sums of the entering rows are formed, so the LDS
            // latency overlaps those ~50 VALU ops instead of preceding the running-sum update
            typedef TC v4 __attribute__((ext_vector_type(4)));
            const int slot_b = slot_old + 1 >= K ? slot_old + 1 - K : slot_old + 1;
            v4* cell_a = reinterpret_cast<v4*>(hcache + (slot_old * kWave + lane) * 4);
            v4* cell_b = reinterpret_cast<v4*>(hcache + (slot_b * kWave + lane) * 4);
            const v4 old_a = *cell_a, old_b = *cell_b;
            TC hA1[2], hA2[2], hB1[2], hB2[2];
            row_box_packed<K>(ra, hA1, hA2);
            row_box_packed<K>(rb, hB1, hB2);
            v4 cur_a, cur_b;
            cur_a.x = hA1[0]; cur_a.y = hA1[1]; cur_a.z = hA2[0]; cur_a.w = hA2[1];
            cur_b.x = hB1[0]; cur_b.y = hB1[1]; cur_b.z = hB2[0]; cur_b.w = hB2[1];
            *cell_a = cur_a;
            *cell_b = cur_b;
            b1[0] += (double)(hA1[0] - old_a.x);
            b1[1] += (double)(hA1[1] - old_a.y);
            b2[0] += (double)(hA2[0] - old_a.z);
            b2[1] += (double)(hA2[1] - old_a.w);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                sA1[c] = (TC)b1[c];
                sA2[c] = (TC)b2[c];
            }
            b1[0] += (double)(hB1[0] - old_b.x);
            b1[1] += (double)(hB1[1] - old_b.y);
            b2[0] += (double)(hB2[0] - old_b.z);
            b2[1] += (double)(hB2[1] - old_b.w);
        } else {
            TC h1[2], h2[2], g1[2], g2[2], gc[2] = {TC(0), TC(0)}, hc[2] = {TC(0), TC(0)};
            row_box<TC, K>(ra, h1, h2);
            row_sums(slot_old, g1, g2, gc);
            if constexpr (MASKED) row_flags(ra, hc);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                b1[c] += (double)(h1[c] - g1[c]);
                b2[c] += (double)(h2[c] - g2[c]);
                sA1[c] = (TC)b1[c];
                sA2[c] = (TC)b2[c];
                nmiss[c] += hc[c] - gc[c];
                nmA[c] = nmiss[c];
            }
            row_box<TC, K>(rb, h1, h2);
            row_sums(ring_next(slot_old, 1), g1, g2, gc);
            if constexpr (MASKED) row_flags(rb, hc);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                b1[c] += (double)(h1[c] - g1[c]);
                b2[c] += (double)(h2[c] - g2[c]);
                nmiss[c] += hc[c] - gc[c];
            }
        }

        // issue the global loads of rows t+2, t+3 now, commit them to LDS after the FMAs
        TC nx0[4], nx1[4];
        fetch(t + 2, nx0);
        fetch(t + 3, nx1);

        TC fxA[2][3] = {{TC(0), TC(0), TC(0)}, {TC(0), TC(0), TC(0)}}, fxB[2][3] = {{TC(0), TC(0), TC(0)}, {TC(0), TC(0), TC(0)}};
        if constexpr (REG) {
            if (strip_needs_fix && t >= K - 1) {
                const int oi = i0 + t - (K - 1);
                // wave-uniform: can a pixel of these two output rows need a correction at all?  (a tall
                // strip of a wide band crosses an edge diagonal range only in part of its rows)
                const int dmin = j0 - (oi + 1), dmax = j0 + G::TW - 1 - oi;
                const bool rows_need = (oi < A.fix_top) | (oi + 1 >= A.fix_bot0) | (A.fix_cols != nullptr) |
                                       (A.fix_on && ((dmin < K - 1 && dmax >= 0) |
                                                     (dmax >= A.fix_hi_d0 && dmin < A.fix_hi_d0 + A.fix_hi_w)));
                if (rows_need) {
                    fix_fetch(oi, d_out, fxA);
                    fix_fetch(oi + 1, d_out - 1, fxB);
                }
            }
        }

        acc_t<TC> outA, outB;
        // template rows that reach output rows of the strip (see steps2_rec)
        const int q_need = __builtin_amdgcn_readfirstlane(min(min((K + 1) / 2, t + 2), K - max(0, t - rows_out + 1)));
        CS_STAMP(0)      // box sums, cache swap, issue of the global loads
        steps2<TC, K, 0, SYM, SKIP>(ra, rb, A.w, X, outA, outB, QA, QB, q_need);
        CS_STAMP(1)      // FMA block incl. the waits for the weight rows

        acc_t<TC> kaA = acc_zero<TC>(), kaB = acc_zero<TC>(), kbA = acc_zero<TC>(), kbB = acc_zero<TC>();
        if constexpr (MASKED) {
            // the row registers are dead: turn them into the 0/1 missing flags in place and run the
            // same scheme with the two mask weight sets
            ra.to_missing_flags();
            rb.to_missing_flags();
            steps2<TC, K, K * K * (int)sizeof(TC), SYM, false>(ra, rb, A.w, XA, kaA, kaB, QA, QB, q_need);
            steps2<TC, K, 2 * K * K * (int)sizeof(TC), SYM, false>(ra, rb, A.w, XB, kbA, kbB, QA, QB, q_need);
        }

        auto advance = [&]() {
            commit(ring_next(slot_a, 2), nx0);
            commit(ring_next(slot_a, 3), nx1);
            slot_a = ring_next(slot_a, 2);
            if constexpr (HCACHE) slot_old = slot_old + 2 >= K ? slot_old + 2 - K : slot_old + 2;
            else slot_old = ring_next(slot_old, 2);
        };
        if constexpr (PIPE) {   // rows t+2, t+3 for the next iteration (zeros past the strip)
            advance();
            ra.load(ring + slot_a * G::LWP + lane * 2);
            rb.load(ring + ring_next(slot_a, 1) * G::LWP + lane * 2);
        }

        if (t >= K - 1) {
            const int oi = i0 + t - (K - 1);
            const TC sB1[2] = {(TC)b1[0], (TC)b1[1]}, sB2[2] = {(TC)b2[0], (TC)b2[1]};
            if constexpr (REG) {
                // row terms from the lane-resident header chunk
                if (oi >= hdr_base + kWave) {            // wave-uniform: every 32 iterations
                    hdr_base += kWave;
                    hdr_load();
                }
                const int la = oi - hdr_base;            // even, so la + 1 stays inside the chunk
                const TC nrA = lane_value(hdr[0], la), raA = lane_value(hdr[1], la), rbA = lane_value(hdr[2], la);
                const TC nrB = lane_value(hdr[0], la + 1), raB = lane_value(hdr[1], la + 1), rbB = lane_value(hdr[2], la + 1);
                const unsigned bitsA = (unsigned)lane_value(hdr[3], la), bitsB = (unsigned)lane_value(hdr[3], la + 1);
                acc_t<TC> xaA = acc_zero<TC>(), xbA = acc_zero<TC>(), xaB = acc_zero<TC>(), xbB = acc_zero<TC>();
                if (bitsA | bitsB) {
                    // cross term sum_kj c[j+kj] U_i[kj] = sum over the flagged rows ki of the window of
                    // (column flags (*) template row ki), with the template rows read from the weight
                    // sets themselves: they are hot in the scalar cache, the per-row U vectors were not
                    RowRegs<TC, K> cf;
                    cf.load(cfl + lane * 2);
                    auto cross = [&](unsigned bits, acc_t<TC>& xa, acc_t<TC>& xb) {
                        while (bits) {
                            const int ki = __builtin_ctz(bits);
                            bits &= bits - 1;
                            WRow<TC, K> wa, wb;
                            wa.issue_at(A.w, (unsigned)((K * K + ki * K) * sizeof(TC)));
                            wb.issue_at(A.w, (unsigned)((2 * K * K + ki * K) * sizeof(TC)));
                            wa.wait();
                            wb.wait();
                            one_chain<0, TC, K>(cf, wa, xa);
                            one_chain<0, TC, K>(cf, wb, xb);
                        }
                    };
                    cross(bitsA, xaA, xbA);
                    cross(bitsB, xaB, xbB);
                }
                TC nA[2], nB[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    nA[c] = TC(K) * nrA + (TC(K) - nrA) * ncol[c];
                    nB[c] = TC(K) * nrB + (TC(K) - nrB) * ncol[c];
                }
                kaA.x = raA + ca_col[0] - xaA.x;  kaA.y = raA + ca_col[1] - xaA.y;
                kbA.x = rbA + cb_col[0] - xbA.x;  kbA.y = rbA + cb_col[1] - xbA.y;
                kaB.x = raB + ca_col[0] - xaB.x;  kaB.y = raB + ca_col[1] - xaB.y;
                kbB.x = rbB + cb_col[0] - xbB.x;  kbB.y = rbB + cb_col[1] - xbB.y;
                emit(oi, outA, sA1, sA2, nA, kaA, kbA, fxA);
                emit(oi + 1, outB, sB1, sB2, nB, kaB, kbB, fxB);
            } else {
                emit(oi, outA, sA1, sA2, nmA, kaA, kbA, fxA);
                emit(oi + 1, outB, sB1, sB2, nmiss, kaB, kbB, fxB);
            }
        }
        if constexpr (!PIPE) advance();
        CS_STAMP(2)      // commit, LDS row loads, epilogue
    }
#ifdef CS_PROFILE
    if (lane == 0) {
        atomicAdd(&cs_prof[0], prof_[0]);
        atomicAdd(&cs_prof[1], prof_[1]);
        atomicAdd(&cs_prof[2], prof_[2]);
        atomicAdd(&cs_prof[3], prof_[3]);
        atomicAdd(&cs_prof[4], __builtin_readcyclecounter() - tstart_);
        atomicAdd(&cs_prof[5], (unsigned long long)(tstart_ != 0));
        atomicAdd(&cs_prof[6], (unsigned long long)((n_staged + 1) / 2));
    }
#endif
}

template <int K, typename TC, int MODE>
constexpr size_t corr_stream_smem_bytes()
{
    using G = StreamGeom<K>;
    return sizeof(TC) * G::NWAVES * G::wave_elems(sizeof(TC) == 4 && MODE != 1);
}

}  // namespace cs
