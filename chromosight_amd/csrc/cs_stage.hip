// cs_stage.hip -- ContactMap.create_mat of ALL intra-chromosomal blocks of a genome in three launches
// (reference contacts_map.py:527-548, 603-638; preprocessing.py:129-197, 256-310):
//
//   stage_law_kernel     balance (count * w[bin1] * w[bin2]), slice the block, trim to the diagonals 0 .. keep, and
//                        reduce the strictly positive pixels of every diagonal -- one pass over the pixel table
//   stage_finish_kernel  law[d] = sum / count per block (0 for an empty diagonal)
//   stage_tile_kernel    detrend by the law, >= max_val -> 1, NaN -> 0, and write the diagonal band (or the dense
//                        map of a short chromosome) ONCE, in float64 (exact re-scoring, windows) and / or float32
//                        (what the matrix-core tile kernel stages by LDS-DMA)
//
// The pixel table of a .cool (upper triangle of the whole genome, sorted by bin1, bin2) is one CSR matrix whose row
// r holds the pixels (r, c >= r); the intra block of chromosome b is the prefix of every row up to its last bin, so a
// "view" needs no copy and no binary search: a wave walks the row until the column leaves the band.
//
// Why these and not one launch chain per block (cs_aux.hip, still used for inter blocks, row windows, smoothing):
// a genome is 23 blocks of 3 000 - 16 000 rows; per block the old chain is 4 launches that each under-fill the chip,
// and its kernels were bound by latency, not by HBM (distance_law_kernel 1.25 TB/s: one dependent load chain per
// wave and 16 waves per CU).  Here work is cut into groups of 128 rows of one block, workgroups take groups in
// order, every wave keeps 4 x 64 pixels in flight, partial sums live in LDS and leave the workgroup as plain stores
// (one slot per group, summed by stage_finish_kernel in a fixed order: no global atomics).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cs_device.h"
#include "cs_launch_aux.h"

namespace cs {

namespace {

static_assert(sizeof(LazyBand) <= 128, "CS_LAZY_BAND_BYTES of include/chromosight_hip.h");

constexpr int kStageThreads = 512;
constexpr int kStageWaves = kStageThreads / 64;

// (stage_detrend_rcp: cs_device.h -- the lazily evaluated float64 bands recompute the same value)

// Latency, not bandwidth, bounded the first version (2.1 TB/s): per row a wave waited for the row pointers, then for the
// pixels they delimit, then for the gathered column weights -- three dependent round trips for ~270 stored pixels.  Now
// the row pointers and row weights of all the wave's rows of a group arrive together (one lane per row), the column
// weights the group can touch sit in LDS (rows r0 .. r0 + rows, columns up to keep further: one coalesced load per
// group), and the first kStageUnroll x 64 pixels of the NEXT row are requested before the current row is reduced.
constexpr int kStageGroupRows = 128;             // upper bound of rows per group (cs_api_entries.cpp cs_stage_blocks picks 64 .. 128)

__device__ __forceinline__ void stage_wave_sync()
{
    // LDS operations of one wave execute in order; this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// COUNTS: blocks flagged StageBlock::counts get their band of raw counts (CS_LAYOUT_BAND_COUNTS) written by THIS pass -- a wave has
// the stored pixels of its row in registers anyway: it drops the counts into a zeroed LDS row piece (slot = diagonal) and
// streams the piece out, 16 bytes per lane.  The detrended band of the tiler (a second pass over the pixel table, and the
// division by a law that this pass is only reducing) is not written at all: whoever reads a pixel detrends it (cs_device.h).
#ifndef CS_COUNTS_PIECE
#define CS_COUNTS_PIECE 576
#endif
constexpr int kCountsPiece = CS_COUNTS_PIECE;               // slots per LDS row piece (float32): 2.25 KB per wave

template <typename TV, int kStageUnroll, bool COUNTS>
__global__ __launch_bounds__(kStageThreads) void stage_law_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                                  const TV* __restrict__ data, const double* __restrict__ weight,
                                                                  const StageBlock* __restrict__ blocks,
                                                                  const StageGroup* __restrict__ groups, int n_groups, int pitch,
                                                                  double* __restrict__ part_sum, unsigned* __restrict__ part_cnt,
                                                                  long long* __restrict__ row_stop)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* l_sum = reinterpret_cast<double*>(smem_raw);
    unsigned* l_cnt = reinterpret_cast<unsigned*>(smem_raw + sizeof(double) * (size_t)pitch);
    double* l_w = reinterpret_cast<double*>(smem_raw + (sizeof(double) + sizeof(unsigned)) * (size_t)pitch);   // pitch + kStageGroupRows
    // (a store per row would sit between a row's pixel request and its use: memory operations retire in order, so every
    // row would wait a full write round trip -- the group's row ends leave together instead)
    long long* l_stop = reinterpret_cast<long long*>(l_w + pitch + kStageGroupRows);                             // kStageGroupRows
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* l_row = reinterpret_cast<float*>(l_stop + kStageGroupRows) + (size_t)wv * kCountsPiece;               // COUNTS: per wave
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const StageGroup G = groups[g];
        const StageBlock B = blocks[G.block];
        const long long last_col = B.row0 + B.n - 1;
        const long long r0 = B.row0 + G.row_begin;                         // first row of the group = first column it can touch
        const int n_w = (int)min((long long)(G.row_end - G.row_begin) + B.keep, last_col - r0 + 1);
        for (int d = tid; d < B.n_diags; d += kStageThreads) {
            l_sum[d] = 0.0;
            l_cnt[d] = 0u;
        }
        for (int t = tid; t < n_w; t += kStageThreads) l_w[t] = weight[r0 + t];
        if constexpr (COUNTS) {
            // the float32 copy of the block's weights (cs_device.h CountsHeader::weight32): every row belongs to one group
            if (B.counts) {
                float* w32 = reinterpret_cast<float*>(B.law + 2 * B.n_diags + 2) + (B.n_diags + 2 + 1) / 2 * 2;
                for (int t = tid; t < G.row_end - G.row_begin; t += kStageThreads) w32[G.row_begin + t] = (float)weight[r0 + t];
            }
        }
        // this wave's rows: G.row_begin + wv + 8 i, i < n_mine (<= 16) -- lane i holds row i's pointers
        const int n_mine = (G.row_end - G.row_begin - wv + kStageWaves - 1) / kStageWaves;
        long long my_b = 0, my_e = 0;
        if (lane < n_mine) {
            const long long r = r0 + wv + (long long)kStageWaves * lane;
            my_b = indptr[r];
            my_e = min(indptr[r + 1], my_b + (long long)B.keep + 1);       // columns are distinct and >= r
        }
        __syncthreads();
        auto row_b = [&](int i) { return __shfl(my_b, i); };
        auto row_e = [&](int i) { return __shfl(my_e, i); };
        int cn[kStageUnroll];                                              // the next row's first pixels, in flight
        TV xn[kStageUnroll];
        auto request = [&](int i) {
            const long long b = row_b(i), e = row_e(i);
#pragma unroll
            for (int u = 0; u < kStageUnroll; ++u) {
                const long long k = b + lane + 64 * u;
                const bool ok = k < e;
                cn[u] = ok ? indices[k] : 0x7fffffff;
                xn[u] = ok ? data[k] : (TV)0;
            }
        };
        if (n_mine > 0) request(0);
        for (int i = 0; i < n_mine; ++i) {
            const int rl = G.row_begin + wv + kStageWaves * i;
            const long long r = B.row0 + rl;
            const double wr = l_w[rl - G.row_begin];
            const long long b = row_b(i), e = row_e(i);
            const long long c_hi = min(r + (long long)B.keep, last_col);
            // (columns relative to the group's first row: 32-bit tests and LDS indices; the sentinel of a lane beyond the row
            // stays above every column)
            const int c_base = (int)r0, c_hi_rel = (int)(c_hi - r0), c_r = (int)(r - r0);
            int c[kStageUnroll];
            TV x[kStageUnroll];
#pragma unroll
            for (int u = 0; u < kStageUnroll; ++u) {
                c[u] = cn[u];
                x[u] = xn[u];
            }
            if (i + 1 < n_mine) request(i + 1);
            int n_in = 0;
            // COUNTS: the row's band leaves in pieces of kCountsPiece slots (slot = diagonal); the stored pixels arrive in column
            // order, so a piece is complete when a pixel beyond it shows up (or the row ends)
            const bool counts_row = COUNTS && B.counts;
            float* out_row = counts_row ? B.band32 + (size_t)rl * B.ld : nullptr;
            const int ld_row = (int)B.ld;
            int piece0 = 0;
            auto piece_clear = [&]() {
                for (int x = 4 * lane; x < kCountsPiece; x += 256) *reinterpret_cast<f4*>(l_row + x) = f4{0.0f, 0.0f, 0.0f, 0.0f};
            };
            auto piece_flush = [&]() {
                stage_wave_sync();
                const int len = min(ld_row - piece0, kCountsPiece);          // (ld is a multiple of 4)
                for (int x = 4 * lane; x < len; x += 256) *reinterpret_cast<f4*>(out_row + piece0 + x) = *reinterpret_cast<const f4*>(l_row + x);
                stage_wave_sync();
                piece0 += kCountsPiece;
            };
            if (counts_row) {
                piece_clear();
                stage_wave_sync();
            }
            for (long long k0 = b;;) {
#pragma unroll
                for (int u = 0; u < kStageUnroll; ++u) {
                    const int rel = c[u] - c_base;
                    const bool in = rel <= c_hi_rel;
                    n_in += __builtin_popcountll(__builtin_amdgcn_ballot_w64(in));
                    const double wc = l_w[in ? rel : 0];              // (unconditional read: behind a condition it is an exec-mask branch)
                    const double v = ((double)x[u] * wr) * wc;        // csr_value: cooler's matrix(balance=True)
                    if (in && v > 0.0) {                              // also drops NaN (preprocessing.py:188)
                        const int d = rel - c_r;
                        atomicAdd(&l_sum[d], v);
                        atomicAdd(&l_cnt[d], 1u);
                    }
                    if constexpr (COUNTS) {
                        if (counts_row) {
                            const int d = in ? rel - c_r : 0x7fffffff;
                            while (__builtin_amdgcn_ballot_w64(in && d >= piece0 + kCountsPiece)) {      // (rows wider than one piece)
                                if (in && d >= piece0 && d < piece0 + kCountsPiece) l_row[d - piece0] = (float)x[u];
                                piece_flush();
                                piece_clear();
                                stage_wave_sync();
                            }
                            if (in && d >= piece0 && d < piece0 + kCountsPiece) l_row[d - piece0] = (float)x[u];
                        }
                    }
                }
                k0 += 64 * kStageUnroll;
                if (k0 >= e) break;
                // a row with more stored pixels inside the band than one request holds: the rest, request by request
#pragma unroll
                for (int u = 0; u < kStageUnroll; ++u) {
                    const long long k = k0 + lane + 64 * u;
                    const bool ok = k < e;
                    c[u] = ok ? indices[k] : 0x7fffffff;
                    x[u] = ok ? data[k] : (TV)0;
                }
            }
            if (lane == 0) l_stop[rl - G.row_begin] = b + n_in;
            if constexpr (COUNTS) {
                if (counts_row) {
                    while (piece0 < ld_row) {
                        piece_flush();
                        if (piece0 < ld_row) {
                            piece_clear();
                            stage_wave_sync();
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (int d = tid; d < B.n_diags; d += kStageThreads) {
            part_sum[(size_t)g * pitch + d] = l_sum[d];
            part_cnt[(size_t)g * pitch + d] = l_cnt[d];
        }
        for (int t = tid; t < G.row_end - G.row_begin; t += kStageThreads) row_stop[r0 + t] = l_stop[t];
        __syncthreads();
    }
}

// 16 diagonals x 64 group phases per workgroup: the partial sums of a block's groups are added in a fixed order
// (phase by phase, the phases eight at a time, then the eight), so a law does not depend on how the groups were
// scheduled.  (64 x 16 before: a 50 000-bin block is 781 groups -- 49 dependent-latency loads per thread in 4 workgroups,
// 10 us; now 13 loads per thread in 16 workgroups.)
constexpr int kFinishDiags = 16;
constexpr int kFinishPhases = 64;
// PPT phases per thread: the same additions in the same order whatever the launch shape -- 64 phase sums per diagonal, then
// eight sums of eight, then the eight.  PPT = 1 (1024 threads) for blocks of hundreds of groups (C3: 781 groups, 13 loads
// per thread); PPT = 4 (256 threads, a thread walks four phases) when every block has at most 256 groups -- the 23 blocks of
// a genome are 1472 workgroups whose threads hold one or two partials each, and as 1024-thread workgroups they took three
// rounds of residence for 42 us between the law pass and the tiler (profiles/r04b_genome_timeline.txt).
template <int PPT>
__global__ __launch_bounds__(kFinishDiags * kFinishPhases / PPT) void stage_finish_kernel(const StageBlock* __restrict__ blocks, int pitch,
                                                           const double* __restrict__ part_sum, const unsigned* __restrict__ part_cnt,
                                                           const LazySource src)
{
    __shared__ double s_sum[kFinishPhases][kFinishDiags];
    __shared__ unsigned long long s_cnt[kFinishPhases][kFinishDiags];
    const StageBlock B = blocks[blockIdx.x];
    if (B.lazy && blockIdx.y == 0 && threadIdx.x == 0) {
        // the block's float64 band as a function of the pixel table (cs_device.h LazyBand): only its first w64 diagonals
        // are stored
        LazyBand L;
        L.indptr = src.indptr;
        L.indices = src.indices;
        L.data = src.data;
        L.weight = src.weight;
        L.law = B.law;
        L.near_ = B.band64;
        L.row0 = B.row0;
        L.near_ld = B.ld64;
        L.max_val = src.max_val;
        L.n = B.n;
        L.n_diags = B.n_diags;
        L.near_w = B.band64 ? (B.w64 > 0 ? min(B.w64, B.width) : B.width) : 0;
        L.data_is_f64 = src.data_is_f64;
        L.counts = nullptr;
        L.counts_ld = 0;
        if (B.counts) {                 // every kept diagonal is in memory, as counts: nothing is searched
            L.near_ = nullptr;
            L.near_w = B.n_diags;
            L.counts = B.band32;
            L.counts_ld = B.ld;
        }
        *B.lazy = L;
    }
    if (B.counts && blockIdx.y == 0 && threadIdx.x == 0) {
        CountsHeader H;
        H.weight = src.weight;
        H.law = B.law;
        H.rlaw = B.law + B.n_diags + 1;
        H.row0 = B.row0;
        H.max_val = src.max_val;
        H.n = B.n;
        H.n_diags = B.n_diags;
        float* r32 = reinterpret_cast<float*>(B.law + 2 * B.n_diags + 2);
        H.rlaw32 = r32 + 1;
        H.weight32 = r32 + (B.n_diags + 2 + 1) / 2 * 2;
        *reinterpret_cast<CountsHeader*>(reinterpret_cast<char*>(B.band32) - kCountsHeaderBytes) = H;
        B.law[B.n_diags] = 0.0;                       // rlaw[-1], rlaw[n_diags]: read beside a real neighbour, never used
        B.law[2 * B.n_diags + 1] = 0.0;
        r32[0] = 0.0f;
        r32[B.n_diags + 1] = 0.0f;
    }
    const int dx = threadIdx.x % kFinishDiags, t_ph = threadIdx.x / kFinishDiags;
    const int d = blockIdx.y * kFinishDiags + dx;
    const int g_end = B.group0 + B.n_groups;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int ph = t_ph * PPT + j;
        double s = 0.0;
        unsigned long long c = 0;
        if (d < B.n_diags) {
            // eight groups' partials requested together (the order of the additions stays the one of the plain loop)
            int g = B.group0 + ph;
            for (; g + 7 * kFinishPhases < g_end; g += 8 * kFinishPhases) {
                double ps[8];
                unsigned pc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ps[u] = part_sum[(size_t)(g + u * kFinishPhases) * pitch + d];
                    pc[u] = part_cnt[(size_t)(g + u * kFinishPhases) * pitch + d];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    s += ps[u];
                    c += pc[u];
                }
            }
            for (; g < g_end; g += kFinishPhases) {
                s += part_sum[(size_t)g * pitch + d];
                c += part_cnt[(size_t)g * pitch + d];
            }
        }
        s_sum[ph][dx] = s;
        s_cnt[ph][dx] = c;
    }
    __syncthreads();
    double s = 0.0;
    unsigned long long c = 0;
    if (t_ph < 8) {
        for (int k = 8 * t_ph; k < 8 * t_ph + 8; ++k) {
            s += s_sum[k][dx];
            c += s_cnt[k][dx];
        }
    }
    __syncthreads();
    if (t_ph < 8) {
        s_sum[t_ph][dx] = s;
        s_cnt[t_ph][dx] = c;
    }
    __syncthreads();
    if (t_ph == 0 && d < B.n_diags) {
        s = 0.0;
        c = 0;
        for (int k = 0; k < 8; ++k) {
            s += s_sum[k][dx];
            c += s_cnt[k][dx];
        }
        const double y = c > 0 ? s / (double)c : 0.0;       // cs_distance_law_finish
        B.law[d] = y;
        if (B.counts) {
            const double ry = 1.0 / y;                      // (what stage_tile_kernel takes per group: l_law)
            B.law[B.n_diags + 1 + d] = ry;
            reinterpret_cast<float*>(B.law + 2 * B.n_diags + 2)[1 + d] = (float)ry;
        }
    }
}

// A block's rows are written exactly once.  A wave builds its row in LDS -- zero it, scatter the detrended stored pixels
// into their slots (slot = diagonal of a band, column of a dense block) -- and streams it out with 16-byte-per-lane
// stores (1 KB per instruction, float64 and float32 copies from the same LDS row).  Writing gaps from the lane that holds
// the previous stored pixel (the first version, and cs_aux.hip's block-by-block tiler) turns a sparse row -- Hi-C rows
// ARE sparse away from the diagonal: 60 % of the slots of the bench genome are gaps -- into thousands of divergent
// single-element stores: 3.1 ms for the 23-block genome (1.0 TB/s), against 0.3 ms for the same reads in the law kernel.
// Rows are built in pieces of kStageRowMax slots (columns are sorted: a piece continues where the last one stopped), which
// keeps three workgroups per CU resident.
constexpr int kStageRowMax = 544;                // slots per LDS row piece: 4.25 KB per wave, 34 KB per workgroup (+ law and weights: 52 KB, three per CU)

// Per group: the law and the column weights the group can touch go to LDS, lane i of a wave holds the first and last stored
// pixel of the wave's i-th row, and the first 64 x kStageUnroll stored pixels of the NEXT row are requested before the current
// row is assembled -- a row whose stored pixels fit one request (kStageUnroll is picked by the width of the band) is built
// without waiting for a load.  What still serialises: memory operations of a wave retire in order, so the use of a request
// also waits for the previous row's stores.  (Tried: builder waves that only load and assemble + writer waves that only
// store, a workgroup barrier per piece -- 1.27 ms against 0.95 for the 23-block genome; the stores alone take 0.39 ms,
// tools/ubench/write_rate.hip.)
#ifndef CS_STAGE_TILE_WAVES
#define CS_STAGE_TILE_WAVES 4
#endif
template <typename TV, int kStageUnroll>
__global__ __launch_bounds__(kStageThreads, CS_STAGE_TILE_WAVES) void stage_tile_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                                          const TV* __restrict__ data, const double* __restrict__ weight,
                                                                          const StageBlock* __restrict__ blocks,
                                                                          const StageGroup* __restrict__ groups, int n_groups, int pitch,
                                                                          const long long* __restrict__ row_stop, double max_val, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* l_law = reinterpret_cast<double*>(smem_raw);
    double* l_w = l_law + pitch;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double* l_row = l_w + pitch + kStageGroupRows + (size_t)wv * kStageRowMax;
    typedef double d2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const StageGroup G = groups[g];
        const StageBlock B = blocks[G.block];
        if (B.counts && !B.band64) continue;               // (uniform) the law pass wrote this block's band
        const long long last_col = B.row0 + B.n - 1;
        const long long r0 = B.row0 + G.row_begin;
        const int n_w = (int)min((long long)(G.row_end - G.row_begin) + B.keep, last_col - r0 + 1);
        __syncthreads();
        // (the RECIPROCAL of the law: one division per diagonal and group instead of one per stored pixel, cs_device.h
        // stage_detrend_rcp)
        for (int d = tid; d < B.n_diags; d += kStageThreads) l_law[d] = 1.0 / B.law[d];
        for (int t = tid; t < n_w; t += kStageThreads) l_w[t] = weight[r0 + t];
        const int n_mine = (G.row_end - G.row_begin - wv + kStageWaves - 1) / kStageWaves;
        long long my_b = 0, my_e = 0;
        if (lane < n_mine) {
            const long long r = r0 + wv + (long long)kStageWaves * lane;
            my_b = indptr[r];
            my_e = row_stop[r];
        }
        __syncthreads();
        int cn[kStageUnroll];
        TV xn[kStageUnroll];
        auto request = [&](int i) {
            // (a uniform base pointer + a 32-bit lane offset per load: no 64-bit address per load in registers)
            const long long b = __shfl(my_b, i);
            const int n = (int)(__shfl(my_e, i) - b);
            const int* ip = indices + b;
            const TV* dp = data + b;
#pragma unroll
            for (int u = 0; u < kStageUnroll; ++u) {
                const int off = lane + 64 * u;
                const bool ok = off < n;
                cn[u] = ok ? ip[off] : 0x7fffffff;
                xn[u] = ok ? dp[off] : (TV)0;
            }
        };
        if (n_mine > 0) request(0);
        const int ld = (int)B.ld;
        const int n_diags_m1 = max(B.n_diags - 1, 0);
        const double kInf = __longlong_as_double(0x7ff0000000000000ll);
        const int n_pieces = (ld + kStageRowMax - 1) / kStageRowMax;
        const int piece_len = ((ld + n_pieces - 1) / n_pieces + 15) & ~15;
        for (int i = 0; i < n_mine; ++i) {
            const int rl = G.row_begin + wv + kStageWaves * i;
            const long long r = B.row0 + rl;
            const double wr = l_w[rl - G.row_begin];
            const long long b = __shfl(my_b, i), e = __shfl(my_e, i);
            double* out64 = (B.band64 && !(dbg & 1)) ? B.band64 + (size_t)rl * B.ld64 : nullptr;
            const int w64 = B.w64 > 0 ? ((B.w64 + 1) & ~1) : (int)B.ld;               // (stored in pairs; ld64 is even)
            float* out32 = (B.band32 && !B.counts && !(dbg & 2)) ? B.band32 + (size_t)rl * B.ld : nullptr;
            const long long x0 = B.dense ? B.row0 : r;
            double v[kStageUnroll];
            int slot[kStageUnroll];
            const int c_r = (int)(r - r0), c_x0 = (int)(x0 - r0);        // columns relative to the group's first row: 32-bit
            const int c_base = (int)r0;                                  // (pixel tables of fewer than 2^31 bins: the indices are int32)
#pragma unroll
            for (int u = 0; u < kStageUnroll; ++u) {
                __builtin_amdgcn_sched_barrier(0);                       // one division at a time: they would all be in flight
                const bool ok = cn[u] != 0x7fffffff;
                const int rel = ok ? cn[u] - c_base : 0;                 // column - r0
                const int d = rel - c_r;
                // (both LDS reads unconditional, from clamped indices: behind a condition each became an exec-mask branch)
                const double wc = l_w[rel];
                const double iy = l_law[min(max(d, 0), n_diags_m1)];
                v[u] = stage_detrend_rcp(((double)xn[u] * wr) * wc, (ok && d < B.n_diags) ? iy : kInf, max_val, B.law + min(max(d, 0), n_diags_m1));
                slot[u] = ok ? rel - c_x0 : -1;
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool fits = e - b <= 64 * kStageUnroll;
            if (i + 1 < n_mine) request(i + 1);
            long long k_next = b;
            for (int p = 0; p < n_pieces; ++p) {
                const int s0 = p * piece_len, s1 = min(ld, s0 + piece_len);
                for (int x = 2 * lane; x < s1 - s0; x += 128) *reinterpret_cast<d2*>(l_row + x) = d2{0.0, 0.0};
                stage_wave_sync();
                if (fits) {
#pragma unroll
                    for (int u = 0; u < kStageUnroll; ++u)
                        if (slot[u] >= s0 && slot[u] < s1) l_row[slot[u] - s0] = v[u];
                } else {
                    const long long c_end = x0 + s1;
                    int placed = 0;
                    constexpr int kSlow = 2;                            // (this path keeps few registers: it bounds the kernel's)
                    for (long long k0 = k_next + lane; k0 - lane < e; k0 += 64 * kSlow) {
                        int c[kSlow];
                        TV x[kSlow];
#pragma unroll
                        for (int u = 0; u < kSlow; ++u) {
                            const long long k = k0 + 64 * u;
                            const bool ok = k < e;
                            c[u] = ok ? indices[k] : 0x7fffffff;
                            x[u] = ok ? data[k] : (TV)0;
                        }
                        bool more = true;
#pragma unroll
                        for (int u = 0; u < kSlow; ++u) {
                            const bool in = (long long)c[u] < c_end;
                            const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
                            placed += __builtin_popcountll(m);
                            if (in) {
                                const int rel = c[u] - c_base;
                                const int d = rel - c_r;
                                l_row[rel - c_x0 - s0] = stage_detrend_rcp(((double)x[u] * wr) * l_w[rel], d < B.n_diags ? l_law[d] : kInf, max_val,
                                                                           B.law + min(max(d, 0), n_diags_m1));
                            }
                            more = more && (m == ~0ull);                // a lane beyond the piece (or the row): this piece is complete
                        }
                        if (!more) break;
                    }
                    k_next += placed;
                }
                stage_wave_sync();
                // (one read of the piece serving both copies -- 4 slots per lane, float64 stores 32 bytes apart -- was slower:
                // 1.12 against 0.95 ms for the genome; a store instruction should cover one dense kilobyte)
                if (out64)
                    for (int x = 2 * lane; x < min(s1, w64) - s0; x += 128) *reinterpret_cast<d2*>(out64 + s0 + x) = *reinterpret_cast<const d2*>(l_row + x);
                if (out32)
                    for (int x = 4 * lane; x < s1 - s0; x += 256) {
                        const d2 a = *reinterpret_cast<const d2*>(l_row + x), c2 = *reinterpret_cast<const d2*>(l_row + x + 2);
                        *reinterpret_cast<f4*>(out32 + s0 + x) = f4{(float)a[0], (float)a[1], (float)c2[0], (float)c2[1]};
                    }
                stage_wave_sync();
            }
        }
    }
}

}  // namespace

size_t stage_table_bytes(int n_blocks, int n_groups)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al(sizeof(StageBlock) * (size_t)n_blocks) + al(sizeof(StageGroup) * (size_t)n_groups);
}

size_t stage_scratch_bytes(int n_blocks, int n_groups, int pitch, long long n_rows)
{
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al(sizeof(StageBlock) * (size_t)n_blocks) + al(sizeof(StageGroup) * (size_t)n_groups) +
           al(sizeof(double) * (size_t)n_groups * pitch) + al(sizeof(unsigned) * (size_t)n_groups * pitch) +
           al(sizeof(long long) * (size_t)n_rows) + 1024;
}

// h_blocks: group0 / n_groups are filled here.  Everything is enqueued on `stream`; `scratch` holds
// stage_scratch_bytes(...) bytes and must stay untouched until the stream has drained.
int enqueue_stage_blocks(const long long* indptr, const int* indices, const void* data, int data_is_f64, const double* weight,
                         long long n_rows, StageBlock* h_blocks, int n_blocks, double max_val, int rows_per_group, int n_cu,
                         void* scratch, void* h_tables, hipStream_t stream, std::vector<char>* uploaded)
{
    std::vector<StageGroup> groups;
    int pitch = 1;
    for (int b = 0; b < n_blocks; ++b) {
        StageBlock& B = h_blocks[b];
        B.group0 = (int)groups.size();
        for (int r0 = 0; r0 < B.n; r0 += rows_per_group) groups.push_back(StageGroup{b, r0, std::min(B.n, r0 + rows_per_group)});
        B.n_groups = (int)groups.size() - B.group0;
        pitch = std::max(pitch, B.n_diags);
    }
    const int n_groups = (int)groups.size();
    if (n_groups == 0) return 0;
    pitch = (pitch + 63) / 64 * 64;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char* p = (char*)scratch;
    StageBlock* d_blocks = (StageBlock*)p;
    p += al(sizeof(StageBlock) * (size_t)n_blocks);
    StageGroup* d_groups = (StageGroup*)p;
    p += al(sizeof(StageGroup) * (size_t)n_groups);
    double* part_sum = (double*)p;
    p += al(sizeof(double) * (size_t)n_groups * pitch);
    unsigned* part_cnt = (unsigned*)p;
    p += al(sizeof(unsigned) * (size_t)n_groups * pitch);
    long long* row_stop = (long long*)p;
    // the two tables travel from page-locked memory the caller keeps alive (stage_table_bytes; one copy, no synchronisation)
    const size_t blocks_bytes = al(sizeof(StageBlock) * (size_t)n_blocks);
    std::memcpy(h_tables, h_blocks, sizeof(StageBlock) * (size_t)n_blocks);
    std::memcpy((char*)h_tables + blocks_bytes, groups.data(), sizeof(StageGroup) * (size_t)n_groups);
    // (a genome staged again with the same layout and outputs -- every step of a run -- finds its tables on the device:
    // `uploaded` holds what the scratch at this address received last, and the 5 us copy in front of the chain is skipped)
    const size_t table_bytes = blocks_bytes + sizeof(StageGroup) * (size_t)n_groups;
    const size_t key_bytes = table_bytes + sizeof(void*);
    bool same = uploaded && uploaded->size() == key_bytes && std::memcmp(uploaded->data(), &scratch, sizeof(void*)) == 0;
    if (same) {
        // (the padding between the two tables is not initialised: compare them one by one)
        same = std::memcmp(uploaded->data() + sizeof(void*), h_blocks, sizeof(StageBlock) * (size_t)n_blocks) == 0 &&
               std::memcmp(uploaded->data() + sizeof(void*) + blocks_bytes, groups.data(), sizeof(StageGroup) * (size_t)n_groups) == 0;
    }
    if (!same) {
        hipError_t e = hipMemcpyAsync(d_blocks, h_tables, table_bytes, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return (int)e;
        if (uploaded) {
            uploaded->assign(key_bytes, 0);
            std::memcpy(uploaded->data(), &scratch, sizeof(void*));
            std::memcpy(uploaded->data() + sizeof(void*), h_blocks, sizeof(StageBlock) * (size_t)n_blocks);
            std::memcpy(uploaded->data() + sizeof(void*) + blocks_bytes, groups.data(), sizeof(StageGroup) * (size_t)n_groups);
        }
    }
    const int per_cu = 4;        // (6 or 8 workgroups per CU: the law pass 170 against 154 us on the genome)
    const int unroll = 4;
    const int grid = std::min(n_groups, per_cu * n_cu);
    if (rows_per_group > kStageGroupRows) return (int)hipErrorInvalidValue;
    bool any_counts = false, any_tiled = false;
    for (int b = 0; b < n_blocks; ++b) {
        any_counts = any_counts || h_blocks[b].counts;
        any_tiled = any_tiled || !h_blocks[b].counts || h_blocks[b].band64;
    }
    const size_t smem_law = (sizeof(double) + sizeof(unsigned)) * (size_t)pitch + sizeof(double) * ((size_t)pitch + 2 * kStageGroupRows) +
                            (any_counts ? sizeof(float) * (size_t)kStageWaves * kCountsPiece : 0);
    const size_t smem_tile = sizeof(double) * (2 * (size_t)pitch + kStageGroupRows + (size_t)kStageWaves * kStageRowMax);
    int max_b = 0;
    for (int b = 0; b < n_blocks; ++b) max_b = std::max(max_b, h_blocks[b].n_diags);
#define CS_STAGE_LAW_(TV, U, CN)                                                                                                  \
    do {                                                                                                                           \
        if (smem_law > 48 * 1024)       /* laws beyond ~ 2300 diagonals: more dynamic LDS than a launch gets by default */           \
            (void)hipFuncSetAttribute((const void*)stage_law_kernel<TV, U, CN>, hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                      160 * 1024);                                                                                 \
        hipLaunchKernelGGL((stage_law_kernel<TV, U, CN>), dim3(grid), dim3(kStageThreads), smem_law, stream, indptr, indices,       \
                           (const TV*)data, weight, d_blocks, d_groups, n_groups, pitch, part_sum, part_cnt, row_stop);            \
    } while (0)
#define CS_STAGE_LAW(TV, U)                      \
    do {                                         \
        if (any_counts) CS_STAGE_LAW_(TV, U, true);  \
        else CS_STAGE_LAW_(TV, U, false);        \
    } while (0)
#define CS_STAGE_TILE(TV, U)                                                                                                       \
    do {                                                                                                                           \
        if (smem_tile > 48 * 1024)                                                                                                 \
            (void)hipFuncSetAttribute((const void*)stage_tile_kernel<TV, U>, hipFuncAttributeMaxDynamicSharedMemorySize,           \
                                      160 * 1024);                                                                                 \
        hipLaunchKernelGGL((stage_tile_kernel<TV, U>), dim3(grid), dim3(kStageThreads), smem_tile, stream, indptr, indices,          \
                           (const TV*)data, weight, d_blocks, d_groups, n_groups, pitch, row_stop, max_val, stage_dbg);             \
    } while (0)
#define CS_STAGE_BOTH(WHAT, UNROLL)               \
    if (data_is_f64) {                           \
        if (UNROLL >= 8) WHAT(double, 8);        \
        else if (UNROLL >= 4) WHAT(double, 4);   \
        else WHAT(double, 2);                    \
    } else {                                     \
        if (UNROLL >= 8) WHAT(float, 8);         \
        else if (UNROLL >= 4) WHAT(float, 4);    \
        else WHAT(float, 2);                     \
    }
    // the tiler keeps a whole row's stored pixels in registers when they fit its request (64 x unroll): by the widest band
    const int unroll_tile = max_b > 256 ? 8 : 4;
    const int stage_dbg = 0;
    CS_STAGE_BOTH(CS_STAGE_LAW, unroll)
    const LazySource lazy_src{indptr, indices, data, weight, max_val, data_is_f64};
    int max_groups = 0;
    for (int b = 0; b < n_blocks; ++b) max_groups = std::max(max_groups, h_blocks[b].n_groups);
    const dim3 finish_grid(n_blocks, (max_b + kFinishDiags - 1) / kFinishDiags);
    // (few workgroups: they are all resident at once either way, and the 1024-thread shape has the shorter per-thread chain)
    if (max_groups <= 256 && (long long)finish_grid.x * finish_grid.y > 512)
        hipLaunchKernelGGL(stage_finish_kernel<4>, finish_grid, dim3(kFinishDiags * kFinishPhases / 4), 0, stream, d_blocks, pitch, part_sum,
                           part_cnt, lazy_src);
    else
        hipLaunchKernelGGL(stage_finish_kernel<1>, finish_grid, dim3(kFinishDiags * kFinishPhases), 0, stream, d_blocks, pitch, part_sum,
                           part_cnt, lazy_src);
    if (any_tiled) { CS_STAGE_BOTH(CS_STAGE_TILE, unroll_tile) }
#undef CS_STAGE_BOTH
#undef CS_STAGE_LAW_
#undef CS_STAGE_LAW
#undef CS_STAGE_TILE
    (void)n_rows;
    return (int)hipGetLastError();
}

}  // namespace cs
