// cs_api.cpp -- the C ABI (include/chromosight_hip.h): argument validation, template
// statistics, weight upload and kernel dispatch.  No torch types, no retained pointers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <memory>
#include <condition_variable>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <unordered_map>
#include <type_traits>
#include <vector>

#include "../../include/chromosight_hip.h"
#include "cs_device.h"
#include "cs_launch.h"
#include "cs_launch_aux.h"

constexpr int kBlkLanes = 6;        // streams a multi-block tile pass may use (the caller's + side streams)

struct cs_ctx {
    int device = 0;
    int n_cu = 0;
    std::string err;
    // device buffers for the template weights of the call in flight (3 * kk elements), one per
    // arithmetic type: a detect call runs the float32 map kernel and the float64 re-scoring with the
    // same template, and neither upload should evict the other
    void* d_w[2] = {nullptr, nullptr};
    size_t d_w_bytes[2] = {0, 0};
    // host copies of what d_w currently holds, to skip re-uploads of an unchanged template
    std::vector<unsigned char> w_cached[2];
    // the weight sets of the templates used before the current one (calls that alternate between a few templates, or
    // hand several to one launch chain: cs_detect_foci_batch_templates) -- swapped back in instead of uploaded again
    struct ParkedWeights {
        void* d = nullptr;
        size_t bytes = 0;
        std::vector<unsigned char> host;
        unsigned long long stamp = 0;
    };
    ParkedWeights w_parked[2][3];
    unsigned long long w_clock = 0;
    // cs_detect_foci: coefficient map and candidate / foci scratch (grow-only), pinned counters
    void* d_map = nullptr;
    size_t d_map_bytes = 0;
    void* d_stage = nullptr;         // cs_stage_blocks: tables, per-group partial sums, row extents
    size_t d_stage_bytes = 0;
    void* h_stage[2] = {nullptr, nullptr};       // page-locked staging of its tables, two slots
    size_t h_stage_bytes[2] = {0, 0};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    int stage_slot = 0;
    // cs_detect_foci_blocks: the tile kernels of consecutive blocks alternate over the caller's stream and two side
    // streams (each with its own mask-table scratch), so that one block's tail overlaps the next block's ramp
    hipStream_t s_blk[kBlkLanes - 1] = {};
    hipEvent_t ev_blk[kBlkLanes] = {};
    void* ws_alt[kBlkLanes - 1] = {};
    size_t ws_alt_bytes[kBlkLanes - 1] = {};
    int grid_cap = 0;                  // CorrArgs::grid_cap of the launches built next (cs_detect_foci_blocks)
    // one launch for the tiles of all blocks (cs::launch_corr_mfma_blocks): mask tables per block, argument table
    std::vector<void*> ws_tab;
    std::vector<size_t> ws_tab_bytes;
    void* h_tab = nullptr;
    void* d_tab = nullptr;
    size_t tab_bytes = 0;
    // one device word: set by the last workgroup of a multi-block tile launch when it starts (all of them are resident then),
    // consumed by cs_stream_wait_tiles -- the dependency that lets a side chain run in what the tile workgroups leave
    unsigned* d_tiles_started = nullptr;
    // cs_detect_foci_blocks in two calls (cs_foci_params.reserved & 2: the prepare form): what the prepare form enqueued is valid for a
    // call with this key
    bool prep_pending = false, skip_prep_launch = false;
    // the float64 argument blocks a prepare form uploaded from: the copy on a side lane may still be reading them when the
    // prepare form returns, so they live here until the next foci entry on this context
    std::vector<cs::CorrArgs<double>> prep_tab_keep;
    // the mask tables of all blocks of a multi-block tile launch in one launch: while set, prepare_regular_mask<float> appends a
    // block's arguments and its number of workgroups here instead of launching (cs::launch_mask_prep_batch)
    std::vector<cs::MaskPrepArgs<float>>* prep_collect = nullptr;
    std::vector<int> prep_groups;
    unsigned long long prep_key = 0;
    void* d_pool = nullptr;
    size_t d_pool_bytes = 0;
    long long* h_counts = nullptr;   // pinned: [0] candidates, [1] foci
    long long* h_cand_counts = nullptr;   // pinned, 256 entries (cs_detect_foci_blocks, segmented lists): [b] the blocks' own candidate counts, [60] their
                                          // clamped total, [61] status flags; [64 + b] / [128 + b]: the regions' starts / rooms the device reads
    void* d_counts_peak = nullptr;   // cs_normxcorr2_host: largest |pixel| of the map (float bits), and its pinned copy
    unsigned* h_peak = nullptr;
    // grow-only scratch for the mask tables of the streaming kernel (one call in flight per context)
    void* d_ws = nullptr;
    size_t d_ws_bytes = 0;
    // matrix-core kernel: the float32 weight sets as float16 head / tail Toeplitz fragments
    void* d_wfrag = nullptr;
    // cs_detect_foci_batch_templates: host tables of the call in flight (asynchronous mode), its virtual blocks and capacity
    std::vector<cs::CorrArgs<double>> nb_tab;
    std::vector<long long> nb_seg;
    std::vector<int> nb_lo_w;
    int nb_pending = 0;
    long long nb_cap = 0;
    std::vector<char> stage_uploaded;   // cs_stage_blocks: the tables the staging scratch holds (skip the upload of identical ones)
    void* d_rim = nullptr;          // rim tables of the mask weight sets (cs_launch.h MfmaWeights::rim), same key as d_wfrag
    std::vector<unsigned char> wfrag_key;     // the float32 weights the image was built from
    int wfrag_km = 0, wfrag_kn = 0;
    float wfrag_unscale[3] = {1.0f, 1.0f, 1.0f};
    // ... and for the two-pass kernel of the templates of up to 33 x 33 (cs_launch.h MfmaWideWeights), with its own key
    void* d_wfrag_wide = nullptr;
    size_t d_wfrag_wide_bytes = 0;
    std::vector<unsigned char> wfrag_wide_key;
    int wfrag_wide_km = 0, wfrag_wide_kn = 0;
    float wfrag_wide_unscale[3] = {1.0f, 1.0f, 1.0f};
    int last_kernel = 0;     // cs_last_kernel()
    int range_check = 0;     // cs_ctx_set_range_check()
    bool cand_fused = false; // the last candidate-mode call appended its candidates itself (no map was written)
    long long cand_hint = 0, cand_hint_pixels = 0;   // cs_detect_foci_blocks: candidates, pixels and blocks of the previous call
    int cand_hint_blocks = 0;
    bool cand_hint_paced = false;    // ... and its lists needed the host-paced chain (too long for the labelling workgroups' LDS arrays)
    bool allow_lazy = false; // the entry in progress takes CS_LAYOUT_BAND_LAZY signals (check_matrix)
    bool allow_counts = false;   // ... CS_LAYOUT_BAND_COUNTS signals for its float32 tile kernel
    long long uploads = 0;   // template weights / fragments / rim tables copied to the device so far (upload_weights, ensure_wfrag)
    // what build_args derives from a template (statistics, the three weight sets, symmetry, threshold
    // flags), per arithmetic type: a detect run calls with the same template thousands of times
    struct TemplateCache {
        std::vector<double> key;      // km, kn, flags, threshold, then the template arrays as passed
        double stats[6] = {0, 0, 0, 0, 0, 0};
        std::vector<double> w;
        bool sym = false;
        bool rank1 = false;           // template == u v^T exactly: u, v appended to w (float32 kernels)
        int zk_possible = 1, snap_possible = 1;
    } tcache[2];
    long long* h_blk_counts = nullptr;     // page-locked: total + per-block foci counts of cs_detect_foci_batch
    size_t h_blk_bytes = 0;
    void* d_narrow = nullptr;       // float32 copy of a float64 dense signal for the matrix-core kernel
    size_t d_narrow_bytes = 0;
    // cs_normxcorr2_host: device staging of the map, pinned bounce buffer of the float32 result, three
    // streams (upload / kernels / download) and one event pair per row slab, all grow-only
    void* d_host_in = nullptr;
    void* d_host_out = nullptr;
    size_t d_host_bytes = 0;
    void* h_bounce = nullptr;
    size_t h_bounce_bytes = 0;
    hipStream_t s_up = nullptr, s_run = nullptr, s_down = nullptr;
    std::vector<hipEvent_t> ev_up, ev_run, ev_down;
};

static int ensure_scratch(cs_ctx* ctx, void** buf, size_t* have, size_t need);

namespace {

// CHROMOSIGHT_HIP_TIMING=1: host-side lap times of the batched entries on stderr (where a call's microseconds go before its
// kernels are on the device)
// Worker threads of the host-side passes (cs_accept_records), kept between calls: starting seven threads for the 7 000 records
// of a rank's share cost more than their work (128 us for 30 us of arithmetic), and on a genome the 1-D pattern's 56 000
// records are the last thing a step waits for once its launch chain runs behind the tile kernels.  Tasks are taken from a
// shared counter by the workers AND the caller; a second caller at the same time runs a short job itself and waits with a long one.
class HostPool {
public:
    static HostPool& get()
    {
        static HostPool* p = new HostPool();           // (never destroyed: the detached workers may be waiting at exit)
        return *p;
    }
    template <typename F>
    void run(int n_tasks, int max_threads, const F& fn)
    {
        if (n_tasks <= 0) return;
        if (n_tasks == 1 || max_threads <= 1) {          // (a single task never takes the pool from a caller that has many)
            for (int t = 0; t < n_tasks; ++t) fn(t);
            return;
        }
        // A second caller at the same time: a short job runs its tasks itself; a long one WAITS for the pool -- the two patterns of a
        // genome step end within microseconds of each other every few steps, and the 1-D pattern's 56 000 records then took 640 us
        // on the calling thread alone instead of 150 us on the pool behind the 2-D pattern's 40 us (profiles/r05_genome_step_modes.txt)
        std::unique_lock<std::mutex> busy(busy_mu_, std::try_to_lock);
        if (!busy.owns_lock()) {
            if (n_tasks <= 4) {
                for (int t = 0; t < n_tasks; ++t) fn(t);
                return;
            }
            busy.lock();
        }
        const int want = std::min(std::min(max_threads, n_tasks) - 1, kMaxWorkers);
        grow(want);
        Job job;
        job.fn = [](const void* f, int t) { (*static_cast<const F*>(f))(t); };
        job.ctx = &fn;
        job.n_tasks = n_tasks;
        job.next.store(0, std::memory_order_relaxed);
        job.active.store(0, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job;
            wanted_ = want;
            ++generation_;
        }
        cv_.notify_all();
        for (int t; (t = job.next.fetch_add(1, std::memory_order_relaxed)) < n_tasks;) fn(t);
        {
            // no worker joins from here on (they register under the same lock); the ones that did are waited for -- a worker
            // that wakes up late finds no job instead of holding the caller up
            std::lock_guard<std::mutex> lk(mu_);
            job_ = nullptr;
        }
        int spins = 0;
        while (job.active.load(std::memory_order_acquire) != 0)
            if (++spins > 2000) std::this_thread::yield();
    }

private:
    static constexpr int kMaxWorkers = 63;
    struct Job {
        void (*fn)(const void*, int) = nullptr;
        const void* ctx = nullptr;
        int n_tasks = 0;
        std::atomic<int> next{0};
        std::atomic<int> active{0};        // workers that took the job and have not finished with it
    };
    void grow(int n)
    {
        while ((int)threads_ < n) {
            const int id = (int)threads_++;
            std::thread([this, id] { loop(id); }).detach();
        }
    }
    void loop(int id)
    {
        long long seen = 0;
        for (;;) {
            Job* job = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                if (id < wanted_ && job_) {
                    job = job_;
                    job->active.fetch_add(1, std::memory_order_relaxed);
                }
            }
            if (!job) continue;
            for (int t; (t = job->next.fetch_add(1, std::memory_order_relaxed)) < job->n_tasks;) job->fn(job->ctx, t);
            job->active.fetch_sub(1, std::memory_order_release);
        }
    }
    std::mutex busy_mu_, mu_;
    std::condition_variable cv_;
    Job* job_ = nullptr;
    int wanted_ = 0;
    long long generation_ = 0;
    size_t threads_ = 0;
};

// cs_stream_wait_tiles: one wave that sleeps until the word has reached `epoch` (the tile workgroups of the launch that carries this
// epoch are resident) and gives up after `ticks` of the constant-rate counter (100 MHz) -- the word is a scheduling hint, never a lock
__global__ void cs_wait_tiles_kernel(const unsigned* word, unsigned epoch, long long ticks)
{
    // (epochs only grow: a word left by an earlier launch never lets a later wait through; the difference is taken modulo 2^32)
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0 && wall_clock64() - t0 < ticks)
        __builtin_amdgcn_s_sleep(16);
}

struct Laps {
    bool on;
    std::chrono::steady_clock::time_point t0, last;
    const char* what;
    explicit Laps(const char* w) : on(std::getenv("CHROMOSIGHT_HIP_TIMING") != nullptr), what(w) { t0 = last = std::chrono::steady_clock::now(); }
    void lap(const char* name)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] %s: %-28s +%7.1f us (at %7.1f)\n", what, name,
                std::chrono::duration<double, std::micro>(now - last).count(), std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};

int fail(cs_ctx* ctx, int code, const char* fmt, ...)
{
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}

#define CS_HIP(ctx, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ctx, CS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));     \
    } while (0)

// every entry point that launches, copies or allocates first makes the context's GPU current: a
// process may hold contexts on several devices
#define CS_ENTER(ctx)                                                                        \
    do {                                                                                     \
        if (!(ctx)) return CS_ERR_INVALID;                                                   \
        CS_HIP(ctx, hipSetDevice((ctx)->device));                                            \
    } while (0)

// entries whose float64 kernels all read their signal through load_signal take lazily evaluated bands (cs_stage_block)
struct AllowLazy {
    cs_ctx* c;
    bool was, was_counts;
    explicit AllowLazy(cs_ctx* c_);
    ~AllowLazy();
};
inline bool is_band(int layout) { return layout == CS_LAYOUT_BAND || layout == CS_LAYOUT_BAND_LAZY || layout == CS_LAYOUT_BAND_PADDED || layout == CS_LAYOUT_BAND_COUNTS || layout == CS_LAYOUT_BAND_COUNTS_VIEW; }

cs::MatView view_of(const cs_matrix* m)
{
    cs::MatView v;
    v.ptr = m ? m->d_ptr : nullptr;
    v.ld = m ? m->ld : 0;
    v.layout = m ? ((m->layout == CS_LAYOUT_BAND_PADDED || m->layout == CS_LAYOUT_BAND_COUNTS || m->layout == CS_LAYOUT_BAND_COUNTS_VIEW) ? CS_LAYOUT_BAND : m->layout) : 0;
    v.band_lo = m ? m->band_lo : 0;
    v.band_w = m ? m->band_w : 0;
    v.row0 = m ? m->row0 : 0;
    v.pad = (m && (m->layout == CS_LAYOUT_BAND_PADDED || m->layout == CS_LAYOUT_BAND_COUNTS)) ? 1 : 0;
    v.counts = (m && (m->layout == CS_LAYOUT_BAND_COUNTS || m->layout == CS_LAYOUT_BAND_COUNTS_VIEW)) ? 1 : 0;
    return v;
}

AllowLazy::AllowLazy(cs_ctx* c_) : c(c_), was(c_ ? c_->allow_lazy : false), was_counts(c_ ? c_->allow_counts : false)
{
    if (c) c->allow_lazy = c->allow_counts = true;
}
AllowLazy::~AllowLazy()
{
    if (c) {
        c->allow_lazy = was;
        c->allow_counts = was_counts;
    }
}

// the entry in progress takes a CS_LAYOUT_BAND_COUNTS signal (cs_normxcorr2: the masked tile kernel or an error)
struct AllowCounts {
    cs_ctx* c;
    bool was;
    explicit AllowCounts(cs_ctx* c_) : c(c_), was(c_ ? c_->allow_counts : false)
    {
        if (c) c->allow_counts = true;
    }
    ~AllowCounts()
    {
        if (c) c->allow_counts = was;
    }
};

struct HostStats {
    double n, kmean, kstd, kvar, ksum, k2sum;
};

// statistics of the exact template, float64, in the reference's order of operations
// (detection.py:1002-1003: kernel.mean(), kernel.std(); :1023-1026: sums and means)
HostStats template_stats(const double* k, int kk)
{
    HostStats s;
    s.n = (double)kk;
    double sum = 0, sum2 = 0;
    for (int t = 0; t < kk; ++t) {
        sum += k[t];
        sum2 += k[t] * k[t];
    }
    s.ksum = sum;
    s.k2sum = sum2;
    s.kmean = sum / kk;
    s.kvar = sum2 / kk - s.kmean * s.kmean;
    double dev = 0;
    for (int t = 0; t < kk; ++t) dev += (k[t] - s.kmean) * (k[t] - s.kmean);
    s.kstd = std::sqrt(dev / kk);
    return s;
}

template <typename TC>
int upload_weights(cs_ctx* ctx, hipStream_t stream, const std::vector<double>& w64)
{
    constexpr int slot = sizeof(TC) == 8 ? 1 : 0;
    std::vector<TC> w(w64.size());
    for (size_t t = 0; t < w64.size(); ++t) w[t] = (TC)w64[t];
    const size_t bytes = w.size() * sizeof(TC);
    if (ctx->w_cached[slot].size() == bytes && std::memcmp(ctx->w_cached[slot].data(), w.data(), bytes) == 0) return CS_OK;
    // a set used before?  swap it back in; otherwise the current set is parked in the slot used longest ago and that
    // slot's buffer takes the upload (work queued by earlier calls has drained: every entry point ends synchronised)
    {
        auto swap_in = [&](cs_ctx::ParkedWeights& pk) {
            std::swap(ctx->d_w[slot], pk.d);
            std::swap(ctx->d_w_bytes[slot], pk.bytes);
            ctx->w_cached[slot].swap(pk.host);
            pk.stamp = ++ctx->w_clock;
        };
        cs_ctx::ParkedWeights* oldest = &ctx->w_parked[slot][0];
        for (auto& pk : ctx->w_parked[slot]) {
            if (pk.d && pk.host.size() == bytes && std::memcmp(pk.host.data(), w.data(), bytes) == 0) {
                swap_in(pk);
                return CS_OK;
            }
            if (pk.stamp < oldest->stamp) oldest = &pk;
        }
        if (ctx->d_w[slot]) swap_in(*oldest);
    }
    // + 64 bytes of slack: the fast kernels' scalar row loads over-read (cs_corr_stream.h WRow)
    if (bytes + 64 > ctx->d_w_bytes[slot]) {
        if (ctx->d_w[slot]) {
            CS_HIP(ctx, hipDeviceSynchronize());   // a queued kernel may still read the old buffer
            CS_HIP(ctx, hipFree(ctx->d_w[slot]));
        }
        ctx->d_w[slot] = nullptr;
        ctx->d_w_bytes[slot] = 0;
        CS_HIP(ctx, hipMalloc(&ctx->d_w[slot], bytes + 64));
        CS_HIP(ctx, hipMemsetAsync(ctx->d_w[slot], 0, bytes + 64, stream));   // same stream as the upload below
        ctx->d_w_bytes[slot] = bytes + 64;
        ctx->w_cached[slot].clear();
    }
    if (ctx->w_cached[slot].size() == bytes && std::memcmp(ctx->w_cached[slot].data(), w.data(), bytes) == 0)
        return CS_OK;
    CS_HIP(ctx, hipMemcpyAsync(ctx->d_w[slot], w.data(), bytes, hipMemcpyHostToDevice, stream));
    ++ctx->uploads;
    // `w` is pageable and dies here: on a non-default stream the runtime may still be reading it after
    // the call returns, so wait (uploads only happen when the template changes)
    CS_HIP(ctx, hipStreamSynchronize(stream));
    ctx->w_cached[slot].assign((unsigned char*)w.data(), (unsigned char*)w.data() + bytes);
    return CS_OK;
}

// ---- float16 head / tail fragments of the weight sets for the matrix-core kernel -----------------
uint16_t f32_to_f16_bits(float f)
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t be = (x >> 23) & 0xffu;
    uint32_t m = x & 0x7fffffu;
    if (be == 0xffu) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0u));
    const int e = (int)be - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int shift = 14 - e;
        uint32_t half = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;     // may carry into the exponent: still right
    return (uint16_t)(sign | half);
}

float f16_bits_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const int e = (h >> 10) & 0x1f;
    const uint32_t m = h & 0x3ffu;
    float out;
    if (e == 0) {
        out = std::ldexp((float)m, -24);
    } else if (e == 31) {
        uint32_t x = 0x7f800000u | (m << 13);
        std::memcpy(&out, &x, 4);
    } else {
        out = std::ldexp((float)(m | 0x400u), e - 25);
    }
    uint32_t x;
    std::memcpy(&x, &out, 4);
    x |= sign;
    std::memcpy(&out, &x, 4);
    return out;
}

// Build (or reuse) the fragment image of the float32 weights currently in ctx->d_w[0] (layout in
// cs_launch.h MfmaWeights).  Each set is scaled by the power of two that puts its largest magnitude in
// [64, 128), so that heads and tails stay in float16's normal range.
int ensure_wfrag(cs_ctx* ctx, hipStream_t stream, int km, int kn, cs::MfmaWeights* E)
{
    const std::vector<unsigned char>& key = ctx->w_cached[0];
    const int kk = km * kn;
    const size_t n_floats = key.size() / 4;
    if (kk <= 0 || n_floats < (size_t)kk) return fail(ctx, CS_ERR_INVALID, "weights missing for the matrix-core kernel");
    if (km > 17 || kn > 17) return fail(ctx, CS_ERR_INVALID, "the matrix-core weight image holds templates of up to 17 x 17");
    const int nsets = (int)std::min<size_t>(3, n_floats / kk);
    constexpr size_t kImage = 3 * 17 * 2 * 1024;
    if (!ctx->d_wfrag) CS_HIP(ctx, hipMalloc(&ctx->d_wfrag, kImage));
    if (!(ctx->wfrag_km == km && ctx->wfrag_kn == kn && ctx->wfrag_key == key)) {
        const float* w = reinterpret_cast<const float*>(key.data());
        std::vector<uint16_t> img(kImage / 2, 0);
        for (int set = 0; set < nsets; ++set) {
            float amax = 0.0f;
            for (int t = 0; t < kk; ++t) amax = std::max(amax, std::fabs(w[set * kk + t]));
            int ew = 0;
            if (amax > 0.0f && std::isfinite(amax)) {
                int e2;
                (void)std::frexp(amax, &e2);          // amax = f * 2^e2, f in [0.5, 1)
                ew = 7 - e2;                           // amax * 2^ew in [64, 128)
            }
            ew = std::max(-100, std::min(100, ew));
            ctx->wfrag_unscale[set] = std::ldexp(1.0f, -ew);
            for (int s = 0; s < km; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int t = 8 * (lane >> 4) + e - (lane & 15);
                        if (t < 0 || t >= kn) continue;
                        const float v = std::ldexp(w[set * kk + s * kn + t], ew);
                        const uint16_t hb = f32_to_f16_bits(v);
                        const uint16_t lb = f32_to_f16_bits(v - f16_bits_to_f32(hb));
                        const size_t base = ((size_t)set * km + s) * 2 * 64 * 8;
                        img[base + (size_t)lane * 8 + e] = hb;
                        img[base + 64 * 8 + (size_t)lane * 8 + e] = lb;
                    }
        }
        CS_HIP(ctx, hipMemcpyAsync(ctx->d_wfrag, img.data(), kImage, hipMemcpyHostToDevice, stream));
        ++ctx->uploads;
        // rim tables of the two mask weight sets (cs_launch.h MfmaWeights::rim), square templates only
        std::vector<float> rim(cs::kRimFloats, 0.0f);
        if (nsets == 3 && km == kn) {
            const int K = km;
            for (int set = 0; set < 2; ++set) {
                const float* ws = w + (1 + set) * kk;
                for (int r = 0; r < K; ++r) {
                    double sp = 0.0, sq = 0.0;
                    for (int m = 0; m <= K; ++m) {
                        rim[cs::kRimPW + (set * 17 + r) * 18 + m] = (float)sp;
                        rim[cs::kRimQW + (set * 17 + r) * 18 + m] = (float)sq;
                        if (m < K) {
                            sp += ws[r * K + m];
                            sq += ws[m * K + r];
                        }
                    }
                    for (int c = 0; c < K; ++c) rim[cs::kRimW + set * 289 + r * 17 + c] = ws[r * K + c];
                }
            }
            for (int D = 0; D < K - 1; ++D) {          // the whole lower triangle of diagonal D: row ki has its first ki - D pixels in it
                double n = 0.0, a = 0.0, b = 0.0;
                for (int ki = 0; ki < K; ++ki) {
                    const int L = std::min(K, std::max(0, ki - D));
                    n += L;
                    for (int kj = 0; kj < L; ++kj) {
                        a += w[kk + ki * K + kj];
                        b += w[2 * kk + ki * K + kj];
                    }
                }
                rim[cs::kRimBase + 0 * 17 + D] = (float)n;
                rim[cs::kRimBase + 1 * 17 + D] = (float)a;
                rim[cs::kRimBase + 2 * 17 + D] = (float)b;
            }
        }
        if (!ctx->d_rim) CS_HIP(ctx, hipMalloc(&ctx->d_rim, sizeof(float) * cs::kRimFloats));
        CS_HIP(ctx, hipMemcpyAsync(ctx->d_rim, rim.data(), sizeof(float) * cs::kRimFloats, hipMemcpyHostToDevice, stream));
        ++ctx->uploads;
        CS_HIP(ctx, hipStreamSynchronize(stream));    // pageable sources die here
        ctx->wfrag_key = key;
        ctx->wfrag_km = km;
        ctx->wfrag_kn = kn;
    }
    E->frag = reinterpret_cast<const uint4*>(ctx->d_wfrag);
    E->rim = reinterpret_cast<const float*>(ctx->d_rim);
    for (int set = 0; set < 3; ++set) E->unscale[set] = ctx->wfrag_unscale[set];
    return CS_OK;
}

// The image of the two-pass kernel (cs_corr_wide.hip; layout in cs_launch.h MfmaWideWeights): three sets x km rows x
// two passes x {head, tail} x 1 KiB, then the 4 x 33 row / column sums of the two mask weight sets (float64 sums of the
// float32 weights the device holds, rounded once).
int ensure_wfrag_wide(cs_ctx* ctx, hipStream_t stream, int km, int kn, cs::MfmaWideWeights* E)
{
    const std::vector<unsigned char>& key = ctx->w_cached[0];
    const int kk = km * kn;
    const size_t n_floats = key.size() / 4;
    if (kk <= 0 || n_floats < (size_t)kk) return fail(ctx, CS_ERR_INVALID, "weights missing for the matrix-core kernel");
    if (!cs::corr_mfma_wide_fits(km, kn)) return fail(ctx, CS_ERR_INVALID, "the two-pass matrix-core kernel holds templates of up to 33 x 33");
    const int nsets = (int)std::min<size_t>(3, n_floats / kk);
    const size_t frag_halfs = (size_t)3 * km * 2 * 2 * 512;
    const size_t sums_off = frag_halfs * 2;                    // bytes (a multiple of 16)
    const size_t bytes = sums_off + 4 * 33 * sizeof(float);
    if (bytes > ctx->d_wfrag_wide_bytes) {
        if (ctx->d_wfrag_wide) {
            CS_HIP(ctx, hipDeviceSynchronize());   // a queued kernel may still read the old image
            CS_HIP(ctx, hipFree(ctx->d_wfrag_wide));
        }
        ctx->d_wfrag_wide = nullptr;
        ctx->d_wfrag_wide_bytes = 0;
        ctx->wfrag_wide_key.clear();
        CS_HIP(ctx, hipMalloc(&ctx->d_wfrag_wide, bytes));
        ctx->d_wfrag_wide_bytes = bytes;
    }
    if (!(ctx->wfrag_wide_km == km && ctx->wfrag_wide_kn == kn && ctx->wfrag_wide_key == key)) {
        const float* w = reinterpret_cast<const float*>(key.data());
        std::vector<unsigned char> img(bytes, 0);
        uint16_t* halfs = reinterpret_cast<uint16_t*>(img.data());
        float* sums = reinterpret_cast<float*>(img.data() + sums_off);
        for (int set = 0; set < nsets; ++set) {
            float amax = 0.0f;
            for (int t = 0; t < kk; ++t) amax = std::max(amax, std::fabs(w[set * kk + t]));
            int ew = 0;
            if (amax > 0.0f && std::isfinite(amax)) {
                int e2;
                (void)std::frexp(amax, &e2);          // amax = f * 2^e2, f in [0.5, 1)
                ew = 7 - e2;                           // amax * 2^ew in [64, 128)
            }
            ew = std::max(-100, std::min(100, ew));
            ctx->wfrag_wide_unscale[set] = std::ldexp(1.0f, -ew);
            for (int s = 0; s < km; ++s)
                for (int pass = 0; pass < 2; ++pass)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int k = 8 * (lane >> 4) + e;
                            const size_t base = ((((size_t)set * km + s) * 2 + pass) * 2) * 512;
                            const int t = 32 * pass + k - (lane & 15);
                            if (t >= 0 && t < kn) {
                                const float v = std::ldexp(w[set * kk + s * kn + t], ew);
                                const uint16_t hb = f32_to_f16_bits(v);
                                const uint16_t lb = f32_to_f16_bits(v - f16_bits_to_f32(hb));
                                halfs[base + (size_t)lane * 8 + e] = hb;
                                halfs[base + 512 + (size_t)lane * 8 + e] = lb;
                            }
                            // the second pass of the template itself (set 0): k = 16 .. 31 meet no weight (t >= 33), and the kernel
                            // sends the signal's TAILS of the columns of k - 16 through them: a second copy of the heads
                            const int t2 = 32 + (k - 16) - (lane & 15);
                            if (set == 0 && pass == 1 && k >= 16 && t2 >= 0 && t2 < kn)
                                halfs[base + (size_t)lane * 8 + e] = f32_to_f16_bits(std::ldexp(w[s * kn + t2], ew));
                        }
        }
        if (nsets == 3)
            for (int set = 0; set < 2; ++set) {
                const float* ws = w + (1 + set) * kk;
                for (int s = 0; s < km; ++s) {
                    double acc = 0.0;
                    for (int t = 0; t < kn; ++t) acc += ws[s * kn + t];
                    sums[set * 33 + s] = (float)acc;
                }
                for (int t = 0; t < kn; ++t) {
                    double acc = 0.0;
                    for (int s = 0; s < km; ++s) acc += ws[s * kn + t];
                    sums[(2 + set) * 33 + t] = (float)acc;
                }
            }
        CS_HIP(ctx, hipMemcpyAsync(ctx->d_wfrag_wide, img.data(), bytes, hipMemcpyHostToDevice, stream));
        ++ctx->uploads;
        CS_HIP(ctx, hipStreamSynchronize(stream));    // the pageable source dies here
        ctx->wfrag_wide_key = key;
        ctx->wfrag_wide_km = km;
        ctx->wfrag_wide_kn = kn;
    }
    E->frag = reinterpret_cast<const uint4*>(ctx->d_wfrag_wide);
    E->sums = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ctx->d_wfrag_wide) + sums_off);
    for (int set = 0; set < 3; ++set) E->unscale[set] = ctx->wfrag_wide_unscale[set];
    E->plane_only = std::getenv("CHROMOSIGHT_HIP_WIDE_PLANE") ? 1 : 0;
    E->plane_only_staging = std::getenv("CHROMOSIGHT_HIP_WIDE_SLOW") ? 1 : 0;
    E->one_launch = std::getenv("CHROMOSIGHT_HIP_WIDE_ONE_LAUNCH") ? 1 : std::getenv("CHROMOSIGHT_HIP_WIDE_TWO_LAUNCHES") ? 2 : 0;
    E->tile_mode = 0;
    return CS_OK;
}

bool fast_available(int km, int kn, int* K);

// Which float32 calls take the two-pass matrix-core kernel (cs_corr_wide.hip): templates with a side of 18 .. 33 -- what
// `--win-size` makes (cli/chromosight.py:365-370) and the 19 x 19 .. 33 x 33 templates of API users -- in every container
// the runtime-size kernel served (bands and dense maps, float32 and float64, any mask, n_obs, plain cross-correlations).
// CHROMOSIGHT_HIP_NO_MFMA=1 / CHROMOSIGHT_HIP_NO_WIDE=1: never (the runtime-size kernel: the in-library cross-check).
bool mfma_wide_wanted(const cs::CorrArgs<float>& A)
{
    if (!cs::corr_mfma_wide_fits(A.km, A.kn) || (A.km <= 17 && A.kn <= 17)) return false;
    if (A.sig.counts || A.sig.layout == CS_LAYOUT_BAND_LAZY || !A.out.ptr) return false;
    if (std::getenv("CHROMOSIGHT_HIP_NO_MFMA") || std::getenv("CHROMOSIGHT_HIP_NO_WIDE")) return false;
    return true;
}

// Which float32 calls go to the matrix cores (cs_corr_mfma.hip).  Default: unmasked dense float32 maps
// (cs_normxcorr2 without a mask: the API / benchmark configuration) whenever the template is large
// (>= 13 x 13 entries: every template costs a full 17-row pass there, so small ones are cheaper on the
// packed-FMA kernel -- measured 4096^2: 17x17 0.115 vs 0.185 ms (no symmetry), 13x13 0.124 vs 0.127,
// 11x11 0.129 vs 0.108), the map is small (the strips of the streaming kernel under-fill the chip:
// 2048^2 11x11 0.035 vs 0.042 ms, 1024^2 9x9 0.014 vs 0.019) or the template has no streaming kernel
// (rectangular sizes).  CHROMOSIGHT_HIP_MFMA=1: every call with a template of up to 17 x 17
// (masked / banded maps run the general, slower, matrix-core kernel -- a test switch);
// CHROMOSIGHT_HIP_NO_MFMA=1: never.  Read per call so that tests can flip them.
// per-bin masks (detect / quantify configuration): opt-in for now (CHROMOSIGHT_HIP_MFMA_REG=1)
bool mfma_reg_wanted(const cs::CorrArgs<float>& A)
{
    if (A.mask_mode != CS_MASK_BINS || !A.full || A.km != A.kn || A.km > 17 || A.km < 3 || !(A.km & 1)) return false;
    if (A.xcorr_only) return false;            // (float64 containers are narrowed row by row first, see launch_corr)
    if ((A.sig.layout == CS_LAYOUT_BAND ? A.sig.band_w : A.ns) < 4) return false;      // 16-byte staging pieces
    if (std::getenv("CHROMOSIGHT_HIP_NO_MFMA")) return false;
    // default: templates of 15 x 15 and 17 x 17 -- measured on the 234- and 1001-diagonal bands (tools/time_templates.py,
    // profiles/*_template_kernels.txt): 17 x 17 mirrored rows 2.01 vs 2.56 ms, 17 x 17 general 2.27 vs 3.01, 15 x 15 2.24 vs
    // 2.72; at 13 x 13 it wins only on the wide band (2.29 vs 2.51, 0.251 vs 0.235 on the narrow one), below that the
    // streaming kernel does (the tile kernel always walks 17 template rows).  CHROMOSIGHT_HIP_MFMA_REG=1 sends every
    // compatible call here, =0 none
    const char* e = std::getenv("CHROMOSIGHT_HIP_MFMA_REG");
    if (e && e[0] == '1') return true;
    if (e && e[0] == '0') return false;
    const char* general = std::getenv("CHROMOSIGHT_HIP_MFMA");
    if (general && general[0] == '1') return false;      // the general matrix-core kernel was asked for by name
    return A.km >= 15;
}

bool mfma_wanted(const cs::CorrArgs<float>& A)
{
    if (A.km < 1 || A.kn < 1 || A.km > 17 || A.kn > 17) return false;
    if (std::getenv("CHROMOSIGHT_HIP_NO_MFMA")) return false;
    const char* e = std::getenv("CHROMOSIGHT_HIP_MFMA");
    if (e && e[0] == '1') return true;
    const bool dense_f32 = A.mask_mode == 0 && A.sig.layout == 0 && A.out.layout == 0 && !A.nobs.ptr;
    int K = 0;
    const long long px = (long long)(A.row_end - A.row_begin) * A.ns;
    return dense_f32 && (A.km * A.kn >= 169 || px <= 6000000 || !fast_available(A.km, A.kn, &K));
}

bool fast_available(int km, int kn, int* K)
{
    (void)km;
    (void)kn;
    (void)K;
#ifdef CS_HAVE_FAST
    if (km != kn) return false;
    switch (km) {
        case 7: case 9: case 11: case 13: case 15: case 17:
            *K = km;
            return true;
        default:
            return false;
    }
#else
    return false;
#endif
}

// aligned_x: the generic kernel starts the x tiles of a row block at a multiple of the tile width,
// the streaming kernel at the block's first in-band column
template <typename TC>
void fill_grid(cs::CorrArgs<TC>& A, int tw, int th, bool aligned_x = true)
{
    A.tile_w = tw;
    A.tile_h = th;
    A.tiles_y = (A.row_end - A.row_begin + th - 1) / th;
    if (A.out.layout == CS_LAYOUT_BAND) {
        A.out_lo = A.out.band_lo;
        A.out_hi = A.out.band_lo + A.out.band_w - 1;
        const long long span = (long long)(A.out_hi - A.out_lo) + th + tw - 1;
        A.tiles_x = aligned_x ? (int)(span / tw) + 2 : (int)(span / tw);
        const int max_x = (A.ns + tw - 1) / tw;
        if (A.tiles_x > max_x) A.tiles_x = max_x;
    } else {
        A.out_lo = -(1 << 30);
        A.out_hi = (1 << 30);
        A.tiles_x = (A.ns + tw - 1) / tw;
    }
}

// Decide whether the factorised per-bin mask path applies and, if so, build its tables in the
// context's scratch buffer (cs_mask_prep.hip).  K = template size served by a streaming kernel.
template <typename TC>
int prepare_regular_mask(cs_ctx* ctx, cs::CorrArgs<TC>& A, int K, hipStream_t stream, bool rim_in_kernel = false)
{
    A.reg_mode = 0;
    A.fix_on = 0;
    A.rim_in_kernel = 0;
    if (std::getenv("CHROMOSIGHT_HIP_DEBUG"))
        fprintf(stderr, "[chromosight_hip] mask_mode=%d full=%d sym_upper=%d ms=%d ns=%d out_layout=%d out_lo=%d out_hi=%d max_dist=%d\n",
                A.mask_mode, A.full, A.sym_upper, A.ms, A.ns, A.out.layout, A.out_lo, A.out_hi, A.max_dist);
    if (A.mask_mode != CS_MASK_BINS || !A.full || std::getenv("CHROMOSIGHT_HIP_NO_REGULAR_MASK")) return CS_OK;
    if (A.ms < 2 * K || A.ns < 2 * K) return CS_OK;
    const int KH = (K - 1) / 2;
    const bool band_out = A.out.layout == CS_LAYOUT_BAND;
    int hi_d0 = 0, hi_w = 0, bot0, width, side = 0;
    bool edge_tables = false;
    if (A.sym_upper && !band_out) {
        // small dense maps (API users, short chromosomes): one correction per pixel, all from the
        // general predicate
        if ((long long)A.ms * A.ns > (1 << 22)) return CS_OK;
        bot0 = KH;
        width = A.ns;
    } else if (A.sym_upper) {
        // band outputs that end near max_dist: the edge tables cover the diagonals whose windows
        // leave 0..max_dist
        if (A.max_dist < 0 || A.out_lo < 0) return CS_OK;
        edge_tables = true;
        hi_d0 = A.max_dist - K + 2;
        hi_w = A.out_hi - hi_d0 + 1;
        if (hi_w < 0) hi_w = 0;
        if (hi_w > 64) return CS_OK;
        // rows whose in-band pixels reach the right or the bottom frame
        bot0 = std::max(KH, std::min(A.ms - KH, A.ns - KH - A.out_hi));
        width = A.out.band_w;
    } else {
        if (band_out) return CS_OK;
        bot0 = A.ms - KH;
        width = A.ns;
        side = KH;
    }
    const int top = KH;
    auto align = [](size_t n) { return (n + 63) & ~(size_t)63; };   // elements
    const size_t n_row = align((size_t)A.ms * 4 + 64), n_col = align(3 * (size_t)A.ns);
    // (rim_in_kernel: the masked matrix-core tile kernel forms the corrections of the edge diagonals itself)
    // ... when the two edge ranges hold at most 16 diagonals each and lie more than the 79 diagonals of a wave's
    // 16 rows x 64 columns apart (its epilogue handles one range per wave); narrow bands keep the records
    const bool edge_records = edge_tables && !(rim_in_kernel && K - 1 <= 16 && hi_w <= 16 && hi_d0 >= 96);
    const size_t n_lo = edge_records ? align((size_t)A.ms * (K - 1) * 4) : 0;
    const size_t n_hi = edge_records ? align((size_t)A.ms * hi_w * 4 + 4) : 0;
    const size_t n_frows = align((size_t)(top + A.ms - bot0) * width * 4);
    const size_t n_fcols = side ? align((size_t)A.ms * 2 * side * 4) : 0;
    const size_t bytes = (n_row + n_col + n_lo + n_hi + n_frows + n_fcols) * sizeof(TC);
    if (bytes > ctx->d_ws_bytes) {
        if (ctx->d_ws) CS_HIP(ctx, hipFree(ctx->d_ws));
        ctx->d_ws = nullptr;
        ctx->d_ws_bytes = 0;
        CS_HIP(ctx, hipMalloc(&ctx->d_ws, bytes));
        ctx->d_ws_bytes = bytes;
    }
    TC* rowtab = reinterpret_cast<TC*>(ctx->d_ws);
    TC* coltab = rowtab + n_row;
    TC* fix_lo = coltab + n_col;
    TC* fix_hi = fix_lo + n_lo;
    TC* fix_rows = fix_hi + n_hi;
    TC* fix_cols = side ? fix_rows + n_frows : nullptr;
    cs::MaskPrepArgs<TC> P;
    std::memset(&P, 0, sizeof(P));
    P.rr = A.miss_row;
    P.cc = A.miss_col;
    P.ms = A.ms;
    P.ns = A.ns;
    P.K = K;
    P.sym_upper = A.sym_upper;
    P.max_dist = A.max_dist;
    P.w = A.w;
    P.rowtab = rowtab;
    P.coltab = coltab;
    P.edge = edge_tables ? 1 : 0;
    P.skip_edge = edge_records ? 0 : 1;
    P.hi_d0 = hi_d0;
    P.hi_w = hi_w;
    P.fix_lo = fix_lo;
    P.fix_hi = fix_hi;
    P.top = top;
    P.bot0 = bot0;
    P.width = width;
    P.x_band = band_out ? 1 : 0;
    P.x_lo = A.out_lo;
    P.side = side;
    P.fix_rows = fix_rows;
    P.fix_cols = fix_cols;
    bool collected = false;
    if constexpr (std::is_same<TC, float>::value) {
        if (ctx->prep_collect) {             // (cs_detect_foci_blocks, one tile launch: the tables of all blocks from ONE launch)
            const int n_wg = cs::mask_prep_blocks<float>(P);
            if (n_wg < 0) return fail(ctx, CS_ERR_HIP, "mask tables: template too large");
            ctx->prep_collect->push_back(P);
            ctx->prep_groups.push_back(n_wg);
            collected = true;
        }
    }
    if (!collected && !ctx->skip_prep_launch) {       // (skip: after the call's own prepare form the tables are in place, only A is wanted)
        int rc = cs::launch_mask_prep<TC>(P, stream);
        if (rc != 0) return fail(ctx, CS_ERR_HIP, "mask table kernel failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (edge_tables) A.fix_on = 1;
    A.rim_in_kernel = (edge_tables && !edge_records) ? 1 : 0;
    A.reg_mode = 1;
    A.rowtab = rowtab;
    A.coltab = coltab;
    A.fix_lo = fix_lo;
    A.fix_hi = fix_hi;
    A.fix_hi_w = hi_w;
    A.fix_hi_d0 = hi_d0;
    A.fix_rows = fix_rows;
    A.fix_cols = fix_cols;
    A.fix_top = top;
    A.fix_bot0 = bot0;
    A.fix_width = width;
    A.fix_xband = band_out ? 1 : 0;
    A.fix_xlo = band_out ? A.out_lo : 0;
    A.fix_side = side;
    return CS_OK;
}

template <typename TC>
int launch_corr(cs_ctx* ctx, cs::CorrArgs<TC>& A, hipStream_t stream, bool allow_fast);

// internal status of launch_corr<float>: a candidate sink was given without a map, and the kernel that would serve the
// call writes maps (nothing was launched that matters: the caller allocates the map and calls again)
constexpr int CS_NEED_MAP = 1000;

// the streaming kernels write n_obs next to the coefficient (same index), so both maps must share
// one geometry; plain cross-correlations run their unmasked instance
template <typename TC>
bool fast_compatible(const cs::CorrArgs<TC>& A)
{
    if (A.nobs.ptr && (A.nobs.layout != A.out.layout || A.nobs.ld != A.out.ld || A.nobs.band_lo != A.out.band_lo ||
                       A.nobs.band_w != A.out.band_w || A.nobs.row0 != A.out.row0))
        return false;
    if (A.xcorr_only && (A.mask_mode != 0 || A.full || A.sym_upper)) return false;
    return true;
}

template <>
int launch_corr<float>(cs_ctx* ctx, cs::CorrArgs<float>& A, hipStream_t stream, bool allow_fast)
{
    int K = 0, tw, th, rc;
    A.n_cu = ctx->n_cu;
    A.grid_cap = ctx->grid_cap;
    A.reg_mode = 0;
    // per-bin masks on the matrix cores: the factorised mask tables + the persistent tile kernel
    // (candidate mode without a sink -- the map fallback of find_candidates -- runs on the kernels that decide the screen
    // at run time: the tile kernel's candidate instance has no map output)
    if (allow_fast && mfma_reg_wanted(A) && !(A.ks.cand_cmin > 0.0f && !A.cand_keys)) {
        if (A.out.layout == CS_LAYOUT_BAND) {
            A.out_lo = A.out.band_lo;
            A.out_hi = A.out.band_lo + A.out.band_w - 1;
        } else {
            A.out_lo = -(1 << 30);
            A.out_hi = (1 << 30);
        }
        if (A.cand_keys) {                   // candidate sink: only the scanned diagonals (the map path trims in the compaction)
            A.out_lo = std::max(A.out_lo, A.cand_dlo);
            A.out_hi = std::min(A.out_hi, A.cand_dhi);
        }
        // (the mirrored-row instance of the tile kernel forms the corrections of the edge diagonals itself: no records)
        const bool rim_in_kernel = A.w_sym && A.km == 17 && A.kn == 17 && !std::getenv("CHROMOSIGHT_HIP_MFMA_NORSYM");
        rc = prepare_regular_mask<float>(ctx, A, A.km, stream, rim_in_kernel);
        if (rc != CS_OK) return rc;
        if (A.reg_mode == 1 && A.sig_is_f64) {
            // float64 container (the pipeline keeps the detrended band in float64 for the exact re-scoring of the
            // candidates), float32 arithmetic: the rows the windows reach are rounded into context scratch -- what
            // the streaming kernel does pixel by pixel -- 12 B of traffic per pixel for a kernel 1.3x faster
            const int kh = (A.km - 1) / 2;
            const int p_lo = std::max(0, A.row_begin - kh), p_hi = std::min(A.ms, A.row_end + (A.km - 1) - kh);
            const int width = A.sig.layout == CS_LAYOUT_BAND ? A.sig.band_w : A.ns;
            const long long ld = ((long long)width + 15) / 16 * 16;
            rc = ensure_scratch(ctx, &ctx->d_narrow, &ctx->d_narrow_bytes, (size_t)(p_hi - p_lo) * (size_t)ld * 4);
            if (rc != CS_OK) return rc;
            const double* src = reinterpret_cast<const double*>(A.sig.ptr) + ((long long)p_lo - A.sig.row0) * A.sig.ld;
            rc = cs::launch_narrow_rows(src, A.sig.ld, reinterpret_cast<float*>(ctx->d_narrow), ld, p_hi - p_lo, width, ctx->n_cu, stream);
            if (rc != 0) return fail(ctx, CS_ERR_HIP, "narrowing kernel failed: %s", hipGetErrorString((hipError_t)rc));
            A.sig.ptr = ctx->d_narrow;
            A.sig.ld = ld;
            A.sig.row0 = p_lo;
            A.sig_is_f64 = 0;
        }
        if (A.reg_mode == 1) {
            cs::MfmaWeights E;
            rc = ensure_wfrag(ctx, stream, A.km, A.kn, &E);
            if (rc != CS_OK) return rc;
            int path = 0;
            rc = cs::launch_corr_mfma_f32(A, E, stream, &path);
            ctx->last_kernel = path == 2 ? CS_KERNEL_MFMA_REG : CS_KERNEL_MFMA;
            if (rc == -5) return CS_NEED_MAP;
            if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
            if (path == 2 && A.cand_keys) ctx->cand_fused = true;
            return CS_OK;
        }
    }
    if (A.sig.counts) return fail(ctx, CS_ERR_UNSUPPORTED, "a band of counts (CS_LAYOUT_BAND_COUNTS) is read by the masked float32 tile kernel only: per-bin masks, full mode, odd square template of up to 17");
    if (!A.out.ptr) return CS_NEED_MAP;      // every other kernel writes a map
    if (allow_fast && mfma_wanted(A)) {
        if (A.sig_is_f64 && A.mask_mode == 0 && A.sig.layout == 0 && A.out.layout == 0 && !A.nobs.ptr) {
            // float64 container, float32 arithmetic: narrow the rows the windows reach into context scratch
            // (what the kernels do pixel by pixel anyway) so that the persistent tile kernel can stage them
            const int kh = (A.km - 1) / 2;
            const int p_lo = std::max(0, A.row_begin - kh), p_hi = std::min(A.ms, A.row_end + (A.km - 1) - kh);
            const long long ld = ((long long)A.ns + 15) / 16 * 16;
            const size_t bytes = (size_t)(p_hi - p_lo) * (size_t)ld * 4;
            rc = ensure_scratch(ctx, &ctx->d_narrow, &ctx->d_narrow_bytes, bytes);
            if (rc != CS_OK) return rc;
            const double* src = reinterpret_cast<const double*>(A.sig.ptr) + ((long long)p_lo - A.sig.row0) * A.sig.ld;
            rc = cs::launch_narrow_rows(src, A.sig.ld, reinterpret_cast<float*>(ctx->d_narrow), ld, p_hi - p_lo, A.ns, ctx->n_cu, stream);
            if (rc != 0) return fail(ctx, CS_ERR_HIP, "narrowing kernel failed: %s", hipGetErrorString((hipError_t)rc));
            A.sig.ptr = ctx->d_narrow;
            A.sig.ld = ld;
            A.sig.row0 = p_lo;
            A.sig_is_f64 = 0;
        }
        cs::MfmaWeights E;
        rc = ensure_wfrag(ctx, stream, A.km, A.kn, &E);
        if (rc != CS_OK) return rc;
        int dense_path = 0;
        rc = cs::launch_corr_mfma_f32(A, E, stream, &dense_path);
        ctx->last_kernel = dense_path == 1 ? CS_KERNEL_MFMA_DENSE : dense_path == 2 ? CS_KERNEL_MFMA_REG : CS_KERNEL_MFMA;
    } else if (allow_fast && fast_compatible(A) && fast_available(A.km, A.kn, &K)) {
        ctx->last_kernel = CS_KERNEL_STREAM;
#ifdef CS_HAVE_FAST
#define CS_CASE(KK)                          \
    case KK:                                 \
        cs::corr_fast_tile_k##KK(A.row_end - A.row_begin, A.ns, A.out.layout == 1 ? A.out.band_w : 0, ctx->n_cu, &tw, &th);  \
        fill_grid(A, tw, th, false);         \
        rc = prepare_regular_mask<float>(ctx, A, KK, stream); \
        if (rc != CS_OK) return rc;          \
        rc = cs::launch_corr_fast_f32_k##KK(A, stream); \
        break;
        switch (K) {
            CS_CASE(7) CS_CASE(9) CS_CASE(11) CS_CASE(13) CS_CASE(15) CS_CASE(17)
            default: rc = -1;
        }
#undef CS_CASE
#else
        rc = -1;
#endif
    } else if (allow_fast && A.w_rank1 && !A.xcorr_only && !std::getenv("CHROMOSIGHT_HIP_NO_SEPARABLE") &&
               cs::corr_sep_fits(A.km, A.kn, A.mask_mode != 0) &&
               !(mfma_wide_wanted(A) && A.out.layout == CS_LAYOUT_BAND && A.out.band_w >= 512 && !std::getenv("CHROMOSIGHT_HIP_SEPARABLE_FIRST"))) {
        // (an outer product that also fits the two-pass matrix-core kernel: that one on wide bands -- measured on the
        // 31 x 31 stripes, profiles/r06_template_kernels.txt: 1001 diagonals 5.40 vs 6.92 ms, 234 diagonals 0.577 vs 0.517)
        // templates without an unrolled instance that are an outer product (31 x 31 stripes): separable sums
        ctx->last_kernel = CS_KERNEL_SEPARABLE;
        cs::corr_sep_tile(&tw, &th);
        fill_grid(A, tw, th);
        rc = cs::launch_corr_sep_f32(A, stream);
    } else if (allow_fast && mfma_wide_wanted(A)) {
        cs::MfmaWideWeights E;
        rc = ensure_wfrag_wide(ctx, stream, A.km, A.kn, &E);
        if (rc != CS_OK) return rc;
        ctx->last_kernel = CS_KERNEL_MFMA_WIDE;
        rc = cs::launch_corr_mfma_wide_f32(A, E, stream);
    } else {
        ctx->last_kernel = CS_KERNEL_GENERIC;
        cs::corr_generic_tile(A.km, A.kn, &tw, &th);
        fill_grid(A, tw, th);
        rc = cs::launch_corr_generic_f32(A, stream);
    }
    if (rc == -3) return fail(ctx, CS_ERR_UNSUPPORTED, "template %dx%d needs more than 160 KiB of LDS", A.km, A.kn);
    if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

template <>
int launch_corr<double>(cs_ctx* ctx, cs::CorrArgs<double>& A, hipStream_t stream, bool allow_fast)
{
    int K = 0, tw, th, rc;
    if (A.sig.counts) return fail(ctx, CS_ERR_UNSUPPORTED, "a band of counts (CS_LAYOUT_BAND_COUNTS) is read by the masked float32 tile kernel only");
    A.n_cu = ctx->n_cu;
    A.grid_cap = ctx->grid_cap;
    ctx->last_kernel = CS_KERNEL_GENERIC;
    if (allow_fast && fast_compatible(A) && fast_available(A.km, A.kn, &K)) {
        ctx->last_kernel = CS_KERNEL_STREAM;
#ifdef CS_HAVE_FAST
#define CS_CASE(KK)                          \
    case KK:                                 \
        cs::corr_fast_tile_k##KK(A.row_end - A.row_begin, A.ns, A.out.layout == 1 ? A.out.band_w : 0, ctx->n_cu, &tw, &th);  \
        fill_grid(A, tw, th, false);         \
        rc = prepare_regular_mask<double>(ctx, A, KK, stream); \
        if (rc != CS_OK) return rc;          \
        rc = cs::launch_corr_fast_f64_k##KK(A, stream); \
        break;
        switch (K) {
            CS_CASE(7) CS_CASE(9) CS_CASE(11) CS_CASE(13) CS_CASE(15) CS_CASE(17)
            default: rc = -1;
        }
#undef CS_CASE
#else
        rc = -1;
#endif
    } else {
        cs::corr_generic_tile(A.km, A.kn, &tw, &th);
        fill_grid(A, tw, th);
        rc = cs::launch_corr_generic_f64(A, stream);
    }
    if (rc == -3) return fail(ctx, CS_ERR_UNSUPPORTED, "template %dx%d needs more than 160 KiB of LDS", A.km, A.kn);
    if (rc != 0) return fail(ctx, CS_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int check_matrix(cs_ctx* ctx, const cs_matrix* m, const char* what, int ns)
{
    if (!m || !m->d_ptr) return fail(ctx, CS_ERR_INVALID, "%s: null matrix", what);
    if (m->dtype != CS_F32 && m->dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "%s: bad dtype", what);
    if (m->layout == CS_LAYOUT_DENSE) {
        if (m->ld < ns) return fail(ctx, CS_ERR_INVALID, "%s: ld < number of columns", what);
    } else if (m->layout == CS_LAYOUT_BAND || m->layout == CS_LAYOUT_BAND_PADDED) {
        if (m->band_w <= 0 || m->ld < m->band_w) return fail(ctx, CS_ERR_INVALID, "%s: bad band geometry", what);
        if (m->layout == CS_LAYOUT_BAND_PADDED && m->ld < (int64_t)m->band_w + 4) return fail(ctx, CS_ERR_INVALID, "%s: a padded band keeps 4 zero slots per row", what);
    } else if ((m->layout == CS_LAYOUT_BAND_COUNTS || m->layout == CS_LAYOUT_BAND_COUNTS_VIEW) && ctx->allow_counts) {
        // a band of raw counts (cs_stage_block.band32_counts): only the masked tile kernel reads it (launch_corr<float>)
        if (m->band_w <= 0 || m->band_lo != 0 || m->row0 != 0 || m->dtype != CS_F32 || m->ld < (int64_t)m->band_w + (m->layout == CS_LAYOUT_BAND_COUNTS ? 4 : 0))
            return fail(ctx, CS_ERR_INVALID, "%s: bad band of counts", what);
    } else if (m->layout == CS_LAYOUT_BAND_LAZY && ctx->allow_lazy) {
        // a lazily evaluated float64 band (cs_stage_block): only the float64 kernels behind the batched foci entries read it
        if (m->band_w <= 0 || m->band_lo != 0 || m->row0 != 0 || m->dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "%s: bad lazy band", what);
    } else {
        return fail(ctx, CS_ERR_INVALID, "%s: bad layout", what);
    }
    return CS_OK;
}

// Build the device argument block shared by cs_normxcorr2 and cs_rescore_f64.
template <typename TC>
int build_args(cs_ctx* ctx, hipStream_t stream, const cs_matrix* signal, const cs_kernel* kernel,
               const cs_normxcorr2_params* p, cs::CorrArgs<TC>* out)
{
    if (!signal || !kernel || !p) return fail(ctx, CS_ERR_INVALID, "null argument");
    const int km = kernel->km, kn = kernel->kn, kk = km * kn;
    if (km <= 0 || kn <= 0 || !kernel->h_kernel) return fail(ctx, CS_ERR_INVALID, "bad template");
    if (!(km & 1) || !(kn & 1))
        return fail(ctx, CS_ERR_INVALID, "template dimensions must be odd (reference preprocessing.py:774-775)");
    if (p->ms <= 0 || p->ns <= 0) return fail(ctx, CS_ERR_INVALID, "empty signal");
    int rc = check_matrix(ctx, signal, "signal", p->ns);
    if (rc) return rc;
    if (p->mask_mode != CS_MASK_NONE) {
        // detection.py:880-881
        if (std::min(km, kn) >= std::max(p->ms, p->ns))
            return fail(ctx, CS_ERR_INVALID, "cannot have kernel bigger than signal");
        if (p->mask_mode == CS_MASK_BINS && (!p->d_miss_row || !p->d_miss_col))
            return fail(ctx, CS_ERR_INVALID, "missing-bin vectors are null");
        if (p->mask_mode == CS_MASK_EXPLICIT && !p->d_mask)
            return fail(ctx, CS_ERR_INVALID, "explicit mask is null");
    }
    const double* kconv = kernel->h_kernel_conv ? kernel->h_kernel_conv : kernel->h_kernel;
    const bool want_sym = !std::getenv("CHROMOSIGHT_HIP_NO_SYMMETRY");
    cs_ctx::TemplateCache& tc = ctx->tcache[sizeof(TC) == 8 ? 1 : 0];
    {
        std::vector<double> key;
        key.reserve(6 + 3 * (size_t)kk);
        key.push_back(km);
        key.push_back(kn);
        key.push_back(kernel->h_kernel_conv ? 1 : 0);
        key.push_back(kernel->h_kernel_sq ? 1 : 0);
        key.push_back(want_sym ? 1 : 0);
        key.push_back(p->xcorr_threshold);
        key.insert(key.end(), kernel->h_kernel, kernel->h_kernel + kk);
        if (kernel->h_kernel_conv) key.insert(key.end(), kernel->h_kernel_conv, kernel->h_kernel_conv + kk);
        if (kernel->h_kernel_sq) key.insert(key.end(), kernel->h_kernel_sq, kernel->h_kernel_sq + kk);
        const bool hit = key.size() == tc.key.size() && std::memcmp(key.data(), tc.key.data(), key.size() * sizeof(double)) == 0;
        if (!hit) {
            const HostStats st0 = template_stats(kernel->h_kernel, kk);
            if (!(st0.kstd > 0)) return fail(ctx, CS_ERR_INVALID, "Cannot have flat kernel.");  // detection.py:887-888
            tc.stats[0] = st0.n; tc.stats[1] = st0.kmean; tc.stats[2] = st0.kstd;
            tc.stats[3] = st0.kvar; tc.stats[4] = st0.ksum; tc.stats[5] = st0.k2sum;
            std::vector<double>& w0 = tc.w;
            w0.assign(3 * (size_t)kk, 0.0);
            for (int t = 0; t < kk; ++t) {
                const double kc = kconv[t];
                const double k2 = kernel->h_kernel_sq ? kernel->h_kernel_sq[t] : kc * kc;
                w0[t] = kc - st0.kmean;
                if (sizeof(TC) == 8) {
                    // float64 kernels use the reference's literal (un-centred) mask sums
                    w0[kk + t] = kc;
                    w0[2 * kk + t] = k2;
                } else {
                    w0[kk + t] = kc - st0.kmean;
                    // chosen so that kb + 2*kmean*ka + kmean^2*nm == sum over missing pixels of k2
                    w0[2 * kk + t] = k2 - 2.0 * st0.kmean * kc + st0.kmean * st0.kmean;
                }
            }
            // vertical symmetry of the template (loops, stripes): lets the streaming kernel fold template
            // rows.  Rounding-level asymmetries (truncated-SVD reconstructions) are symmetrised for the
            // float32 kernels, whose own rounding is 1e5 times larger; float64 needs exact symmetry.
            bool sym0 = want_sym;
            double wmax = 0;
            for (size_t t = 0; t < w0.size(); ++t) wmax = std::max(wmax, std::fabs(w0[t]));
            const double tol = sizeof(TC) == 8 ? 0.0 : 1e-12 * wmax;
            for (int set = 0; set < 3 && sym0; ++set)
                for (int r = 0; r < km / 2 && sym0; ++r)
                    for (int c = 0; c < kn; ++c)
                        if (std::fabs(w0[set * kk + r * kn + c] - w0[set * kk + (km - 1 - r) * kn + c]) > tol) {
                            sym0 = false;
                            break;
                        }
            if (sym0 && sizeof(TC) == 4)
                for (int set = 0; set < 3; ++set)
                    for (int r = 0; r < km / 2; ++r)
                        for (int c = 0; c < kn; ++c) w0[set * kk + (km - 1 - r) * kn + c] = w0[set * kk + r * kn + c];
            tc.sym = sym0;
            // Exactly rank-1 templates (the 31 x 31 stripes templates): u and v for the separable kernel
            // (cs_corr_sep.hip).  Pivot on the largest entry; the outer product must reproduce the template to
            // rounding (1e-12 relative) -- truncated-SVD templates carry their own squares and stay on the full kernels.
            tc.rank1 = false;
            if (sizeof(TC) == 4 && !kernel->h_kernel_sq) {
                int pa = 0, pb = 0;
                double kmax = 0;
                for (int a = 0; a < km; ++a)
                    for (int b = 0; b < kn; ++b)
                        if (std::fabs(kconv[a * kn + b]) > kmax) {
                            kmax = std::fabs(kconv[a * kn + b]);
                            pa = a;
                            pb = b;
                        }
                if (kmax > 0) {
                    std::vector<double> u(km), v(kn);
                    for (int b = 0; b < kn; ++b) v[b] = kconv[pa * kn + b];
                    for (int a = 0; a < km; ++a) u[a] = kconv[a * kn + pb] / kconv[pa * kn + pb];
                    double worst = 0;
                    for (int a = 0; a < km; ++a)
                        for (int b = 0; b < kn; ++b) worst = std::max(worst, std::fabs(u[a] * v[b] - kconv[a * kn + b]));
                    if (worst <= 1e-12 * kmax) {
                        tc.rank1 = true;
                        w0.insert(w0.end(), u.begin(), u.end());
                        w0.insert(w0.end(), v.begin(), v.end());
                    }
                }
            }
            // can sum_missing K' or sum_missing K'^2 of a non-empty set fall under the zeroing threshold?
            // not if every entry alone exceeds it and all have one sign (the built-in templates: >= 0.5)
            double lo = 1e300, lo2 = 1e300;
            bool pos = true, pos2 = true;
            std::vector<double> distinct;
            for (int t = 0; t < kk; ++t) {
                const double kc = kconv[t];
                const double k2 = kernel->h_kernel_sq ? kernel->h_kernel_sq[t] : kc * kc;
                lo = std::min(lo, std::fabs(kc));
                lo2 = std::min(lo2, std::fabs(k2));
                pos = pos && kc > 0;
                pos2 = pos2 && k2 > 0;
                bool seen = false;
                for (double d : distinct) seen = seen || std::fabs(d - kc) <= 1e-9 * std::max(1.0, std::fabs(d));
                if (!seen && distinct.size() < 64) distinct.push_back(kc);
            }
            tc.zk_possible = !(pos && pos2 && lo > 2 * p->xcorr_threshold && lo2 > 2 * p->xcorr_threshold);
            // piecewise-constant templates (borders, hairpins: 2 levels) have windows whose present pixels
            // are all equal; smooth templates never do, but with up to 75 % of a window missing a handful of
            // levels can still coincide, so only clearly many-valued templates skip the snap
            tc.snap_possible = distinct.size() < 64;
            tc.key.swap(key);
        }
    }
    HostStats st;
    st.n = tc.stats[0]; st.kmean = tc.stats[1]; st.kstd = tc.stats[2];
    st.kvar = tc.stats[3]; st.ksum = tc.stats[4]; st.k2sum = tc.stats[5];
    const std::vector<double>& w = tc.w;
    const bool sym = tc.sym;
    rc = upload_weights<TC>(ctx, stream, w);
    if (rc) return rc;

    cs::CorrArgs<TC> A;
    std::memset(&A, 0, sizeof(A));
    A.sig = view_of(signal);
    A.sig_is_f64 = signal->dtype == CS_F64;
    A.ms = p->ms;
    A.ns = p->ns;
    A.km = km;
    A.kn = kn;
    A.full = p->full ? 1 : 0;
    A.sym_upper = p->sym_upper ? 1 : 0;
    A.max_dist = p->max_dist;
    A.mask_mode = p->mask_mode;
    A.miss_row = p->d_miss_row;
    A.miss_col = p->d_miss_col;
    A.mask = A.sig;
    A.mask.ptr = (void*)p->d_mask;
    A.w = reinterpret_cast<const TC*>(ctx->d_w[sizeof(TC) == 8 ? 1 : 0]);
    A.ks.n = (TC)st.n;
    A.ks.inv_n = (TC)(1.0 / st.n);
    A.ks.kmean = (TC)st.kmean;
    A.ks.kstd = (TC)st.kstd;
    A.ks.kvar = (TC)st.kvar;
    A.ks.ksum = (TC)st.ksum;
    A.ks.k2sum = (TC)st.k2sum;
    A.ks.thr = (TC)p->xcorr_threshold;
    A.ks.eps = (TC)p->denom_eps;
    A.ks.cut = (TC)p->min_present;
    A.ks.thr_n = (TC)(p->xcorr_threshold * st.n);
    A.ks.nkvar = (TC)(st.n * st.kvar);
    A.ks.eps2 = (TC)(p->denom_eps * p->denom_eps);
    A.ks.den2_min = (TC)(p->denom_eps * p->denom_eps * st.n * st.n);
    A.ks.zk_possible = tc.zk_possible;
    A.ks.snap_possible = tc.snap_possible;
    A.ks.cand_cmin = (TC)0;              // candidate mode: corr_candidates_f32 below
    A.ks.cand_thr = (TC)0;
    A.xcorr_only = 0;
    A.w_sym = sym ? 1 : 0;
    A.w_rank1 = tc.rank1 ? 1 : 0;
    A.row_begin = 0;
    A.row_end = p->ms;
    if (p->row_end > p->row_begin) {
        if (p->row_begin < 0 || p->row_end > p->ms) return fail(ctx, CS_ERR_INVALID, "row window outside the matrix");
        if ((p->row_begin != 0 || p->row_end != p->ms) && p->mask_mode == CS_MASK_EXPLICIT)
            return fail(ctx, CS_ERR_UNSUPPORTED, "row windows need per-bin masks or none");
        A.row_begin = p->row_begin;
        A.row_end = p->row_end;
    }
    *out = A;
    return CS_OK;
}

}  // namespace

// ============================================================================================
extern "C" {

const char* cs_version(void) { return "chromosight_hip 0.1 (gfx950)"; }

int cs_last_kernel(const cs_ctx* ctx) { return ctx ? ctx->last_kernel : 0; }

int cs_ctx_set_range_check(cs_ctx* ctx, int32_t on)
{
    if (!ctx) return CS_ERR_INVALID;
    ctx->range_check = on ? 1 : 0;
    return CS_OK;
}

int cs_ctx_create(int device, cs_ctx** out)
{
    if (!out) return CS_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return CS_ERR_HIP;
    if (device < 0 || device >= count) return CS_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return CS_ERR_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return CS_ERR_HIP;
    cs_ctx* ctx = new cs_ctx();
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    if (hipMalloc((void**)&ctx->d_tiles_started, 256) != hipSuccess || hipMemset(ctx->d_tiles_started, 0, 256) != hipSuccess) {
        delete ctx;
        return CS_ERR_HIP;
    }
    *out = ctx;
    return CS_OK;
}

void cs_ctx_destroy(cs_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int slot = 0; slot < 2; ++slot) {
        if (ctx->d_w[slot]) (void)hipFree(ctx->d_w[slot]);
        for (auto& pk : ctx->w_parked[slot])
            if (pk.d) (void)hipFree(pk.d);
    }
    if (ctx->d_ws) (void)hipFree(ctx->d_ws);
    if (ctx->d_tiles_started) (void)hipFree(ctx->d_tiles_started);
    if (ctx->d_wfrag) (void)hipFree(ctx->d_wfrag);
    if (ctx->d_rim) (void)hipFree(ctx->d_rim);
    if (ctx->d_wfrag_wide) (void)hipFree(ctx->d_wfrag_wide);
    if (ctx->d_map) (void)hipFree(ctx->d_map);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    for (int k = 0; k < 2; ++k) {
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
        if (ctx->ev_stage[k]) (void)hipEventDestroy(ctx->ev_stage[k]);
    }
    for (int k = 0; k < kBlkLanes - 1; ++k) {
        if (ctx->ws_alt[k]) (void)hipFree(ctx->ws_alt[k]);
        if (ctx->s_blk[k]) (void)hipStreamDestroy(ctx->s_blk[k]);
    }
    for (int k = 0; k < kBlkLanes; ++k)
        if (ctx->ev_blk[k]) (void)hipEventDestroy(ctx->ev_blk[k]);
    for (void* w : ctx->ws_tab)
        if (w) (void)hipFree(w);
    if (ctx->h_tab) (void)hipHostFree(ctx->h_tab);
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    if (ctx->d_pool) (void)hipFree(ctx->d_pool);
    if (ctx->h_counts) (void)hipHostFree(ctx->h_counts);
    if (ctx->h_cand_counts) (void)hipHostFree(ctx->h_cand_counts);
    if (ctx->d_counts_peak) (void)hipFree(ctx->d_counts_peak);
    if (ctx->h_peak) (void)hipHostFree(ctx->h_peak);
    if (ctx->d_narrow) (void)hipFree(ctx->d_narrow);
    if (ctx->h_blk_counts) (void)hipHostFree(ctx->h_blk_counts);
    if (ctx->d_host_in) (void)hipFree(ctx->d_host_in);
    if (ctx->d_host_out) (void)hipFree(ctx->d_host_out);
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    for (hipStream_t st : {ctx->s_up, ctx->s_run, ctx->s_down})
        if (st) (void)hipStreamDestroy(st);
    for (auto* v : {&ctx->ev_up, &ctx->ev_run, &ctx->ev_down})
        for (hipEvent_t e : *v) (void)hipEventDestroy(e);
    delete ctx;
}

const char* cs_last_error(const cs_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int cs_device_cu_count(const cs_ctx* ctx) { return ctx ? ctx->n_cu : 0; }

int cs_malloc(cs_ctx* ctx, size_t bytes, void** d_ptr)
{
    if (!d_ptr) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipMalloc(d_ptr, bytes ? bytes : 1));
    return CS_OK;
}

int cs_free(cs_ctx* ctx, void* d_ptr)
{
    CS_ENTER(ctx);
    if (d_ptr) CS_HIP(ctx, hipFree(d_ptr));
    return CS_OK;
}

int cs_memcpy_h2d(cs_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, void* stream)
{
    CS_ENTER(ctx);
    if (bytes) {
        CS_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
        // the caller may reuse h_src as soon as this returns (pageable sources are not always staged
        // before an asynchronous copy returns)
        CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    }
    return CS_OK;
}

int cs_memcpy_d2h(cs_ctx* ctx, void* h_dst, const void* d_src, size_t bytes, void* stream)
{
    CS_ENTER(ctx);
    if (bytes) {
        CS_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    }
    return CS_OK;
}

int cs_memset(cs_ctx* ctx, void* d_dst, int value, size_t bytes, void* stream)
{
    CS_ENTER(ctx);
    if (bytes) CS_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, (hipStream_t)stream));
    return CS_OK;
}

int cs_stream_sync(cs_ctx* ctx, void* stream)
{
    CS_ENTER(ctx);
    CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CS_OK;
}

int cs_stream_create(cs_ctx* ctx, void** stream)
{
    if (!stream) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    hipStream_t s;
    CS_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return CS_OK;
}

int cs_stream_create_priority(cs_ctx* ctx, int32_t high, void** stream)
{
    if (!stream) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    int least = 0, greatest = 0;                     // numerically lower = served first
    CS_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s;
    CS_HIP(ctx, hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high ? greatest : least));
    *stream = (void*)s;
    return CS_OK;
}

int cs_stream_destroy(cs_ctx* ctx, void* stream)
{
    CS_ENTER(ctx);
    if (stream) CS_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    return CS_OK;
}

int cs_event_create(cs_ctx* ctx, void** event)
{
    if (!event) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    hipEvent_t e;
    CS_HIP(ctx, hipEventCreate(&e));
    *event = (void*)e;
    return CS_OK;
}

int cs_event_destroy(cs_ctx* ctx, void* event)
{
    CS_ENTER(ctx);
    if (event) CS_HIP(ctx, hipEventDestroy((hipEvent_t)event));
    return CS_OK;
}

int cs_event_record(cs_ctx* ctx, void* event, void* stream)
{
    if (!event) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return CS_OK;
}

int cs_stream_wait_event(cs_ctx* ctx, void* stream, void* event)
{
    if (!event) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return CS_OK;
}

int cs_stream_wait_tiles(cs_ctx* ctx, void* stream, cs_ctx* tiles_ctx, int32_t epoch, int32_t timeout_us)
{
    CS_ENTER(ctx);
    if (!tiles_ctx || !tiles_ctx->d_tiles_started || tiles_ctx->device != ctx->device) return fail(ctx, CS_ERR_INVALID, "no tile context on this device");
    if (epoch <= 0 || epoch >= (1 << 23)) return fail(ctx, CS_ERR_INVALID, "tile epochs are 1 .. 2^23 - 1");
    hipLaunchKernelGGL(cs_wait_tiles_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned*)tiles_ctx->d_tiles_started,
                       (unsigned)epoch, (long long)std::max(1, std::min(timeout_us, 5000)) * 100);
    CS_HIP(ctx, hipGetLastError());
    return CS_OK;
}

int cs_event_elapsed_ms(cs_ctx* ctx, void* start, void* stop, float* ms)
{
    if (!start || !stop || !ms) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipEventSynchronize((hipEvent_t)stop));
    CS_HIP(ctx, hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return CS_OK;
}

// --------------------------------------------------------------------------------------------
int cs_normxcorr2(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                  const cs_normxcorr2_params* p, const cs_matrix* out_corr, const cs_matrix* out_nobs)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!p) return fail(ctx, CS_ERR_INVALID, "null params");
    int rc = check_matrix(ctx, out_corr, "out_corr", p->ns);
    AllowCounts allow_counts(ctx);                   // (the signal only: the outputs were just checked without it)
    if (rc) return rc;
    const bool want_nobs = out_nobs && out_nobs->d_ptr;
    if (want_nobs) {
        rc = check_matrix(ctx, out_nobs, "out_nobs", p->ns);
        if (rc) return rc;
        if (out_nobs->dtype != CS_F32) return fail(ctx, CS_ERR_INVALID, "out_nobs must be float32");
    }
    const bool allow_fast = getenv("CHROMOSIGHT_HIP_FORCE_GENERIC") == nullptr;
    if (ctx->range_check && signal && signal->d_ptr && kernel) {
        // the guard of cs_ctx_set_range_check: largest |pixel| of the rows this call reads (the reduction of
        // cs_normxcorr2_host's slabs), then CS_ERR_RANGE for a non-finite pixel or, in float32, a magnitude beyond 1e15
        const int kh = (kernel->km - 1) / 2;
        const int rb = (p->row_begin == 0 && p->row_end == 0) ? 0 : p->row_begin, re = (p->row_begin == 0 && p->row_end == 0) ? p->ms : p->row_end;
        const int p_lo = std::max(0, rb - kh), p_hi = std::min(p->ms, re + (kernel->km - 1) - kh);
        const int width = is_band(signal->layout) ? signal->band_w : p->ns;
        if (!ctx->d_counts_peak) {
            CS_HIP(ctx, hipMalloc(&ctx->d_counts_peak, 256));
            CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_peak, 256, hipHostMallocDefault));
        }
        CS_HIP(ctx, hipMemsetAsync(ctx->d_counts_peak, 0, 4, stream));
        const size_t esz = signal->dtype == CS_F64 ? 8 : 4;
        const char* src = (const char*)signal->d_ptr + ((long long)p_lo - signal->row0) * signal->ld * (long long)esz;
        if (p_hi > p_lo && cs::launch_peak_rows(src, signal->dtype == CS_F64, signal->ld, p_hi - p_lo, width, ctx->n_cu,
                                                reinterpret_cast<unsigned*>(ctx->d_counts_peak), stream) != 0)
            return fail(ctx, CS_ERR_HIP, "range reduction failed to launch");
        CS_HIP(ctx, hipMemcpyAsync(ctx->h_peak, ctx->d_counts_peak, 4, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        const float limit = p->compute_dtype == CS_F32 ? 1e15f : 3.4e38f;
        unsigned limit_bits;
        std::memcpy(&limit_bits, &limit, 4);
        if (*ctx->h_peak > limit_bits)
            return fail(ctx, CS_ERR_RANGE, p->compute_dtype == CS_F32 ? "the map holds non-finite pixels or magnitudes beyond 1e15: float64 path"
                                                                      : "the map holds non-finite pixels");
    }
    if (p->compute_dtype == CS_F64) {
        cs::CorrArgs<double> A;
        rc = build_args<double>(ctx, stream, signal, kernel, p, &A);
        if (rc) return rc;
        A.out = view_of(out_corr);
        A.out_is_f64 = out_corr->dtype == CS_F64;
        A.nobs = want_nobs ? view_of(out_nobs) : cs::MatView{nullptr, 0, 0, 0, 0, 0};
        return launch_corr<double>(ctx, A, stream, allow_fast);
    } else if (p->compute_dtype == CS_F32) {
        cs::CorrArgs<float> A;
        rc = build_args<float>(ctx, stream, signal, kernel, p, &A);
        if (rc) return rc;
        A.out = view_of(out_corr);
        A.out_is_f64 = out_corr->dtype == CS_F64;
        A.nobs = want_nobs ? view_of(out_nobs) : cs::MatView{nullptr, 0, 0, 0, 0, 0};
        return launch_corr<float>(ctx, A, stream, allow_fast);
    }
    return fail(ctx, CS_ERR_INVALID, "bad compute dtype");
}

// Host map in, host map out, pipelined over PCIe in row slabs (see the header).
int cs_normxcorr2_host(cs_ctx* ctx, const void* h_signal, int32_t sig_dtype, int64_t ld_in, const cs_kernel* kernel,
                       const cs_normxcorr2_params* p, void* h_out, int32_t out_dtype, int64_t ld_out)
{
    CS_ENTER(ctx);
    if (!p || !h_signal || !h_out || !kernel) return fail(ctx, CS_ERR_INVALID, "null argument");
    if (p->mask_mode != CS_MASK_NONE) return fail(ctx, CS_ERR_UNSUPPORTED, "cs_normxcorr2_host takes unmasked maps");
    if (p->compute_dtype != CS_F32) return fail(ctx, CS_ERR_UNSUPPORTED, "cs_normxcorr2_host computes in float32");
    if (out_dtype != CS_F32 && out_dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "bad output dtype");
    if (sig_dtype != CS_F32 && sig_dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "bad signal dtype");
    const size_t esz = sig_dtype == CS_F64 ? 8 : 4;          // float64 maps are narrowed on the device, slab by slab
    const int ms = p->ms, ns = p->ns;
    if (ms <= 0 || ns <= 0 || ld_in < ns || ld_out < ns) return fail(ctx, CS_ERR_INVALID, "bad geometry");
    const int km = kernel->km;
    const int kh = (km - 1) / 2, kt = km - 1 - kh;          // rows a window reaches above / below its pixel
    const int64_t ld = ((int64_t)ns + 15) / 16 * 16;
    const size_t map_bytes = (size_t)ms * (size_t)ld * 8;     // sized for either input type
    if (map_bytes > ctx->d_host_bytes) {
        CS_HIP(ctx, hipDeviceSynchronize());
        if (ctx->d_host_in) CS_HIP(ctx, hipFree(ctx->d_host_in));
        if (ctx->d_host_out) CS_HIP(ctx, hipFree(ctx->d_host_out));
        ctx->d_host_in = ctx->d_host_out = nullptr;
        ctx->d_host_bytes = 0;
        CS_HIP(ctx, hipMalloc(&ctx->d_host_in, map_bytes));
        CS_HIP(ctx, hipMalloc(&ctx->d_host_out, map_bytes));
        ctx->d_host_bytes = map_bytes;
    }
    const size_t bounce_bytes = (size_t)ms * (size_t)ns * 4;
    if (bounce_bytes > ctx->h_bounce_bytes) {
        if (ctx->h_bounce) CS_HIP(ctx, hipHostFree(ctx->h_bounce));
        ctx->h_bounce = nullptr;
        ctx->h_bounce_bytes = 0;
        CS_HIP(ctx, hipHostMalloc(&ctx->h_bounce, bounce_bytes, hipHostMallocDefault));
        ctx->h_bounce_bytes = bounce_bytes;
    }
    if (!ctx->s_up) {
        CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_up, hipStreamNonBlocking));
        CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_run, hipStreamNonBlocking));
        CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->s_down, hipStreamNonBlocking));
    }
    // slabs of ~1/12 of the map, whole 64-row tiles
    int rows = std::max(64, ((ms + 11) / 12 + 63) / 64 * 64);
    const int n_slabs = (ms + rows - 1) / rows;
    while ((int)ctx->ev_up.size() < n_slabs) {
        hipEvent_t a, b, c;
        CS_HIP(ctx, hipEventCreateWithFlags(&a, hipEventDisableTiming));
        CS_HIP(ctx, hipEventCreateWithFlags(&b, hipEventDisableTiming));
        CS_HIP(ctx, hipEventCreateWithFlags(&c, hipEventDisableTiming));
        ctx->ev_up.push_back(a);
        ctx->ev_run.push_back(b);
        ctx->ev_down.push_back(c);
    }
    cs_matrix m_in{ctx->d_host_in, sig_dtype, CS_LAYOUT_DENSE, ld, 0, 0, 0};
    cs_matrix m_out{ctx->d_host_out, CS_F32, CS_LAYOUT_DENSE, ld, 0, 0, 0};
    // warm the template upload (it synchronises) before the pipeline starts
    {
        cs_normxcorr2_params p0 = *p;
        p0.row_begin = 0;
        p0.row_end = 0;
        cs::CorrArgs<float> A;
        int rc0 = build_args<float>(ctx, ctx->s_run, &m_in, kernel, &p0, &A);
        if (rc0) return rc0;
    }
    // ---- host side of the drain: workers convert / copy each slab out of the bounce buffer as soon as
    //      its download has completed
    const int n_workers = (int)std::min<unsigned>(8, std::max(1u, std::thread::hardware_concurrency()));
    std::vector<std::thread> workers;
    std::vector<int> worker_rc(n_workers, 0);
    // slabs whose download has been ENQUEUED (an event that was never recorded, or still carries the
    // previous call's record, would let hipEventSynchronize return at once); -1 = give up
    std::atomic<int> enqueued{0};
    const float* bounce = reinterpret_cast<const float*>(ctx->h_bounce);
    const int device = ctx->device;
    for (int w = 0; w < n_workers; ++w) {
        workers.emplace_back([=, &worker_rc, &enqueued]() {
            (void)hipSetDevice(device);
            for (int k = 0; k < n_slabs; ++k) {
                int seen;
                while ((seen = enqueued.load(std::memory_order_acquire)) <= k && seen >= 0) std::this_thread::yield();
                if (seen < 0) return;
                if (hipEventSynchronize(ctx->ev_down[k]) != hipSuccess) {
                    worker_rc[w] = 1;
                    return;
                }
                const int r0 = k * rows, r1 = std::min(ms, r0 + rows);
                const int span = r1 - r0, lo = r0 + (int)((long long)span * w / n_workers),
                          hi = r0 + (int)((long long)span * (w + 1) / n_workers);
                for (int r = lo; r < hi; ++r) {
                    const float* src = bounce + (size_t)r * ns;
                    if (out_dtype == CS_F64) {
                        double* dst = reinterpret_cast<double*>(h_out) + (size_t)r * ld_out;
                        for (int c = 0; c < ns; ++c) dst[c] = (double)src[c];
                    } else {
                        std::memcpy(reinterpret_cast<float*>(h_out) + (size_t)r * ld_out, src, (size_t)ns * 4);
                    }
                }
            }
        });
    }
    // ---- enqueue: upload slab k + 1, kernel of slab k (its windows reach into slab k + 1), download slab k
    int rc = CS_OK;
    auto upload = [&](int k) -> hipError_t {
        const int r0 = k * rows, r1 = std::min(ms, r0 + rows);
        hipError_t e;
        const char* src = reinterpret_cast<const char*>(h_signal) + (size_t)r0 * ld_in * esz;
        if (ld_in == ns && ld == ns)
            e = hipMemcpyAsync((char*)ctx->d_host_in + (size_t)r0 * ld * esz, src, (size_t)(r1 - r0) * ns * esz,
                               hipMemcpyHostToDevice, ctx->s_up);
        else
            e = hipMemcpy2DAsync((char*)ctx->d_host_in + (size_t)r0 * ld * esz, (size_t)ld * esz, src, (size_t)ld_in * esz,
                                 (size_t)ns * esz, (size_t)(r1 - r0), hipMemcpyHostToDevice, ctx->s_up);
        if (e != hipSuccess) return e;
        return hipEventRecord(ctx->ev_up[k], ctx->s_up);
    };
    // the windows of slab k's last row reach kt rows down: with slabs of `rows` rows that is `ahead` slabs (1 unless the
    // template is taller than two slabs), all of which must have landed before the kernel of slab k starts (uploads are
    // issued in order on one stream, so waiting for the furthest one covers the others)
    (void)kh;
    if (!ctx->d_counts_peak) {
        CS_HIP(ctx, hipMalloc(&ctx->d_counts_peak, 256));
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_peak, 256, hipHostMallocDefault));
    }
    CS_HIP(ctx, hipMemsetAsync(ctx->d_counts_peak, 0, 4, ctx->s_run));
    const int ahead = std::max(1, (kt + rows - 1) / rows);
    int uploaded = -1;
    hipError_t he = hipSuccess;
    for (int k = 0; k < n_slabs && he == hipSuccess && rc == CS_OK; ++k) {
        const int need = std::min(k + ahead, n_slabs - 1);
        while (uploaded < need && he == hipSuccess) he = upload(++uploaded);
        if (he != hipSuccess) break;
        const int r0 = k * rows, r1 = std::min(ms, r0 + rows);
        he = hipStreamWaitEvent(ctx->s_run, ctx->ev_up[need], 0);
        if (he != hipSuccess) break;
        cs_normxcorr2_params pk = *p;
        pk.row_begin = r0;
        pk.row_end = r1;
        rc = cs_normxcorr2(ctx, ctx->s_run, &m_in, kernel, &pk, &m_out, nullptr);
        if (rc != CS_OK) break;
        // the slab's largest |pixel| on the side (16 us for the whole 4096^2 map): see CS_ERR_RANGE
        if (cs::launch_peak_rows((const char*)ctx->d_host_in + (size_t)r0 * ld * esz, sig_dtype == CS_F64, ld, r1 - r0, ns, ctx->n_cu,
                                 reinterpret_cast<unsigned*>(ctx->d_counts_peak), ctx->s_run) != 0) {
            he = hipErrorLaunchFailure;
            break;
        }
        he = hipEventRecord(ctx->ev_run[k], ctx->s_run);
        if (he != hipSuccess) break;
        he = hipStreamWaitEvent(ctx->s_down, ctx->ev_run[k], 0);
        if (he != hipSuccess) break;
        if (ld == ns)
            he = hipMemcpyAsync((char*)ctx->h_bounce + (size_t)r0 * ns * 4, (char*)ctx->d_host_out + (size_t)r0 * ld * 4,
                                (size_t)(r1 - r0) * ns * 4, hipMemcpyDeviceToHost, ctx->s_down);
        else
            he = hipMemcpy2DAsync((char*)ctx->h_bounce + (size_t)r0 * ns * 4, (size_t)ns * 4,
                                  (char*)ctx->d_host_out + (size_t)r0 * ld * 4, (size_t)ld * 4, (size_t)ns * 4,
                                  (size_t)(r1 - r0), hipMemcpyDeviceToHost, ctx->s_down);
        if (he != hipSuccess) break;
        he = hipEventRecord(ctx->ev_down[k], ctx->s_down);
        if (he == hipSuccess) enqueued.store(k + 1, std::memory_order_release);
    }
    unsigned peak_bits = 0u;
    if (he == hipSuccess && rc == CS_OK) {
        he = hipMemcpyAsync(ctx->h_peak, ctx->d_counts_peak, 4, hipMemcpyDeviceToHost, ctx->s_run);
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->s_run);
        if (he == hipSuccess) peak_bits = *ctx->h_peak;
    }
    if (he != hipSuccess || rc != CS_OK) enqueued.store(-1, std::memory_order_release);     // release the workers
    for (auto& t : workers) t.join();
    if (rc != CS_OK) return rc;
    if (he != hipSuccess) return fail(ctx, CS_ERR_HIP, "pipelined call failed: %s", hipGetErrorString(he));
    for (int w = 0; w < n_workers; ++w)
        if (worker_rc[w]) return fail(ctx, CS_ERR_HIP, "download wait failed");
    {
        const float limit = 1e15f;
        unsigned limit_bits;
        std::memcpy(&limit_bits, &limit, 4);
        if (p->compute_dtype == CS_F32 && peak_bits > limit_bits)
            return fail(ctx, CS_ERR_RANGE, "the map holds non-finite pixels or magnitudes beyond 1e15: float64 path");
    }
    return CS_OK;
}

int cs_xcorr2(cs_ctx* ctx, void* stream_, const cs_matrix* signal, int32_t ms, int32_t ns,
              const double* h_weights, int32_t km, int32_t kn, double threshold, int32_t compute_dtype,
              const cs_matrix* out)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!h_weights || km <= 0 || kn <= 0) return fail(ctx, CS_ERR_INVALID, "bad weights");
    if (ms < km || ns < kn) return fail(ctx, CS_ERR_INVALID, "signal smaller than kernel");
    int rc = check_matrix(ctx, signal, "signal", ns);
    if (rc) return rc;
    rc = check_matrix(ctx, out, "out", ns);
    if (rc) return rc;
    const int kk = km * kn;
    std::vector<double> w(3 * (size_t)kk, 0.0);
    for (int t = 0; t < kk; ++t) w[t] = h_weights[t];
    // exactly vertically symmetric weights: the folded chain of the streaming kernel applies
    bool sym = !std::getenv("CHROMOSIGHT_HIP_NO_SYMMETRY");
    for (int r = 0; r < km / 2 && sym; ++r)
        for (int c = 0; c < kn; ++c)
            if (w[r * kn + c] != w[(km - 1 - r) * kn + c]) {
                sym = false;
                break;
            }
#define CS_XC(TC)                                                         \
    {                                                                     \
        rc = upload_weights<TC>(ctx, stream, w);                          \
        if (rc) return rc;                                                \
        cs::CorrArgs<TC> A;                                               \
        std::memset(&A, 0, sizeof(A));                                    \
        A.sig = view_of(signal);                                          \
        A.sig_is_f64 = signal->dtype == CS_F64;                           \
        A.out = view_of(out);                                             \
        A.out_is_f64 = out->dtype == CS_F64;                              \
        A.ms = ms; A.ns = ns; A.km = km; A.kn = kn;                       \
        A.row_begin = 0; A.row_end = ms;                                  \
        A.max_dist = -1;                                                  \
        A.w = reinterpret_cast<const TC*>(ctx->d_w[sizeof(TC) == 8 ? 1 : 0]); \
        A.ks.n = (TC)kk; A.ks.thr = (TC)threshold;                        \
        A.xcorr_only = 1;                                                 \
        A.w_sym = sym ? 1 : 0;                                            \
        return launch_corr<TC>(ctx, A, stream, getenv("CHROMOSIGHT_HIP_FORCE_GENERIC") == nullptr); \
    }
    if (compute_dtype == CS_F64) CS_XC(double)
    if (compute_dtype == CS_F32) CS_XC(float)
#undef CS_XC
    return fail(ctx, CS_ERR_INVALID, "bad compute dtype");
}

int cs_rescore_f64(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                   const cs_normxcorr2_params* p, const int32_t* d_rows, const int32_t* d_cols,
                   int64_t n_px, double* d_out_corr, double* d_out_nobs)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (n_px < 0 || (n_px > 0 && (!d_rows || !d_cols || !d_out_corr)))
        return fail(ctx, CS_ERR_INVALID, "bad pixel list");
    cs::CorrArgs<double> A;
    int rc = build_args<double>(ctx, stream, signal, kernel, p, &A);
    if (rc) return rc;
    rc = cs::launch_rescore_f64(A, d_rows, d_cols, n_px, d_out_corr, d_out_nobs, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "rescore launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_compact_ge(cs_ctx* ctx, void* stream_, const cs_matrix* corr, int32_t ms, int32_t ns,
                  double threshold, int32_t lo_diag, int32_t hi_diag, int32_t* d_rows, int32_t* d_cols,
                  double* d_vals, int64_t cap, int64_t* d_count)
{
    CS_ENTER(ctx);
    int rc = check_matrix(ctx, corr, "corr", ns);
    if (rc) return rc;
    if (!d_rows || !d_cols || !d_vals || !d_count || cap < 0) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    rc = cs::launch_compact_ge(view_of(corr), corr->dtype == CS_F64, ms, ns, threshold, lo_diag, hi_diag, d_rows,
                               d_cols, d_vals, cap, (long long*)d_count, ctx->n_cu, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "compact launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

static int csr_view(cs_ctx* ctx, const cs_csr* m, cs::CsrView* v)
{
    if (!m) return fail(ctx, CS_ERR_INVALID, "null csr");
    if (m->n_rows < 0 || m->n_cols < 0 || m->nnz < 0) return fail(ctx, CS_ERR_INVALID, "bad csr shape");
    if (m->dtype != CS_F32 && m->dtype != CS_F64) return fail(ctx, CS_ERR_INVALID, "bad csr dtype");
    if (!m->d_indptr || (m->nnz > 0 && (!m->d_indices || !m->d_data))) return fail(ctx, CS_ERR_INVALID, "null csr arrays");
    v->n_rows = m->n_rows;
    v->n_cols = m->n_cols;
    v->nnz = m->nnz;
    v->indptr = (const long long*)m->d_indptr;
    v->row_end = m->d_row_end ? (const long long*)m->d_row_end : (const long long*)m->d_indptr + 1;
    v->col0 = m->col0;
    v->row_w = m->d_row_weight;
    v->col_w = m->d_row_weight ? m->d_col_weight : nullptr;
    if (m->d_row_weight && !m->d_col_weight) return fail(ctx, CS_ERR_INVALID, "row weights without column weights");
    v->indices = m->d_indices;
    v->data = m->d_data;
    v->is_f64 = m->dtype == CS_F64;
    return CS_OK;
}

int cs_distance_law_csr(cs_ctx* ctx, void* stream_, const cs_csr* mat, const uint8_t* d_detectable,
                        int32_t n_diags, double* d_sum, int64_t* d_cnt)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (n_diags < 0 || (n_diags > 0 && (!d_sum || !d_cnt))) return fail(ctx, CS_ERR_INVALID, "bad law buffers");
    rc = cs::launch_distance_law(v, d_detectable, n_diags, d_sum, (long long*)d_cnt, ctx->n_cu, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "distance law launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_detrend_csr(cs_ctx* ctx, void* stream_, const cs_csr* mat, const double* d_law, int32_t n_law,
                   double max_val, void* d_out)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (!d_law || n_law < 0 || (v.nnz > 0 && !d_out)) return fail(ctx, CS_ERR_INVALID, "bad detrend buffers");
    rc = cs::launch_detrend_csr(v, d_law, n_law, max_val, d_out, ctx->n_cu, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "detrend launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_csr_to_band(cs_ctx* ctx, void* stream_, const cs_csr* mat, const double* d_law, int32_t n_law,
                   double max_val, const cs_matrix* band)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (!band || !band->d_ptr) return fail(ctx, CS_ERR_INVALID, "null output matrix");
    if (band->dtype != CS_F32 && band->dtype != CS_F64 && band->dtype != CS_U8)
        return fail(ctx, CS_ERR_INVALID, "bad output dtype");
    if (band->layout == CS_LAYOUT_DENSE ? band->ld < v.n_cols : (band->band_w <= 0 || band->ld < band->band_w))
        return fail(ctx, CS_ERR_INVALID, "bad output geometry");
    rc = cs::launch_csr_to_band(v, d_law, n_law, max_val, view_of(band), band->dtype, ctx->n_cu,
                                (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "csr_to_band launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}


int cs_csr_band_extent(cs_ctx* ctx, void* stream_, const cs_csr* mat, int32_t lo_diag, int32_t hi_diag,
                       int64_t* d_begin, int64_t* d_end)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    if (v.n_rows > 0 && (!d_begin || !d_end)) return fail(ctx, CS_ERR_INVALID, "null extent buffers");
    rc = cs::launch_csr_band_extent(v, lo_diag, hi_diag, (long long*)d_begin, (long long*)d_end, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "band extent launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_distance_law_finish(cs_ctx* ctx, void* stream_, const double* d_sum, const int64_t* d_cnt, int32_t n_diags,
                           double* d_law)
{
    CS_ENTER(ctx);
    if (n_diags < 0 || (n_diags > 0 && (!d_sum || !d_cnt || !d_law))) return fail(ctx, CS_ERR_INVALID, "bad law buffers");
    int rc = cs::launch_law_finish(d_sum, (const long long*)d_cnt, n_diags, d_law, (hipStream_t)stream_);
    if (rc) return fail(ctx, CS_ERR_HIP, "law finish launch failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_remove_neighbours(const int64_t* h_bin1, const int64_t* h_bin2, const int64_t* h_order, int64_t n, int64_t win,
                         uint8_t* h_keep)
{
    if (n < 0 || win < 1 || (n > 0 && (!h_bin1 || !h_bin2 || !h_order || !h_keep))) return CS_ERR_INVALID;
    // kept patterns bucketed by (bin1 / win, bin2 / win): a neighbour closer than win on both axes lies in one of the 3 x 3
    // surrounding cells -- and a cell holds at most ONE kept pattern (two patterns of one cell are closer than win on both axes),
    // so the grid is a flat open-addressing table of (cell, pattern): nine probes of a few nanoseconds per pattern (the
    // node-based map of vectors this replaces took 140 ns per pattern: 2.3 of the 4.8 ms of a borders table of the C4 genome)
    size_t cap = 16;
    while (cap < 2 * (size_t)n + 2) cap <<= 1;
    std::vector<uint64_t> keys(cap, ~0ull);
    std::vector<int64_t> vals(cap);
    auto cell = [](int64_t a, int64_t b) { return ((uint64_t)(a + (1ll << 30)) << 32) | (uint64_t)(uint32_t)(b + (1ll << 30)); };
    auto slot_of = [&](uint64_t key) {
        size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 17) & (cap - 1);
        while (keys[h] != ~0ull && keys[h] != key) h = (h + 1) & (cap - 1);
        return h;
    };
    for (int64_t t = 0; t < n; ++t) h_keep[t] = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t i = h_order[t];
        if (i < 0 || i >= n) return CS_ERR_INVALID;
        const int64_t b1 = h_bin1[i], b2 = h_bin2[i];
        const int64_t c1 = b1 >= 0 ? b1 / win : -((-b1 + win - 1) / win), c2 = b2 >= 0 ? b2 / win : -((-b2 + win - 1) / win);
        bool close = false;
        for (int64_t d1 = -1; d1 <= 1 && !close; ++d1)
            for (int64_t d2 = -1; d2 <= 1 && !close; ++d2) {
                const size_t h = slot_of(cell(c1 + d1, c2 + d2));
                if (keys[h] == ~0ull) continue;
                const int64_t j = vals[h];
                const int64_t e1 = h_bin1[j] - b1, e2 = h_bin2[j] - b2;
                close = (e1 < 0 ? -e1 : e1) < win && (e2 < 0 ? -e2 : e2) < win;
            }
        if (!close) {
            h_keep[i] = 1;
            const size_t h = slot_of(cell(c1, c2));          // (empty: a kept pattern of this cell would have been close)
            keys[h] = cell(c1, c2);
            vals[h] = i;
        }
    }
    return CS_OK;
}

// 2 * Phi(-a), a >= 0 or NaN: the two-sided tail of stats.py:43-81 with the case split of the normal distribution
// function the reference calls (scipy.special.ndtr)
static double two_sided_tail(double a)
{
    const double x = -a * M_SQRT1_2, z = std::fabs(x);
    double y;
    if (z < M_SQRT1_2) y = 0.5 + 0.5 * std::erf(x);
    else {
        y = 0.5 * std::erfc(z);
        if (x > 0) y = 1.0 - y;
    }
    return 2.0 * y;
}

int cs_accept_records(const cs_focus* h_rec, int64_t n_blocks, const int64_t* h_counts, const int32_t* h_rows,
                      const int32_t* h_cols, const int32_t* h_max_dist, int32_t inter, int32_t km, int32_t kn,
                      double missing_tol, double zero_tol, int32_t full, int32_t flags, double* h_table, uint8_t* h_ok,
                      int64_t* h_kept)
{
    const bool compact = (flags & 1) != 0, have_p = (flags & 2) != 0;
    if (n_blocks < 0 || km < 1 || kn < 1 || (n_blocks > 0 && (!h_counts || !h_rows || !h_cols || !h_kept))) return CS_ERR_INVALID;
    const double tot = (double)km * (double)kn;
    int64_t n = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        if (h_counts[b] < 0) return CS_ERR_INVALID;
        n += h_counts[b];
    }
    if (n > 0 && (!h_rec || !h_table || !h_ok)) return CS_ERR_INVALID;
    // pass 1, record by record (a few transcendental functions each: threads beyond a couple of thousand records):
    // the row of every record at its own slot
    auto rows_piece = [&](int64_t b, int64_t at, int64_t cnt) {
        {
            const int64_t ms = h_rows[b], ns = h_cols[b];
            const bool limited = !inter && h_max_dist && h_max_dist[b] >= 0;
            for (int64_t t = at; t < at + cnt; ++t) {
                const cs_focus& f = h_rec[t];
                const int64_t r = f.bin1, c = f.bin2;
                // coefficient on the trimmed map (detection.py:269-270) ...
                bool in_band = r >= 0 && r < ms && c >= 0 && c < ns;
                if (!inter) in_band = in_band && c - r >= 0 && (!limited || c - r <= (int64_t)h_max_dist[b]);
                // acceptance rules of validate_patterns (:121-141) on the window statistics
                const double undetected = (double)f.n_missing / tot;
                const double zero = (double)f.n_zero / (tot - (double)f.n_missing);   // 0 / 0 -> NaN -> rejected
                const bool ok = f.inside != 0 && undetected < missing_tol && zero < zero_tol;
                h_ok[t] = ok ? 1 : 0;
                if (!ok && compact) continue;
                // ... p-value on the untrimmed one (:332-336), Fisher z (stats.py:43-81)
                double n_obs = full ? f.n_obs : tot;
                if (n_obs == 0) n_obs = tot;
                double pval = 1.0;                                // 10 ** 0 where the coefficient is exactly 0
                if (have_p) {
                    pval = f.pval;                                // formed by the kernel that wrote the record (cs_foci.hip focus_pval)
                } else if (f.score != 0) {
                    const double zz = std::atanh(f.score) * std::sqrt(n_obs - 3.0);
                    // (the reference forms 10 ** log10(p): p again within two units in the last place -- the round trip through
                    // log10 and pow was 40 % of a record's arithmetic, and the p-values are pinned to 1e-12)
                    pval = two_sided_tail(std::fabs(zz));
                }
                double* row = h_table + 4 * t;
                row[0] = (double)r;
                row[1] = (double)c;
                row[2] = ok ? (in_band ? f.score : 0.0) : std::numeric_limits<double>::quiet_NaN();
                row[3] = pval;
            }
        }
    };
    // tasks: a block's records, long blocks in pieces of 512 (waking a sleeping worker costs the caller ~ 4 us: 16 threads at most --
    // with 64 the 56 000 records of a genome's 1-D pattern took 266 us, with 32 and pieces of 1024 180 us); the pool's workers and this thread take them from a counter
    // (records that carry their p-values cost a few nanoseconds each: pieces of 4096, so that a share's few thousand stay on the
    // calling thread)
    const int64_t piece_n = have_p ? 4096 : 512;
    struct Piece { int64_t b, at, n; };
    std::vector<Piece> pieces;
    {
        int64_t at = 0;
        for (int64_t b = 0; b < n_blocks; ++b) {
            for (int64_t o = 0; o < h_counts[b]; o += piece_n) pieces.push_back({b, at + o, std::min<int64_t>(piece_n, h_counts[b] - o)});
            at += h_counts[b];
        }
    }
    static const int cores = (int)std::max(1u, std::thread::hardware_concurrency());
    const int max_threads = (int)std::min<int64_t>(std::min(16, std::max(1, cores / 2)), n / piece_n);
    HostPool::get().run((int)pieces.size(), max_threads, [&](int t) {
        const Piece& pc = pieces[(size_t)t];
        rows_piece(pc.b, pc.at, pc.n);
    });
    // pass 2: counts, and in compact mode the accepted rows packed to the front (in place: a row never moves backwards)
    int64_t at = 0, out = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        int64_t kept = 0;
        for (int64_t t = at; t < at + h_counts[b]; ++t) {
            if (!h_ok[t]) continue;
            ++kept;
            if (compact) {
                if (out != t) std::memcpy(h_table + 4 * out, h_table + 4 * t, 4 * sizeof(double));
                ++out;
            }
        }
        h_kept[b] = kept;
        at += h_counts[b];
    }
    return CS_OK;
}

// grow-only device scratch; growing waits for the device, since queued work may still use the old block
static int ensure_scratch(cs_ctx* ctx, void** buf, size_t* have, size_t need)
{
    if (need <= *have) return CS_OK;
    if (*buf) {
        CS_HIP(ctx, hipDeviceSynchronize());
        CS_HIP(ctx, hipFree(*buf));
        *buf = nullptr;
        *have = 0;
    }
    const size_t want = need + need / 4;
    CS_HIP(ctx, hipMalloc(buf, want));
    *have = want;
    return CS_OK;
}

int cs_csr_median(cs_ctx* ctx, void* stream_, const cs_csr* mat, double* h_median)
{
    CS_ENTER(ctx);
    if (!h_median) return fail(ctx, CS_ERR_INVALID, "null output");
    cs::CsrView v;
    int rc = csr_view(ctx, mat, &v);
    if (rc) return rc;
    auto grow = [](void* user, size_t bytes) -> void* {
        cs_ctx* c = (cs_ctx*)user;
        return ensure_scratch(c, &c->d_pool, &c->d_pool_bytes, bytes) == CS_OK ? c->d_pool : nullptr;
    };
    rc = cs::csr_median(v, ctx->n_cu, (hipStream_t)stream_, grow, ctx, h_median);
    if (rc) return fail(ctx, CS_ERR_HIP, "median failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_csr_median_many(cs_ctx* ctx, void* stream_, const cs_csr* mats, int32_t n, double* h_medians)
{
    CS_ENTER(ctx);
    if (n < 0 || (n > 0 && (!mats || !h_medians))) return fail(ctx, CS_ERR_INVALID, "bad view list");
    std::vector<cs::CsrView> views((size_t)n);
    for (int i = 0; i < n; ++i) {
        int rc = csr_view(ctx, mats + i, &views[(size_t)i]);
        if (rc) return rc;
    }
    auto grow = [](void* user, size_t bytes) -> void* {
        cs_ctx* c = (cs_ctx*)user;
        return ensure_scratch(c, &c->d_pool, &c->d_pool_bytes, bytes) == CS_OK ? c->d_pool : nullptr;
    };
    int rc = cs::csr_median_many(views.data(), n, ctx->n_cu, (hipStream_t)stream_, grow, ctx, h_medians);
    if (rc) return fail(ctx, CS_ERR_HIP, "medians failed: %s", hipGetErrorString((hipError_t)rc));
    return CS_OK;
}

int cs_stage_blocks(cs_ctx* ctx, void* stream_, const cs_csr* genome, const cs_stage_block* blocks, int32_t n_blocks, double max_val)
{
    CS_ENTER(ctx);
    cs::CsrView v;
    int rc = csr_view(ctx, genome, &v);
    if (rc) return rc;
    if (n_blocks < 0 || (n_blocks > 0 && !blocks)) return fail(ctx, CS_ERR_INVALID, "bad block table");
    if (n_blocks == 0) return CS_OK;
    if (genome->d_row_end || genome->col0 != 0 || !v.row_w || v.row_w != v.col_w || v.n_rows != v.n_cols)
        return fail(ctx, CS_ERR_INVALID, "cs_stage_blocks takes the whole-genome pixel table (square, one weight vector, plain row pointers)");
    // rows per group: enough groups to keep every CU's waves on different rows (a wave walks its rows one after the other,
    // each a chain of dependent loads), 64 to 128 rows (measured: 50 000-bin block 0.353 -> 0.335 ms per C3 step with 64,
    // the 200 000-bin genome 1.59 -> 1.47 ms with 128; a group costs an LDS clear, a flush and a slot for the finish pass)
    long long total_rows = 0;
    for (int b = 0; b < n_blocks; ++b) total_rows += std::max(blocks[b].n, 0);
    const int kRowsPerGroup = (int)std::min<long long>(128, std::max<long long>(64, (total_rows / (4LL * ctx->n_cu) + 7) / 8 * 8));
    std::vector<cs::StageBlock> tab((size_t)n_blocks);
    int n_groups = 0, pitch = 1;
    for (int b = 0; b < n_blocks; ++b) {
        const cs_stage_block& s = blocks[b];
        cs::StageBlock& B = tab[b];
        if (s.n <= 0 || s.row0 < 0 || s.row0 + s.n > v.n_rows || s.keep < 0) return fail(ctx, CS_ERR_INVALID, "block %d outside the genome", b);
        if (s.layout != CS_LAYOUT_BAND && s.layout != CS_LAYOUT_DENSE) return fail(ctx, CS_ERR_INVALID, "block %d: bad layout", b);
        B.row0 = s.row0;
        B.n = s.n;
        B.keep = s.keep;
        B.n_diags = (int)std::min<int64_t>(s.n, (int64_t)s.keep + 1);
        B.dense = s.layout == CS_LAYOUT_DENSE;
        B.width = B.dense ? s.n : B.n_diags;
        if (!B.dense && s.band_w != B.n_diags) return fail(ctx, CS_ERR_INVALID, "block %d: band_w must be min(keep, n - 1) + 1", b);
        if (s.ld < B.width) return fail(ctx, CS_ERR_INVALID, "block %d: ld < stored slots", b);
        if (!s.d_law || (!s.d_band64 && !s.d_band32)) return fail(ctx, CS_ERR_INVALID, "block %d: null outputs", b);
        B.ld = s.ld;
        B.band64 = (double*)s.d_band64;
        B.band32 = (float*)s.d_band32;
        B.law = s.d_law;
        B.ld64 = s.ld;
        B.w64 = 0;
        B.counts = 0;
        B.lazy = nullptr;
        if (s.band32_counts) {
            // CS_LAYOUT_BAND_COUNTS: d_band32 receives the raw counts (the caller vouches that they are exact in float32)
            if (B.dense || !s.d_band32 || (s.ld & 3) || s.ld < (int64_t)B.width + 4 || ((uintptr_t)s.d_band32 & 15) || s.d_band64 || s.f64_diags > 0)
                return fail(ctx, CS_ERR_INVALID, "block %d: a band of counts needs the band layout, a 16-byte aligned d_band32 behind its header, ld a multiple of 4 and >= band_w + 4, and no float64 band", b);
            B.counts = 1;
            B.lazy = reinterpret_cast<cs::LazyBand*>(s.d_lazy);          // (or null: float32 consumers only)
        } else if (s.f64_diags > 0 || s.d_lazy) {
            if (B.dense || !s.d_lazy || !s.d_band64 || s.f64_diags <= 0 || s.ld64 < s.f64_diags || (s.ld64 & 1))
                return fail(ctx, CS_ERR_INVALID, "block %d: a lazy float64 band needs the band layout, d_lazy, d_band64 and an even ld64 >= f64_diags > 0", b);
            B.ld64 = s.ld64;
            B.w64 = std::min<int>(s.f64_diags, B.width);
            B.lazy = reinterpret_cast<cs::LazyBand*>(s.d_lazy);
        }
        B.group0 = B.n_groups = 0;
        n_groups += (s.n + kRowsPerGroup - 1) / kRowsPerGroup;
        pitch = std::max(pitch, B.n_diags);
    }
    if (pitch > 4096) return fail(ctx, CS_ERR_UNSUPPORTED, "distance laws of more than 4096 diagonals: stage block by block");
    pitch = (pitch + 63) / 64 * 64;
    const size_t stage_had = ctx->d_stage_bytes;
    rc = ensure_scratch(ctx, &ctx->d_stage, &ctx->d_stage_bytes, cs::stage_scratch_bytes(n_blocks, n_groups, pitch, v.n_rows));
    if (rc) return rc;
    if (ctx->d_stage_bytes != stage_had) ctx->stage_uploaded.clear();       // a fresh allocation holds no tables
    // the block / group tables go through one of two page-locked slots; a slot is reused two calls later, after the
    // event recorded behind its copy has fired (normally long ago): no synchronisation on the way in
    const int slot = ctx->stage_slot ^= 1;
    const size_t tbytes = cs::stage_table_bytes(n_blocks, n_groups);
    if (!ctx->ev_stage[slot]) CS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_stage[slot], hipEventDisableTiming));
    else CS_HIP(ctx, hipEventSynchronize(ctx->ev_stage[slot]));
    if (tbytes > ctx->h_stage_bytes[slot]) {
        if (ctx->h_stage[slot]) CS_HIP(ctx, hipHostFree(ctx->h_stage[slot]));
        ctx->h_stage[slot] = nullptr;
        ctx->h_stage_bytes[slot] = 0;
        CS_HIP(ctx, hipHostMalloc(&ctx->h_stage[slot], 2 * tbytes, hipHostMallocDefault));
        ctx->h_stage_bytes[slot] = 2 * tbytes;
    }
    rc = cs::enqueue_stage_blocks(v.indptr, v.indices, v.data, v.is_f64, v.row_w, v.n_rows, tab.data(), n_blocks, max_val,
                                  kRowsPerGroup, ctx->n_cu, ctx->d_stage, ctx->h_stage[slot], (hipStream_t)stream_, &ctx->stage_uploaded);
    if (rc) return fail(ctx, CS_ERR_HIP, "staging kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipEventRecord(ctx->ev_stage[slot], (hipStream_t)stream_));
    return CS_OK;
}

int cs_host_alloc(cs_ctx* ctx, size_t bytes, void** h_ptr)
{
    if (!h_ptr) return CS_ERR_INVALID;
    CS_ENTER(ctx);
    CS_HIP(ctx, hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return CS_OK;
}

int cs_host_free(cs_ctx* ctx, void* h_ptr)
{
    CS_ENTER(ctx);
    if (h_ptr) CS_HIP(ctx, hipHostFree(h_ptr));
    return CS_OK;
}

static int check_foci_args(cs_ctx* ctx, const cs_matrix* signal, const cs_kernel* kernel, const cs_normxcorr2_params* p,
                           const cs_foci_params* fp)
{
    if (!signal || !kernel || !p || !fp) return fail(ctx, CS_ERR_INVALID, "null argument");
    // (every foci entry shares d_pool, the candidate counters and the argument tables with a pending prepare form of
    // cs_detect_foci_blocks: whatever runs in between invalidates it -- that entry notes the flag before its own checks)
    ctx->prep_pending = false;
    // (every foci entry shares the context's count words, block tables and record pool with an asynchronous batch)
    if (ctx->nb_pending) return fail(ctx, CS_ERR_INVALID, "an asynchronous batch is pending on this context: cs_detect_foci_batch_finish first");
    if (p->mask_mode == CS_MASK_EXPLICIT) return fail(ctx, CS_ERR_UNSUPPORTED, "device foci need per-bin masks or none");
    if (signal->layout == CS_LAYOUT_BAND_COUNTS || signal->layout == CS_LAYOUT_BAND_COUNTS_VIEW)
        return fail(ctx, CS_ERR_INVALID, "a band of counts is the float32 twin of a block (signals_f32), not its signal: hand the block's CS_LAYOUT_BAND_LAZY descriptor");
    if (fp->min_size < 1) return fail(ctx, CS_ERR_INVALID, "min_size must be >= 1");
    if (!ctx->h_counts) CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_counts, 64, hipHostMallocDefault));
    return CS_OK;
}

namespace {
// Where the masked matrix-core tile kernel appends its candidates (keys tag + row * ns + col); see CorrArgs::cand_keys.
struct CandSink {
    unsigned long long* keys;
    unsigned long long* count;       // zeroed by the caller
    long long cap;
    unsigned long long tag;
    int lo_diag, hi_diag;            // scanned diagonals
    void* defer_args = nullptr;      // see CorrArgs::defer_args: prepare the tile kernel's launch, do not launch
    int* defer_rsym = nullptr;
};

// float32 correlation in candidate mode (cs_device.h cand_screen_*: margin + conditioning screen, sentinel 2.0).  With a
// sink and the masked tile kernel the candidates are appended to the sink and no map is written (ctx->cand_fused);
// otherwise the map goes to `out` -- CS_NEED_MAP when that has no storage, before anything that matters was launched.
int corr_candidates_f32(cs_ctx* ctx, hipStream_t stream, const cs_matrix* signal, const cs_kernel* kernel,
                        const cs_normxcorr2_params* p, const cs_matrix* out, double margin, double thr, const CandSink* sink)
{
    cs::CorrArgs<float> A;
    int rc = build_args<float>(ctx, stream, signal, kernel, p, &A);
    if (rc) return rc;
    // windows conditioned at least 8 n 2^-24 / margin have a float32 error below margin / 4 (2 gamma / conditioning)
    A.ks.cand_cmin = (float)std::min(0.5, 8.0 * (double)A.ks.n * 0x1p-24 / margin);
    A.ks.cand_thr = (float)thr;
    A.out = view_of(out);
    A.out_is_f64 = 0;
    A.nobs = cs::MatView{nullptr, 0, 0, 0, 0, 0};
    if (sink) {
        A.cand_keys = sink->keys;
        A.cand_count = sink->count;
        A.cand_cap = sink->cap;
        A.cand_tag = sink->tag;
        A.cand_dlo = sink->lo_diag;
        A.cand_dhi = sink->hi_diag;
        A.defer_args = sink->defer_args;
        A.defer_rsym = sink->defer_rsym;
    } else if (!out || !out->d_ptr) {
        return fail(ctx, CS_ERR_INVALID, "candidate mode needs a sink or a map");
    }
    ctx->cand_fused = false;
    return launch_corr<float>(ctx, A, stream, getenv("CHROMOSIGHT_HIP_FORCE_GENERIC") == nullptr);
}

// 1-D patterns (cs_foci_params.diag_only): the reference forces bin1 = bin2 AFTER shifting the coordinates by (kh, kw)
// into a map padded by (kw, kh) (detection.py:287-315, preprocessing.py:636-676), so with a non-square template in full
// mode the row ends up kw - kh away from the column.  The kernels take an odd code whose upper bits hold that offset.
inline int diag_code(const cs_foci_params* fp, const cs_kernel* kernel, const cs_normxcorr2_params* p)
{
    if (!fp->diag_only) return 0;
    const int shift = p->full ? (kernel->kn - 1) / 2 - (kernel->km - 1) / 2 : 0;
    return shift * 2 + 1;
}

// Candidate stage shared by cs_detect_foci and cs_candidates: the coefficient map of the row window in
// context scratch, thresholded compaction (or, for 1-D patterns, the enumeration of the few scanned
// diagonals).  The pool is laid out as rows | cols | vals | counters | windows | tail.
struct CandPlan {
    size_t off_cols = 0, off_vals = 0, off_cnt = 0, off_win = 0, off_tail = 0;
    long long n_cand = 0;
    int row_major = 0;       // the candidate list is already sorted row-major (1-D patterns)
};

int find_candidates(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                    const cs_normxcorr2_params* p, const cs_foci_params* fp, size_t win_bytes,
                    size_t (*tail_bytes)(long long), CandPlan* P)
{
    hipStream_t stream = (hipStream_t)stream_;
    int rc;
    int rb = 0, re = p->ms;
    if (p->row_end > p->row_begin) {
        if (p->row_begin < 0 || p->row_end > p->ms) return fail(ctx, CS_ERR_INVALID, "row window outside the matrix");
        rb = p->row_begin;
        re = p->row_end;
    }
    // ---- coefficient map in context-owned scratch, with the signal's layout
    const bool f64 = p->compute_dtype == CS_F64;
    cs_matrix map;
    map.d_ptr = nullptr;
    map.dtype = f64 ? CS_F64 : CS_F32;
    map.layout = signal->layout == CS_LAYOUT_BAND_PADDED ? CS_LAYOUT_BAND : signal->layout;
    map.row0 = rb;
    if (signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED) {
        if (fp->hi_diag < fp->lo_diag) return fail(ctx, CS_ERR_INVALID, "empty diagonal range");
        map.band_lo = fp->lo_diag;
        map.band_w = fp->hi_diag - fp->lo_diag + 1;
        map.ld = ((int64_t)map.band_w + 63) / 64 * 64;
    } else {
        map.band_lo = map.band_w = 0;
        map.ld = ((int64_t)p->ns + 15) / 16 * 16;
    }
    // 1-D patterns (borders, hairpins: max_dist = 0 in the config, 2 scanned diagonals): a streamed
    // 128-column strip would compute 64 columns for every one it keeps.  Every pixel of the few diagonals
    // is a candidate instead and goes straight to the float64 evaluation (one wave per pixel).
    const bool narrow = (signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED) && map.band_w <= 4;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    auto layout = [&](size_t c_cap) {
        P->off_cols = al(4 * c_cap);
        P->off_vals = P->off_cols + al(4 * c_cap);
        P->off_cnt = P->off_vals + al(8 * c_cap);
        P->off_win = P->off_cnt + 256;
        P->off_tail = P->off_win + al(win_bytes);
        return ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, P->off_tail + tail_bytes((long long)c_cap));
    };
    P->n_cand = 0;
    if (narrow) {
        const long long n_cand = cs::narrow_band_pixels(rb, re, p->ns, map.band_lo, map.band_w);
        if (n_cand > INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%lld)", n_cand);
        if (n_cand > 0) {
            rc = layout((size_t)n_cand);
            if (rc) return rc;
            char* pool = (char*)ctx->d_pool;
            rc = cs::enqueue_enumerate_band(rb, re, p->ns, map.band_lo, map.band_w, n_cand, (int*)pool,
                                            (int*)(pool + P->off_cols), stream, &P->row_major);
            if (rc) return fail(ctx, CS_ERR_HIP, "enumerate launch failed: %s", hipGetErrorString((hipError_t)rc));
        }
        P->n_cand = n_cand;
        return CS_OK;
    }
    const size_t map_bytes = (size_t)(re - rb) * (size_t)map.ld * (f64 ? 8 : 4);
    // float32 maps are computed in candidate mode (cs_device.h cand_screen_*): a pixel keeps its value only if that is
    // below pearson - margin on a window conditioned well enough for the float32 error to stay under margin / 4; every
    // other pixel holds 2.0 and is re-evaluated.  float64 maps hold the coefficient.
    const double margin = std::max(fp->rescore_margin, 1e-4);
    const double thr = f64 ? fp->pearson : fp->pearson - margin;
    size_t c_cap = std::max<size_t>(1 << 16, (size_t)(re - rb) * (size_t)((signal->layout == CS_LAYOUT_BAND || signal->layout == CS_LAYOUT_BAND_PADDED) ? map.band_w : p->ns) / 256);
    // ---- float32, masked tile kernel: the kernel appends the candidates itself (no map, no compaction pass)
    bool fused_ok = !f64;
    while (fused_ok) {
        if (c_cap > (size_t)INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%zu)", c_cap);
        rc = layout(c_cap);
        if (rc) return rc;
        char* pool = (char*)ctx->d_pool;
        long long* d_cnt = (long long*)(pool + P->off_cnt);
        CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, stream));
        CandSink sink{(unsigned long long*)(pool + P->off_vals), (unsigned long long*)d_cnt, (long long)c_cap, 0ull,
                      fp->lo_diag, fp->hi_diag};
        rc = corr_candidates_f32(ctx, stream, signal, kernel, p, &map, margin, thr, &sink);
        if (rc == CS_NEED_MAP) {
            fused_ok = false;
            break;
        }
        if (rc) return rc;
        CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts, d_cnt, 8, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        P->n_cand = ctx->h_counts[0];
        if ((size_t)P->n_cand <= c_cap) {
            if (P->n_cand > 0) {
                rc = cs::launch_decode_keys((const long long*)(pool + P->off_vals), P->n_cand, p->ns, (int*)pool,
                                            (int*)(pool + P->off_cols), stream);
                if (rc) return fail(ctx, CS_ERR_HIP, "key decoding failed: %s", hipGetErrorString((hipError_t)rc));
            }
            return CS_OK;
        }
        c_cap = (size_t)P->n_cand + (size_t)P->n_cand / 8;      // the list overflowed: once more with room for all
    }
    // ---- coefficient map in context scratch, then thresholded compaction
    rc = ensure_scratch(ctx, &ctx->d_map, &ctx->d_map_bytes, map_bytes);
    if (rc) return rc;
    map.d_ptr = ctx->d_map;
    if (f64) rc = cs_normxcorr2(ctx, stream_, signal, kernel, p, &map, nullptr);
    else rc = corr_candidates_f32(ctx, stream, signal, kernel, p, &map, margin, thr, nullptr);
    if (rc) return rc;
    while (true) {
        if (c_cap > (size_t)INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%zu)", c_cap);
        rc = layout(c_cap);
        if (rc) return rc;
        char* pool = (char*)ctx->d_pool;
        long long* d_cnt = (long long*)(pool + P->off_cnt);
        CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, stream));
        rc = cs::launch_compact_ge(view_of(&map), f64, re, p->ns, thr, fp->lo_diag, fp->hi_diag, (int*)pool,
                                   (int*)(pool + P->off_cols), (double*)(pool + P->off_vals), (long long)c_cap, d_cnt,
                                   ctx->n_cu, stream);
        if (rc) return fail(ctx, CS_ERR_HIP, "compact launch failed: %s", hipGetErrorString((hipError_t)rc));
        CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts, d_cnt, 8, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        P->n_cand = ctx->h_counts[0];
        if ((size_t)P->n_cand <= c_cap) break;
        c_cap = (size_t)P->n_cand + (size_t)P->n_cand / 8;
    }
    return CS_OK;
}
}  // namespace

int cs_detect_foci(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                   const cs_normxcorr2_params* p, const cs_foci_params* fp, cs_focus* h_foci, int64_t cap,
                   int64_t* n_foci, double* h_windows)
{
    CS_ENTER(ctx);
    static_assert(sizeof(cs_focus) == sizeof(cs::FocusRec), "record layouts must agree");
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_foci_args(ctx, signal, kernel, p, fp);
    if (rc) return rc;
    if (!n_foci || cap < 0 || (cap > 0 && !h_foci)) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    if (p->row_end > p->row_begin && (p->row_begin != 0 || p->row_end != p->ms))
        return fail(ctx, CS_ERR_INVALID, "foci of a row window: use cs_candidates + cs_label_foci");
    *n_foci = 0;
    const int kk = kernel->km * kernel->kn;
    const size_t win_pat = fp->want_windows ? (size_t)std::max<int64_t>(cap, 1) : 0;
    CandPlan P;
    rc = find_candidates(ctx, stream_, signal, kernel, p, fp, 8 * win_pat * kk, cs::foci_scratch_bytes, &P);
    if (rc) return rc;
    const long long n_cand = P.n_cand;
    if (n_cand == 0) return CS_OK;
    // ---- foci
    cs::CorrArgs<double> A64;
    rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    long long* d_cnt = (long long*)(pool + P.off_cnt);
    double* d_win = fp->want_windows ? (double*)(pool + P.off_win) : nullptr;
    cs::FocusRec* d_rec = nullptr;
    // Page-locked output buffers (what cs_host_alloc hands out) are written by the last kernel itself: one
    // stream synchronisation per call instead of a count round trip plus two copies.
    auto device_view = [&](const void* h) -> void* {
        if (!h) return nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
    };
    cs::FocusRec* rec_direct = cap > 0 ? reinterpret_cast<cs::FocusRec*>(device_view(h_foci)) : nullptr;
    double* win_direct = (fp->want_windows && h_windows) ? reinterpret_cast<double*>(device_view(h_windows)) : nullptr;
    const bool direct = rec_direct && (!fp->want_windows || !h_windows || win_direct);
    if (direct) {
        ctx->h_counts[1] = -1;
        rc = cs::enqueue_foci(A64, (const int*)pool, (const int*)(pool + P.off_cols), n_cand, fp->pearson, fp->min_size,
                              diag_code(fp, kernel, p), fp->inter, pool + P.off_tail, &d_rec, win_direct, win_direct ? (long long)cap : 0,
                              d_cnt + 1, stream, P.row_major, rec_direct, (long long)cap, ctx->h_counts + 1);
        if (rc) return fail(ctx, CS_ERR_HIP, "foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
        CS_HIP(ctx, hipStreamSynchronize(stream));
        const long long n = ctx->h_counts[1];
        if (n < 0) return fail(ctx, CS_ERR_HIP, "foci kernels did not report a count");
        *n_foci = n;
        if (n > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", n, (long long)cap);
        return CS_OK;
    }
    rc = cs::enqueue_foci(A64, (const int*)pool, (const int*)(pool + P.off_cols), n_cand, fp->pearson, fp->min_size,
                          diag_code(fp, kernel, p), fp->inter, pool + P.off_tail, &d_rec, d_win, (long long)win_pat, d_cnt + 1, stream,
                          P.row_major, nullptr, 0, nullptr);
    if (rc) return fail(ctx, CS_ERR_HIP, "foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts + 1, d_cnt + 1, 8, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    const long long n = ctx->h_counts[1];
    *n_foci = n;
    if (n > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", n, (long long)cap);
    if (n > 0) {
        CS_HIP(ctx, hipMemcpyAsync(h_foci, d_rec, sizeof(cs_focus) * (size_t)n, hipMemcpyDeviceToHost, stream));
        if (d_win && h_windows)
            CS_HIP(ctx, hipMemcpyAsync(h_windows, d_win, 8 * (size_t)n * kk, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
    }
    return CS_OK;
}

int cs_detect_foci_batch_templates(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, int32_t n_kernels,
                                   const cs_kernel* kernels, const cs_normxcorr2_params* params, const cs_foci_params* foci,
                                   cs_focus* h_foci, int64_t cap, int64_t* h_n_foci, double* h_windows)
{
    CS_ENTER(ctx);
    AllowLazy allow_lazy(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks <= 0 || !signals || !kernels || !params || !foci || !h_n_foci || cap < 0 || (cap > 0 && !h_foci))
        return fail(ctx, CS_ERR_INVALID, "bad batch arguments");
    // (the weight sets of all templates stay resident side by side: the current one + the parked ones of upload_weights)
    if (n_kernels < 1 || n_kernels > 4) return fail(ctx, CS_ERR_UNSUPPORTED, "1 to 4 templates per batch");
    for (int t = 1; t < n_kernels; ++t)
        if (kernels[t].km != kernels[0].km || kernels[t].kn != kernels[0].kn)
            return fail(ctx, CS_ERR_INVALID, "the templates of a batch share their size");
    // virtual block v = t * n_blocks + b: sub-matrix b under template t
    const int n_virtual = n_blocks * n_kernels;
    // (the host tables live in the context: an asynchronous call -- foci[0].reserved & 1 -- returns while their copies may
    // still be in flight; cs_detect_foci_batch_finish ends the call)
    if (ctx->nb_pending) return fail(ctx, CS_ERR_INVALID, "an asynchronous batch is pending on this context: cs_detect_foci_batch_finish first");
    std::vector<cs::CorrArgs<double>>& tab = ctx->nb_tab;
    std::vector<long long>& seg = ctx->nb_seg;
    std::vector<int>& lo_w = ctx->nb_lo_w;
    tab.assign((size_t)n_virtual, cs::CorrArgs<double>{});
    seg.assign((size_t)n_virtual + 1, 0);
    lo_w.assign(2 * (size_t)n_virtual, 0);
    for (int t = 0; t < n_kernels; ++t)
        for (int b = 0; b < n_blocks; ++b) {
            const int v = t * n_blocks + b;
            const cs_normxcorr2_params* p = params + b;
            const cs_foci_params* fp = foci + b;
            int rc = check_foci_args(ctx, signals + b, kernels + t, p, fp);
            if (rc) return rc;
            const int w = fp->hi_diag - fp->lo_diag + 1;
            if (!is_band(signals[b].layout) || w < 1 || w > 4 || fp->lo_diag < 0 || (p->row_end > p->row_begin))
                return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry takes 1-D patterns (<= 4 scanned diagonals from 0 up) of whole banded blocks");
            if (fp->pearson != foci[0].pearson || fp->min_size != foci[0].min_size || fp->diag_only != foci[0].diag_only ||
                fp->inter != foci[0].inter || fp->want_windows != foci[0].want_windows)
                return fail(ctx, CS_ERR_INVALID, "the blocks of a batch share the detection parameters");
            const long long n_b = cs::narrow_band_pixels(0, p->ms, p->ns, fp->lo_diag, w);
            if (n_b > cs::kFociSmallMax) return fail(ctx, CS_ERR_UNSUPPORTED, "block %d has %lld candidate pixels", b, n_b);
            seg[v + 1] = seg[v] + n_b;
            lo_w[2 * v] = fp->lo_diag;
            lo_w[2 * v + 1] = w;
            rc = build_args<double>(ctx, stream, signals + b, kernels + t, p, &tab[v]);      // (uploads template t once: b == 0)
            if (rc) return rc;
        }
    const long long n_total = seg[n_virtual];
    if (n_total > INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%lld)", n_total);
    // results straight into page-locked caller buffers (cs_host_alloc); anything else goes through cs_detect_foci
    auto device_view = [&](const void* h) -> void* {
        if (!h) return nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
    };
    cs::FocusRec* rec = cap > 0 ? reinterpret_cast<cs::FocusRec*>(device_view(h_foci)) : nullptr;
    double* win = (foci[0].want_windows && h_windows) ? reinterpret_cast<double*>(device_view(h_windows)) : nullptr;
    if ((cap > 0 && !rec) || (foci[0].want_windows && h_windows && !win))
        return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry writes into page-locked buffers (cs_host_alloc)");
    // per-block counts through a page-locked array owned by the context
    const size_t cnt_bytes = 8 * ((size_t)n_virtual + 2);
    if (cnt_bytes > ctx->h_blk_bytes) {
        if (ctx->h_blk_counts) CS_HIP(ctx, hipHostFree(ctx->h_blk_counts));
        ctx->h_blk_counts = nullptr;
        ctx->h_blk_bytes = 0;
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_blk_counts, 2 * cnt_bytes, hipHostMallocDefault));
        ctx->h_blk_bytes = 2 * cnt_bytes;
    }
    ctx->h_blk_counts[0] = -1;
    int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, cs::narrow_batch_scratch_bytes(n_virtual, n_total));
    if (rc) return rc;
    rc = cs::enqueue_foci_narrow_batch(tab.data(), seg.data(), lo_w.data(), n_virtual, foci[0].pearson, foci[0].min_size,
                                       diag_code(foci, kernels, params), foci[0].inter, ctx->d_pool, rec, (long long)cap, win,
                                       win ? (long long)cap : 0, ctx->h_blk_counts, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "batched foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
    ctx->nb_pending = n_virtual;
    ctx->nb_cap = cap;
    if (foci[0].reserved & 1) return CS_OK;          // asynchronous: everything is enqueued, cs_detect_foci_batch_finish waits
    return cs_detect_foci_batch_finish(ctx, stream_, h_n_foci);
}

int cs_detect_foci_batch_finish(cs_ctx* ctx, void* stream_, int64_t* h_n_foci)
{
    CS_ENTER(ctx);
    if (!h_n_foci) return fail(ctx, CS_ERR_INVALID, "null counts");
    if (!ctx->nb_pending) return fail(ctx, CS_ERR_INVALID, "no batch is pending on this context");
    const int n_virtual = ctx->nb_pending;
    ctx->nb_pending = 0;
    CS_HIP(ctx, hipStreamSynchronize((hipStream_t)stream_));            // also: the host tables were consumed
    const long long total = ctx->h_blk_counts[0];
    if (total < 0) return fail(ctx, CS_ERR_HIP, "batched foci kernels did not report a count");
    for (int v = 0; v < n_virtual; ++v) h_n_foci[v] = ctx->h_blk_counts[1 + v];
    if (total > ctx->nb_cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", total, (long long)ctx->nb_cap);
    return CS_OK;
}

int cs_detect_foci_batch(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, const cs_kernel* kernel,
                         const cs_normxcorr2_params* params, const cs_foci_params* foci, cs_focus* h_foci, int64_t cap,
                         int64_t* h_n_foci, double* h_windows)
{
    return cs_detect_foci_batch_templates(ctx, stream_, n_blocks, signals, 1, kernel, params, foci, h_foci, cap, h_n_foci, h_windows);
}

// 2-D patterns (loops, stripes: a band of scanned diagonals) of MANY sub-matrices with one launch chain: the masked
// matrix-core tile kernel of every block appends its candidates to one list (composite keys block | row | col), then
// ONE sort, ONE float64 re-scoring, one labelling workgroup per block, one statistics pass -- instead of ~20 small
// launches and three synchronisations per block (23 blocks of a human genome: the loops pass was bound by them).
int cs_detect_foci_blocks(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, const cs_matrix* signals_f32,
                          const cs_kernel* kernel, const cs_normxcorr2_params* params, const cs_foci_params* foci, cs_focus* h_foci,
                          int64_t cap, int64_t* h_n_foci, double* h_windows)
{
    CS_ENTER(ctx);
    AllowLazy allow_lazy(ctx);
    Laps laps("detect_foci_blocks");
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks <= 0 || !signals || !kernel || !params || !foci || !h_n_foci || cap < 0 || (cap > 0 && !h_foci))
        return fail(ctx, CS_ERR_INVALID, "bad batch arguments");
    const bool prepare_only = (foci[0].reserved & 2) != 0;
    const unsigned tile_epoch = (unsigned)foci[0].reserved >> 8;          // (0: nobody waits for this call's tile launch)
    const bool was_pending = ctx->prep_pending;
    ctx->prep_pending = false;                     // (whatever a prepare form left is used by the very next call or not at all)
    constexpr int kKeyShift = 40;                 // row * ns + col < 2^40: sub-matrices of up to 2^20 bins
    bool all_narrow = true;
    long long pixels = 0;
    std::vector<long long> block_pixels((size_t)std::max(n_blocks, 1), 0);
    for (int b = 0; b < n_blocks; ++b) {
        const cs_normxcorr2_params* p = params + b;
        const cs_foci_params* fp = foci + b;
        int rc = check_foci_args(ctx, signals + b, kernel, p, fp);
        if (rc) return rc;
        if (fp->pearson != foci[0].pearson || fp->min_size != foci[0].min_size || fp->diag_only != foci[0].diag_only ||
            fp->inter != foci[0].inter || fp->want_windows != foci[0].want_windows || fp->rescore_margin != foci[0].rescore_margin)
            return fail(ctx, CS_ERR_INVALID, "the blocks of a batch share the detection parameters");
        if (p->row_end > p->row_begin) return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry takes whole blocks");
        if (fp->hi_diag < fp->lo_diag) return fail(ctx, CS_ERR_INVALID, "empty diagonal range");
        const int w = fp->hi_diag - fp->lo_diag + 1;
        all_narrow = all_narrow && is_band(signals[b].layout) && w <= 4 && fp->lo_diag >= 0;
        if (signals[b].layout == CS_LAYOUT_BAND_LAZY && !(signals_f32 && signals_f32[b].d_ptr) && !(is_band(signals[b].layout) && w <= 4 && fp->lo_diag >= 0))
            return fail(ctx, CS_ERR_INVALID, "block %d: a lazily evaluated float64 band needs its float32 twin for the tile kernel", b);
        if ((long long)p->ms * p->ns >= (1ll << kKeyShift)) return fail(ctx, CS_ERR_UNSUPPORTED, "block %d is too large for the batch keys", b);
        pixels += (long long)p->ms * std::min<long long>(w, p->ns);
        block_pixels[(size_t)b] = (long long)p->ms * std::min<long long>(w, p->ns);
    }
    if (all_narrow)
        return prepare_only ? CS_OK : cs_detect_foci_batch(ctx, stream_, n_blocks, signals, kernel, params, foci, h_foci, cap, h_n_foci, h_windows);
    if (params[0].compute_dtype != CS_F32) return fail(ctx, CS_ERR_UNSUPPORTED, "the 2-D batch runs the float32 tile kernel");
    if (kernel->km != kernel->kn || kernel->km > 17 || kernel->km < 3 || !(kernel->km & 1))
        return fail(ctx, CS_ERR_UNSUPPORTED, "the masked tile kernel takes odd square templates of 3 .. 17 (caller: block by block)");
    // results straight into page-locked caller buffers (cs_host_alloc)
    auto device_view = [&](const void* h) -> void* {
        if (!h) return nullptr;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return attr.type == hipMemoryTypeHost ? attr.devicePointer : nullptr;
    };
    cs::FocusRec* rec = cap > 0 ? reinterpret_cast<cs::FocusRec*>(device_view(h_foci)) : nullptr;
    double* win = (foci[0].want_windows && h_windows) ? reinterpret_cast<double*>(device_view(h_windows)) : nullptr;
    if ((cap > 0 && !rec) || (foci[0].want_windows && h_windows && !win))
        return fail(ctx, CS_ERR_UNSUPPORTED, "the batch entry writes into page-locked buffers (cs_host_alloc)");
    const size_t cnt_bytes = 8 * ((size_t)n_blocks + 2);
    if (cnt_bytes > ctx->h_blk_bytes) {
        if (ctx->h_blk_counts) CS_HIP(ctx, hipHostFree(ctx->h_blk_counts));
        ctx->h_blk_counts = nullptr;
        ctx->h_blk_bytes = 0;
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_blk_counts, 2 * cnt_bytes, hipHostMallocDefault));
        ctx->h_blk_bytes = 2 * cnt_bytes;
    }
    laps.lap("checks, pinned views");
    std::vector<cs::CorrArgs<double>> tab;       // the float64 argument blocks of the chain behind the tile kernels
    // ---- candidates of every block into one list
    const double margin = std::max(foci[0].rescore_margin, 1e-4);
    const double thr = foci[0].pearson - margin;
    size_t c_cap = std::max<size_t>(1 << 16, (size_t)(pixels / 256));
    // (tests of the retry and fall-back paths: a first room for the candidates / a bound for the deferred chain's launches that
    // this call's lists outgrow)
    long long seg_min = 2048;
    if (const char* t = std::getenv("CHROMOSIGHT_HIP_TEST_CAND_CAP")) {
        c_cap = (size_t)std::max(1, atoi(t));
        seg_min = 64;
    }
    const long long test_bound = std::getenv("CHROMOSIGHT_HIP_TEST_DEFER_BOUND") ? atoll(std::getenv("CHROMOSIGHT_HIP_TEST_DEFER_BOUND")) : 0;
    // what the previous call on this context saw: when it scanned the same layout (a run's steps, an iterated template's
    // passes) its candidate count sizes the LAUNCHES of the chain that is enqueued before this call's count is known (below);
    // nothing but a size is carried over
    const bool same_layout = ctx->cand_hint > 0 && ctx->cand_hint_pixels == (long long)pixels && ctx->cand_hint_blocks == n_blocks;
    const long long hint = same_layout ? ctx->cand_hint : 0;
    struct HintUpdate {                     // every successful exit records what this call saw
        cs_ctx* c;
        long long* n;
        long long px;
        int nb;
        bool on;
        bool paced;
        ~HintUpdate() { if (on) { c->cand_hint = *n; c->cand_hint_pixels = px; c->cand_hint_blocks = nb; c->cand_hint_paced = paced; } }
    };
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    long long n_total = 0;
    HintUpdate hint_update{ctx, &n_total, (long long)pixels, n_blocks, !prepare_only, same_layout && ctx->cand_hint_paced};
    // SEGMENTED candidate lists: every block appends to a region of its own with a counter of its own (room in proportion to
    // its pixels), so the chain behind the tile kernels knows the blocks' segments from n_blocks counts instead of sorting
    // all candidates by block and position -- the labelling workgroup of a block sorts its own few thousand in LDS (cs_foci.hip
    // lds_sort_pairs): no device-wide sort, no segments / split kernels.  CHROMOSIGHT_HIP_NO_SEGMENTED=1: one list, sorted on the device.
    const bool segmented_ok = n_blocks <= 60 && !std::getenv("CHROMOSIGHT_HIP_NO_SEGMENTED");
    if (segmented_ok && !ctx->h_cand_counts)
        CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_cand_counts, 256 * sizeof(long long), hipHostMallocDefault));
    // (seg_tab: the n_blocks + 1 segment starts of the compact numbering, then the n_blocks region starts: one upload)
    std::vector<long long> seg_tab(2 * (size_t)n_blocks + 2, 0), seg_cap((size_t)n_blocks, 0);
    long long* const seg_off = seg_tab.data();
    long long* const seg_base = seg_tab.data() + n_blocks + 1;
    constexpr size_t kCntBytes = 8 * 64;           // the one list's counter (and a spare word) + up to 60 blocks' own
    size_t off_tail_now = 0;                        // where the chain's scratch starts in the current layout
    bool segmented = false;
    bool pass_again = false;                        // a list outgrew its room: this pass does everything itself, whatever was prepared
    while (true) {
        if (c_cap > (size_t)INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%zu)", c_cap);
        size_t list_cap = c_cap;
        if (segmented_ok) {
            list_cap = 0;
            for (int b = 0; b < n_blocks; ++b) {
                seg_cap[(size_t)b] = std::max<long long>(seg_min, (long long)((double)c_cap * (double)block_pixels[(size_t)b] / (double)std::max<long long>(pixels, 1)) + 1);
                seg_base[(size_t)b] = (long long)list_cap;
                list_cap += (size_t)seg_cap[(size_t)b];
            }
        }
        const size_t off_cnt = al(8 * list_cap), off_tail = off_cnt + 1024;
        off_tail_now = off_tail;
        segmented = segmented_ok;
        int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, off_tail + cs::keyed_batch_scratch_bytes(n_blocks, (long long)list_cap));
        if (rc) return rc;
        char* pool = (char*)ctx->d_pool;
        unsigned long long* d_cnt = (unsigned long long*)(pool + off_cnt);       // [0] the one list's counter; [2 + b] block b's
        // (the counter is zeroed right before the tile kernels go out: on a side lane when the lanes carry the call's
        // preparations, see `early_upload` below)
        // side streams only when no block needs the (single) narrowing scratch
        bool twins = signals_f32 != nullptr;
        for (int b = 0; b < n_blocks && twins; ++b) twins = signals_f32[b].d_ptr != nullptr;
        // Lanes: the blocks' persistent launches run side by side on n_lanes streams -- a launch then walks n_lanes times as
        // many tiles per workgroup (pipeline fill / drain and the rounding to whole tiles per workgroup are paid per launch),
        // and no launch waits for slots another one holds.
        constexpr int lanes_env = 3;       // (2 .. 6 lanes measured on the 23-block genome, tools/c4_mode_sweep.sh: flat between 3 and 6)
        const int n_lanes = (twins && n_blocks > 1) ? std::min(std::min(std::max(lanes_env, 1), kBlkLanes), n_blocks) : 1;
        // one persistent launch for the tiles of all blocks (the lanes then only carry the blocks' mask tables): when the caller
        // says nothing else is queued beside it, or for a few blocks (a rank's share of a genome on 8 GPUs: 3 blocks -- one
        // launch beats three that fight for the slots: 1.40 -> 1.29 ms per rank)
        const bool table = n_lanes > 1 && (foci[0].exclusive != 0 || n_blocks <= 4);
        // either way the blocks' launches are PREPARED first (mask tables of every block on the lanes, argument blocks in a host
        // table) and the tile kernels go out afterwards: a mask-table launch queued behind persistent tile kernels would wait
        // for their workgroups to drain, and its lane's next tile kernel with it
        const bool prepared = n_lanes > 1;
        if (!prepared) CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, stream));      // (tile kernels go out block by block below)
        int table_rsym = -1;
        if (prepared) {
            const size_t need = al(cs::mfma_blocks_table_bytes(n_blocks)) + cs::mask_prep_table_bytes(n_blocks);
            if (need > ctx->tab_bytes) {
                CS_HIP(ctx, hipDeviceSynchronize());
                if (ctx->h_tab) CS_HIP(ctx, hipHostFree(ctx->h_tab));
                if (ctx->d_tab) CS_HIP(ctx, hipFree(ctx->d_tab));
                ctx->h_tab = ctx->d_tab = nullptr;
                ctx->tab_bytes = 0;
                CS_HIP(ctx, hipHostMalloc(&ctx->h_tab, 2 * need, hipHostMallocDefault));
                CS_HIP(ctx, hipMalloc(&ctx->d_tab, 2 * need));
                ctx->tab_bytes = 2 * need;
            }
            if (ctx->ws_tab.size() < (size_t)n_blocks) {
                ctx->ws_tab.resize((size_t)n_blocks, nullptr);
                ctx->ws_tab_bytes.resize((size_t)n_blocks, 0);
            }
        }
        // (a layout whose lists went to the host-paced chain last time -- a block with more candidates than the labelling
        // workgroup's LDS arrays hold -- goes there directly)
        const bool deferred = segmented && cs::keyed_batch_deferred_available() && !(same_layout && ctx->cand_hint_paced) &&
                              !std::getenv("CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN");
        if (tab.empty()) {
            // the float64 argument blocks of the chain behind the tile kernels: built (and, for the chain that is enqueued ahead
            // of the counts, uploaded) while the caller's stream is still staging the maps
            tab.resize((size_t)n_blocks);
            for (int b = 0; b < n_blocks; ++b) {
                int rc2 = build_args<double>(ctx, stream, signals + b, kernel, params + b, &tab[b]);
                if (rc2) return rc2;
            }
        }
        bool early_tables = false;
        // blocks to lanes: largest first onto the least loaded lane
        std::vector<int> lane_of((size_t)n_blocks, 0);
        if (n_lanes > 1) {
            std::vector<int> order((size_t)n_blocks);
            for (int b = 0; b < n_blocks; ++b) order[b] = b;
            auto cost = [&](int b) { return (long long)params[b].ms * (foci[b].hi_diag - foci[b].lo_diag + 1); };
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
            long long load[kBlkLanes] = {};
            for (int b : order) {
                int best = 0;
                for (int k = 1; k < n_lanes; ++k)
                    if (load[k] < load[best]) best = k;
                lane_of[b] = best;
                load[best] += cost(b);
            }
        }
        if (n_lanes > 1) {
            for (int k = 0; k < n_lanes - 1; ++k)
                if (!ctx->s_blk[k]) {
                    // (the lanes carry the short mask-table launches beside the caller's staging kernels, which fill every
                    // wave slot of the chip: served first, they are done when the staging is)
                    int lo_p = 0, hi_p = 0;
                    CS_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
                    CS_HIP(ctx, hipStreamCreateWithPriority(&ctx->s_blk[k], hipStreamNonBlocking, hi_p));
                }
            for (int k = 0; k < kBlkLanes; ++k)
                if (!ctx->ev_blk[k]) CS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_blk[k], hipEventDisableTiming));
            // template weights / matrix-core fragments are uploaded on the caller's stream (once per template): before the
            // side streams are released
            const long long uploads_before = ctx->uploads;
            {
                cs::CorrArgs<float> A0;
                int rc0 = build_args<float>(ctx, stream, signals_f32, kernel, params, &A0);
                if (rc0) return rc0;
                cs::MfmaWeights E0;
                rc0 = ensure_wfrag(ctx, stream, A0.km, A0.kn, &E0);
                if (rc0) return rc0;
            }
            // The mask tables depend on the bins' flags and the template only -- not on the maps, which the caller's stream
            // may still be staging (a genome step enqueues this call right behind cs_stage_blocks): with the launches
            // prepared first, every block's tables are built on the SIDE lanes at once, beside whatever the caller's stream is
            // doing, and the lanes wait for that stream only when this call uploaded the template.  (The tables' scratch is
            // free: the previous call's tile kernels were synchronised before it returned.)
            early_tables = prepared && !std::getenv("CHROMOSIGHT_HIP_NO_EARLY_TABLES");
            // (without early tables the lanes' only link to this stream is the event below: the counter is zeroed before it)
            if (prepared && !early_tables) CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, stream));
            if (!early_tables || ctx->uploads != uploads_before) {
                CS_HIP(ctx, hipEventRecord(ctx->ev_blk[kBlkLanes - 1], stream));  // the counter is zero, earlier work is done
                for (int k = 0; k < n_lanes - 1; ++k) CS_HIP(ctx, hipStreamWaitEvent(ctx->s_blk[k], ctx->ev_blk[kBlkLanes - 1], 0));
            }
        }
        // The PREPARE form (foci[0].reserved & 2): everything of this call that does not depend on the maps -- the blocks' mask
        // tables, the zeroed counters, the tile kernels' argument table and the chain's -- is enqueued on the side lanes NOW,
        // and the call returns; the same call without the flag, made next on this context, finds it done.  A genome step makes
        // the prepare form BEFORE cs_stage_blocks: the lanes' work is then long finished when the staging is, and the tile
        // launch behind the staging waits for events that have fired -- a wait for an event that fires later costs the queue
        // 20-40 us (profiles/r05_rank_share_timeline.txt: staging done at 76 us, lanes at 93, tile kernel at 134).
        const bool can_split = table && prepared && early_tables && deferred;
        unsigned long long key = 1469598103934665603ull;
        {
            auto mix = [&](const void* p, size_t n) {
                const unsigned char* q = (const unsigned char*)p;
                for (size_t k = 0; k < n; ++k) key = (key ^ q[k]) * 1099511628211ull;
            };
            mix(params, sizeof(cs_normxcorr2_params) * (size_t)n_blocks);
            for (int b = 0; b < n_blocks; ++b) {
                cs_foci_params f = foci[b];
                f.reserved = 0;
                mix(&f, sizeof(f));
            }
            mix(signals, sizeof(cs_matrix) * (size_t)n_blocks);
            if (signals_f32) mix(signals_f32, sizeof(cs_matrix) * (size_t)n_blocks);
            mix(kernel, sizeof(cs_kernel));
            const long long extra[4] = {ctx->uploads, (long long)c_cap, (long long)n_blocks, (long long)(uintptr_t)ctx->d_pool};
            mix(extra, sizeof(extra));
        }
        if (prepare_only && !can_split) return CS_OK;
        const bool reuse = !prepare_only && was_pending && can_split && key == ctx->prep_key && !pass_again;
        struct SkipLaunch {
            cs_ctx* c;
            SkipLaunch(cs_ctx* c_, bool on) : c(c_) { c->skip_prep_launch = on; }
            ~SkipLaunch() { c->skip_prep_launch = false; }
        } skip_launch(ctx, reuse);
        // (one persistent tile launch: ONE side lane carries all of it -- every event the launch waits for costs the caller's
        // queue ~ 7 us between the staging and the tile kernel, and the lane's work is off the critical path)
        const int side_lanes = (table && early_tables) ? 1 : n_lanes - 1;
        laps.lap("pool, weights, events");
        // One tile launch + the lanes busy with the mask tables while the caller's stream is still staging the maps: the zeroed
        // counter (first) and the argument table (behind the tables of its lane) travel on the LAST side lane, which carries the
        // fewest tables -- ordered before the launch by the lanes' events below -- instead of sitting between the staging and
        // the tile kernel
        const bool early_upload = table && early_tables && n_lanes > 1;
        if (early_upload && !reuse) CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, ctx->s_blk[side_lanes - 1]));
        // (largest blocks first on every lane: the short ones fill the end)
        std::vector<int> launch_order((size_t)n_blocks);
        for (int b = 0; b < n_blocks; ++b) launch_order[b] = b;
        if (n_lanes > 1)
            std::stable_sort(launch_order.begin(), launch_order.end(), [&](int x, int y) {
                return (long long)params[x].ms * (foci[x].hi_diag - foci[x].lo_diag + 1) > (long long)params[y].ms * (foci[y].hi_diag - foci[y].lo_diag + 1);
            });
        // one tile launch: the mask tables of all blocks from ONE launch on the side lane (collected in the loop, launched
        // behind it) -- a launch per block takes 30-90 us each beside a genome's staging kernels
        const bool one_prep = table && early_tables;
        std::vector<cs::MaskPrepArgs<float>> prep_list;
        struct PrepCollect {
            cs_ctx* c;
            PrepCollect(cs_ctx* c_, std::vector<cs::MaskPrepArgs<float>>* v) : c(c_)
            {
                c->prep_collect = v;
                c->prep_groups.clear();
            }
            ~PrepCollect() { c->prep_collect = nullptr; }
        } prep_collect(ctx, one_prep ? &prep_list : nullptr);
        int next_side = 0;
        for (int b : launch_order) {
            // (early tables: on the side lanes only, dealt round-robin in launch order -- largest first)
            const int lane = early_tables ? 1 + (next_side++) % side_lanes : lane_of[b];
            hipStream_t stream = lane == 0 ? (hipStream_t)stream_ : ctx->s_blk[lane - 1];
            struct WsSwap {                   // the side lanes build their mask tables in their own scratch
                cs_ctx* c;
                int k;
                WsSwap(cs_ctx* c_, int k_) : c(c_), k(k_) { swap(); }
                ~WsSwap() { swap(); }
                void swap()
                {
                    if (k < 0) return;
                    std::swap(c->d_ws, c->ws_alt[k]);
                    std::swap(c->d_ws_bytes, c->ws_alt_bytes[k]);
                }
            } ws_swap(ctx, prepared ? -1 : lane - 1);
            struct TabSwap {                  // table mode: every block keeps its own mask tables until the one launch is done
                cs_ctx* c;
                int b;
                TabSwap(cs_ctx* c_, int b_) : c(c_), b(b_) { swap(); }
                ~TabSwap() { swap(); }
                void swap()
                {
                    if (b < 0) return;
                    std::swap(c->d_ws, c->ws_tab[(size_t)b]);
                    std::swap(c->d_ws_bytes, c->ws_tab_bytes[(size_t)b]);
                }
            } tab_swap(ctx, prepared ? b : -1);
            const cs_matrix* sig = (signals_f32 && signals_f32[b].d_ptr) ? signals_f32 + b : signals + b;
            cs_matrix map;                        // geometry of the (virtual) coefficient map: the scanned diagonals
            map.d_ptr = nullptr;
            map.dtype = CS_F32;
            map.layout = is_band(signals[b].layout) ? CS_LAYOUT_BAND : signals[b].layout;
            map.row0 = 0;
            if (map.layout == CS_LAYOUT_BAND) {
                map.band_lo = foci[b].lo_diag;
                map.band_w = foci[b].hi_diag - foci[b].lo_diag + 1;
                map.ld = ((int64_t)map.band_w + 63) / 64 * 64;
            } else {
                map.band_lo = map.band_w = 0;
                map.ld = ((int64_t)params[b].ns + 15) / 16 * 16;
            }
            CandSink sink{(unsigned long long*)pool, d_cnt, (long long)c_cap, (unsigned long long)b << kKeyShift, foci[b].lo_diag,
                          foci[b].hi_diag};
            if (segmented) {
                sink.keys = (unsigned long long*)pool + seg_base[(size_t)b];
                sink.count = d_cnt + 2 + b;
                sink.cap = seg_cap[(size_t)b];
            }
            int rsym = 0;
            if (prepared) {
                sink.defer_args = (char*)ctx->h_tab + cs::mfma_blocks_arg_offset(n_blocks) + (size_t)b * cs::mfma_blocks_arg_bytes();
                sink.defer_rsym = &rsym;
            }
            rc = corr_candidates_f32(ctx, stream, sig, kernel, params + b, &map, margin, thr, &sink);
            if (!rc && prepared) {
                if (!ctx->cand_fused) rc = CS_NEED_MAP;             // another kernel than the masked tile kernel took the block
                else if (table_rsym >= 0 && table_rsym != rsym) rc = fail(ctx, CS_ERR_UNSUPPORTED, "blocks need different tile kernels");
                table_rsym = rsym;
            }
            if (rc) {
                if (n_lanes > 1) (void)hipDeviceSynchronize();      // nothing of this call may still be running on a side stream
                if (rc == CS_NEED_MAP) return fail(ctx, CS_ERR_UNSUPPORTED, "block %d is not served by the masked tile kernel", b);
                return rc;
            }
        }
        if (one_prep) {
            ctx->prep_collect = nullptr;
            if (!reuse) {
                const size_t tile_tab_bytes = al(cs::mfma_blocks_table_bytes(n_blocks));
                int rcp = cs::launch_mask_prep_batch(prep_list.data(), ctx->prep_groups.data(), (int)prep_list.size(), (char*)ctx->h_tab + tile_tab_bytes,
                                                     (char*)ctx->d_tab + tile_tab_bytes, ctx->s_blk[0]);
                if (rcp) return fail(ctx, CS_ERR_HIP, "mask table kernel failed: %s", hipGetErrorString((hipError_t)rcp));
            }
        }
        laps.lap("mask tables + arguments");
        bool tab_uploaded = false;
        if (early_upload) {
            // (reuse: both tables were uploaded by the prepare form; the host-side table is still filled in -- the launch reads
            // its block count and tile ranges from it)
            rc = cs::launch_corr_mfma_blocks(ctx->h_tab, ctx->d_tab, n_blocks, table_rsym, ctx->n_cu, stream, ctx->s_blk[side_lanes - 1], !reuse, false);
            if (rc) return fail(ctx, CS_ERR_HIP, "tile kernel table upload failed: %s", hipGetErrorString((hipError_t)rc));
            if (deferred) {
                if (!reuse) rc = cs::upload_keyed_batch_table(tab.data(), n_blocks, (long long)list_cap, pool + off_tail, ctx->s_blk[side_lanes - 1]);
                if (rc) return fail(ctx, CS_ERR_HIP, "argument table upload failed: %s", hipGetErrorString((hipError_t)rc));
                tab_uploaded = true;
            }
            if (prepare_only) {
                ctx->prep_tab_keep.swap(tab);           // (the source of the asynchronous upload above outlives this call)
                ctx->prep_key = key;
                ctx->prep_pending = true;
                return CS_OK;
            }
        } else if (prepared && early_tables) {
            CS_HIP(ctx, hipMemsetAsync(d_cnt, 0, kCntBytes, stream));      // (before the event the lanes' tile kernels wait for, below)
        }
        if (early_tables) {
            // the tile kernels read the maps and the zeroed counter: behind the caller's stream.  A block whose tables were
            // built on another lane than the one that launches its tiles: every lane waits for every lane's tables.
            if (!table) {
                CS_HIP(ctx, hipEventRecord(ctx->ev_blk[kBlkLanes - 1], stream));
                for (int k = 0; k < n_lanes - 1; ++k) CS_HIP(ctx, hipEventRecord(ctx->ev_blk[k], ctx->s_blk[k]));
                for (int k = 0; k < n_lanes - 1; ++k) {
                    CS_HIP(ctx, hipStreamWaitEvent(ctx->s_blk[k], ctx->ev_blk[kBlkLanes - 1], 0));
                    for (int j = 0; j < n_lanes - 1; ++j)
                        if (j != k) CS_HIP(ctx, hipStreamWaitEvent(ctx->s_blk[k], ctx->ev_blk[j], 0));
                    CS_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_blk[k], 0));
                }
            }
        }
        if (prepared && !table) {
            // one persistent launch per block, each on the lane that built its mask tables
            for (int b : launch_order) {
                hipStream_t s_lane = lane_of[b] == 0 ? stream : ctx->s_blk[lane_of[b] - 1];
                const void* arg = (const char*)ctx->h_tab + cs::mfma_blocks_arg_offset(n_blocks) + (size_t)b * cs::mfma_blocks_arg_bytes();
                rc = cs::launch_corr_mfma_prepared(arg, table_rsym, ctx->n_cu, 0, s_lane);
                if (rc) {
                    (void)hipDeviceSynchronize();
                    return fail(ctx, CS_ERR_HIP, "tile kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
                }
            }
        }
        if (n_lanes > 1) {
            for (int k = 0; k < (table && early_tables ? side_lanes : n_lanes - 1); ++k) {
                CS_HIP(ctx, hipEventRecord(ctx->ev_blk[k], ctx->s_blk[k]));
                CS_HIP(ctx, hipStreamWaitEvent(stream, ctx->ev_blk[k], 0));
            }
        }
        if (table) {
            rc = cs::launch_corr_mfma_blocks(ctx->h_tab, ctx->d_tab, n_blocks, table_rsym, ctx->n_cu, stream, stream, !early_upload, true,
                                             tile_epoch ? ctx->d_tiles_started : nullptr, tile_epoch);
            if (rc) return fail(ctx, CS_ERR_HIP, "tile kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        } else if (tile_epoch) {
            // (the per-block launches carry no start word: whoever waits for this epoch -- cs_stream_wait_tiles -- is let go here)
            CS_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->d_tiles_started, (int)tile_epoch, 1, stream));
        }
        laps.lap("tile kernels launched");
        // ---- The chain behind the tile kernels, enqueued BEFORE they have finished (segmented lists): the blocks' candidate
        // counts stay on the device -- one tiny kernel turns them into the segments of the compact numbering --, the launches
        // are sized for a bound (the previous call's count on the same layout + 50 %; without one, a share of the lists'
        // room), and the host reads counts, status and records after ONE synchronisation.  What the round trip in the middle
        // cost a rank's share of a genome: the wake-up, ~ 30 us of enqueueing with the device idle, of a 160 us tail
        // (profiles/r05_rank_share_timeline.txt).  A list that outgrew its room sends the call round again as before; lists too
        // long for the labelling workgroups' LDS arrays, or more candidates than the bound, take the host-paced chain below
        // on the same lists.  CHROMOSIGHT_HIP_NO_DEFERRED_CHAIN=1: always the host-paced chain.
        bool counts_known = false;
        if (deferred) {
            long long bound = std::min<long long>((long long)list_cap, hint > 0 ? hint + hint / 2 + 4096
                                                                               : std::max<long long>(16384, (long long)list_cap / 8));
            if (test_bound > 0) bound = std::min(bound, test_bound);
            cs::DeferredSegments D;
            D.d_counts = (const long long*)(d_cnt + 2);
            for (int b = 0; b < n_blocks; ++b) {                 // (the device reads the two tables where they are: page-locked)
                ctx->h_cand_counts[64 + b] = seg_base[(size_t)b];
                ctx->h_cand_counts[128 + b] = seg_cap[(size_t)b];
            }
            D.h_base = ctx->h_cand_counts + 64;
            D.h_cap = ctx->h_cand_counts + 128;
            D.bound = bound;
            D.tab_uploaded = tab_uploaded;
            D.h_counts_out = ctx->h_cand_counts;             // [0, n_blocks): the blocks' counts; [60], [61]: total, status flags
            ctx->h_blk_counts[0] = -1;
            ctx->h_cand_counts[61] = -1;
            int rc1 = cs::enqueue_foci_keyed_batch(tab.data(), n_blocks, (const long long*)pool, (long long)list_cap, kKeyShift, foci[0].pearson,
                                                   foci[0].min_size, diag_code(foci, kernel, params), foci[0].inter, pool + off_tail, rec,
                                                   (long long)cap, win, win ? (long long)cap : 0, ctx->h_blk_counts, stream, nullptr, nullptr, &D);
            if (rc1) return fail(ctx, CS_ERR_HIP, "batched foci kernels failed: %s", hipGetErrorString((hipError_t)rc1));
            laps.lap("foci chain enqueued");
            CS_HIP(ctx, hipStreamSynchronize(stream));
            laps.lap("wait: records");
            counts_known = true;
        } else if (segmented) {
            CS_HIP(ctx, hipMemcpyAsync(ctx->h_cand_counts, d_cnt + 2, 8 * (size_t)n_blocks, hipMemcpyDeviceToHost, stream));
        } else {
            CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts, d_cnt, 8, hipMemcpyDeviceToHost, stream));
        }
        if (!counts_known) {
            CS_HIP(ctx, hipStreamSynchronize(stream));
            laps.lap("wait: candidates");
        }
        if (segmented) {
            // the blocks' own counts: segments of the compact numbering; a block that outgrew its room sends the call round again
            bool fits = true;
            long long sum = 0, worst = 1;
            for (int b = 0; b < n_blocks; ++b) {
                const long long nb_ = ctx->h_cand_counts[b];
                seg_off[(size_t)b] = sum;
                sum += std::min(nb_, seg_cap[(size_t)b]);
                if (nb_ > seg_cap[(size_t)b]) {
                    fits = false;
                    // (room in proportion to the pixels: the total that would have given this block enough)
                    worst = std::max(worst, (long long)((double)nb_ * (double)std::max<long long>(pixels, 1) / (double)std::max<long long>(block_pixels[(size_t)b], 1)) + 1);
                }
            }
            seg_off[(size_t)n_blocks] = sum;
            n_total = sum;
            if (!fits) {
                c_cap = (size_t)worst + (size_t)worst / 8;
                pass_again = true;
                continue;
            }
            if (counts_known && ctx->h_cand_counts[61] == 0) {
                // the chain ran on this call's lists as they are: done
                const long long total = ctx->h_blk_counts[0];
                if (total < 0) return fail(ctx, CS_ERR_HIP, "batched foci kernels did not report a count");
                for (int b = 0; b < n_blocks; ++b) h_n_foci[b] = ctx->h_blk_counts[1 + b];
                if (total > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", total, (long long)cap);
                return CS_OK;
            }
            if (counts_known) hint_update.paced = (ctx->h_cand_counts[61] & 2) != 0;
            break;
        }
        n_total = ctx->h_counts[0];
        if ((size_t)n_total <= c_cap) break;
        c_cap = (size_t)n_total + (size_t)n_total / 8;
        pass_again = true;
    }
    // ---- exact scores, foci, statistics: one chain for all blocks (its argument table was built while the tile kernels ran),
    // sized by the counts the host has read
    char* pool = (char*)ctx->d_pool;
    const size_t off_tail = off_tail_now;
    ctx->h_blk_counts[0] = -1;
    int rc = cs::enqueue_foci_keyed_batch(tab.data(), n_blocks, (const long long*)pool, n_total, kKeyShift, foci[0].pearson,
                                          foci[0].min_size, diag_code(foci, kernel, params), foci[0].inter, pool + off_tail, rec,
                                          (long long)cap, win, win ? (long long)cap : 0, ctx->h_blk_counts, stream,
                                          segmented ? seg_base : nullptr, segmented ? seg_off : nullptr, nullptr);
    if (rc) return fail(ctx, CS_ERR_HIP, "batched foci kernels failed: %s", hipGetErrorString((hipError_t)rc));
    laps.lap("foci chain enqueued");
    CS_HIP(ctx, hipStreamSynchronize(stream));            // also: the host table above was consumed
    laps.lap("wait: records");
    const long long total = ctx->h_blk_counts[0];
    if (total < 0) return fail(ctx, CS_ERR_HIP, "batched foci kernels did not report a count");
    for (int b = 0; b < n_blocks; ++b) h_n_foci[b] = ctx->h_blk_counts[1 + b];
    if (total > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", total, (long long)cap);
    return CS_OK;
}

int cs_candidates(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                  const cs_normxcorr2_params* p, const cs_foci_params* fp, int32_t* h_rows, int32_t* h_cols,
                  double* h_vals, int64_t cap, int64_t* n_out)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_foci_args(ctx, signal, kernel, p, fp);
    if (rc) return rc;
    if (!n_out || cap < 0 || (cap > 0 && (!h_rows || !h_cols || !h_vals))) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    *n_out = 0;
    CandPlan P;
    rc = find_candidates(ctx, stream_, signal, kernel, p, fp, 0, cs::keep_scratch_bytes, &P);
    if (rc) return rc;
    if (P.n_cand == 0) return CS_OK;
    cs::CorrArgs<double> A64;
    rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    int *d_rows = nullptr, *d_cols = nullptr, *d_n = nullptr;
    double* d_vals = nullptr;
    rc = cs::enqueue_keep(A64, (const int*)pool, (const int*)(pool + P.off_cols), P.n_cand, fp->pearson, pool + P.off_tail,
                          &d_rows, &d_cols, &d_vals, &d_n, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "candidate kernels failed: %s", hipGetErrorString((hipError_t)rc));
    int* h_n = reinterpret_cast<int*>(ctx->h_counts + 2);
    CS_HIP(ctx, hipMemcpyAsync(h_n, d_n, 4, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    const long long n = *h_n;
    *n_out = n;
    if (n > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld candidates, room for %lld", n, (long long)cap);
    if (n > 0) {
        CS_HIP(ctx, hipMemcpyAsync(h_rows, d_rows, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_cols, d_cols, 4 * (size_t)n, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_vals, d_vals, 8 * (size_t)n, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
    }
    return CS_OK;
}

int cs_label_foci(cs_ctx* ctx, void* stream_, int32_t ms, int32_t ns, const int32_t* h_rows, const int32_t* h_cols,
                  const double* h_vals, int64_t n, int32_t min_size, int32_t diag_only, int32_t* h_foci_rows,
                  int32_t* h_foci_cols, int32_t* h_foci_size, int64_t cap, int64_t* n_foci)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (!n_foci || n < 0 || ms <= 0 || ns <= 0 || min_size < 1 || (n > 0 && (!h_rows || !h_cols || !h_vals)))
        return fail(ctx, CS_ERR_INVALID, "bad candidate list");
    if (cap < 0 || (cap > 0 && (!h_foci_rows || !h_foci_cols || !h_foci_size))) return fail(ctx, CS_ERR_INVALID, "bad output buffers");
    if (n > INT32_MAX / 2) return fail(ctx, CS_ERR_OVERFLOW, "too many candidate pixels (%lld)", (long long)n);
    *n_foci = 0;
    if (n == 0) return CS_OK;
    if (!ctx->h_counts) CS_HIP(ctx, hipHostMalloc((void**)&ctx->h_counts, 64, hipHostMallocDefault));
    for (int64_t t = 0; t < n; ++t)
        if (h_rows[t] < 0 || h_rows[t] >= ms || h_cols[t] < 0 || h_cols[t] >= ns)
            return fail(ctx, CS_ERR_INVALID, "candidate %lld outside the matrix", (long long)t);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t un = (size_t)n;
    const size_t off_cols = al(4 * un), off_vals = off_cols + al(4 * un), off_cnt = off_vals + al(8 * un),
                 off_tail = off_cnt + 256;
    int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, off_tail + cs::label_scratch_bytes(n));
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    CS_HIP(ctx, hipMemcpyAsync(pool, h_rows, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_cols, h_cols, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_vals, h_vals, 8 * un, hipMemcpyHostToDevice, stream));
    long long* d_cnt = (long long*)(pool + off_cnt);
    int *f_rows = nullptr, *f_cols = nullptr, *f_size = nullptr;
    rc = cs::enqueue_label((const int*)pool, (const int*)(pool + off_cols), (const double*)(pool + off_vals), n, ns, min_size,
                           diag_only, pool + off_tail, &f_rows, &f_cols, &f_size, d_cnt, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "labelling kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(ctx->h_counts + 1, d_cnt, 8, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    const long long k = ctx->h_counts[1];
    *n_foci = k;
    if (k > cap) return fail(ctx, CS_ERR_OVERFLOW, "%lld foci, room for %lld", k, (long long)cap);
    if (k > 0) {
        CS_HIP(ctx, hipMemcpyAsync(h_foci_rows, f_rows, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_foci_cols, f_cols, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipMemcpyAsync(h_foci_size, f_size, 4 * (size_t)k, hipMemcpyDeviceToHost, stream));
        CS_HIP(ctx, hipStreamSynchronize(stream));
    }
    return CS_OK;
}

int cs_quantify_pixels(cs_ctx* ctx, void* stream_, const cs_matrix* signal, const cs_kernel* kernel,
                       const cs_normxcorr2_params* p, const cs_foci_params* fp, const int32_t* h_rows,
                       const int32_t* h_cols, int64_t n, cs_focus* h_out, double* h_windows)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    int rc = check_foci_args(ctx, signal, kernel, p, fp);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!h_rows || !h_cols || !h_out))) return fail(ctx, CS_ERR_INVALID, "bad pixel list");
    if (n == 0) return CS_OK;
    const int kk = kernel->km * kernel->kn;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t un = (size_t)n;
    const size_t off_cols = al(4 * un), off_score = off_cols + al(4 * un), off_nobs = off_score + al(8 * un),
                 off_rec = off_nobs + al(8 * un), off_win = off_rec + al(sizeof(cs::FocusRec) * un),
                 total = off_win + (fp->want_windows ? al(8 * un * kk) : 0);
    rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, total);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    CS_HIP(ctx, hipMemcpyAsync(pool, h_rows, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_cols, h_cols, 4 * un, hipMemcpyHostToDevice, stream));
    cs::CorrArgs<double> A64;
    rc = build_args<double>(ctx, stream, signal, kernel, p, &A64);
    if (rc) return rc;
    double* d_win = fp->want_windows ? (double*)(pool + off_win) : nullptr;
    rc = cs::enqueue_quantify(A64, (const int*)pool, (const int*)(pool + off_cols), n, fp->inter, (double*)(pool + off_score),
                              (double*)(pool + off_nobs), (cs::FocusRec*)(pool + off_rec), d_win, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "quantify kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(h_out, pool + off_rec, sizeof(cs_focus) * un, hipMemcpyDeviceToHost, stream));
    if (d_win && h_windows) CS_HIP(ctx, hipMemcpyAsync(h_windows, d_win, 8 * un * kk, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));
    return CS_OK;
}

// quantify mode over the sub-matrices of a genome in ONE launch chain (one call per template instead of one per
// sub-matrix and template: cli/chromosight.py:229-260 scores one sub-matrix per task)
int cs_quantify_blocks(cs_ctx* ctx, void* stream_, int32_t n_blocks, const cs_matrix* signals, const cs_kernel* kernel,
                       const cs_normxcorr2_params* params, const cs_foci_params* foci, const int32_t* h_blk, const int32_t* h_rows,
                       const int32_t* h_cols, int64_t n, cs_focus* h_out, double* h_windows)
{
    CS_ENTER(ctx);
    hipStream_t stream = (hipStream_t)stream_;
    if (n_blocks <= 0 || !signals || !kernel || !params || !foci) return fail(ctx, CS_ERR_INVALID, "bad batch arguments");
    if (n < 0 || (n > 0 && (!h_blk || !h_rows || !h_cols || !h_out))) return fail(ctx, CS_ERR_INVALID, "bad pixel list");
    for (int b = 0; b < n_blocks; ++b) {
        int rc = check_foci_args(ctx, signals + b, kernel, params + b, foci + b);
        if (rc) return rc;
        if (params[b].compute_dtype != CS_F64) return fail(ctx, CS_ERR_UNSUPPORTED, "quantify scores in float64");
    }
    for (int64_t t = 0; t < n; ++t)
        if (h_blk[t] < 0 || h_blk[t] >= n_blocks) return fail(ctx, CS_ERR_INVALID, "pixel %lld names sub-matrix %d of %d", (long long)t, h_blk[t], n_blocks);
    if (n == 0) return CS_OK;
    const bool want_windows = foci[0].want_windows && h_windows;
    const int kk = kernel->km * kernel->kn;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t un = (size_t)n, nb = (size_t)n_blocks;
    const size_t off_rows = al(4 * un), off_cols = off_rows + al(4 * un), off_inter = off_cols + al(4 * un),
                 off_tab = off_inter + al(4 * nb), off_score = off_tab + al(sizeof(cs::CorrArgs<double>) * nb),
                 off_nobs = off_score + al(8 * un), off_rec = off_nobs + al(8 * un), off_win = off_rec + al(sizeof(cs::FocusRec) * un),
                 total = off_win + (want_windows ? al(8 * un * kk) : 0);
    int rc = ensure_scratch(ctx, &ctx->d_pool, &ctx->d_pool_bytes, total);
    if (rc) return rc;
    char* pool = (char*)ctx->d_pool;
    std::vector<cs::CorrArgs<double>> tab(nb);
    std::vector<int> inter(nb);
    for (int b = 0; b < n_blocks; ++b) {
        rc = build_args<double>(ctx, stream, signals + b, kernel, params + b, &tab[(size_t)b]);
        if (rc) return rc;
        inter[(size_t)b] = foci[b].inter;
    }
    CS_HIP(ctx, hipMemcpyAsync(pool, h_blk, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_rows, h_rows, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_cols, h_cols, 4 * un, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_inter, inter.data(), 4 * nb, hipMemcpyHostToDevice, stream));
    CS_HIP(ctx, hipMemcpyAsync(pool + off_tab, tab.data(), sizeof(cs::CorrArgs<double>) * nb, hipMemcpyHostToDevice, stream));
    double* d_win = want_windows ? (double*)(pool + off_win) : nullptr;
    rc = cs::enqueue_quantify_batch((const cs::CorrArgs<double>*)(pool + off_tab), (const int*)(pool + off_inter), (const int*)pool,
                                    (const int*)(pool + off_rows), (const int*)(pool + off_cols), n, (double*)(pool + off_score),
                                    (double*)(pool + off_nobs), (cs::FocusRec*)(pool + off_rec), d_win, stream);
    if (rc) return fail(ctx, CS_ERR_HIP, "quantify kernels failed: %s", hipGetErrorString((hipError_t)rc));
    CS_HIP(ctx, hipMemcpyAsync(h_out, pool + off_rec, sizeof(cs_focus) * un, hipMemcpyDeviceToHost, stream));
    if (d_win) CS_HIP(ctx, hipMemcpyAsync(h_windows, d_win, 8 * un * kk, hipMemcpyDeviceToHost, stream));
    CS_HIP(ctx, hipStreamSynchronize(stream));       // (the pageable tables above were consumed)
    return CS_OK;
}

}  // extern "C"

// ---- cs_run_calls: a list of the library's own calls, natively (see the header) ---------------------------------------
namespace {
struct CallRun {
    cs_call* calls = nullptr;
    int n = 0;
    std::atomic<int>* done = nullptr;      // per call: 1 once it has returned (or was skipped)
    bool timing = false;
    std::chrono::steady_clock::time_point t0;
};

int dispatch_call(cs_call& c)
{
    void** p = c.p;
    const int64_t* i = c.i;
    switch (c.fn) {
        case CS_CALL_STAGE_BLOCKS:
            return cs_stage_blocks((cs_ctx*)p[0], p[1], (const cs_csr*)p[2], (const cs_stage_block*)p[3], (int32_t)i[0], c.d[0]);
        case CS_CALL_EVENT_RECORD:
            return cs_event_record((cs_ctx*)p[0], p[1], p[2]);
        case CS_CALL_STREAM_WAIT_EVENT:
            return cs_stream_wait_event((cs_ctx*)p[0], p[1], p[2]);
        case CS_CALL_DETECT_FOCI_BLOCKS:
            return cs_detect_foci_blocks((cs_ctx*)p[0], p[1], (int32_t)i[0], (const cs_matrix*)p[2], (const cs_matrix*)p[3],
                                         (const cs_kernel*)p[4], (const cs_normxcorr2_params*)p[5], (const cs_foci_params*)p[6],
                                         (cs_focus*)p[7], i[1], (int64_t*)p[8], (double*)p[9]);
        case CS_CALL_DETECT_FOCI_BATCH_TEMPLATES:
            return cs_detect_foci_batch_templates((cs_ctx*)p[0], p[1], (int32_t)i[0], (const cs_matrix*)p[2], (int32_t)i[1],
                                                  (const cs_kernel*)p[3], (const cs_normxcorr2_params*)p[4],
                                                  (const cs_foci_params*)p[5], (cs_focus*)p[6], i[2], (int64_t*)p[7], (double*)p[8]);
        case CS_CALL_ACCEPT_RECORDS:
            return cs_accept_records((const cs_focus*)p[0], i[0], (const int64_t*)p[1], (const int32_t*)p[2], (const int32_t*)p[3],
                                     (const int32_t*)p[4], (int32_t)i[1], (int32_t)i[2], (int32_t)i[3], c.d[0], c.d[1], (int32_t)i[4],
                                     (int32_t)i[5], (double*)p[5], (uint8_t*)p[6], (int64_t*)p[7]);
        case CS_CALL_DETECT_FOCI_BATCH_FINISH:
            return cs_detect_foci_batch_finish((cs_ctx*)p[0], p[1], (int64_t*)p[2]);
        case CS_CALL_STREAM_WAIT_TILES:
            return cs_stream_wait_tiles((cs_ctx*)p[0], p[1], (cs_ctx*)p[2], (int32_t)i[0], (int32_t)i[1]);
        default:
            return CS_ERR_INVALID;
    }
}

void run_lane(const CallRun& R, int lane)
{
    bool failed = false;
    for (int k = 0; k < R.n; ++k) {
        cs_call& c = R.calls[k];
        if (c.lane != lane) continue;
        if (!failed && c.after >= 0 && c.after < R.n) {
            int spins = 0;
            while (R.done[c.after].load(std::memory_order_acquire) == 0)
                if (++spins > 2000) std::this_thread::yield();
            if (R.calls[c.after].rc != 0) failed = true;            // what it waited for did not happen
        }
        const auto t_begin = std::chrono::steady_clock::now();
        if (failed && c.fn == CS_CALL_DETECT_FOCI_BATCH_FINISH) {
            // a lane that failed between the asynchronous batch and its finish must not leave the context "pending" (every later
            // foci call on it would be refused): end the batch whatever it holds; "nothing pending" is as good
            (void)dispatch_call(c);
        }
        c.rc = failed ? CS_ERR_INVALID : dispatch_call(c);
        if (R.timing)       // CHROMOSIGHT_HIP_TIMING: the host timeline of the list (lane, entry, begin and end since the list began)
            fprintf(stderr, "[timing] run_calls: lane %d call %2d fn %d  %7.1f -> %7.1f us\n", lane, k, c.fn,
                    std::chrono::duration<double, std::micro>(t_begin - R.t0).count(),
                    std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - R.t0).count());
        if (c.rc != 0) failed = true;
        R.done[k].store(1, std::memory_order_release);
    }
}

// worker threads of the extra lanes: kept between calls (starting a thread costs more than a lane's host work), spinning
// briefly after a job -- the next step of a loop is usually microseconds away -- before they sleep
struct LaneWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> state{0};             // 0 idle, 1 job posted, 2 job done
    const CallRun* job = nullptr;
    int lane = 0;
    bool quit = false;
    void loop()
    {
        for (;;) {
            int spins = 0;
            while (state.load(std::memory_order_acquire) != 1) {
                if (++spins < 20000) continue;
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return state.load(std::memory_order_acquire) == 1 || quit; });
                if (quit) return;
            }
            run_lane(*job, lane);
            state.store(2, std::memory_order_release);
        }
    }
};

std::mutex g_run_mu;                       // one cs_run_calls at a time (the workers are shared)
// (never destroyed: the detached workers may be waiting on their condition variables when the process exits)
std::vector<LaneWorker*>& lane_workers()
{
    static std::vector<LaneWorker*>* v = new std::vector<LaneWorker*>();
    return *v;
}
}  // namespace

extern "C" int cs_run_calls(cs_call* calls, int32_t n_calls)
{
    if (n_calls < 0 || (n_calls > 0 && !calls)) return CS_ERR_INVALID;
    if (n_calls == 0) return CS_OK;
    int lanes = 1;
    for (int k = 0; k < n_calls; ++k) {
        if (calls[k].lane < 0 || calls[k].lane > 7 || calls[k].after >= k) return CS_ERR_INVALID;     // (waits only look back)
        lanes = std::max(lanes, calls[k].lane + 1);
        calls[k].rc = 0;
    }
    std::lock_guard<std::mutex> lock(g_run_mu);
    std::vector<std::atomic<int>> done((size_t)n_calls);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    CallRun R;
    R.calls = calls;
    R.n = n_calls;
    R.done = done.data();
    R.timing = std::getenv("CHROMOSIGHT_HIP_TIMING") != nullptr;
    R.t0 = std::chrono::steady_clock::now();
    std::vector<LaneWorker*>& g_workers = lane_workers();
    while ((int)g_workers.size() < lanes - 1) {
        LaneWorker* w = new LaneWorker;
        g_workers.push_back(w);
        w->th = std::thread([w] { w->loop(); });
        w->th.detach();
    }
    for (int l = 1; l < lanes; ++l) {
        LaneWorker* w = g_workers[(size_t)l - 1];
        w->job = &R;
        w->lane = l;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->state.store(1, std::memory_order_release);
        }
        w->cv.notify_one();
    }
    run_lane(R, 0);
    for (int l = 1; l < lanes; ++l) {
        LaneWorker* w = g_workers[(size_t)l - 1];
        int spins = 0;
        while (w->state.load(std::memory_order_acquire) != 2)
            if (++spins > 2000) std::this_thread::yield();
        w->state.store(0, std::memory_order_release);
    }
    for (int k = 0; k < n_calls; ++k)
        if (calls[k].rc != 0) return calls[k].rc;
    return CS_OK;
}
