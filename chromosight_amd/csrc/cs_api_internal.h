// cs_api_internal.h -- what the translation units of the C ABI share (cs_api.cpp: context, weights, dispatch; cs_api_entries.cpp:
// correlation, staging and host-side entries, call lists; cs_api_foci.cpp: the foci / quantify entries): the context, the host
// thread pool, error plumbing and the declarations of the helpers defined in cs_api.cpp.  Internal to the library.
#pragma once
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
    void* h_small = nullptr;              // page-locked: count + the heads of a short candidate list (cs_candidates, one download)
    long long* d_cand_cnt = nullptr;      // two candidate counters (256 B apart) that alternate between calls of cs_candidates: the
    int cand_cnt_phase = 0;               // re-scoring kernel of one call clears the counter of the next (no memset in the chain);
    bool cand_cnt_clean = false;          // false until a call has gone through: both are cleared before use
    long long cand_prev = 0;              // candidates of the previous cs_candidates call on this context (sizes the next call's one launch)
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

namespace csapi {

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

int fail(cs_ctx* ctx, int code, const char* fmt, ...);

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
cs::MatView view_of(const cs_matrix* m);

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

// grow-only device scratch of a context
int ensure_scratch(cs_ctx* ctx, void** buf, size_t* have, size_t need);
int check_matrix(cs_ctx* ctx, const cs_matrix* m, const char* what, int ns);
// template weights of the call in flight (cs_api.cpp): float32 / float64 instances
template <typename TC>
int upload_weights(cs_ctx* ctx, hipStream_t stream, const std::vector<double>& w64);
extern template int upload_weights<float>(cs_ctx*, hipStream_t, const std::vector<double>&);
extern template int upload_weights<double>(cs_ctx*, hipStream_t, const std::vector<double>&);
int ensure_wfrag(cs_ctx* ctx, hipStream_t stream, int km, int kn, cs::MfmaWeights* E);
// kernel dispatch of one correlation call (cs_api.cpp).  CS_NEED_MAP: a candidate sink was given without a map, and the kernel
// that would serve the call writes maps (nothing was launched that matters: the caller allocates the map and calls again)
constexpr int CS_NEED_MAP = 1000;
template <typename TC>
int launch_corr(cs_ctx* ctx, cs::CorrArgs<TC>& A, hipStream_t stream, bool allow_fast);
template <>
int launch_corr<float>(cs_ctx* ctx, cs::CorrArgs<float>& A, hipStream_t stream, bool allow_fast);
template <>
int launch_corr<double>(cs_ctx* ctx, cs::CorrArgs<double>& A, hipStream_t stream, bool allow_fast);
// validation + template statistics + weight upload + argument block of one correlation call (cs_api.cpp)
template <typename TC>
int build_args(cs_ctx* ctx, hipStream_t stream, const cs_matrix* signal, const cs_kernel* kernel,
               const cs_normxcorr2_params* p, cs::CorrArgs<TC>* out);
extern template int build_args<float>(cs_ctx*, hipStream_t, const cs_matrix*, const cs_kernel*, const cs_normxcorr2_params*, cs::CorrArgs<float>*);
extern template int build_args<double>(cs_ctx*, hipStream_t, const cs_matrix*, const cs_kernel*, const cs_normxcorr2_params*, cs::CorrArgs<double>*);

}  // namespace csapi
