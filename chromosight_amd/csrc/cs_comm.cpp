// cs_comm.cpp -- the exchange step of the sharded path on RCCL (xGMI inside a node), behind the C ABI.
//
// The path shards over independent sub-matrices (reference cli/chromosight.py:748-752: Pool.imap over sub-matrices), so
// there is no data-path collective; what ranks exchange is
//   * the pattern records at the end of a detect call: variable-length lists of fixed-size float64 records ->
//     ncclAllGather of the per-rank counts, then ONE padded ncclAllGather of the records;
//   * the pileup of an iterated template (cli/chromosight.py:791) and the per-diagonal (sum, count) of a sub-matrix
//     split over ranks -> ncclAllReduce(sum) of a small float64 vector.
// Records live on the host on both sides (the acceptance rules are numpy), so the entry points take host arrays and do
// the device staging themselves: no torch tensors, no Python between the two collectives.
//
// librccl is loaded on first use (dlopen): a single-GPU process never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/chromosight_hip.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
    std::string err_create;      // why the last cs_comm_create failed (cs_comm_last_error(NULL))
};

Rccl* rccl()
{
    static Rccl R;
    if (R.lib || !R.err.empty()) return &R;
    // The librccl that belongs to the HIP runtime THIS library is bound to: a process may hold two (PyTorch wheels bundle
    // their own libamdhip64 / librccl next to /opt/rocm's, same sonames), and a communicator only works with streams of
    // its own runtime.  So: the directory of the libamdhip64 that hipGetDeviceCount resolves to, then the loader's defaults.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t cut = dir.rfind('/');
        if (cut != std::string::npos) {
            dir.resize(cut + 1);
            names.push_back(dir + "librccl.so.1");
            names.push_back(dir + "librccl.so");
        }
    }
    for (const char* name : {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"}) names.push_back(name);
    // First of all: a librccl that is ALREADY mapped into the process (torch.distributed's, when the ranks were started
    // under it) -- by soname with RTLD_NOLOAD, which loads nothing.  Opening another copy by path next to it would put two
    // RCCL instances (two sets of IPC state, two proxy threads) into one process; the mapped one is bound to the same HIP
    // runtime this library resolved, the loader keeps one libamdhip64 per soname.
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (R.lib) break;
    }
    for (const std::string& name : names) {
        if (R.lib) break;
        R.lib = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!R.lib) {
        R.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
        return &R;
    }
#define CS_SYM(field, name)                                                   \
    R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, name));        \
    if (!R.field) R.err = std::string("librccl lacks ") + name;
    CS_SYM(GetUniqueId, "ncclGetUniqueId")
    CS_SYM(CommInitRank, "ncclCommInitRank")
    CS_SYM(CommDestroy, "ncclCommDestroy")
    CS_SYM(AllGather, "ncclAllGather")
    CS_SYM(AllReduce, "ncclAllReduce")
    CS_SYM(GetErrorString, "ncclGetErrorString")
#undef CS_SYM
    return &R;
}

}  // namespace

struct cs_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void* d_buf = nullptr;        // staging: send block | receive blocks
    size_t d_bytes = 0;
    void* h_pin = nullptr;        // page-locked mirror of the staging buffer
    size_t h_bytes = 0;
    std::string err;
};

namespace {

int cfail(cs_comm* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

int ensure(cs_comm* c, size_t bytes)
{
    if (bytes <= c->d_bytes) return CS_OK;
    bytes = std::max(bytes + bytes / 2, (size_t)1 << 16);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    c->d_buf = c->h_pin = nullptr;
    c->d_bytes = c->h_bytes = 0;
    if (hipMalloc(&c->d_buf, bytes) != hipSuccess) return cfail(c, CS_ERR_HIP, "hipMalloc of %zu staging bytes failed", bytes);
    if (hipHostMalloc(&c->h_pin, bytes, hipHostMallocDefault) != hipSuccess) return cfail(c, CS_ERR_HIP, "hipHostMalloc failed");
    c->d_bytes = c->h_bytes = bytes;
    return CS_OK;
}

#define CS_NCCL(c, call)                                                                                      \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) return cfail(c, CS_ERR_HIP, "%s: %s", #call, rccl()->GetErrorString(r_));     \
    } while (0)
#define CS_HIPC(c, call)                                                                             \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return cfail(c, CS_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));   \
    } while (0)

}  // namespace

extern "C" {

int cs_comm_available(void)
{
    return rccl()->err.empty() ? CS_OK : CS_ERR_UNSUPPORTED;
}

int cs_comm_unique_id(void* out128)
{
    if (!out128) return CS_ERR_INVALID;
    Rccl* R = rccl();
    if (!R->err.empty()) return CS_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return CS_ERR_HIP;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out128, &id, sizeof(id));
    return CS_OK;
}

int cs_comm_create(int device, int rank, int world, const void* unique_id128, cs_comm** out)
{
    if (!out || !unique_id128 || world < 1 || rank < 0 || rank >= world) return CS_ERR_INVALID;
    *out = nullptr;
    Rccl* R = rccl();
    if (!R->err.empty()) return CS_ERR_UNSUPPORTED;
    cs_comm* c = new cs_comm;
    c->device = device;
    c->rank = rank;
    c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    hipError_t he = hipSetDevice(device);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    ncclResult_t ne = ncclSuccess;
    size_t stack_before = 0, stack_after = 0;
    (void)hipDeviceGetLimit(&stack_before, hipLimitStackSize);
    if (he == hipSuccess) ne = R->CommInitRank(&c->comm, world, id, rank);
    (void)hipDeviceGetLimit(&stack_after, hipLimitStackSize);
    if (std::getenv("CHROMOSIGHT_HIP_DEBUG"))
        fprintf(stderr, "[chromosight_hip] ncclCommInitRank: hipLimitStackSize %zu -> %zu bytes per lane\n", stack_before, stack_after);
    if (he != hipSuccess || ne != ncclSuccess) {
        R->err_create = he != hipSuccess ? std::string("HIP: ") + hipGetErrorString(he)
                                         : std::string("ncclCommInitRank: ") + R->GetErrorString(ne);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return CS_ERR_HIP;
    }
    *out = c;
    return CS_OK;
}

void cs_comm_destroy(cs_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)rccl()->CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    delete c;
}

const char* cs_comm_last_error(const cs_comm* c)
{
    if (c) return c->err.c_str();
    Rccl* R = rccl();
    return !R->err.empty() ? R->err.c_str() : R->err_create.c_str();      // NULL: why loading / the last create failed
}

int cs_comm_rank(const cs_comm* c) { return c ? c->rank : -1; }
int cs_comm_world(const cs_comm* c) { return c ? c->world : 0; }

int cs_comm_allreduce_f64(cs_comm* c, double* h_values, int64_t n)
{
    if (!c || n < 0 || (n > 0 && !h_values)) return CS_ERR_INVALID;
    if (n == 0) return CS_OK;
    CS_HIPC(c, hipSetDevice(c->device));
    int rc = ensure(c, 8 * (size_t)n);
    if (rc) return rc;
    std::memcpy(c->h_pin, h_values, 8 * (size_t)n);
    CS_HIPC(c, hipMemcpyAsync(c->d_buf, c->h_pin, 8 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    CS_NCCL(c, rccl()->AllReduce(c->d_buf, c->d_buf, (size_t)n, ncclDouble, ncclSum, c->comm, c->stream));
    CS_HIPC(c, hipMemcpyAsync(c->h_pin, c->d_buf, 8 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    CS_HIPC(c, hipStreamSynchronize(c->stream));
    std::memcpy(h_values, c->h_pin, 8 * (size_t)n);
    return CS_OK;
}

int cs_comm_allgather_rows(cs_comm* c, const double* h_rows, int64_t n_rows, int32_t width, double* h_out, int64_t cap_rows,
                           int64_t* h_counts)
{
    if (!c || n_rows < 0 || width <= 0 || !h_counts || (n_rows > 0 && !h_rows) || cap_rows < 0 || (cap_rows > 0 && !h_out))
        return CS_ERR_INVALID;
    CS_HIPC(c, hipSetDevice(c->device));
    const int W = c->world;
    // 1. counts AND capacities (two int64 per rank): whether the records fit must come out the same on every rank --
    // a rank that returned "overflow" on its own while the others went on to the record exchange would leave the
    // communicator with mismatched collectives.  Every rank sees min(capacity) and takes the same branch.
    int rc = ensure(c, 16 * (size_t)(W + 1));
    if (rc) return rc;
    long long* hp = reinterpret_cast<long long*>(c->h_pin);
    hp[0] = n_rows;
    hp[1] = cap_rows;
    CS_HIPC(c, hipMemcpyAsync(c->d_buf, hp, 16, hipMemcpyHostToDevice, c->stream));
    CS_NCCL(c, rccl()->AllGather(c->d_buf, (char*)c->d_buf + 16, 2, ncclInt64, c->comm, c->stream));
    CS_HIPC(c, hipMemcpyAsync(hp + 2, (char*)c->d_buf + 16, 16 * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    CS_HIPC(c, hipStreamSynchronize(c->stream));
    long long total = 0, widest = 0, room = cap_rows;
    for (int r = 0; r < W; ++r) {
        h_counts[r] = hp[2 + 2 * r];
        total += hp[2 + 2 * r];
        widest = std::max(widest, hp[2 + 2 * r]);
        room = std::min(room, hp[3 + 2 * r]);
    }
    if (total > room) return cfail(c, CS_ERR_OVERFLOW, "%lld rows, room for %lld on the tightest rank", total, room);
    if (total == 0) return CS_OK;
    // 2. the records, padded to the longest list
    const size_t block = 8 * (size_t)widest * (size_t)width;
    rc = ensure(c, block * (size_t)(W + 1));
    if (rc) return rc;
    std::memset(c->h_pin, 0, block);
    if (n_rows) std::memcpy(c->h_pin, h_rows, 8 * (size_t)n_rows * (size_t)width);
    CS_HIPC(c, hipMemcpyAsync(c->d_buf, c->h_pin, block, hipMemcpyHostToDevice, c->stream));
    CS_NCCL(c, rccl()->AllGather(c->d_buf, (char*)c->d_buf + block, (size_t)widest * (size_t)width, ncclDouble, c->comm, c->stream));
    CS_HIPC(c, hipMemcpyAsync((char*)c->h_pin + block, (char*)c->d_buf + block, block * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    CS_HIPC(c, hipStreamSynchronize(c->stream));
    double* dst = h_out;
    for (int r = 0; r < W; ++r) {
        const size_t bytes = 8 * (size_t)h_counts[r] * (size_t)width;
        std::memcpy(dst, (char*)c->h_pin + block * (size_t)(1 + r), bytes);
        dst += (size_t)h_counts[r] * (size_t)width;
    }
    return CS_OK;
}

// The same exchange in ONE collective: every rank sends a block of slot_rows + 1 rows whose first row carries its count; a rank
// whose rows do not fit the slot (or a total beyond cap_rows) shows in the gathered headers, which every rank reads alike: all of
// them get CS_ERR_OVERFLOW (counts set) and call again with a slot for the longest list.  A step of a sharded run gathers a few
// thousand records per rank, about as many as the step before: with the previous counts as the slot the count exchange and its
// synchronisation (half of an exchange's 80 us) are gone.
int cs_comm_allgather_rows_once(cs_comm* c, const double* h_rows, int64_t n_rows, int32_t width, int64_t slot_rows, double* h_out,
                                int64_t cap_rows, int64_t* h_counts)
{
    if (!c || n_rows < 0 || width <= 0 || slot_rows < 0 || !h_counts || (n_rows > 0 && !h_rows) || cap_rows < 0 || (cap_rows > 0 && !h_out))
        return CS_ERR_INVALID;
    CS_HIPC(c, hipSetDevice(c->device));
    const int W = c->world;
    const size_t block = 8 * (size_t)(slot_rows + 1) * (size_t)width;
    int rc = ensure(c, block * (size_t)(W + 1));
    if (rc) return rc;
    double* hp = reinterpret_cast<double*>(c->h_pin);
    std::memset(hp, 0, 8 * (size_t)width);
    hp[0] = (double)n_rows;                                   // (exact: counts are far below 2^53)
    const size_t sent = (size_t)std::min<int64_t>(n_rows, slot_rows);
    if (sent) std::memcpy(hp + width, h_rows, 8 * sent * (size_t)width);
    // (only the header and the rows that are there travel to the device; the rest of the slot is whatever the buffer held)
    CS_HIPC(c, hipMemcpyAsync(c->d_buf, hp, 8 * (sent + 1) * (size_t)width, hipMemcpyHostToDevice, c->stream));
    CS_NCCL(c, rccl()->AllGather(c->d_buf, (char*)c->d_buf + block, (size_t)(slot_rows + 1) * (size_t)width, ncclDouble, c->comm, c->stream));
    CS_HIPC(c, hipMemcpyAsync((char*)c->h_pin + block, (char*)c->d_buf + block, block * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    CS_HIPC(c, hipStreamSynchronize(c->stream));
    long long total = 0, widest = 0;
    for (int r = 0; r < W; ++r) {
        const double* head = reinterpret_cast<const double*>((char*)c->h_pin + block * (size_t)(1 + r));
        h_counts[r] = (int64_t)head[0];
        total += h_counts[r];
        widest = std::max<long long>(widest, h_counts[r]);
    }
    if (widest > slot_rows || total > cap_rows)
        return cfail(c, CS_ERR_OVERFLOW, "%lld rows (longest list %lld), slot of %lld rows, room for %lld", total, widest, (long long)slot_rows,
                     (long long)cap_rows);
    double* dst = h_out;
    for (int r = 0; r < W; ++r) {
        const size_t bytes = 8 * (size_t)h_counts[r] * (size_t)width;
        if (bytes) std::memcpy(dst, (char*)c->h_pin + block * (size_t)(1 + r) + 8 * (size_t)width, bytes);
        dst += (size_t)h_counts[r] * (size_t)width;
    }
    return CS_OK;
}

}  // extern "C"
