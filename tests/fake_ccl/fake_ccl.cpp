// tests/fake_ccl/fake_ccl.cpp -- TEST HARNESS ONLY.  The nine RCCL entry points cuda-sfm_amd/csrc/comm.cpp uses, implemented over a
// POSIX shared-memory segment between processes of ONE machine that may all sit on the SAME GPU.  Linked with comm.cpp INSTEAD of
// librccl into tests/fake_ccl/libsfm_amd_fakeccl.so, it lets a 1-GPU box run the C exchange code (sfm_estimate_E_sharded and its
// pipelined form, the count-sized feature exchange of sfm_process_views_sharded) with TWO real ranks: two RCCL ranks on one GPU
// are refused by the runtime (profiles/same_gpu_rccl_probe.py), so without this the multi-rank logic of comm.cpp -- offsets,
// counts, the order of the grouped broadcasts, who waits for what -- first runs on a multi-GPU node nobody can log into.
// What it does NOT test: RCCL itself, xGMI, the overlap of a collective with compute (every call here synchronises its stream,
// exchanges through host memory and returns when the result is in place).  Nothing in the product loads this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {
constexpr size_t kHeader = 4096;
constexpr size_t kCapacity = (size_t)192 << 20;          // data area (sparse until touched)
constexpr double kTimeoutSeconds = 120.0;

struct Header {
    std::atomic<uint32_t> arrived, generation, attached;
};

struct Deferred { int kind; const void *send; void *recv; size_t count; int elem; int root; hipStream_t stream; };
thread_local int g_group_depth = 0;
}  // namespace

struct ncclComm {
    int rank = 0, nranks = 1;
    char name[64] = "";
    unsigned char *base = nullptr;
    Header *hdr = nullptr;
    unsigned char *data = nullptr;
    std::vector<unsigned char> host;                     // staging between the device and the segment
};

namespace {
int elem_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

bool barrier(ncclComm *c)
{
    const uint32_t gen = c->hdr->generation.load(std::memory_order_acquire);
    if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
        c->hdr->arrived.store(0, std::memory_order_relaxed);
        c->hdr->generation.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) {
            std::fprintf(stderr, "fake_ccl: rank %d waited %.0f s at a barrier: the ranks did not issue the same collectives\n", c->rank, kTimeoutSeconds);
            return false;
        }
    }
    return true;
}

// kind 0: all-reduce (u64 max), 1: all-gather, 2: broadcast
ncclResult_t run(ncclComm *c, const Deferred &d)
{
    const size_t bytes = d.count * (size_t)d.elem;
    const size_t need = d.kind == 2 ? bytes : bytes * (size_t)c->nranks;
    if (need > kCapacity) { std::fprintf(stderr, "fake_ccl: %zu bytes exceed the segment\n", need); return ncclInvalidArgument; }
    if (hipStreamSynchronize(d.stream) != hipSuccess) return ncclUnhandledCudaError;
    if (d.kind == 2) {
        if (c->rank == d.root && bytes && hipMemcpy(c->data, d.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (!barrier(c)) return ncclInternalError;
        if (bytes && (c->rank != d.root || d.send != d.recv) && hipMemcpy(d.recv, c->data, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        return barrier(c) ? ncclSuccess : ncclInternalError;
    }
    if (bytes && hipMemcpy(c->data + (size_t)c->rank * bytes, d.send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclInternalError;
    if (d.kind == 1) {
        if (bytes && hipMemcpy(d.recv, c->data, bytes * (size_t)c->nranks, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    } else {
        c->host.resize(bytes);
        uint64_t *out = reinterpret_cast<uint64_t *>(c->host.data());
        for (size_t i = 0; i < d.count; ++i) {
            uint64_t m = 0;
            for (int r = 0; r < c->nranks; ++r) {
                uint64_t v;
                std::memcpy(&v, c->data + (size_t)r * bytes + 8 * i, 8);
                m = v > m ? v : m;
            }
            out[i] = m;
        }
        if (bytes && hipMemcpy(d.recv, out, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return barrier(c) ? ncclSuccess : ncclInternalError;
}

struct Queued { ncclComm *c; Deferred d; };
thread_local std::vector<Queued> g_queue;

ncclResult_t issue(ncclComm *c, const Deferred &d)
{
    if (g_group_depth > 0) { g_queue.push_back(Queued{ c, d }); return ncclSuccess; }
    return run(c, d);
}
}  // namespace

extern "C" {
#define FAKE_API __attribute__((visibility("default")))

FAKE_API ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof(*id));
    unsigned long long r = (unsigned long long)getpid() * 1000003ull ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof(id->internal), "/sfm_fakeccl_%016llx", r);
    return ncclSuccess;
}

FAKE_API ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm *c = new ncclComm;
    c->rank = rank; c->nranks = nranks;
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)(kHeader + kCapacity)) != 0) { std::perror("fake_ccl: shm_open"); delete c; return ncclSystemError; }
    void *p = mmap(nullptr, kHeader + kCapacity, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { std::perror("fake_ccl: mmap"); delete c; return ncclSystemError; }
    c->base = static_cast<unsigned char *>(p);
    c->hdr = reinterpret_cast<Header *>(c->base);         // (a fresh segment is zero-filled: all three counters start at 0)
    c->data = c->base + kHeader;
    c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->attached.load(std::memory_order_acquire) < (uint32_t)nranks) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) { std::fprintf(stderr, "fake_ccl: rank %d: the other ranks never attached\n", rank); return ncclInternalError; }
    }
    *out = c;
    return ncclSuccess;
}

FAKE_API ncclResult_t ncclCommCount(const ncclComm_t c, int *count) { if (!c || !count) return ncclInvalidArgument; *count = c->nranks; return ncclSuccess; }

FAKE_API ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    (void)barrier(c);                                    // nobody unmaps while another rank still reads
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->base, kHeader + kCapacity);
    delete c;
    return ncclSuccess;
}

FAKE_API const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled hip error (fake_ccl)";
    case ncclSystemError: return "system error (fake_ccl)";
    case ncclInternalError: return "internal error: barrier timeout (fake_ccl)";
    case ncclInvalidArgument: return "invalid argument (fake_ccl)";
    default: return "error (fake_ccl)";
    }
}

FAKE_API ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s)
{
    if (!c || t != ncclUint64 || op != ncclMax) return ncclInvalidArgument;       // the one reduction comm.cpp performs
    return issue(c, Deferred{ 0, send, recv, count, 8, 0, s });
}

FAKE_API ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm_t c, hipStream_t s)
{
    if (!c || elem_bytes(t) == 0) return ncclInvalidArgument;
    return issue(c, Deferred{ 1, send, recv, sendcount, elem_bytes(t), 0, s });
}

FAKE_API ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s)
{
    if (!c || elem_bytes(t) == 0 || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    return issue(c, Deferred{ 2, send, recv, count, elem_bytes(t), root, s });
}

FAKE_API ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }

FAKE_API ncclResult_t ncclGroupEnd()
{
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t rc = ncclSuccess;
    for (const Queued &q : g_queue) {
        const ncclResult_t r = run(q.c, q.d);
        if (r != ncclSuccess && rc == ncclSuccess) rc = r;
    }
    g_queue.clear();
    return rc;
}
}  // extern "C"
