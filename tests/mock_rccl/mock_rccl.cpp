// TEST INFRASTRUCTURE ONLY -- never shipped, never measured.
//
// A host-staged stand-in for the few RCCL entry points libsigma_hip.so binds with dlopen
// (sgm_dist.hip: ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / AllGather / Send / Recv /
// GroupStart / GroupEnd / GetErrorString).  RCCL refuses two ranks on one device ("Duplicate GPU
// detected"), and the GPU boxes of this project have ONE GPU; with SGM_RCCL_LIB pointing here the
// product's whole multi-rank code path -- planning, request-list swap, halo exchange on the
// communication stream, all-reduced dots inside the device-resident Krylov loops -- runs as 2-4
// real processes sharing that GPU.  Only the transport differs: messages go device -> host ->
// POSIX shared memory -> host -> device instead of over xGMI.
//
// Semantics kept: operations are ordered on the stream they are given (the data is read after
// everything queued on the stream before the call, and written with a stream-ordered copy, so a
// consumer on ANOTHER stream still needs the event the product records); sends and receives of a
// pair match in issue order; a group is progressed as a whole (no deadlock on crossing sends).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

// The few device operations the transport needs, in one place.  -DMOCK_RCCL_HOST_ONLY (tools/asan: the AddressSanitizer /
// UBSan build, run on a CPU box by tests/test_asan_cpu.py) makes "device" buffers plain host memory, so that the rings, the
// barrier and the collectives -- the index work of this file -- run as several processes without a GPU.
#include <cstdlib>
#include <cstring>
#ifdef MOCK_RCCL_HOST_ONLY
static inline hipError_t dev_sync(hipStream_t) { return hipSuccess; }
static inline hipError_t dev_copy(void *dst, const void *src, size_t n, hipMemcpyKind) { memcpy(dst, src, n); return hipSuccess; }
static inline hipError_t dev_copy_async(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { memcpy(dst, src, n); return hipSuccess; }
static inline hipError_t pinned_alloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline void pinned_free(void *p) { free(p); }
#else
// Staging buffers are PAGEABLE host memory and every copy is synchronous.  Until round 6 they were hipHostMalloc'ed (recycled by
// the runtime from one message to the next) with an asynchronous copy back: in three full-suite runs one message of an 8-rank
// solve left its sender as the PREVIOUS content of the recycled buffer -- the device-to-host copy had not landed when the bytes
// went into the ring (the per-message trace, MOCK_RCCL_TRACE, showed sender and receiver agreeing on a payload that was an old
// message's).  A stand-in has no use for the overlap pinned memory buys: plain memory, hipMemcpy, and a wait on the device.
static inline hipError_t dev_sync(hipStream_t st) { return hipStreamSynchronize(st); }
static inline hipError_t dev_copy(void *dst, const void *src, size_t n, hipMemcpyKind k)
{
    const hipError_t e = hipMemcpy(dst, src, n, k);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
static inline hipError_t dev_copy_async(void *dst, const void *src, size_t n, hipMemcpyKind k, hipStream_t)
{
    // (the stream was drained before the operation started and this thread launches nothing meanwhile: a copy that has
    //  completed before the call returns is ordered in front of everything the stream gets afterwards)
    const hipError_t e = hipMemcpy(dst, src, n, k);
    return e != hipSuccess ? e : hipDeviceSynchronize();
}
static inline hipError_t pinned_alloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline void pinned_free(void *p) { free(p); }
#endif

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

constexpr size_t kRing = 1u << 20;       // bytes per ordered (src, dst) pair
constexpr size_t kColl = 1u << 16;       // bytes per rank for all-reduce / all-gather
constexpr int kMaxRanks = 16;
constexpr double kTimeoutS = 120.0;

struct alignas(64) Ring {
    std::atomic<uint64_t> head, tail;    // bytes written / read so far
    alignas(64) char buf[kRing];
};
struct Shared {
    std::atomic<int> attached;
    std::atomic<int> bar_count, bar_gen;
    // (64-byte aligned: the reduction reads these slots as doubles -- UBSan found them at offset 76 of the segment, tools/asan)
    alignas(64) char coll[kMaxRanks][kColl];
    alignas(64) Ring ring[1];            // nranks * nranks
};

struct Comm {
    Shared *sh = nullptr;
    size_t bytes = 0;
    int rank = 0, n = 1;
    Ring &ring(int src, int dst) { return *reinterpret_cast<Ring *>(reinterpret_cast<char *>(sh->ring) + sizeof(Ring) * ((size_t)src * n + dst)); }
};

struct Op {
    bool send;
    void *dev;
    size_t bytes, done;
    int peer;
    Comm *c;
    hipStream_t st;
    char *host;
};
std::vector<Op> g_ops;
int g_depth = 0;
struct Pinned { void *p; hipStream_t st; };
std::vector<Pinned> g_graveyard;

// MOCK_RCCL_TRACE=<directory>: every rank appends one line per operation to <directory>/rank<r>.trace -- collectives with their
// first inputs / outputs as hex floats, point-to-point messages with a checksum of their payload -- so that a wrong result of a
// multi-rank test can be told apart: the transport delivered something else than was sent (compare the files), or the library
// computed something else from the same messages.  tests/test_gpu_multirank.py keeps the files of a failed attempt.
FILE *g_trace = nullptr;
long g_trace_seq = 0;
void trace_open(int rank)
{
    if (g_trace) return;
    const char *d = getenv("MOCK_RCCL_TRACE");
    if (!d) return;
    char path[512];
    snprintf(path, sizeof path, "%s/rank%d.trace", d, rank);
    g_trace = fopen(path, "a");
}
uint64_t checksum(const void *p, size_t bytes)
{
    uint64_t h = 1469598103934665603ull;
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void reap()
{
    for (auto &g : g_graveyard) { (void)dev_sync(g.st); pinned_free(g.p); }
    g_graveyard.clear();
}

bool barrier(Comm *c)
{
    Shared *s = c->sh;
    const int gen = s->bar_gen.load();
    if (s->bar_count.fetch_add(1) + 1 == c->n) {
        s->bar_count.store(0);
        s->bar_gen.fetch_add(1);
        return true;
    }
    const double t0 = now();
    while (s->bar_gen.load() == gen) {
        sched_yield();
        if (now() - t0 > kTimeoutS) return false;
    }
    return true;
}

size_t type_size(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

// Test hook (tests/test_gpu_multirank.py: the bench watchdog): MOCK_RCCL_STALL="<rank>:<groups>" makes that rank stop dead
// inside its (groups+1)-th send/recv group -- what a rank stuck in a halo exchange over a broken link looks like to its peers.
void maybe_stall(int rank)
{
    static int want_rank = -2, after = 0, seen = 0;
    if (want_rank == -2) {
        want_rank = -1;
        if (const char *e = getenv("MOCK_RCCL_STALL")) sscanf(e, "%d:%d", &want_rank, &after);
    }
    if (rank != want_rank) return;
    if (++seen <= after) return;
    fprintf(stderr, "[mock_rccl] rank %d: stalling in send/recv group %d (MOCK_RCCL_STALL)\n", rank, seen);
    for (;;) sleep(1);
}

ncclResult_t run_ops()
{
    if (g_ops.empty()) return ncclSuccess;
    maybe_stall(g_ops[0].c->rank);
    reap();
    for (auto &o : g_ops) if (dev_sync(o.st) != hipSuccess) return ncclUnhandledCudaError;
    for (auto &o : g_ops) {
        if (pinned_alloc((void **)&o.host, o.bytes ? o.bytes : 1) != hipSuccess) return ncclSystemError;
        if (o.send && o.bytes && dev_copy(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    }
    const double t0 = now();
    for (;;) {
        bool all = true, moved = false;
        for (size_t oi = 0; oi < g_ops.size(); ++oi) {
            Op &o = g_ops[oi];
            if (o.done == o.bytes) continue;
            all = false;
            // several messages to / from one peer share a ring: they move strictly in issue order
            bool earlier = false;
            for (size_t e = 0; e < oi && !earlier; ++e)
                earlier = g_ops[e].send == o.send && g_ops[e].peer == o.peer && g_ops[e].c == o.c && g_ops[e].done < g_ops[e].bytes;
            if (earlier) continue;
            Ring &r = o.send ? o.c->ring(o.c->rank, o.peer) : o.c->ring(o.peer, o.c->rank);
            const uint64_t head = r.head.load(std::memory_order_acquire), tail = r.tail.load(std::memory_order_acquire);
            size_t chunk = o.send ? kRing - (size_t)(head - tail) : (size_t)(head - tail);
            if (chunk > o.bytes - o.done) chunk = o.bytes - o.done;
            if (!chunk) continue;
            size_t pos = (size_t)((o.send ? head : tail) % kRing), left = chunk, off = o.done;
            while (left) {
                const size_t run = left < kRing - pos ? left : kRing - pos;
                if (o.send) memcpy(r.buf + pos, o.host + off, run); else memcpy(o.host + off, r.buf + pos, run);
                pos = (pos + run) % kRing; off += run; left -= run;
            }
            if (o.send) r.head.store(head + chunk, std::memory_order_release);
            else r.tail.store(tail + chunk, std::memory_order_release);
            o.done += chunk;
            moved = true;
        }
        if (all) break;
        if (!moved) {
            sched_yield();
            if (now() - t0 > kTimeoutS) { fprintf(stderr, "[mock_rccl] send/recv timed out\n"); return ncclSystemError; }
        }
    }
    if (g_trace) {
        for (auto &o : g_ops)
            fprintf(g_trace, "%ld %s peer %d bytes %zu sum %016llx\n", ++g_trace_seq, o.send ? "send" : "recv", o.peer, o.bytes,
                    (unsigned long long)checksum(o.host, o.bytes));
        fflush(g_trace);
    }
    for (auto &o : g_ops) {
        if (!o.send && o.bytes && dev_copy_async(o.dev, o.host, o.bytes, hipMemcpyHostToDevice, o.st) != hipSuccess)
            return ncclUnhandledCudaError;
        g_graveyard.push_back(Pinned{o.host, o.st});
    }
    g_ops.clear();
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/sgm_mock_%d_%lld", (int)getpid(), (long long)(now() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->rank = rank;
    c->n = nranks;
    c->bytes = sizeof(Shared) + sizeof(Ring) * (size_t)nranks * nranks;
    char name[128];
    memcpy(name, id.internal, sizeof name);
    name[127] = 0;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared *)p;                       // a fresh segment is zero-filled: counters start at 0
    c->sh->attached.fetch_add(1);
    const double t0 = now();
    while (c->sh->attached.load() < nranks) {
        sched_yield();
        if (now() - t0 > kTimeoutS) { delete c; return ncclSystemError; }
    }
    if (!barrier(c)) { delete c; return ncclSystemError; }
    trace_open(rank);
    if (rank == 0) shm_unlink(name);            // the mappings keep it alive; nothing is left in /dev/shm
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    reap();
    munmap(c->sh, c->bytes);
    delete c;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "mock_rccl: HIP error";
        case ncclSystemError: return "mock_rccl: system error / timeout";
        case ncclInvalidArgument: return "mock_rccl: invalid argument";
        case ncclInvalidUsage: return "mock_rccl: a group mixing send / recv with an all-reduce is refused (MOCK_RCCL_REFUSE_MIXED_GROUP)";
        default: return "mock_rccl: error";
    }
}

// Test hook: MOCK_RCCL_REFUSE_MIXED_GROUP=1 makes this transport refuse a group that holds BOTH point-to-point operations
// and an all-reduce (ncclInvalidUsage from the all-reduce and from ncclGroupEnd, the queued pairs dropped) -- on every rank
// alike, so nobody waits for a peer.  What sgm_comm_init's probe must find and the solvers must then work around.
static bool refuse_mixed()
{
    static const bool on = getenv("MOCK_RCCL_REFUSE_MIXED_GROUP") != nullptr;
    return on;
}
static bool g_refused = false;

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd()
{
    if (--g_depth > 0) return ncclSuccess;
    g_depth = 0;
    if (g_refused) { g_refused = false; g_ops.clear(); return ncclInvalidUsage; }
    return run_ops();
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st)
{
    g_ops.push_back(Op{true, const_cast<void *>(buf), count * type_size(t), 0, peer, (Comm *)comm, st, nullptr});
    return g_depth ? ncclSuccess : run_ops();
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st)
{
    g_ops.push_back(Op{false, buf, count * type_size(t), 0, peer, (Comm *)comm, st, nullptr});
    return g_depth ? ncclSuccess : run_ops();
}

static ncclResult_t collective(const void *send, void *recv, size_t bytes_each, bool reduce_f64, Comm *c, hipStream_t st)
{
    if (bytes_each > kColl || (reduce_f64 && bytes_each % 8)) return ncclInvalidArgument;
    reap();
    if (dev_sync(st) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes_each && dev_copy(c->sh->coll[c->rank], send, bytes_each, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    const size_t out_bytes = reduce_f64 ? bytes_each : bytes_each * c->n;
    char *host = nullptr;
    if (pinned_alloc((void **)&host, out_bytes ? out_bytes : 1) != hipSuccess) return ncclSystemError;
    if (reduce_f64) {
        double *o = (double *)host;
        for (size_t i = 0; i < bytes_each / 8; ++i) {
            double s = 0.0;
            for (int r = 0; r < c->n; ++r) s += ((const double *)c->sh->coll[r])[i];     // rank order: same bits everywhere
            o[i] = s;
        }
    } else {
        for (int r = 0; r < c->n; ++r) memcpy(host + (size_t)r * bytes_each, c->sh->coll[r], bytes_each);
    }
    if (g_trace) {
        const double *in = (const double *)c->sh->coll[c->rank], *out = (const double *)host;
        if (reduce_f64) fprintf(g_trace, "%ld allreduce count %zu in %a out %a insum %016llx outsum %016llx\n", ++g_trace_seq, bytes_each / 8,
                                bytes_each ? in[0] : 0.0, bytes_each ? out[0] : 0.0, (unsigned long long)checksum(in, bytes_each),
                                (unsigned long long)checksum(out, out_bytes));
        else fprintf(g_trace, "%ld allgather bytes %zu outsum %016llx\n", ++g_trace_seq, bytes_each, (unsigned long long)checksum(host, out_bytes));
        fflush(g_trace);
    }
    if (!barrier(c)) return ncclSystemError;        // nobody overwrites a slot that is still being read
    if (out_bytes && dev_copy_async(recv, host, out_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
    g_graveyard.push_back(Pinned{host, st});
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t st)
{
    if (t != ncclFloat64 || op != ncclSum) return ncclInvalidArgument;
    if (g_depth > 0 && !g_ops.empty() && refuse_mixed()) { g_refused = true; return ncclInvalidUsage; }
    return collective(send, recv, count * 8, true, (Comm *)comm, st);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st)
{
    return collective(send, recv, count * type_size(t), false, (Comm *)comm, st);
}

}  // extern "C"
