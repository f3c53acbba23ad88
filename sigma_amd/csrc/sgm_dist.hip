// Row-partitioned SpMV / Krylov: halo planning (host index work, bit-exact), halo
// exchange and scalar all-reduce.  Two transports behind one structure:
//   * RCCL over xGMI, one process per GPU (ncclSend/ncclRecv between neighbour ranks for
//     the halo, ncclAllReduce(sum, fp64) for dot products);
//   * all P row blocks inside one process on one GPU (device gathers / a tiny sum kernel)
//     -- the same partition, renumbering and reduction code, testable on a 1-GPU box.
// Nothing like this exists in the reference (no MPI/NCCL/coarrays); the
// only hint is "This loop can be parallelized" (sparse_matrix_composites.f90:1086).
//
// RCCL is bound at run time with dlopen("librccl.so.1") so that the single-GPU path has
// no link-time dependency on it and a process that already loaded torch's RCCL shares it.
#include "sgm_internal.hpp"
#include "sgm_plan_host.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <mutex>
#include <chrono>

namespace sgm {

void free_part(Part &p);
void set_interior_range(Part &p, const int32_t *ptr1, const int32_t *node1);
__global__ void k_gather(double *__restrict__ dst, const double *__restrict__ src,
                         const int32_t *__restrict__ idx, int32_t count);

// ------------------------------------------------------------------ RCCL binding
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static Rccl g_nccl;

static int load_rccl()
{
    if (g_nccl.h) return SGM_OK;
    // SGM_RCCL_LIB names another build of the library (same ncclXxx entry points): a site's own
    // RCCL, or the host-staged stand-in under tests/mock_rccl that lets `pytest -m gpu` run the
    // multi-rank code with several processes on ONE GPU (RCCL itself refuses two ranks per device)
    if (const char *e = getenv("SGM_RCCL_LIB")) {
        g_nccl.h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
        if (!g_nccl.h) return fail(SGM_ERR_RCCL, "cannot dlopen SGM_RCCL_LIB=%s: %s", e, dlerror());
    }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        if (g_nccl.h) break;
        g_nccl.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_nccl.h) return fail(SGM_ERR_RCCL, "cannot dlopen librccl.so.1: %s", dlerror());
#define SYM(field, name)                                                         \
    *(void **)(&g_nccl.field) = dlsym(g_nccl.h, name);                           \
    if (!g_nccl.field) return fail(SGM_ERR_RCCL, "librccl: missing symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(AllGather, "ncclAllGather");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return SGM_OK;
}

#define SGM_NCCL(call)                                                                   \
    do {                                                                                 \
        ncclResult_t r__ = (call);                                                       \
        if (r__ != ncclSuccess)                                                          \
            return sgm::fail(SGM_ERR_RCCL, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, \
                             g_nccl.GetErrorString ? g_nccl.GetErrorString(r__) : "?");  \
    } while (0)

// ------------------------------------------------------------------ phase timers
namespace {
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> pool;            // every event ever created (reused after a read)
    size_t used = 0;
    struct Span { int phase; hipEvent_t a, b; };
    std::vector<Span> spans;
    hipEvent_t open[PH_COUNT] = {nullptr};
} g_prof;
}  // namespace
bool prof_on() { return g_prof.on; }
hipEvent_t prof_event(hipStream_t st)
{
    if (!g_prof.on) return nullptr;
    if (g_prof.used == g_prof.pool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_prof.pool.push_back(e);
    }
    hipEvent_t e = g_prof.pool[g_prof.used++];
    (void)hipEventRecord(e, st);
    return e;
}
void prof_begin(int phase, hipStream_t st) { if (g_prof.on) g_prof.open[phase] = prof_event(st); }
void prof_end(int phase, hipStream_t st)
{
    if (!g_prof.on || !g_prof.open[phase]) return;
    hipEvent_t b = prof_event(st);
    if (b) g_prof.spans.push_back({phase, g_prof.open[phase], b});
    g_prof.open[phase] = nullptr;
}
void prof_span(int phase, hipEvent_t a, hipEvent_t b) { if (g_prof.on && a && b) g_prof.spans.push_back({phase, a, b}); }

// ------------------------------------------------------------------ halo exchange
// the links of an in-process partition, by value in the kernel's arguments: link l copies count[l] entries idx[l][.] of part
// src[l]'s vector into part dst[l]'s halo slots from dst_off[l] on (k_gather's statement)
constexpr int kLinkMax = 64, kLinkParts = 64;
struct GatherLinks {
    double *x[kLinkParts];
    const int32_t *idx[kLinkMax];
    int64_t dst_off[kLinkMax];
    int32_t count[kLinkMax], src[kLinkMax], dst[kLinkMax];
};
__global__ __launch_bounds__(kBlock) void k_gather_links(GatherLinks g)
{
    const int l = blockIdx.y;
    const int32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < g.count[l]) g.x[g.dst[l]][g.dst_off[l] + i] = g.x[g.src[l]][g.idx[l][i]];
}

int halo_exchange(sgm_mat A, double *const *xext, hipStream_t st)
{
    if (A->comm) {
        Part &p = A->parts[0];
        if (p.nbrs.empty()) return SGM_OK;
        // (a second communicator, when one was attached, keeps the halo pairs out of the queue the dots' all-reduces use)
        ncclComm_t comm = (ncclComm_t)(A->comm->nccl_halo ? A->comm->nccl_halo : A->comm->nccl);
        for (auto &nb : p.nbrs)
            if (nb.send_count)
                hipLaunchKernelGGL(k_gather, dim3((nb.send_count + kBlock - 1) / kBlock), dim3(kBlock), 0,
                                   st, nb.send_buf, (const double *)xext[0], nb.send_idx, nb.send_count);
        const int hb_prev = g_hb.phase;
        hb_phase(HB_HALO_POST);
        g_hb.halo_posts = g_hb.halo_posts + 1;
        SGM_NCCL(g_nccl.GroupStart());
        for (auto &nb : p.nbrs) {
            if (nb.send_count) SGM_NCCL(g_nccl.Send(nb.send_buf, nb.send_count, ncclFloat64, nb.peer, comm, st));
            if (nb.recv_count)
                SGM_NCCL(g_nccl.Recv(xext[0] + p.ncol_own + nb.recv_offset, nb.recv_count, ncclFloat64,
                                     nb.peer, comm, st));
        }
        SGM_NCCL(g_nccl.GroupEnd());
        hb_phase(hb_prev);
        return SGM_OK;
    }
    // in-process partitions: the sender's list is gathered straight into the peer's halo -- every link of the partition by ONE
    // launch when the table fits the kernel's arguments (blockIdx.y = link), link by link otherwise
    {
        GatherLinks g;
        int nl = 0, maxc = 0;
        bool fits = A->parts.size() <= (size_t)kLinkParts;
        for (size_t ip = 0; fits && ip < A->parts.size(); ++ip) {
            g.x[ip] = xext[ip];
            for (auto &nb : A->parts[ip].nbrs) {
                if (!nb.send_count) continue;
                if (nl == kLinkMax) { fits = false; break; }
                g.idx[nl] = nb.send_idx;
                g.dst_off[nl] = (int64_t)A->parts[nb.peer].ncol_own + nb.recv_offset;
                g.count[nl] = nb.send_count;
                g.src[nl] = (int32_t)ip;
                g.dst[nl] = nb.peer;
                maxc = std::max(maxc, (int)nb.send_count);
                ++nl;
            }
        }
        if (fits && nl > 1) {
            hipLaunchKernelGGL(k_gather_links, dim3((maxc + kBlock - 1) / kBlock, nl), dim3(kBlock), 0, st, g);
            SGM_HIP(hipGetLastError());
            return SGM_OK;
        }
    }
    for (size_t ip = 0; ip < A->parts.size(); ++ip) {
        Part &p = A->parts[ip];
        for (auto &nb : p.nbrs) {
            if (!nb.send_count) continue;
            Part &q = A->parts[nb.peer];
            hipLaunchKernelGGL(k_gather, dim3((nb.send_count + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                               xext[nb.peer] + q.ncol_own + nb.recv_offset, (const double *)xext[ip],
                               nb.send_idx, nb.send_count);
        }
    }
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

__global__ void k_sum_parts(double *const *slots, int nparts, int count)
{
    const int t = threadIdx.x;
    if (t >= count) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += slots[p][t];     // fixed order: deterministic
    for (int p = 0; p < nparts; ++p) slots[p][t] = s;
}

// testing aid (option "dist_force_collectives"): with ONE rank the all-reduce is still issued, so that the fixed cost of the
// RCCL code path can be measured on a single-GPU box
int g_force_collectives = 0;

// In-process parts: the device table of the parts' slot pointers, one per distinct set of pointers (a solver's slot arrays
// never move while it lives, so a solve uploads each of its few tables once and no call waits for the host afterwards).
// A table is found again only by EXACTLY the pointers it holds, so a recycled address can never name a stale table.
// While the launch stream is being captured (GraphBatch::ensure) nothing here may allocate, copy or synchronise: a table that
// is not there yet is refused (the capture is given up and the solve stays on the launch loop -- in practice the 64+ launched
// iterations in front of the first capture have uploaded every table a replay needs), and the cache is never flushed.
static int slot_table(double *const *slot_ptrs, size_t P, double ***out)
{
    static std::mutex mu;
    static std::vector<std::pair<std::vector<double *>, double **>> tabs;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &t : tabs)
        if (t.first.size() == P && std::equal(t.first.begin(), t.first.end(), slot_ptrs)) { *out = t.second; return SGM_OK; }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(g_rt.stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs != hipStreamCaptureStatusNone)
        return fail(SGM_ERR_UNSUPPORTED, "slot_table: a table of %zu slot pointers is missing while the stream is being captured", P);
    if (tabs.size() >= 256) {                    // (solvers come and go: start over rather than grow without bound)
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
        for (auto &t : tabs) dfree(t.second);
        tabs.clear();
    }
    double **dev = nullptr;
    SGM_TRY(dalloc(&dev, P));
    SGM_HIP(hipMemcpy(dev, slot_ptrs, P * sizeof(double *), hipMemcpyHostToDevice));
    tabs.emplace_back(std::vector<double *>(slot_ptrs, slot_ptrs + P), dev);
    *out = dev;
    return SGM_OK;
}

int allreduce_slots(sgm_mat A, double *const *slot_ptrs, int count)
{
    hipStream_t st = g_rt.stream;
    if (A->comm) {
        if (A->comm->nranks == 1 && !g_force_collectives) return SGM_OK;
        prof_begin(PH_ALLREDUCE, st);
        const int hb_prev = g_hb.phase;
        hb_phase(HB_ALLREDUCE_POST);
        g_hb.allreduce_posts = g_hb.allreduce_posts + 1;
        SGM_NCCL(g_nccl.AllReduce(slot_ptrs[0], slot_ptrs[0], (size_t)count, ncclFloat64, ncclSum,
                                  (ncclComm_t)A->comm->nccl, st));
        hb_phase(hb_prev);
        prof_end(PH_ALLREDUCE, st);
        return SGM_OK;
    }
    const size_t P = A->parts.size();
    if (P <= 1) return SGM_OK;
    if (count > 64) return fail(SGM_ERR_BAD_ARG, "allreduce_slots: count %d > 64", count);
    double **tab = nullptr;
    SGM_TRY(slot_table(slot_ptrs, P, &tab));
    prof_begin(PH_ALLREDUCE, st);
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(64), 0, st, (double *const *)tab, (int)P, count);
    prof_end(PH_ALLREDUCE, st);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// The boundary rows of an extended vector to the neighbours' halo slots AND the sum of `count` scalar slots, in one step on
// the launch stream -- what a CG iteration needs between "r -= alpha q" and "p = r + beta p" once p's halo is formed locally
// (run_cg): over RCCL the send / recv pairs and the all-reduce are ONE group (one launch of the communication kernel, the
// pairs on their direct links beside the all-reduce's ring); `grouped` = false posts the same operations one after the other.
int halo_exchange_allreduce(sgm_mat A, double *const *uext, double *const *slot_ptrs, int count, bool grouped)
{
    hipStream_t st = g_rt.stream;
    if (!A->comm) {
        prof_begin(PH_HALO, st);
        SGM_TRY(halo_exchange(A, uext, st));
        prof_end(PH_HALO, st);
        return allreduce_slots(A, slot_ptrs, count);
    }
    Part &p = A->parts[0];
    const bool reduce = A->comm->nranks > 1 || g_force_collectives;
    grouped = grouped && A->comm->group_ok;                // (what sgm_comm_init found this transport to take)
    if (!grouped || p.nbrs.empty() || !reduce) {
        if (!p.nbrs.empty()) {
            prof_begin(PH_HALO, st);
            SGM_TRY(halo_exchange(A, uext, st));
            prof_end(PH_HALO, st);
        }
        return allreduce_slots(A, slot_ptrs, count);
    }
    ncclComm_t comm = (ncclComm_t)A->comm->nccl;          // (one group = one communicator: the all-reduce's)
    for (auto &nb : p.nbrs)
        if (nb.send_count)
            hipLaunchKernelGGL(k_gather, dim3((nb.send_count + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nb.send_buf,
                               (const double *)uext[0], nb.send_idx, nb.send_count);
    const int hb_prev = g_hb.phase;
    hb_phase(HB_HALO_POST);
    g_hb.halo_posts = g_hb.halo_posts + 1;
    g_hb.allreduce_posts = g_hb.allreduce_posts + 1;
    prof_begin(PH_ALLREDUCE, st);                          // (the group is timed as the all-reduce it contains)
    SGM_NCCL(g_nccl.GroupStart());
    for (auto &nb : p.nbrs) {
        if (nb.send_count) SGM_NCCL(g_nccl.Send(nb.send_buf, nb.send_count, ncclFloat64, nb.peer, comm, st));
        if (nb.recv_count)
            SGM_NCCL(g_nccl.Recv(uext[0] + p.ncol_own + nb.recv_offset, nb.recv_count, ncclFloat64, nb.peer, comm, st));
    }
    SGM_NCCL(g_nccl.AllReduce(slot_ptrs[0], slot_ptrs[0], (size_t)count, ncclFloat64, ncclSum, comm, st));
    SGM_NCCL(g_nccl.GroupEnd());
    prof_end(PH_ALLREDUCE, st);
    hb_phase(hb_prev);
    return SGM_OK;
}

// ------------------------------------------------------------------ re-ordered halo slots (colour-ordered parts, sgm_pc.hip)
// A receiver that re-orders the slots of its halo (halo_attach_order) tells every sender where entry j of their link goes:
// out[k][j] = new position (relative to the link's first slot) of entry j this rank sends to nbrs[k].peer.  One grouped
// send / recv of int32 lists over the communicator the halo itself travels on; collective over the ranks (every rank sets its
// preconditioner up).
int exchange_halo_orders(sgm_mat A, const std::vector<int32_t> &mine, std::vector<int32_t *> &out)
{
    Part &p = A->parts[0];
    out.assign(p.nbrs.size(), nullptr);
    if (!A->comm || p.nbrs.empty()) return SGM_OK;
    hipStream_t st = g_rt.stream;
    ncclComm_t comm = (ncclComm_t)(A->comm->nccl_halo ? A->comm->nccl_halo : A->comm->nccl);
    std::vector<int32_t *> give(p.nbrs.size(), nullptr);
    struct Tmp { std::vector<int32_t *> &v; ~Tmp() { for (int32_t *q : v) dfree(q); } } tmp{give};
    for (size_t k = 0; k < p.nbrs.size(); ++k) {
        const HaloNbr &nb = p.nbrs[k];
        if (nb.recv_count) {
            std::vector<int32_t> rel((size_t)nb.recv_count);
            for (int32_t t = 0; t < nb.recv_count; ++t) rel[(size_t)t] = mine[(size_t)nb.recv_offset + t] - nb.recv_offset;
            SGM_TRY(dalloc(&give[k], (size_t)nb.recv_count));
            SGM_HIP(hipMemcpyAsync(give[k], rel.data(), (size_t)nb.recv_count * 4, hipMemcpyHostToDevice, st));
            SGM_HIP(hipStreamSynchronize(st));          // (on the stream the sends read it on; `rel` goes out of scope)
        }
        if (nb.send_count) SGM_TRY(dalloc(&out[k], (size_t)nb.send_count));
    }
    const int hb_prev = g_hb.phase;
    hb_phase(HB_HALO_POST);
    SGM_NCCL(g_nccl.GroupStart());
    for (size_t k = 0; k < p.nbrs.size(); ++k) {
        const HaloNbr &nb = p.nbrs[k];
        if (nb.recv_count) SGM_NCCL(g_nccl.Send(give[k], (size_t)nb.recv_count, ncclInt32, nb.peer, comm, st));
        if (nb.send_count) SGM_NCCL(g_nccl.Recv(out[k], (size_t)nb.send_count, ncclInt32, nb.peer, comm, st));
    }
    SGM_NCCL(g_nccl.GroupEnd());
    SGM_HIP(hipStreamSynchronize(st));
    hb_phase(hb_prev);
    return SGM_OK;
}

// ------------------------------------------------------------------ dot_order = 1 across ranks
int seq_chain_recv(sgm_mat A, double *run)
{
    if (!A->comm || A->comm->nranks <= 1 || A->comm->rank == 0) return SGM_OK;
    SGM_NCCL(g_nccl.Recv(run, 1, ncclFloat64, A->comm->rank - 1, (ncclComm_t)A->comm->nccl, g_rt.stream));
    return SGM_OK;
}
int seq_chain_share(sgm_mat A, double *run)
{
    if (!A->comm || A->comm->nranks <= 1) return SGM_OK;
    const int r = A->comm->rank, R = A->comm->nranks;
    hipStream_t st = g_rt.stream;
    if (r + 1 < R) {
        SGM_NCCL(g_nccl.Send(run, 1, ncclFloat64, r + 1, (ncclComm_t)A->comm->nccl, st));
        SGM_HIP(hipMemsetAsync(run, 0, sizeof(double), st));      // +0.0: total + 0.0 + ... is the total, bit for bit
    }
    SGM_NCCL(g_nccl.AllReduce(run, run, 1, ncclFloat64, ncclSum, (ncclComm_t)A->comm->nccl, st));
    return SGM_OK;
}

// ------------------------------------------------------------------ host planning
// The host-only index work (sorted unique halo lists, renumbering, request lists, neighbour tables, links of an in-process
// partition, the nnz-balanced row split) lives in sgm_plan_host.hpp: plain C++ without a HIP call, exported through the C ABI
// so that the CPU test suite drives the SAME code the RCCL path runs (tests/test_dist_cpu.py, world_size-2 gloo) -- also under
// AddressSanitizer / UBSan (tools/asan) -- and sgm_csr_create_partitioned / sgm_csr_create_dist are both built on it.

// frees a half-built matrix (and scratch device buffers) on every early return
struct MatGuard {
    sgm_mat A = nullptr;
    std::vector<void *> scratch;
    ~MatGuard()
    {
        for (void *p : scratch) dfree(p);
        if (A) sgm_mat_destroy(A);
    }
    sgm_mat release() { sgm_mat a = A; A = nullptr; return a; }
};

}  // namespace sgm

using namespace sgm;

extern "C" {

int sgm_halo_plan_host(int32_t n_own, int64_t col_begin, int64_t nnz, const int32_t *node,
                       int32_t *node_local, int32_t *halo_cols, int32_t *n_halo)
{
    return host_halo_plan_host(n_own, col_begin, nnz, node, node_local, halo_cols, n_halo);
}
int sgm_dist_plan_host(int32_t rank, int32_t nranks, const int64_t *row_starts, int32_t n_halo,
                       const int32_t *halo_cols, int32_t *want, int32_t *want_off, int32_t *req)
{
    return host_dist_plan_host(rank, nranks, row_starts, n_halo, halo_cols, want, want_off, req);
}
int sgm_dist_neighbors_host(int32_t rank, int32_t nranks, const int32_t *want_all, int32_t *peer,
                            int32_t *send_count, int32_t *recv_count, int32_t *recv_offset, int32_t *n_nbrs)
{
    return host_dist_neighbors_host(rank, nranks, want_all, peer, send_count, recv_count, recv_offset, n_nbrs);
}
int sgm_partition_links_host(int32_t nparts, const int64_t *row_starts, const int32_t *ptr, const int32_t *node,
                             int32_t *n_links, int32_t *sender, int32_t *receiver, int32_t *recv_offset,
                             int32_t *count, int32_t *idx_concat, int64_t idx_capacity, int64_t *idx_needed)
{
    return host_partition_links_host(nparts, row_starts, ptr, node, n_links, sender, receiver, recv_offset, count, idx_concat,
                                     idx_capacity, idx_needed);
}
int sgm_partition_rows_by_nnz(int32_t nrow, const int32_t *ptr, int32_t nparts, int32_t align, int64_t *row_starts)
{
    return host_partition_rows_by_nnz(nrow, ptr, nparts, align, row_starts);
}
int sgm_ell_degrees_host(int32_t n, int32_t max_d, const int32_t *node, int32_t *deg)
{
    return host_ell_degrees_host(n, max_d, node, deg);
}
int sgm_left_permute_rows_host(int32_t n, const int32_t *p, const int32_t *ptr, const int32_t *node, const double *val, int64_t r0,
                               int64_t r1, int32_t *lptr, int32_t *lnode, double *lval, int64_t capacity, int64_t *needed)
{
    return host_left_permute_rows_host(n, p, ptr, node, val, r0, r1, lptr, lnode, lval, capacity, needed);
}

int sgm_csr_create_partitioned(sgm_mat *out, int32_t nparts, const int64_t *row_starts, int32_t nrow,
                               int32_t ncol, int64_t nnz, const int32_t *ptr, const int32_t *node,
                               const double *val)
{
    SGM_TRY(require_init());
    if (!out || nparts < 1 || !row_starts || !ptr || nrow != ncol || row_starts[0] != 0 ||
        row_starts[nparts] != nrow)
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned: bad argument (square matrices only)");
    // host arrays, cut up on the host below: check them here (sgm_csr_create's checks, same codes)
    if (ptr[0] != 1) return fail(SGM_ERR_BAD_ARG, "csr create: ptr(1) = %d, expected 1 (1-based row pointers)", ptr[0]);
    for (int32_t i = 0; i < nrow; ++i)
        if (ptr[i + 1] < ptr[i])
            return fail(SGM_ERR_BAD_ARG, "csr create: row pointers decrease at row %d: ptr(%d) = %d > ptr(%d) = %d", i + 1, i + 1, ptr[i],
                        i + 2, ptr[i + 1]);
    if ((int64_t)ptr[nrow] - 1 != nnz)
        return fail(SGM_ERR_DIMS, "csr create: ptr(%d) - 1 = %lld entries, but nnz = %lld", nrow + 1, (long long)ptr[nrow] - 1, (long long)nnz);
    if (nnz && (!node || !val)) return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned: null arrays");
    for (int p = 0; p < nparts; ++p)
        if (row_starts[p + 1] < row_starts[p]) return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned: row_starts must not decrease");
    for (int32_t i = 0; i < nrow; ++i)
        for (int64_t k = ptr[i] - 1; k < ptr[i + 1] - 1; ++k)
            if (node[k] < 1 || node[k] > ncol)
                return fail(SGM_ERR_DIMS, "csr create: node(%lld) = %d in row %d is outside 1..%d", (long long)k + 1, node[k], i + 1, ncol);
    MatGuard g;
    sgm_mat A = g.A = new sgm_mat_s;
    A->fmt = SGM_FMT_CSR;
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz = nnz;
    A->parts.resize(nparts);
    std::vector<std::vector<int32_t>> halos(nparts);
    for (int ip = 0; ip < nparts; ++ip) {
        const int64_t r0 = row_starts[ip], r1 = row_starts[ip + 1];
        const int32_t n = (int32_t)(r1 - r0);
        const int64_t k0 = ptr[r0] - 1, k1 = ptr[r1] - 1;
        std::vector<int32_t> lptr(n + 1), lnode(std::max<int64_t>(k1 - k0, 1));
        for (int32_t i = 0; i <= n; ++i) lptr[i] = (int32_t)(ptr[r0 + i] - k0);
        halo_plan(n, r0, k1 - k0, node + k0, lnode.data(), halos[ip]);
        Part &p = A->parts[ip];
        SGM_TRY(build_csr_part(p, n, n, (int32_t)halos[ip].size(), k1 - k0, lptr.data(), lnode.data(), val + k0, SGM_HOST));
        SGM_TRY(dalloc(&p.xext, (size_t)p.xlen()));
        p.row_begin = r0;
        set_interior_range(p, lptr.data(), lnode.data());
    }
    // send lists: part q sends to part p the entries of p's halo that q owns, in p's halo order
    std::vector<Link> links;
    SGM_TRY(partition_links(nparts, row_starts, halos, links));
    for (const Link &l : links) {
        HaloNbr nb;
        nb.peer = l.receiver;                 // stored on the SENDER
        nb.send_count = (int32_t)l.idx.size();
        nb.recv_offset = l.recv_offset;       // offset in the receiver's halo region
        SGM_TRY(dalloc(&nb.send_idx, l.idx.size()));
        A->parts[l.sender].nbrs.push_back(nb);          // owned by the part from here on
        SGM_HIP(hipMemcpyAsync(nb.send_idx, l.idx.data(), l.idx.size() * 4, hipMemcpyHostToDevice, g_rt.stream));
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
    }
    *out = g.release();
    return SGM_OK;
}

/* sgm_csr_create_partitioned_parts: the same in-process row partition, handed over PART BY PART -- for every part its rows as
 * sgm_csr_create_dist takes a rank's (local 1-based row pointers, GLOBAL 1-based columns, values; host or device arrays) -- so
 * that a matrix too large to be assembled whole on the host (7-point 464^3: 8.8 GB of arrays) can be built from row blocks
 * generated on the device.  Same planners, same parts, same products as sgm_csr_create_partitioned on the whole arrays. */
int sgm_csr_create_partitioned_parts(sgm_mat *out, int32_t nparts, const int64_t *row_starts, const int64_t *nnz_of_part,
                                     const int32_t *const *ptr_of_part, const int32_t *const *node_of_part,
                                     const double *const *val_of_part, int where)
{
    SGM_TRY(require_init());
    if (!out || nparts < 1 || !row_starts || !nnz_of_part || !ptr_of_part || !node_of_part || !val_of_part || row_starts[0] != 0 ||
        row_starts[nparts] > INT32_MAX)
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned_parts: bad argument");
    for (int p = 0; p < nparts; ++p) {
        if (row_starts[p + 1] < row_starts[p]) return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned_parts: row_starts must not decrease");
        if (p && (row_starts[p] & 1)) return fail(SGM_ERR_UNSUPPORTED, "partition boundaries must be even rows (16-B vector access)");
        if (!ptr_of_part[p] || (nnz_of_part[p] && (!node_of_part[p] || !val_of_part[p])))
            return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned_parts: null arrays of part %d", p);
    }
    hipStream_t st = g_rt.stream;
    const int32_t nrow = (int32_t)row_starts[nparts];
    MatGuard g;
    sgm_mat A = g.A = new sgm_mat_s;
    A->fmt = SGM_FMT_CSR;
    A->nrow = A->ncol = nrow;
    A->nnz = 0;
    A->parts.resize(nparts);
    std::vector<std::vector<int32_t>> halos(nparts);
    for (int ip = 0; ip < nparts; ++ip) {
        const int64_t r0 = row_starts[ip], nnz = nnz_of_part[ip];
        const int32_t n = (int32_t)(row_starts[ip + 1] - r0);
        std::vector<int32_t> hnode, hptr, lnode((size_t)std::max<int64_t>(nnz, 1));
        const int32_t *node_h = node_of_part[ip], *ptr_h = ptr_of_part[ip];
        if (where == SGM_DEVICE) {
            hnode.resize((size_t)std::max<int64_t>(nnz, 1));
            hptr.resize((size_t)n + 1);
            if (nnz) SGM_TRY(copy_big(hnode.data(), node_of_part[ip], (size_t)nnz * 4, hipMemcpyDeviceToHost));
            SGM_HIP(hipMemcpyAsync(hptr.data(), ptr_of_part[ip], ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
            SGM_HIP(hipStreamSynchronize(st));
            node_h = hnode.data();
            ptr_h = hptr.data();
        }
        if (ptr_h[0] != 1 || (int64_t)ptr_h[n] - 1 != nnz)
            return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned_parts: part %d: ptr(1) = %d, ptr(n+1)-1 = %lld, nnz = %lld", ip, ptr_h[0],
                        (long long)ptr_h[n] - 1, (long long)nnz);
        for (int64_t k = 0; k < nnz; ++k)
            if (node_h[k] < 1 || node_h[k] > nrow)
                return fail(SGM_ERR_DIMS, "csr create: part %d: node(%lld) = %d is outside 1..%d", ip, (long long)k + 1, node_h[k], nrow);
        halo_plan(n, r0, nnz, node_h, lnode.data(), halos[ip]);
        Part &p = A->parts[ip];
        if (where == SGM_DEVICE) {
            int32_t *dnode = nullptr;                  // values stay on the device; only the renumbered columns are uploaded
            SGM_TRY(dalloc(&dnode, (size_t)std::max<int64_t>(nnz, 1)));
            g.scratch.push_back(dnode);
            if (nnz) SGM_TRY(copy_big(dnode, lnode.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
            SGM_TRY(build_csr_part(p, n, n, (int32_t)halos[ip].size(), nnz, ptr_of_part[ip], dnode, val_of_part[ip], SGM_DEVICE));
            SGM_HIP(hipStreamSynchronize(st));
            dfree(dnode);
            g.scratch.pop_back();
        } else {
            SGM_TRY(build_csr_part(p, n, n, (int32_t)halos[ip].size(), nnz, ptr_h, lnode.data(), val_of_part[ip], SGM_HOST));
        }
        SGM_TRY(dalloc(&p.xext, (size_t)p.xlen()));
        p.row_begin = r0;
        set_interior_range(p, ptr_h, lnode.data());
        A->nnz += nnz;
    }
    std::vector<Link> links;
    SGM_TRY(partition_links(nparts, row_starts, halos, links));
    for (const Link &l : links) {
        HaloNbr nb;
        nb.peer = l.receiver;
        nb.send_count = (int32_t)l.idx.size();
        nb.recv_offset = l.recv_offset;
        SGM_TRY(dalloc(&nb.send_idx, l.idx.size()));
        A->parts[l.sender].nbrs.push_back(nb);
        SGM_HIP(hipMemcpyAsync(nb.send_idx, l.idx.data(), l.idx.size() * 4, hipMemcpyHostToDevice, g_rt.stream));
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
    }
    *out = g.release();
    return SGM_OK;
}

int sgm_comm_unique_id(void *id128)
{
    SGM_TRY(load_rccl());
    ncclUniqueId id;
    SGM_NCCL(g_nccl.GetUniqueId(&id));
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return SGM_OK;
}

// The group CG posts per iteration with option dist_halo_fused = 1 -- send / recv pairs with the neighbours AND an all-reduce in
// ONE ncclGroup -- tried once per communicator before any solve depends on it: every rank sends one double to its
// right neighbour, receives one from its left and all-reduces a 1.0 in the same group.  A transport that REFUSES the group
// (an error from any call of it) or delivers wrong values makes this rank vote 0; the votes are summed by a plain all-reduce,
// and anything short of nranks sets group_ok = 0 on every rank: the solvers then post the pairs and the all-reduce one after
// the other (the order option dist_halo_fused = 2 uses), which every transport that passed sgm_comm_init can do.  A transport
// that HANGS on the group hangs here, at creation, where sgm_heartbeat names the phase -- not inside the first solve.
static int probe_mixed_group(sgm_comm c)
{
    double *d = nullptr;
    SGM_TRY(dalloc(&d, 4));
    struct Tmp { double *&p; ~Tmp() { dfree(p); } } tmp{d};
    const int R = c->nranks, right = (c->rank + 1) % R, left = (c->rank + R - 1) % R;
    const double h[4] = {42.0 + c->rank, -1.0, 1.0, 1.0};
    hipStream_t st = g_rt.stream;
    SGM_HIP(hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));
    ncclComm_t comm = (ncclComm_t)c->nccl;
    const int hb_prev = g_hb.phase;
    hb_phase(HB_HALO_POST);
    bool ok = g_nccl.GroupStart() == ncclSuccess;
    ok = g_nccl.Send(d, 1, ncclFloat64, right, comm, st) == ncclSuccess && ok;
    ok = g_nccl.Recv(d + 1, 1, ncclFloat64, left, comm, st) == ncclSuccess && ok;
    ok = g_nccl.AllReduce(d + 2, d + 2, 1, ncclFloat64, ncclSum, comm, st) == ncclSuccess && ok;
    ok = g_nccl.GroupEnd() == ncclSuccess && ok;
    ok = hipStreamSynchronize(st) == hipSuccess && ok;
    hb_phase(hb_prev);
    double r[4] = {0, 0, 0, 0};
    if (ok) {
        SGM_HIP(hipMemcpy(r, d, sizeof r, hipMemcpyDeviceToHost));
        ok = r[1] == 42.0 + left && r[2] == (double)R;
    } else {
        (void)hipGetLastError();
    }
    const double vote = ok ? 1.0 : 0.0;
    SGM_HIP(hipMemcpyAsync(d + 3, &vote, sizeof vote, hipMemcpyHostToDevice, st));      // (on the stream the all-reduce reads it on)
    SGM_HIP(hipStreamSynchronize(st));
    SGM_NCCL(g_nccl.AllReduce(d + 3, d + 3, 1, ncclFloat64, ncclSum, comm, st));
    SGM_HIP(hipStreamSynchronize(st));
    double votes = 0.0;
    SGM_HIP(hipMemcpy(&votes, d + 3, sizeof votes, hipMemcpyDeviceToHost));
    c->group_ok = votes == (double)R ? 1 : 0;
    if (trace_on() || !c->group_ok)
        fprintf(stderr, "[sigma_hip] rank %d/%d: one group of send/recv pairs + all-reduce: %s (%d of %d ranks)%s\n", c->rank, R,
                c->group_ok ? "taken" : "REFUSED", (int)votes, R,
                c->group_ok ? "" : " -- CG posts the pairs and the all-reduce separately (as with dist_halo_fused = 2)");
    return SGM_OK;
}

int sgm_comm_init(sgm_comm *out, int rank, int nranks, const void *id128)
{
    SGM_TRY(require_init());
    SGM_TRY(load_rccl());
    if (!out || rank < 0 || rank >= nranks || !id128) return fail(SGM_ERR_BAD_ARG, "sgm_comm_init: bad argument");
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t c = nullptr;
    SGM_NCCL(g_nccl.CommInitRank(&c, nranks, id, rank));
    sgm_comm h = new sgm_comm_s;
    h->rank = rank;
    h->nranks = nranks;
    h->nccl = c;
    if (nranks > 1) {
        const int rc = probe_mixed_group(h);
        if (rc != SGM_OK) { sgm_comm_destroy(h); return rc; }
    }
    *out = h;
    return SGM_OK;
}

int sgm_comm_attach_halo_comm(sgm_comm c, const void *id128)
{
    SGM_TRY(require_init());
    SGM_TRY(load_rccl());
    if (!c || !id128) return fail(SGM_ERR_BAD_ARG, "sgm_comm_attach_halo_comm: bad argument");
    if (c->nccl_halo) return fail(SGM_ERR_BAD_ARG, "sgm_comm_attach_halo_comm: a halo communicator is attached already");
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t h = nullptr;
    SGM_NCCL(g_nccl.CommInitRank(&h, c->nranks, id, c->rank));
    c->nccl_halo = h;
    return SGM_OK;
}

int sgm_dist_profile(int on)
{
    SGM_TRY(require_init());
    if (g_rt.comm_stream) SGM_HIP(hipStreamSynchronize(g_rt.comm_stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    g_prof.on = on != 0;
    g_prof.used = 0;
    g_prof.spans.clear();
    for (auto &e : g_prof.open) e = nullptr;
    return SGM_OK;
}

int sgm_dist_profile_read(double *ms_out, int64_t *count_out)
{
    SGM_TRY(require_init());
    if (!ms_out || !count_out) return fail(SGM_ERR_BAD_ARG, "sgm_dist_profile_read: null argument");
    if (g_rt.comm_stream) SGM_HIP(hipStreamSynchronize(g_rt.comm_stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    for (int k = 0; k < PH_COUNT; ++k) { ms_out[k] = 0.0; count_out[k] = 0; }
    for (const auto &sp : g_prof.spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) { (void)hipGetLastError(); continue; }
        ms_out[sp.phase] += ms > 0.f ? (double)ms : 0.0;
        count_out[sp.phase] += 1;
    }
    g_prof.used = 0;
    g_prof.spans.clear();
    return SGM_OK;
}

// sgm_comm_group_selftest: ONE ncclGroup that holds a send / recv pair (this rank to itself) AND an in-place all-reduce of one
// double -- the shape of the group the CG loop posts with option dist_halo_fused = 1 -- on the transport this communicator
// was made with.  out3 = {what the pair delivered (must be 42 + rank), the all-reduced 1.0 (must be nranks), microseconds}.
// What it is for: a single-GPU box can show that the REAL librccl takes a group mixing point-to-point and collective
// operations before the first multi-GPU run depends on it.
int sgm_comm_group_selftest(sgm_comm c, double *out3)
{
    SGM_TRY(require_init());
    if (!c || !out3) return fail(SGM_ERR_BAD_ARG, "sgm_comm_group_selftest: null argument");
    double *d = nullptr;
    SGM_TRY(dalloc(&d, 4));
    struct Tmp { double *&p; ~Tmp() { dfree(p); } } tmp{d};
    const double h[4] = {42.0 + c->rank, -1.0, 1.0, 0.0};
    hipStream_t st = g_rt.stream;
    SGM_HIP(hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));
    ncclComm_t comm = (ncclComm_t)c->nccl;
    double us = 0.0;
    for (int rep = 0; rep < 2; ++rep) {                 // (the second pass is the timed one: the first sets the channels up)
        const auto t0 = std::chrono::steady_clock::now();
        SGM_NCCL(g_nccl.GroupStart());
        SGM_NCCL(g_nccl.Send(d, 1, ncclFloat64, c->rank, comm, st));
        SGM_NCCL(g_nccl.Recv(d + 1, 1, ncclFloat64, c->rank, comm, st));
        if (rep == 0) SGM_NCCL(g_nccl.AllReduce(d + 2, d + 2, 1, ncclFloat64, ncclSum, comm, st));
        else SGM_NCCL(g_nccl.AllReduce(d + 3, d + 3, 1, ncclFloat64, ncclSum, comm, st));
        SGM_NCCL(g_nccl.GroupEnd());
        SGM_HIP(hipStreamSynchronize(st));
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    double r[4] = {0, 0, 0, 0};
    SGM_HIP(hipMemcpy(r, d, sizeof r, hipMemcpyDeviceToHost));
    out3[0] = r[1]; out3[1] = r[2]; out3[2] = us;
    return SGM_OK;
}

int sgm_comm_group_ok(sgm_comm c)
{
    return c ? c->group_ok : 0;
}

int sgm_comm_destroy(sgm_comm c)
{
    if (!c) return SGM_OK;
    if (c->nccl_halo && g_nccl.CommDestroy) (void)g_nccl.CommDestroy((ncclComm_t)c->nccl_halo);
    if (c->nccl && g_nccl.CommDestroy) (void)g_nccl.CommDestroy((ncclComm_t)c->nccl);
    delete c;
    return SGM_OK;
}

int sgm_csr_create_dist(sgm_mat *out, sgm_comm comm, const int64_t *row_starts, int64_t nnz,
                        const int32_t *ptr, const int32_t *node, const double *val, int where)
{
    return sgm_csr_create_dist_rect(out, comm, row_starts, row_starts, nnz, ptr, node, val, where);
}

// rows partitioned by row_starts, x (the columns) by col_starts: an off-diagonal block of a composite whose
// block rows / block columns have partitions of their own
int sgm_csr_create_dist_rect(sgm_mat *out, sgm_comm comm, const int64_t *row_starts, const int64_t *col_starts, int64_t nnz,
                             const int32_t *ptr, const int32_t *node, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || !comm || !row_starts || !col_starts || !ptr) return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_dist: bad argument");
    const int R = comm->nranks, me = comm->rank;
    const int64_t r0 = row_starts[me], r1 = row_starts[me + 1];
    const int64_t c0 = col_starts[me], c1 = col_starts[me + 1];
    const int32_t n = (int32_t)(r1 - r0), nc = (int32_t)(c1 - c0);
    if (row_starts[0] != 0 || r1 < r0 || row_starts[R] > INT32_MAX || col_starts[0] != 0 || c1 < c0 || col_starts[R] > INT32_MAX)
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_dist: row_starts / col_starts must rise from 0 to the global count (< 2^31)");
    for (int q = 1; q < R; ++q)
        if ((row_starts[q] | col_starts[q]) & 1) return fail(SGM_ERR_UNSUPPORTED, "partition boundaries must be even rows (16-B vector access)");
    hipStream_t st = g_rt.stream;

    // the index work runs on the host (same code as sgm_halo_plan_host / sgm_dist_plan_host).
    // Device arrays are read on the library's stream: the caller may still be producing them there.
    std::vector<int32_t> hnode, lnode((size_t)std::max<int64_t>(nnz, 1)), hptr;
    const int32_t *node_h = node, *ptr_h = ptr;
    if (where == SGM_DEVICE) {
        hnode.resize((size_t)std::max<int64_t>(nnz, 1));
        hptr.resize((size_t)n + 1);
        if (nnz) SGM_HIP(hipMemcpyAsync(hnode.data(), node, (size_t)nnz * 4, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipMemcpyAsync(hptr.data(), ptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipStreamSynchronize(st));
        node_h = hnode.data();
        ptr_h = hptr.data();
    }
    // What a rank finds wrong with ITS rows must not make it leave before the collectives below (its peers would wait in them
    // for ever): the verdict travels in the all-gathered want matrix (a rank never asks itself for anything: its own slot
    // carries -1 = "my rows were rejected"), and every rank returns the same error.
    int local_rc = SGM_OK;
    std::string local_msg;
    std::vector<int32_t> halo;
    std::vector<int32_t> want, want_off, req;
    if ((int64_t)ptr_h[n] - 1 != nnz)
        local_rc = fail(SGM_ERR_BAD_ARG, "sgm_csr_create_dist: ptr(n+1)-1 = %lld but nnz_local = %lld", (long long)ptr_h[n] - 1, (long long)nnz);
    if (local_rc == SGM_OK) {
        halo_plan(nc, c0, nnz, node_h, lnode.data(), halo);
        local_rc = dist_plan(me, R, col_starts, halo, want, want_off, req);
    }
    if (local_rc != SGM_OK) {
        local_msg = g_err;
        if (R == 1) return local_rc;
        want.assign((size_t)R, 0);
        want[me] = -1;
        req.clear();
        halo.clear();
    }

    MatGuard g;
    sgm_mat A = g.A = new sgm_mat_s;
    A->fmt = SGM_FMT_CSR;
    A->nrow = (int32_t)row_starts[R];
    A->ncol = (int32_t)col_starts[R];
    A->nnz = nnz;
    A->comm = comm;
    A->row_starts.assign(row_starts, row_starts + R + 1);
    A->col_starts.assign(col_starts, col_starts + R + 1);
    A->halo_cols = halo;
    A->parts.resize(1);
    Part &p = A->parts[0];
    // Building the part can fail too (a row pointer array that is not monotone, device memory): that verdict has to travel like
    // the ones above -- a rank that returned here would leave its peers waiting in the all-gather below.
    auto build_local = [&]() -> int {
        if (where == SGM_DEVICE) {
            // values stay on the device; only the renumbered node array is re-uploaded
            int32_t *dnode = nullptr;
            SGM_TRY(dalloc(&dnode, (size_t)std::max<int64_t>(nnz, 1)));
            g.scratch.push_back(dnode);
            if (nnz) SGM_HIP(hipMemcpyAsync(dnode, lnode.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, st));
            return build_csr_part(p, n, nc, (int32_t)halo.size(), nnz, ptr, dnode, val, SGM_DEVICE);
        }
        return build_csr_part(p, n, nc, (int32_t)halo.size(), nnz, ptr_h, lnode.data(), val, SGM_HOST);
    };
    if (local_rc == SGM_OK) {       // (rejected rows: nothing is built; the rank only takes part in the all-gather that spreads the verdict)
        local_rc = build_local();
        if (local_rc != SGM_OK) {
            local_msg = g_err;
            if (R == 1) return local_rc;
            want.assign((size_t)R, 0);
            want[me] = -1;
            req.clear();
        }
    }
    p.row_begin = r0;
    if (local_rc == SGM_OK) set_interior_range(p, ptr_h, lnode.data());
    { std::vector<int32_t>().swap(hnode); std::vector<int32_t>().swap(lnode); }

    struct HbScope { int prev; HbScope() : prev(g_hb.phase) { hb_phase(HB_CREATE_DIST); } ~HbScope() { hb_phase(prev); } } hb_scope;
    if (R > 1) {
        // all ranks learn the full want matrix, then neighbours swap their request lists
        int32_t *d_want = nullptr, *d_all = nullptr, *d_req = nullptr;
        SGM_TRY(dalloc(&d_want, (size_t)R));
        g.scratch.push_back(d_want);
        SGM_TRY(dalloc(&d_all, (size_t)R * R));
        g.scratch.push_back(d_all);
        SGM_TRY(dalloc(&d_req, req.size()));
        g.scratch.push_back(d_req);
        SGM_HIP(hipMemcpyAsync(d_want, want.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
        SGM_NCCL(g_nccl.AllGather(d_want, d_all, (size_t)R, ncclInt32, (ncclComm_t)comm->nccl, st));
        std::vector<int32_t> all((size_t)R * R);
        SGM_HIP(hipMemcpyAsync(all.data(), d_all, all.size() * 4, hipMemcpyDeviceToHost, st));
        if (!req.empty()) SGM_HIP(hipMemcpyAsync(d_req, req.data(), req.size() * 4, hipMemcpyHostToDevice, st));
        SGM_HIP(hipStreamSynchronize(st));
        for (int q = 0; q < R; ++q)
            if (all[(size_t)me * R + q] != want[q]) return fail(SGM_ERR_RCCL, "sgm_csr_create_dist: all-gather returned a different want row");
        // the verdicts: a rank whose rows were rejected, or a request for more entries than a rank owns -- every rank sees the
        // same matrix, so every rank returns here or none does
        for (int q = 0; q < R; ++q)
            if (all[(size_t)q * R + q] < 0) {
                if (q == me) { g_err = local_msg; return local_rc; }
                return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_dist: rank %d rejected its rows (see its error message)", q);
            }
        for (int q = 0; q < R; ++q)
            for (int r = 0; r < R; ++r)
                if (all[(size_t)q * R + r] > col_starts[r + 1] - col_starts[r])
                    return fail(SGM_ERR_RCCL, "rank %d asks rank %d for %d entries of the %lld it owns", q, r, all[(size_t)q * R + r],
                                (long long)(col_starts[r + 1] - col_starts[r]));
        std::vector<NbrPlan> plan;
        dist_neighbors(me, R, all.data(), plan);
        for (const NbrPlan &pl : plan) {
            HaloNbr nb;
            nb.peer = pl.peer;
            nb.send_count = pl.send_count;
            nb.recv_count = pl.recv_count;
            nb.recv_offset = pl.recv_offset;
            p.nbrs.push_back(nb);                       // buffers below are owned by the part from here on
            if (pl.send_count) {
                SGM_TRY(dalloc(&p.nbrs.back().send_idx, (size_t)pl.send_count));
                SGM_TRY(dalloc(&p.nbrs.back().send_buf, (size_t)pl.send_count));
            }
        }
        SGM_NCCL(g_nccl.GroupStart());
        for (auto &nb : p.nbrs) {
            if (nb.recv_count)
                SGM_NCCL(g_nccl.Send(d_req + nb.recv_offset, nb.recv_count, ncclInt32, nb.peer,
                                     (ncclComm_t)comm->nccl, st));
            if (nb.send_count)
                SGM_NCCL(g_nccl.Recv(nb.send_idx, nb.send_count, ncclInt32, nb.peer, (ncclComm_t)comm->nccl, st));
        }
        SGM_NCCL(g_nccl.GroupEnd());
        SGM_HIP(hipStreamSynchronize(st));
    }
    *out = g.release();
    return SGM_OK;
}

/* ELLPACK rows of a partitioned matrix (ellpack_matvec_add, ellpack_matrices.f90:640-665, sums ALL max_d
 * slots of a row in order, padding included): held as CSR rows of fixed length max_d whose padding slots
 * are stored entries (value 0.0, column = the row's last neighbour), so the row sums -- and a 0 * Inf in a
 * padding slot -- come out exactly as the reference's.  node / val: (max_d, n_local) column-major like the
 * reference holds them, GLOBAL 1-based columns.  A slot of an empty row (node 0 in the reference, which then
 * reads x(0)) points at the row's own column. */
int sgm_ell_create_dist(sgm_mat *out, sgm_comm comm, const int64_t *row_starts, int32_t max_d, const int32_t *node,
                        const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || !comm || !row_starts || max_d < 0) return fail(SGM_ERR_BAD_ARG, "sgm_ell_create_dist: bad argument");
    const int64_t r0 = row_starts[comm->rank], r1 = row_starts[comm->rank + 1];
    const int32_t n = (int32_t)(r1 - r0);
    const int64_t nnz = (int64_t)n * max_d;
    if (nnz >= INT32_MAX) return fail(SGM_ERR_UNSUPPORTED, "sgm_ell_create_dist: n_local * max_d exceeds int32");
    if (nnz && (!node || !val)) return fail(SGM_ERR_BAD_ARG, "sgm_ell_create_dist: null arrays");
    std::vector<int32_t> hnode((size_t)std::max<int64_t>(nnz, 1)), hptr((size_t)n + 1);
    std::vector<double> hval((size_t)std::max<int64_t>(nnz, 1));
    if (nnz) {
        const hipMemcpyKind kind = where == SGM_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
        SGM_HIP(hipMemcpyAsync(hnode.data(), node, (size_t)nnz * 4, kind, g_rt.stream));
        SGM_HIP(hipMemcpyAsync(hval.data(), val, (size_t)nnz * 8, kind, g_rt.stream));
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
    }
    for (int32_t i = 0; i <= n; ++i) hptr[i] = 1 + i * max_d;
    // degrees(i), recovered from the padding the reference keeps (k_ell_degrees, sgm_mat.hip): what an ILDU(0) setup on these
    // rows goes by -- the reference's pattern pass and fill read the real entries only (ellpack_graphs.f90:310-369)
    std::vector<int32_t> hdeg((size_t)std::max(n, 1), 0);
    SGM_TRY(host_ell_degrees_host(n, max_d, hnode.data(), hdeg.data()));
    for (int32_t i = 0; i < n; ++i)
        for (int32_t k = 0; k < max_d; ++k)
            if (hnode[(size_t)i * max_d + k] <= 0) hnode[(size_t)i * max_d + k] = (int32_t)(r0 + i + 1);
    SGM_TRY(sgm_csr_create_dist(out, comm, row_starts, nnz, hptr.data(), hnode.data(), hval.data(), SGM_HOST));
    Part &p = (*out)->parts[0];
    SGM_TRY(dalloc(&p.edeg, (size_t)std::max(n, 1)));
    SGM_HIP(hipMemcpyAsync(p.edeg, hdeg.data(), (size_t)std::max(n, 1) * 4, hipMemcpyHostToDevice, g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

}  // extern "C"

namespace sgm {

struct Staged {
    double *dev = nullptr;
    bool owned = false;
    ~Staged() { if (owned) dfree(dev); }
};
int stage_in(Staged &s, const double *v, int64_t n, int where, bool copy);
int stage_out(const Staged &s, double *v, int64_t n, int where);

// A^T of a matrix distributed over processes, as ANOTHER distributed matrix: every stored entry
// (row j, column i, value) travels once to the rank that owns column i (counts by all-gather, then one
// grouped send/recv of three arrays per peer).  Ranks own ascending row blocks and send their entries in
// (row, slot) order, so the receiver's concatenation in rank order followed by a STABLE sort by column
// leaves every column's entries in global (row j, slot k) order -- the order the reference's scatter
// y(node(k)) += val(k) x(j) adds them in (cs_matrices.f90:627-647).  The product is then a row sum of
// A^T (chained onto y for matvec_t_add), with A^T's own halo exchange for the x entries of other ranks.
static int ensure_transpose_dist(sgm_mat A)
{
    if (A->T && !A->t_stale) return SGM_OK;
    if (A->T) { sgm_mat_destroy(A->T); A->T = nullptr; }
    const Part &p = A->parts[0];
    sgm_comm comm = A->comm;
    const int R = comm->nranks, me = comm->rank;
    const int64_t *rs = A->row_starts.data(), *cs = A->col_starts.data();      // A^T: rows by cs, x by rs
    const int64_t r0 = rs[me], c0 = cs[me];
    const int32_t n = p.n, nc = p.ncol_own;
    hipStream_t st = g_rt.stream;
    std::vector<int32_t> hptr((size_t)n + 1), hcol((size_t)std::max<int64_t>(p.nnz, 1));
    std::vector<double> hval((size_t)std::max<int64_t>(p.nnz, 1));
    SGM_TRY(csr_need_arrays(p));
    SGM_HIP(hipMemcpyAsync(hptr.data(), p.rowptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
    if (p.nnz) {
        SGM_HIP(hipMemcpyAsync(hcol.data(), p.col, (size_t)p.nnz * 4, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipMemcpyAsync(hval.data(), p.val, (size_t)p.nnz * 8, hipMemcpyDeviceToHost, st));
    }
    SGM_HIP(hipStreamSynchronize(st));
    csr_release_arrays(p);
    // entries bucketed by the owner of their column, each bucket in (row, slot) order
    std::vector<int32_t> cnt(R, 0), off(R + 1, 0), owner((size_t)std::max<int64_t>(p.nnz, 1));
    for (int64_t k = 0; k < p.nnz; ++k) {
        const int32_t c = hcol[k];
        const int64_t g0 = c < nc ? c0 + c : (int64_t)A->halo_cols[c - nc] - 1;      // global column, 0-based
        const int q = c < nc ? me : owner_of(cs, R, g0);
        owner[k] = q;
        hcol[k] = (int32_t)(g0 - cs[q]);                                           // column in its owner's numbering
        cnt[q]++;
    }
    for (int q = 0; q < R; ++q) off[q + 1] = off[q] + cnt[q];
    const size_t ns = (size_t)std::max<int64_t>(p.nnz, 1);
    std::vector<int32_t> si(ns), sj(ns), cur(off.begin(), off.end() - 1);
    std::vector<double> sv(ns);
    for (int32_t jl = 0; jl < n; ++jl)
        for (int32_t k = hptr[jl]; k < hptr[jl + 1]; ++k) {
            const int32_t d = cur[owner[k]]++;
            si[d] = hcol[k];
            sj[d] = (int32_t)(r0 + jl + 1);                                        // global row, 1-based: A^T's column
            sv[d] = hval[k];
        }
    // counts of every rank, then the entries
    struct Scratch { std::vector<void *> v; ~Scratch() { for (void *q : v) dfree(q); } } sc;
    auto dev = [&](size_t bytes, void **out) -> int {
        char *q = nullptr;
        SGM_TRY(dalloc(&q, bytes ? bytes : 1));
        sc.v.push_back(q);
        *out = q;
        return SGM_OK;
    };
    int32_t *d_cnt = nullptr, *d_all = nullptr;
    SGM_TRY(dev((size_t)R * 4, (void **)&d_cnt));
    SGM_TRY(dev((size_t)R * R * 4, (void **)&d_all));
    SGM_HIP(hipMemcpyAsync(d_cnt, cnt.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
    SGM_NCCL(g_nccl.AllGather(d_cnt, d_all, (size_t)R, ncclInt32, (ncclComm_t)comm->nccl, st));
    std::vector<int32_t> all((size_t)R * R);
    SGM_HIP(hipMemcpyAsync(all.data(), d_all, all.size() * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> roff(R + 1, 0);
    for (int q = 0; q < R; ++q) roff[q + 1] = roff[q] + all[(size_t)q * R + me];     // what rank q sends to me
    const int64_t nr = roff[R];
    // (the verdict is the same on every rank -- each sees all the counts --, so nobody is left waiting in the exchange below)
    for (int m = 0; m < R; ++m) {
        int64_t tot = 0;
        for (int q = 0; q < R; ++q) tot += all[(size_t)q * R + m];
        if (tot >= INT32_MAX) return fail(SGM_ERR_UNSUPPORTED, "matvec_t: the transposed row block of rank %d exceeds int32 entries", m);
    }
    int32_t *dsi = nullptr, *dsj = nullptr, *dri = nullptr, *drj = nullptr;
    double *dsv = nullptr, *drv = nullptr;
    SGM_TRY(dev(ns * 4, (void **)&dsi));
    SGM_TRY(dev(ns * 4, (void **)&dsj));
    SGM_TRY(dev(ns * 8, (void **)&dsv));
    SGM_TRY(dev((size_t)nr * 4, (void **)&dri));
    SGM_TRY(dev((size_t)nr * 4, (void **)&drj));
    SGM_TRY(dev((size_t)nr * 8, (void **)&drv));
    SGM_HIP(hipMemcpyAsync(dsi, si.data(), ns * 4, hipMemcpyHostToDevice, st));
    SGM_HIP(hipMemcpyAsync(dsj, sj.data(), ns * 4, hipMemcpyHostToDevice, st));
    SGM_HIP(hipMemcpyAsync(dsv, sv.data(), ns * 8, hipMemcpyHostToDevice, st));
    if (R > 1) {
        SGM_NCCL(g_nccl.GroupStart());
        for (int q = 0; q < R; ++q) {
            if (q == me) continue;
            if (cnt[q]) {
                SGM_NCCL(g_nccl.Send(dsi + off[q], (size_t)cnt[q], ncclInt32, q, (ncclComm_t)comm->nccl, st));
                SGM_NCCL(g_nccl.Send(dsj + off[q], (size_t)cnt[q], ncclInt32, q, (ncclComm_t)comm->nccl, st));
                SGM_NCCL(g_nccl.Send(dsv + off[q], (size_t)cnt[q], ncclFloat64, q, (ncclComm_t)comm->nccl, st));
            }
            const size_t rc = (size_t)(roff[q + 1] - roff[q]);
            if (rc) {
                SGM_NCCL(g_nccl.Recv(dri + roff[q], rc, ncclInt32, q, (ncclComm_t)comm->nccl, st));
                SGM_NCCL(g_nccl.Recv(drj + roff[q], rc, ncclInt32, q, (ncclComm_t)comm->nccl, st));
                SGM_NCCL(g_nccl.Recv(drv + roff[q], rc, ncclFloat64, q, (ncclComm_t)comm->nccl, st));
            }
        }
        SGM_NCCL(g_nccl.GroupEnd());
    }
    if (cnt[me]) {
        SGM_HIP(hipMemcpyAsync(dri + roff[me], dsi + off[me], (size_t)cnt[me] * 4, hipMemcpyDeviceToDevice, st));
        SGM_HIP(hipMemcpyAsync(drj + roff[me], dsj + off[me], (size_t)cnt[me] * 4, hipMemcpyDeviceToDevice, st));
        SGM_HIP(hipMemcpyAsync(drv + roff[me], dsv + off[me], (size_t)cnt[me] * 8, hipMemcpyDeviceToDevice, st));
    }
    std::vector<int32_t> ri((size_t)std::max<int64_t>(nr, 1)), rj((size_t)std::max<int64_t>(nr, 1));
    std::vector<double> rv((size_t)std::max<int64_t>(nr, 1));
    if (nr) {
        SGM_HIP(hipMemcpyAsync(ri.data(), dri, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipMemcpyAsync(rj.data(), drj, (size_t)nr * 4, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipMemcpyAsync(rv.data(), drv, (size_t)nr * 8, hipMemcpyDeviceToHost, st));
    }
    SGM_HIP(hipStreamSynchronize(st));
    // stable counting sort by local column = row of A^T
    std::vector<int32_t> tptr((size_t)nc + 1, 0), tnode((size_t)std::max<int64_t>(nr, 1));
    std::vector<double> tval((size_t)std::max<int64_t>(nr, 1));
    for (int64_t e = 0; e < nr; ++e) {
        if (ri[e] < 0 || ri[e] >= nc) return fail(SGM_ERR_RCCL, "matvec_t: received an entry for column %d of %d", ri[e], nc);
        tptr[ri[e] + 1]++;
    }
    for (int32_t i = 0; i < nc; ++i) tptr[i + 1] += tptr[i];
    {
        std::vector<int32_t> fill(tptr.begin(), tptr.end() - 1);
        for (int64_t e = 0; e < nr; ++e) {
            const int32_t d = fill[ri[e]]++;
            tnode[d] = rj[e];
            tval[d] = rv[e];
        }
    }
    for (auto &v : tptr) v += 1;                                                  // 1-based like the reference's ptr
    sgm_mat T = nullptr;
    SGM_TRY(sgm_csr_create_dist_rect(&T, comm, cs, rs, nr, tptr.data(), tnode.data(), tval.data(), SGM_HOST));
    if (!T->parts[0].xext) {
        const int rc = dalloc(&T->parts[0].xext, (size_t)T->parts[0].xlen() + 2);
        if (rc != SGM_OK) { sgm_mat_destroy(T); return rc; }
    }
    A->T = T;
    A->t_stale = false;
    return SGM_OK;
}

// ------------------------------------------------------------------ re-orderings and permutations over ranks (SURVEY 8(f4) x 8(e))
// The reference's breadth_first_search / greedy_coloring / greedy_color_ordering (permutations.f90:22-205) are sequential walks of
// ONE graph whose numbering depends on the visiting order; A%left_permute / right_permute (cs_matrices.f90:471-490) move whole rows
// / rename columns.  On a matrix distributed over ranks both are SETUP work, done the simple way: every rank receives the whole
// index structure (left_permute: and the values) once -- one grouped exchange, the matrix in host memory for the length of the
// call --, the ordering runs on every rank by the single-GPU passes (the same p everywhere, the reference's bits), a permutation
// cuts this rank's NEW rows out of the gathered arrays and rebuilds the handle's row block with sgm_csr_create_dist (same row
// partition, new halo plan).  right_permute needs no exchange at all.  Not for matrices near a rank's memory; what a solver needs
// per rank without any of this is ldu(reorder = "colour") (sgm_pc.hip).

int sgm_invalidate_transpose(sgm_mat A);          // sgm_layouts.hip

// A device -> host copy of index / value arrays that CHECKS what arrived: the 32-bit words are summed on the device (one atomic
// per workgroup) and on the host; a copy whose sum differs is waited for (hipDeviceSynchronize) and repeated, loudly.  Setup-path
// gathers only (a few per call).  Why: round 6 saw, on the shared GPU boxes under load, a host buffer read right after its
// device-to-host copy still holding its previous content -- in the stand-in transport (profiles/r06) and, it seems, here; a wrong
// index array is a wrong graph, silently.
__global__ void k_word_sum(const uint32_t *__restrict__ w, size_t nwords, unsigned long long *sum)
{
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) s += w[i];
    __shared__ unsigned long long red[kBlock];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = kBlock / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(sum, red[0]);
}
static int download_verified(void *host, const void *dev, size_t bytes, const char *what)
{
    if (!bytes) return SGM_OK;
    hipStream_t st = g_rt.stream;
    unsigned long long *dsum = nullptr, want = 0;
    SGM_TRY(dalloc(&dsum, 1));
    struct Tmp { unsigned long long *&p; ~Tmp() { dfree(p); } } tmp{dsum};
    SGM_HIP(hipMemsetAsync(dsum, 0, 8, st));
    const size_t nw = bytes / 4;
    hipLaunchKernelGGL(k_word_sum, dim3((unsigned)std::min<size_t>((nw + kBlock - 1) / kBlock, 1024)), dim3(kBlock), 0, st, (const uint32_t *)dev, nw, dsum);
    SGM_HIP(hipMemcpyAsync(&want, dsum, 8, hipMemcpyDeviceToHost, st));
    for (int attempt = 0; attempt < 4; ++attempt) {
        SGM_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipStreamSynchronize(st));
        unsigned long long got = 0;
        const uint32_t *hw = (const uint32_t *)host;
        for (size_t i = 0; i < nw; ++i) got += hw[i];
        if (got == want) return SGM_OK;
        fprintf(stderr, "[sigma_hip] WARNING: %s: a device-to-host copy of %zu bytes arrived with word sum %llx, the device holds %llx (attempt %d): "
                        "waiting for the device and copying again\n", what, bytes, got, want, attempt + 1);
        SGM_HIP(hipDeviceSynchronize());
    }
    return fail(SGM_ERR_HIP, "%s: a device-to-host copy does not arrive intact", what);
}

// this rank's rows as host arrays with GLOBAL 1-based columns
static int local_rows_global(sgm_mat A, std::vector<int32_t> &hptr, std::vector<int32_t> &hnode, std::vector<double> *hval)
{
    const Part &p = A->parts[0];
    hipStream_t st = g_rt.stream;
    const int32_t n = p.n, nc = p.ncol_own;
    const int64_t c0 = A->col_starts[A->comm->rank];
    hptr.assign((size_t)n + 1, 0);
    hnode.assign((size_t)std::max<int64_t>(p.nnz, 1), 0);
    if (hval) hval->assign((size_t)std::max<int64_t>(p.nnz, 1), 0.0);
    SGM_TRY(csr_need_arrays(p));
    int rc = download_verified(hptr.data(), p.rowptr, ((size_t)n + 1) * 4, "local rows (ptr)");
    if (rc == SGM_OK && p.nnz) rc = download_verified(hnode.data(), p.col, (size_t)p.nnz * 4, "local rows (node)");
    if (rc == SGM_OK && p.nnz && hval) rc = download_verified(hval->data(), p.val, (size_t)p.nnz * 8, "local rows (val)");
    SGM_HIP(hipStreamSynchronize(st));
    csr_release_arrays(p);
    SGM_TRY(rc);
    for (int64_t k = 0; k < p.nnz; ++k) {
        const int32_t c = hnode[(size_t)k];
        hnode[(size_t)k] = c < nc ? (int32_t)(c0 + c + 1) : A->halo_cols[(size_t)(c - nc)];
    }
    return SGM_OK;
}

// every rank gets the whole matrix: gptr (nrow + 1, 1-based), gnode (global 1-based columns), gval (optional), rows in global
// order, entries in stored order.  Row lengths, columns and values travel in one grouped send / recv per peer.
int gather_global_csr(sgm_mat A, std::vector<int32_t> &gptr, std::vector<int32_t> &gnode, std::vector<double> *gval)
{
    if (!A->comm || A->fmt != SGM_FMT_CSR || A->parts.size() != 1)
        return fail(SGM_ERR_UNSUPPORTED, "this operation needs a CSR matrix distributed over ranks (sgm_csr_create_dist)");
    sgm_comm comm = A->comm;
    const int R = comm->nranks, me = comm->rank;
    const int64_t *rs = A->row_starts.data();
    const int64_t ng = rs[R];
    hipStream_t st = g_rt.stream;
    std::vector<int32_t> hptr, hnode;
    std::vector<double> hval;
    SGM_TRY(local_rows_global(A, hptr, hnode, gval ? &hval : nullptr));
    const int32_t n = A->parts[0].n;
    const int64_t nnz = A->parts[0].nnz;
    struct Scratch { std::vector<void *> v; ~Scratch() { for (void *q : v) dfree(q); } } sc;
    auto dev = [&](size_t bytes, void **out) -> int {
        char *q = nullptr;
        SGM_TRY(dalloc(&q, bytes ? bytes : 1));
        sc.v.push_back(q);
        *out = q;
        return SGM_OK;
    };
    // every rank's entry count
    int32_t *d_cnt = nullptr, *d_all = nullptr;
    SGM_TRY(dev(4, (void **)&d_cnt));
    SGM_TRY(dev((size_t)R * 4, (void **)&d_all));
    // (every copy of this function goes through the library's stream and is waited for: a blocking copy on the null stream is
    //  not ordered against a non-blocking stream)
    const int32_t mine = (int32_t)nnz;
    SGM_HIP(hipMemcpyAsync(d_cnt, &mine, 4, hipMemcpyHostToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));
    SGM_NCCL(g_nccl.AllGather(d_cnt, d_all, 1, ncclInt32, (ncclComm_t)comm->nccl, st));
    std::vector<int32_t> cnt((size_t)R);
    SGM_HIP(hipMemcpyAsync(cnt.data(), d_all, (size_t)R * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    std::vector<int64_t> eoff((size_t)R + 1, 0);
    for (int q = 0; q < R; ++q) eoff[(size_t)q + 1] = eoff[(size_t)q] + cnt[(size_t)q];
    const int64_t nnzg = eoff[(size_t)R];
    if (nnzg >= INT32_MAX - 4) return fail(SGM_ERR_UNSUPPORTED, "the whole matrix has %lld entries: more than one rank can hold as int32 CSR", (long long)nnzg);
    int32_t *d_len = nullptr, *d_node = nullptr;
    double *d_val = nullptr;
    SGM_TRY(dev((size_t)ng * 4, (void **)&d_len));
    SGM_TRY(dev((size_t)nnzg * 4, (void **)&d_node));
    if (gval) SGM_TRY(dev((size_t)nnzg * 8, (void **)&d_val));
    std::vector<int32_t> len((size_t)std::max(n, 1));
    for (int32_t i = 0; i < n; ++i) len[(size_t)i] = hptr[(size_t)i + 1] - hptr[(size_t)i];
    if (n) SGM_HIP(hipMemcpyAsync(d_len + rs[me], len.data(), (size_t)n * 4, hipMemcpyHostToDevice, st));
    if (nnz) SGM_HIP(hipMemcpyAsync(d_node + eoff[(size_t)me], hnode.data(), (size_t)nnz * 4, hipMemcpyHostToDevice, st));
    if (nnz && gval) SGM_HIP(hipMemcpyAsync(d_val + eoff[(size_t)me], hval.data(), (size_t)nnz * 8, hipMemcpyHostToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));
    if (R > 1) {
        const int hb_prev = g_hb.phase;
        hb_phase(HB_CREATE_DIST);
        SGM_NCCL(g_nccl.GroupStart());
        for (int q = 0; q < R; ++q) {
            if (q == me) continue;
            const size_t nq = (size_t)(rs[q + 1] - rs[q]);
            if (n) SGM_NCCL(g_nccl.Send(d_len + rs[me], (size_t)n, ncclInt32, q, (ncclComm_t)comm->nccl, st));
            if (nq) SGM_NCCL(g_nccl.Recv(d_len + rs[q], nq, ncclInt32, q, (ncclComm_t)comm->nccl, st));
            if (nnz) SGM_NCCL(g_nccl.Send(d_node + eoff[(size_t)me], (size_t)nnz, ncclInt32, q, (ncclComm_t)comm->nccl, st));
            if (cnt[(size_t)q]) SGM_NCCL(g_nccl.Recv(d_node + eoff[(size_t)q], (size_t)cnt[(size_t)q], ncclInt32, q, (ncclComm_t)comm->nccl, st));
            if (gval) {
                if (nnz) SGM_NCCL(g_nccl.Send(d_val + eoff[(size_t)me], (size_t)nnz, ncclFloat64, q, (ncclComm_t)comm->nccl, st));
                if (cnt[(size_t)q]) SGM_NCCL(g_nccl.Recv(d_val + eoff[(size_t)q], (size_t)cnt[(size_t)q], ncclFloat64, q, (ncclComm_t)comm->nccl, st));
            }
        }
        SGM_NCCL(g_nccl.GroupEnd());
        SGM_HIP(hipStreamSynchronize(st));
        hb_phase(hb_prev);
    }
    std::vector<int32_t> glen((size_t)std::max<int64_t>(ng, 1));
    gnode.assign((size_t)std::max<int64_t>(nnzg, 1), 0);
    if (gval) gval->assign((size_t)std::max<int64_t>(nnzg, 1), 0.0);
    if (ng) SGM_TRY(download_verified(glen.data(), d_len, (size_t)ng * 4, "gathered row lengths"));
    if (nnzg) SGM_TRY(download_verified(gnode.data(), d_node, (size_t)nnzg * 4, "gathered columns"));
    if (nnzg && gval) SGM_TRY(download_verified(gval->data(), d_val, (size_t)nnzg * 8, "gathered values"));
    SGM_HIP(hipStreamSynchronize(st));
    gptr.assign((size_t)ng + 1, 1);
    for (int64_t i = 0; i < ng; ++i) gptr[(size_t)i + 1] = gptr[(size_t)i] + glen[(size_t)i];
    if (trace_on())          // (one line per gather and rank: what arrived, by rank of origin -- equal lines on every rank)
        for (int q = 0; q < R && q < 8; ++q) {
            unsigned long long h = 1469598103934665603ull;
            for (int64_t k = eoff[(size_t)q]; k < eoff[(size_t)q + 1]; ++k) { h ^= (unsigned)gnode[(size_t)k]; h *= 1099511628211ull; }
            fprintf(stderr, "[sigma_hip] gather_global_csr rank %d: rows %lld..%lld of rank %d, %d entries, columns' hash %016llx\n", me,
                    (long long)rs[q], (long long)rs[q + 1], q, cnt[(size_t)q], h);
        }
    if ((int64_t)gptr[(size_t)ng] - 1 != nnzg) return fail(SGM_ERR_RCCL, "gather_global_csr: %lld entries arrived, the row lengths say %lld", (long long)nnzg, (long long)gptr[(size_t)ng] - 1);
    return SGM_OK;
}

// the handle's row block replaced by `rows` (1-based local ptr, global 1-based columns): same communicator and row partition, new
// halo plan and kernel forms; the old block is freed
static int replace_dist_rows(sgm_mat A, const std::vector<int32_t> &lptr, const std::vector<int32_t> &lnode, const std::vector<double> &lval)
{
    sgm_mat N = nullptr;
    const int64_t nnz = (int64_t)lptr.back() - 1;
    SGM_TRY(sgm_csr_create_dist_rect(&N, A->comm, A->row_starts.data(), A->col_starts.data(), nnz, lptr.data(), lnode.data(), lval.data(), SGM_HOST));
    const bool had_xext = A->parts[0].xext != nullptr;
    std::swap(A->parts, N->parts);
    std::swap(A->halo_cols, N->halo_cols);
    A->nnz = N->nnz;
    sgm_mat_destroy(N);                  // (holds the old row block now)
    if (had_xext && !A->parts[0].xext) SGM_TRY(dalloc(&A->parts[0].xext, (size_t)A->parts[0].xlen() + 2));
    return sgm_invalidate_transpose(A);
}

// A%left_permute(p) (cs_matrices.f90:471-478: row i becomes row p(i), entries in their stored order) / A%right_permute(p)
// (:483-490: column j becomes p(j)) on a matrix distributed over ranks; p = the GLOBAL permutation (1-based, host), the same on
// every rank.  Collective.
int permute_dist(sgm_mat A, const int32_t *p, bool left)
{
    if (!A->comm || A->fmt != SGM_FMT_CSR || A->parts.size() != 1)
        return fail(SGM_ERR_UNSUPPORTED, "permutations over ranks: a CSR matrix made by sgm_csr_create_dist");
    const int R = A->comm->nranks, me = A->comm->rank;
    const int64_t ng = left ? A->row_starts[(size_t)R] : A->col_starts[(size_t)R];
    std::vector<int32_t> pinv((size_t)std::max<int64_t>(ng, 1), 0);
    for (int64_t i = 0; i < ng; ++i) {
        const int32_t t = p[i];
        if (t < 1 || t > ng || pinv[(size_t)t - 1]) return fail(SGM_ERR_BAD_ARG, "%s: p is not a permutation of 1..%lld (p(%lld) = %d)", left ? "left_permute" : "right_permute", (long long)ng, (long long)i + 1, t);
        pinv[(size_t)t - 1] = (int32_t)(i + 1);
    }
    std::vector<int32_t> lptr, lnode;
    std::vector<double> lval;
    if (!left) {                         // my rows, their columns renamed: nothing travels
        SGM_TRY(local_rows_global(A, lptr, lnode, &lval));
        for (auto &v : lptr) v += 1;
        const int64_t nnz = (int64_t)lptr.back() - 1;
        for (int64_t k = 0; k < nnz; ++k) lnode[(size_t)k] = p[(size_t)lnode[(size_t)k] - 1];
        return replace_dist_rows(A, lptr, lnode, lval);
    }
    std::vector<int32_t> gptr, gnode;
    std::vector<double> gval;
    SGM_TRY(gather_global_csr(A, gptr, gnode, &gval));
    const int64_t r0 = A->row_starts[(size_t)me], r1 = A->row_starts[(size_t)me + 1];
    lptr.assign((size_t)(r1 - r0) + 1, 1);
    int64_t nnz = 0;
    SGM_TRY(host_left_permute_rows_host((int32_t)ng, p, gptr.data(), gnode.data(), gval.data(), r0, r1, lptr.data(), nullptr, nullptr, 0, &nnz));
    lnode.assign((size_t)std::max<int64_t>(nnz, 1), 0);
    lval.assign((size_t)std::max<int64_t>(nnz, 1), 0.0);
    SGM_TRY(host_left_permute_rows_host((int32_t)ng, p, gptr.data(), gnode.data(), gval.data(), r0, r1, lptr.data(), lnode.data(), lval.data(), nnz, nullptr));
    return replace_dist_rows(A, lptr, lnode, lval);
}

// the graph of a matrix distributed over ranks as a single-GPU CSR matrix on THIS rank (values 0): what the reference's
// sequential orderings walk.  The caller destroys it.
int gathered_graph(sgm_mat A, sgm_mat *out)
{
    std::vector<int32_t> gptr, gnode;
    SGM_TRY(gather_global_csr(A, gptr, gnode, nullptr));
    const int64_t nnz = (int64_t)gptr.back() - 1;
    std::vector<double> zeros((size_t)std::max<int64_t>(nnz, 1), 0.0);
    return sgm_csr_create(out, A->nrow, A->ncol, nnz, gptr.data(), gnode.data(), zeros.data(), SGM_HOST);
}

int matvec_t_dist(sgm_mat A, const double *x, double *y, int where, bool add)
{
    // (an ELLPACK matrix distributed over ranks IS CSR rows of fixed length here -- sgm_ell_create_dist stores the padding slots
    //  as entries --, so ellpack_matvec_t_add's scatter over all max_d slots, ellpack_matrices.f90:670-693, is this same path)
    if (A->fmt != SGM_FMT_CSR) return fail(SGM_ERR_UNSUPPORTED, "matvec_t: a distributed matrix holds CSR rows (sgm_csr_create_dist / sgm_ell_create_dist)");
    SGM_TRY(ensure_transpose_dist(A));
    sgm_mat T = A->T;
    Part &pt = T->parts[0];
    const hipMemcpyKind kind = where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    SGM_HIP(hipMemcpyAsync(pt.xext, x, (size_t)pt.ncol_own * 8, kind, g_rt.stream));     // A's owned rows; A^T's halo is filled by the exchange
    Staged sy;
    SGM_TRY(stage_in(sy, y, pt.n, where, add));
    const double *xs[1] = {pt.xext};
    double *ys[1] = {sy.dev};
    SGM_TRY(spmv_parts(T, xs, ys, add, nullptr, nullptr, nullptr, 0x7fffffff, /*chain=*/add));
    SGM_TRY(stage_out(sy, y, pt.n, where));
    return finish();
}

}  // namespace sgm

extern "C" {

/* read back the exchange plan of a distributed / partitioned matrix (parity checks): for local
 * part `part`, neighbour `k`: peer, counts, offset and (optionally) the send list (0-based). */
int sgm_mat_halo_nbr(sgm_mat A, int32_t part, int32_t k, int32_t *n_nbrs, int32_t *peer, int32_t *send_count,
                     int32_t *recv_count, int32_t *recv_offset, int32_t *send_idx_host, int32_t capacity)
{
    if (!A || part < 0 || (size_t)part >= A->parts.size()) return fail(SGM_ERR_BAD_ARG, "sgm_mat_halo_nbr: bad argument");
    const Part &p = A->parts[part];
    if (n_nbrs) *n_nbrs = (int32_t)p.nbrs.size();
    if (k < 0) return SGM_OK;
    if ((size_t)k >= p.nbrs.size()) return fail(SGM_ERR_BAD_ARG, "sgm_mat_halo_nbr: neighbour %d of %zu", k, p.nbrs.size());
    const HaloNbr &nb = p.nbrs[k];
    if (peer) *peer = nb.peer;
    if (send_count) *send_count = nb.send_count;
    if (recv_count) *recv_count = nb.recv_count;
    if (recv_offset) *recv_offset = nb.recv_offset;
    if (send_idx_host && nb.send_count) {
        if (capacity < nb.send_count) return fail(SGM_ERR_BAD_ARG, "sgm_mat_halo_nbr: buffer too small");
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
        SGM_HIP(hipMemcpy(send_idx_host, nb.send_idx, (size_t)nb.send_count * 4, hipMemcpyDeviceToHost));
    }
    return SGM_OK;
}

}  // extern "C"
