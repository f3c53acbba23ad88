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

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>

namespace sgm {

int build_csr_part(Part &p, int32_t n, int32_t ncol_own, int32_t n_halo, int64_t nnz,
                   const int32_t *ptr1, const int32_t *node1, const double *val, int where);
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
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        g_nccl.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (g_nccl.h) break;
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

// ------------------------------------------------------------------ halo exchange
int halo_exchange(sgm_mat A, double *const *xext, hipStream_t st)
{
    if (A->comm) {
        Part &p = A->parts[0];
        if (p.nbrs.empty()) return SGM_OK;
        ncclComm_t comm = (ncclComm_t)A->comm->nccl;
        for (auto &nb : p.nbrs)
            if (nb.send_count)
                hipLaunchKernelGGL(k_gather, dim3((nb.send_count + kBlock - 1) / kBlock), dim3(kBlock), 0,
                                   st, nb.send_buf, (const double *)xext[0], nb.send_idx, nb.send_count);
        SGM_NCCL(g_nccl.GroupStart());
        for (auto &nb : p.nbrs) {
            if (nb.send_count) SGM_NCCL(g_nccl.Send(nb.send_buf, nb.send_count, ncclFloat64, nb.peer, comm, st));
            if (nb.recv_count)
                SGM_NCCL(g_nccl.Recv(xext[0] + p.ncol_own + nb.recv_offset, nb.recv_count, ncclFloat64,
                                     nb.peer, comm, st));
        }
        SGM_NCCL(g_nccl.GroupEnd());
        return SGM_OK;
    }
    // in-process partitions: the sender's list is gathered straight into the peer's halo
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

static double **g_slot_ptrs_dev = nullptr;
static size_t g_slot_ptrs_cap = 0;

int allreduce_slots(sgm_mat A, double *const *slot_ptrs, int count)
{
    hipStream_t st = g_rt.stream;
    if (A->comm) {
        if (A->comm->nranks == 1) return SGM_OK;
        SGM_NCCL(g_nccl.AllReduce(slot_ptrs[0], slot_ptrs[0], (size_t)count, ncclFloat64, ncclSum,
                                  (ncclComm_t)A->comm->nccl, st));
        return SGM_OK;
    }
    const size_t P = A->parts.size();
    if (P <= 1) return SGM_OK;
    if (count > 64) return fail(SGM_ERR_BAD_ARG, "allreduce_slots: count %d > 64", count);
    // a fresh pointer table per call keeps in-flight calls independent (tiny, test-only path)
    if (g_slot_ptrs_cap < P) {
        dfree(g_slot_ptrs_dev);
        SGM_TRY(dalloc(&g_slot_ptrs_dev, P * 64));
        g_slot_ptrs_cap = P;
    }
    static size_t ring = 0;
    double **tab = g_slot_ptrs_dev + (ring++ % 64) * P;
    SGM_HIP(hipMemcpyAsync(tab, slot_ptrs, P * sizeof(double *), hipMemcpyHostToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));   // slot_ptrs is a host temporary
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(64), 0, st, (double *const *)tab, (int)P, count);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// ------------------------------------------------------------------ host planning
// Sorted unique list of the non-owned columns + renumbering to [owned | halo].
static void halo_plan(int32_t n_own, int64_t col_begin, int64_t nnz, const int32_t *node,
                      int32_t *node_local, std::vector<int32_t> &halo)
{
    const int64_t lo = col_begin + 1, hi = col_begin + n_own;     // owned 1-based range
    halo.clear();
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t c = node[k];
        if (c < lo || c > hi) halo.push_back((int32_t)c);
    }
    std::sort(halo.begin(), halo.end());
    halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t c = node[k];
        if (c >= lo && c <= hi) {
            node_local[k] = (int32_t)(c - col_begin);
        } else {
            const auto it = std::lower_bound(halo.begin(), halo.end(), (int32_t)c);
            node_local[k] = n_own + 1 + (int32_t)(it - halo.begin());
        }
    }
}

static int owner_of(const int64_t *row_starts, int nparts, int64_t col0 /*0-based*/)
{
    const int64_t *it = std::upper_bound(row_starts, row_starts + nparts + 1, col0);
    return (int)(it - row_starts) - 1;
}

}  // namespace sgm

using namespace sgm;

extern "C" {

int sgm_halo_plan_host(int32_t n_own, int64_t col_begin, int64_t nnz, const int32_t *node,
                       int32_t *node_local, int32_t *halo_cols, int32_t *n_halo)
{
    if (n_own < 0 || nnz < 0 || (nnz && (!node || !node_local)) || !n_halo)
        return fail(SGM_ERR_BAD_ARG, "sgm_halo_plan_host: bad argument");
    std::vector<int32_t> halo;
    halo_plan(n_own, col_begin, nnz, node, node_local, halo);
    *n_halo = (int32_t)halo.size();
    if (halo_cols) std::copy(halo.begin(), halo.end(), halo_cols);
    return SGM_OK;
}

int sgm_csr_create_partitioned(sgm_mat *out, int32_t nparts, const int64_t *row_starts, int32_t nrow,
                               int32_t ncol, int64_t nnz, const int32_t *ptr, const int32_t *node,
                               const double *val)
{
    SGM_TRY(require_init());
    if (!out || nparts < 1 || !row_starts || !ptr || nrow != ncol || row_starts[0] != 0 ||
        row_starts[nparts] != nrow)
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_partitioned: bad argument (square matrices only)");
    sgm_mat A = new sgm_mat_s;
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
        int rc = build_csr_part(p, n, n, (int32_t)halos[ip].size(), k1 - k0, lptr.data(), lnode.data(),
                                val + k0, SGM_HOST);
        if (rc == SGM_OK) rc = dalloc(&p.xext, (size_t)p.xlen());
        if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
        p.row_begin = r0;
        set_interior_range(p, lptr.data(), lnode.data());
    }
    // send lists: part q sends to part p the entries of p's halo that q owns, in p's halo order
    for (int ip = 0; ip < nparts; ++ip) {
        const auto &h = halos[ip];
        size_t a = 0;
        while (a < h.size()) {
            const int q = owner_of(row_starts, nparts, (int64_t)h[a] - 1);
            size_t b = a;
            std::vector<int32_t> idx;
            while (b < h.size() && owner_of(row_starts, nparts, (int64_t)h[b] - 1) == q) {
                idx.push_back((int32_t)(h[b] - 1 - row_starts[q]));
                ++b;
            }
            HaloNbr nb;
            nb.peer = ip;                       // stored on the SENDER q
            nb.send_count = (int32_t)idx.size();
            nb.recv_offset = (int32_t)a;        // offset in the receiver's halo region
            int rc = dalloc(&nb.send_idx, idx.size());
            if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
            SGM_HIP(hipMemcpy(nb.send_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
            A->parts[q].nbrs.push_back(nb);
            a = b;
        }
    }
    *out = A;
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
    *out = h;
    return SGM_OK;
}

int sgm_comm_destroy(sgm_comm c)
{
    if (!c) return SGM_OK;
    if (c->nccl && g_nccl.CommDestroy) (void)g_nccl.CommDestroy((ncclComm_t)c->nccl);
    delete c;
    return SGM_OK;
}

int sgm_csr_create_dist(sgm_mat *out, sgm_comm comm, const int64_t *row_starts, int64_t nnz,
                        const int32_t *ptr, const int32_t *node, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || !comm || !row_starts || !ptr) return fail(SGM_ERR_BAD_ARG, "sgm_csr_create_dist: bad argument");
    const int R = comm->nranks, me = comm->rank;
    const int64_t r0 = row_starts[me], r1 = row_starts[me + 1];
    const int32_t n = (int32_t)(r1 - r0);
    hipStream_t st = g_rt.stream;

    // the index work runs on the host (same code as sgm_halo_plan_host)
    std::vector<int32_t> hnode, lnode((size_t)std::max<int64_t>(nnz, 1)), hptr;
    const int32_t *node_h = node, *ptr_h = ptr;
    if (where == SGM_DEVICE) {
        hnode.resize((size_t)std::max<int64_t>(nnz, 1));
        hptr.resize((size_t)n + 1);
        SGM_HIP(hipMemcpy(hnode.data(), node, (size_t)nnz * 4, hipMemcpyDeviceToHost));
        SGM_HIP(hipMemcpy(hptr.data(), ptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost));
        node_h = hnode.data();
        ptr_h = hptr.data();
    }
    std::vector<int32_t> halo;
    halo_plan(n, r0, nnz, node_h, lnode.data(), halo);

    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_CSR;
    A->nrow = (int32_t)row_starts[R];
    A->ncol = A->nrow;
    A->nnz = nnz;
    A->comm = comm;
    A->parts.resize(1);
    Part &p = A->parts[0];
    int rc;
    if (where == SGM_DEVICE) {
        // values stay on the device; only the renumbered node array is re-uploaded
        int32_t *dnode = nullptr;
        SGM_TRY(dalloc(&dnode, (size_t)std::max<int64_t>(nnz, 1)));
        SGM_HIP(hipMemcpy(dnode, lnode.data(), (size_t)nnz * 4, hipMemcpyHostToDevice));
        rc = build_csr_part(p, n, n, (int32_t)halo.size(), nnz, ptr, dnode, val, SGM_DEVICE);
        dfree(dnode);
    } else {
        rc = build_csr_part(p, n, n, (int32_t)halo.size(), nnz, ptr_h, lnode.data(), val, SGM_HOST);
    }
    if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
    p.row_begin = r0;
    set_interior_range(p, ptr_h, lnode.data());
    { std::vector<int32_t>().swap(hnode); std::vector<int32_t>().swap(lnode); }

    // who needs what: want[q] = number of my halo entries owned by rank q
    std::vector<int32_t> want(R, 0), want_off(R + 1, 0);
    for (int32_t c : halo) want[owner_of(row_starts, R, (int64_t)c - 1)]++;
    for (int q = 0; q < R; ++q) want_off[q + 1] = want_off[q] + want[q];
    if (R > 1) {
        // all ranks learn the full want matrix, then neighbours swap index lists
        int32_t *d_want = nullptr, *d_all = nullptr;
        SGM_TRY(dalloc(&d_want, (size_t)R));
        SGM_TRY(dalloc(&d_all, (size_t)R * R));
        SGM_HIP(hipMemcpy(d_want, want.data(), (size_t)R * 4, hipMemcpyHostToDevice));
        SGM_NCCL(g_nccl.AllGather(d_want, d_all, (size_t)R, ncclInt32, (ncclComm_t)comm->nccl, st));
        SGM_HIP(hipStreamSynchronize(st));
        std::vector<int32_t> all((size_t)R * R);
        SGM_HIP(hipMemcpy(all.data(), d_all, all.size() * 4, hipMemcpyDeviceToHost));
        dfree(d_want);
        dfree(d_all);
        // local indices (0-based, in the owner's numbering) of the entries I want from each owner
        std::vector<int32_t> req(halo.size());
        for (size_t t = 0; t < halo.size(); ++t) {
            const int q = owner_of(row_starts, R, (int64_t)halo[t] - 1);
            req[t] = (int32_t)(halo[t] - 1 - row_starts[q]);
        }
        int32_t *d_req = nullptr;
        SGM_TRY(dalloc(&d_req, req.size()));
        SGM_HIP(hipMemcpy(d_req, req.data(), req.size() * 4, hipMemcpyHostToDevice));
        for (int q = 0; q < R; ++q) {
            const int32_t they_want = all[(size_t)q * R + me];   // rank q wants this many of mine
            if (q == me || (!they_want && !want[q])) continue;
            HaloNbr nb;
            nb.peer = q;
            nb.send_count = they_want;
            nb.recv_count = want[q];
            nb.recv_offset = want_off[q];
            if (they_want) {
                SGM_TRY(dalloc(&nb.send_idx, (size_t)they_want));
                SGM_TRY(dalloc(&nb.send_buf, (size_t)they_want));
            }
            p.nbrs.push_back(nb);
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
        dfree(d_req);
    }
    *out = A;
    return SGM_OK;
}

}  // extern "C"
