// Preconditioners for gfx950: Jacobi (jacobi_solvers.f90:37-81) and ILDU(0)
// (ldu_solvers.f90:95-176, :208-265, :275-440).
//
// Jacobi: idiag(i) = 1/A(i,i) is extracted on the device by a row scan (the reference
// calls A%get_value(i,i) per row, cs_matrices.f90:709-724); apply is one elementwise pass.
//
// ILDU(0): the factorisation runs once on the HOST (the reference's algorithm is a
// sequential IKJ sweep built on get/set/add_value row scans; it is setup, not the hot
// path) and keeps the reference's arithmetic order, so L-I, D, U-I are bit-identical.
// The APPLY is the hot part: x=b ; (I+L)^-1 ; x/D ; (I+U)^-1, each triangular solve a
// row recurrence (ldu_solvers.f90:227-236).  Rows are grouped into dependency LEVELS at
// setup; rows of one level are independent, each lane does its row's
// z = z - val(k)*x(node(k)) left to right, so the result is bit-identical to the
// sequential sweep.  The solve runs in "position space": vectors are permuted into level
// order (xp[pos]), every row is a 64-byte record {count, first 4 (dependency position, value)}
// so that ONE independent load brings a row and can be issued a level ahead, and
// dependencies are positions.  Wide levels get one launch each; runs of narrow levels
// (<= 4096 rows) are walked by ONE 1024-thread workgroup that keeps the last 8192 results in
// an LDS ring: a level then costs LDS reads + a barrier instead of four dependent global
// round trips (3.9 us -> see DESIGN.md).
#include "sgm_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace sgm {
struct Staged {
    double *dev = nullptr;
    bool owned = false;
    ~Staged() { if (owned) dfree(dev); }
};
int stage_in(Staged &s, const double *v, int64_t n, int where, bool copy);
int stage_out(const Staged &s, double *v, int64_t n, int where);
}  // namespace sgm
using namespace sgm;

namespace {

constexpr int kTrsvBlock = 1024;
constexpr int kNarrow = 4096;        // levels with <= this many rows are walked by one workgroup (4 rows per lane)
constexpr int kRing = 8192;          // LDS ring of recent results (64 KiB): covers two narrow levels
constexpr int kInline = 4;           // dependencies stored inside the row record

struct TrsvRec {                     // one row of a triangular factor, in level order (64 bytes)
    int32_t cnt, k0;                 // entries of the row; offset of its entries in pq / pv
    int32_t q[kInline];              // position (in level order) of the first dependencies
    double v[kInline];               // their values
    int32_t pad[2];
};
static_assert(sizeof(TrsvRec) == 64, "TrsvRec is one 64-byte record");

struct TriFactor {                   // strictly triangular factor on the device, level order
    int32_t *order = nullptr;        // device: pos -> row
    TrsvRec *recs = nullptr;         // device: n records
    int32_t *pq = nullptr;           // device: dependency positions of ALL entries, rows in level order
    double *pv = nullptr;            // device: their values
    int32_t *level_ptr_dev = nullptr;
    std::vector<int32_t> level_ptr;  // host: offsets into the level order per level
    std::vector<int32_t> h_order, h_pos, h_src;      // host: pos -> row, row -> pos, level-order entry -> factor entry
    std::vector<TrsvRec> h_recs;
    std::vector<int32_t> h_pq;
    struct Launch { int32_t l0, l1; bool narrow; };
    std::vector<Launch> schedule;
};

struct PartPC {
    double *idiag = nullptr;
};

}  // namespace

struct sgm_pc_s {
    int kind = 0;
    int32_t n = 0;
    std::vector<PartPC> parts;       // jacobi
    // ildu (single part)
    TriFactor L, U;
    double *D = nullptr;
    double *xpL = nullptr, *xpU = nullptr, *Dp = nullptr;   // level-order work vectors, D in U's level order
    int32_t *mapLU = nullptr;                                // U position -> L position of the same row
    std::vector<int32_t> hLptr, hLnode, hUptr, hUnode;      // 1-based, as the reference holds them
    std::vector<double> hLval, hUval, hD, hidiag;
};

namespace {

// ------------------------------------------------------------------------------ kernels
__global__ void k_jacobi_setup_csr(int32_t n, const int32_t *__restrict__ rowptr,
                                   const int32_t *__restrict__ col, const double *__restrict__ val,
                                   double *__restrict__ idiag)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double z = 0.0;                                   // get_value: 0 when the entry is absent
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (col[k] == i) z = val[k];
    idiag[i] = 1.0 / z;
}
__global__ void k_jacobi_setup_ell(int32_t n, int32_t max_d, const int32_t *__restrict__ ecol,
                                   const double *__restrict__ eval, double *__restrict__ idiag)
{
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // ellpack get_value scans the first degrees(i) slots (ellpack_matrices.f90:232-235);
    // padding repeats the last real neighbour with val 0, real neighbours are unique, so
    // the FIRST hit is the real slot.
    double z = 0.0;
    for (int32_t k = 0; k < max_d; ++k)
        if (ecol[(int64_t)k * n + i] == i) { z = eval[(int64_t)k * n + i]; break; }
    idiag[i] = 1.0 / z;
}
__global__ void k_scale_by(int64_t n, const double *__restrict__ d, const double *__restrict__ r,
                           double *__restrict__ z, const int *flag)
{
    if (flag && *flag) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) z[i] = d[i] * r[i];       // x = idiag * b
}
__global__ void k_div_by(int64_t n, const double *__restrict__ d, double *__restrict__ x, const int *flag)
{
    if (flag && *flag) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = x[i] / d[i];       // x = x / D
}
__global__ void k_copy(int64_t n, const double *__restrict__ s, double *__restrict__ d, const int *flag)
{
    if (flag && *flag) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = s[i];
}

__global__ void k_perm_gather(int64_t n, double *__restrict__ xp, const double *__restrict__ src,
                              const int32_t *__restrict__ order, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) xp[p] = src[order[p]];            // x = b, in level order
}
__global__ void k_lu_transition(int64_t n, double *__restrict__ xpU, const double *__restrict__ xpL,
                                const int32_t *__restrict__ mapLU, const double *__restrict__ Dp, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) xpU[p] = xpL[mapLU[p]] / Dp[p];   // x = x / D, re-ordered for the U sweep
}
__global__ void k_perm_scatter(int64_t n, double *__restrict__ dst, const double *__restrict__ xp,
                               const int32_t *__restrict__ order, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) dst[order[p]] = xp[p];
}

// one wide level: one lane per row, dependencies read from global memory
__global__ void k_trsv_wide(const TrsvRec *__restrict__ recs, const int32_t *__restrict__ pq,
                            const double *__restrict__ pv, int32_t begin, int32_t end, double *xp, const int *flag)
{
    if (flag && *flag) return;
    const int32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    const TrsvRec r = recs[p];
    double z = xp[p];
#pragma unroll
    for (int j = 0; j < kInline; ++j)
        if (j < r.cnt) z = z - r.v[j] * xp[r.q[j]];
    for (int32_t k = r.k0 + kInline; k < r.k0 + r.cnt; ++k) z = z - pv[k] * xp[pq[k]];
    xp[p] = z;
}

// a run of narrow levels [l0, l1) walked by ONE workgroup.  The records and right-hand sides of
// level l+1 are requested before level l is computed (independent loads); results of the
// current run live in an LDS ring indexed by position, so the dependencies of the next level
// are LDS reads; anything older than the ring (or produced before this run) is read from xp,
// which the per-level workgroup fence + barrier keeps valid.
__global__ __launch_bounds__(kTrsvBlock) void k_trsv_walk(const TrsvRec *__restrict__ recs,
                                                          const int32_t *__restrict__ pq,
                                                          const double *__restrict__ pv,
                                                          const int32_t *__restrict__ level_ptr, int32_t l0,
                                                          int32_t l1, double *xp, const int *flag)
{
    __shared__ double ring[kRing];
    if (flag && *flag) return;
    constexpr int RPT = kNarrow / kTrsvBlock;
    const int32_t base = level_ptr[l0];
    TrsvRec pre[RPT];
    double z0pre[RPT];
    auto fetch = [&](int32_t l) {
        const int32_t b = level_ptr[l], e = level_ptr[l + 1];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int32_t p = b + threadIdx.x + r * kTrsvBlock;
            if (p < e) {
                pre[r] = recs[p];
                z0pre[r] = xp[p];          // the right-hand side entry: only this row ever writes it
            }
        }
    };
    fetch(l0);
    int32_t fpos = base;      // every position < fpos is visible in xp (written before a workgroup fence)
    for (int32_t l = l0; l < l1; ++l) {
        const int32_t b = level_ptr[l], e = level_ptr[l + 1];
        if (e - fpos > kRing) {
            // the ring is about to lose positions that were never fenced: make all stores of this
            // run visible in xp first (rare: once per ~2 narrow levels at most, usually far less)
            __threadfence_block();
            __syncthreads();
            fpos = b;
        }
        TrsvRec cur[RPT];
        double z0[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) { cur[r] = pre[r]; z0[r] = z0pre[r]; }
        if (l + 1 < l1) fetch(l + 1);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int32_t p = b + threadIdx.x + r * kTrsvBlock;
            if (p >= e) continue;
            double z = z0[r];
#pragma unroll
            for (int j = 0; j < kInline; ++j)
                if (j < cur[r].cnt) {
                    const int32_t q = cur[r].q[j];
                    const double xv = q >= fpos ? ring[q & (kRing - 1)] : xp[q];
                    z = z - cur[r].v[j] * xv;
                }
            for (int32_t k = cur[r].k0 + kInline; k < cur[r].k0 + cur[r].cnt; ++k) {
                const int32_t q = pq[k];
                const double xv = q >= fpos ? ring[q & (kRing - 1)] : xp[q];
                z = z - pv[k] * xv;
            }
            ring[p & (kRing - 1)] = z;
            xp[p] = z;                       // drains in the background; readers use the ring
        }
        // level barrier on the LDS ring only: the global stores above stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// --------------------------------------------------------------------- host factorisation
// Row-scan accessors with the reference's semantics (cs_matrices.f90:709-724, :840-895).
struct HostCsr {
    std::vector<int32_t> *ptr, *node;
    std::vector<double> *val;
    double get(int32_t i, int32_t j) const
    {
        double z = 0.0;
        for (int32_t k = (*ptr)[i - 1]; k < (*ptr)[i]; ++k)
            if ((*node)[k - 1] == j) z = (*val)[k - 1];
        return z;
    }
    void set(int32_t i, int32_t j, double z)
    {
        for (int32_t k = (*ptr)[i - 1]; k < (*ptr)[i]; ++k)
            if ((*node)[k - 1] == j) (*val)[k - 1] = z;
    }
    void add(int32_t i, int32_t j, double z)
    {
        for (int32_t k = (*ptr)[i - 1]; k < (*ptr)[i]; ++k)
            if ((*node)[k - 1] == j) (*val)[k - 1] = (*val)[k - 1] + z;
    }
};

// incomplete_ldu_sparsity_pattern, level 0 (ldu_solvers.f90:397-440): entries of A in
// stored order; i>j -> L, j>i -> U.
void ildu_pattern(sgm_pc pc, int32_t n, const std::vector<int32_t> &ptr, const std::vector<int32_t> &node)
{
    pc->hLptr.assign(n + 1, 1);
    pc->hUptr.assign(n + 1, 1);
    pc->hLnode.clear();
    pc->hUnode.clear();
    for (int32_t i = 1; i <= n; ++i) {
        for (int32_t k = ptr[i - 1]; k < ptr[i]; ++k) {
            const int32_t j = node[k - 1];
            if (i > j) pc->hLnode.push_back(j);
            if (j > i) pc->hUnode.push_back(j);
        }
        pc->hLptr[i] = (int32_t)pc->hLnode.size() + 1;
        pc->hUptr[i] = (int32_t)pc->hUnode.size() + 1;
    }
}

// sparse_static_pattern_ldu_factorization (ldu_solvers.f90:275-387), same statement order.
void ildu_factor(sgm_pc pc, int32_t n, const std::vector<int32_t> &ptr, const std::vector<int32_t> &node,
                 const std::vector<double> &val)
{
    pc->hLval.assign(pc->hLnode.size(), 0.0);
    pc->hUval.assign(pc->hUnode.size(), 0.0);
    pc->hD.assign(n, 0.0);
    HostCsr L{&pc->hLptr, &pc->hLnode, &pc->hLval}, U{&pc->hUptr, &pc->hUnode, &pc->hUval};
    std::vector<double> &D = pc->hD;
    for (int32_t i = 1; i <= n; ++i)
        for (int32_t k = ptr[i - 1]; k < ptr[i]; ++k) {
            const int32_t j = node[k - 1];
            if (i > j) L.set(i, j, val[k - 1]);
            else if (j > i) U.set(i, j, val[k - 1]);
            else D[i - 1] = val[k - 1];
        }
    for (int32_t i = 1; i <= n; ++i) {
        const int32_t lb = pc->hLptr[i - 1] - 1, dl = pc->hLptr[i] - pc->hLptr[i - 1];
        const int32_t ub = pc->hUptr[i - 1] - 1, du = pc->hUptr[i] - pc->hUptr[i - 1];
        for (int32_t a = 0; a < dl; ++a) {
            const int32_t k = pc->hLnode[lb + a];
            double Lik = L.get(i, k);
            const double Uki = U.get(k, i);
            L.set(i, k, Lik / D[k - 1]);
            Lik = Lik / D[k - 1];
            for (int32_t c = 0; c < dl; ++c) {
                const int32_t j = pc->hLnode[lb + c];
                if (j > k) {
                    const double Ukj = U.get(k, j);
                    L.add(i, j, -Lik * D[k - 1] * Ukj);
                }
            }
            D[i - 1] = D[i - 1] - Lik * D[k - 1] * Uki;
            for (int32_t c = 0; c < du; ++c) {
                const int32_t j = pc->hUnode[ub + c];
                const double Ukj = U.get(k, j);
                U.add(i, j, -Lik * D[k - 1] * Ukj);
            }
        }
        for (int32_t c = 0; c < du; ++c) {
            const int32_t k = pc->hUnode[ub + c];
            const double Uik = U.get(i, k);
            U.set(i, k, Uik / D[i - 1]);
        }
    }
}

void free_tri(TriFactor &T)
{
    dfree(T.order); dfree(T.recs); dfree(T.pq); dfree(T.pv); dfree(T.level_ptr_dev);
    T = TriFactor();
}

// upload a strictly triangular factor in level order.  lower: rows depend on smaller rows
// (forward sweep 1..n); upper: on larger rows (backward sweep n..1).
int upload_tri(TriFactor &T, int32_t n, const std::vector<int32_t> &ptr1, const std::vector<int32_t> &node1,
               const std::vector<double> &val, bool lower, bool pattern_changed)
{
    const size_t nnz = node1.size();
    if (pattern_changed) {
        free_tri(T);
        std::vector<int32_t> level(n, 0);
        int32_t nlev = 0;
        auto visit = [&](int32_t i) {
            int32_t lv = 0;
            for (int32_t k = ptr1[i] - 1; k < ptr1[i + 1] - 1; ++k) lv = std::max(lv, level[node1[k] - 1] + 1);
            level[i] = lv;
            nlev = std::max(nlev, lv + 1);
        };
        if (lower) for (int32_t i = 0; i < n; ++i) visit(i);
        else for (int32_t i = n - 1; i >= 0; --i) visit(i);
        T.level_ptr.assign(nlev + 1, 0);
        for (int32_t i = 0; i < n; ++i) T.level_ptr[level[i] + 1]++;
        for (int32_t l = 0; l < nlev; ++l) T.level_ptr[l + 1] += T.level_ptr[l];
        T.h_order.assign(std::max(n, 1), 0);
        T.h_pos.assign(std::max(n, 1), 0);
        std::vector<int32_t> cursor(T.level_ptr.begin(), T.level_ptr.end() - 1);
        for (int32_t i = 0; i < n; ++i) {
            const int32_t p = cursor[level[i]]++;
            T.h_order[p] = i;
            T.h_pos[i] = p;
        }
        // rows in level order: dependency POSITIONS in stored order
        T.h_recs.assign(std::max(n, 1), TrsvRec());
        T.h_pq.assign(std::max<size_t>(nnz, 1), 0);
        T.h_src.assign(std::max<size_t>(nnz, 1), 0);
        int32_t kk = 0;
        for (int32_t p = 0; p < n; ++p) {
            const int32_t i = T.h_order[p];
            TrsvRec &r = T.h_recs[p];
            r.cnt = ptr1[i + 1] - ptr1[i];
            r.k0 = kk;
            for (int32_t k = ptr1[i] - 1; k < ptr1[i + 1] - 1; ++k, ++kk) {
                T.h_pq[kk] = T.h_pos[node1[k] - 1];
                T.h_src[kk] = k;
                if (kk - r.k0 < kInline) r.q[kk - r.k0] = T.h_pq[kk];
            }
        }
        // schedule: wide levels alone, runs of narrow levels together
        static const int narrow = getenv("SGM_TRSV_NARROW") ? std::min(atoi(getenv("SGM_TRSV_NARROW")), kNarrow) : kNarrow;
        for (int32_t l = 0; l < nlev;) {
            const int32_t sz = T.level_ptr[l + 1] - T.level_ptr[l];
            if (sz > narrow) { T.schedule.push_back({l, l + 1, false}); ++l; continue; }
            int32_t e = l;
            while (e < nlev && T.level_ptr[e + 1] - T.level_ptr[e] <= narrow) ++e;
            T.schedule.push_back({l, e, true});
            l = e;
        }
        SGM_TRY(dalloc(&T.order, (size_t)n));
        SGM_TRY(dalloc(&T.recs, (size_t)n));
        SGM_TRY(dalloc(&T.pq, nnz));
        SGM_TRY(dalloc(&T.pv, nnz));
        SGM_TRY(dalloc(&T.level_ptr_dev, T.level_ptr.size()));
        if (n) SGM_HIP(hipMemcpy(T.order, T.h_order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        if (nnz) SGM_HIP(hipMemcpy(T.pq, T.h_pq.data(), nnz * 4, hipMemcpyHostToDevice));
        SGM_HIP(hipMemcpy(T.level_ptr_dev, T.level_ptr.data(), T.level_ptr.size() * 4, hipMemcpyHostToDevice));
    }
    // values (every setup): level-order copy + the inline part of the records
    std::vector<double> hv(std::max<size_t>(nnz, 1));
    for (size_t kk = 0; kk < nnz; ++kk) hv[kk] = val[T.h_src[kk]];
    for (int32_t p = 0; p < n; ++p) {
        TrsvRec &r = T.h_recs[p];
        for (int j = 0; j < kInline && j < r.cnt; ++j) r.v[j] = hv[r.k0 + j];
    }
    if (nnz) SGM_HIP(hipMemcpy(T.pv, hv.data(), nnz * 8, hipMemcpyHostToDevice));
    if (n) SGM_HIP(hipMemcpy(T.recs, T.h_recs.data(), (size_t)n * sizeof(TrsvRec), hipMemcpyHostToDevice));
    return SGM_OK;
}

// triangular solve in position space: xp holds the right-hand side on entry, the solution on exit
void trsv(const TriFactor &T, double *xp, const int *flag)
{
    hipStream_t st = g_rt.stream;
    for (const auto &L : T.schedule) {
        if (L.narrow) {
            hipLaunchKernelGGL(k_trsv_walk, dim3(1), dim3(kTrsvBlock), 0, st, (const TrsvRec *)T.recs,
                               (const int32_t *)T.pq, (const double *)T.pv, (const int32_t *)T.level_ptr_dev, L.l0, L.l1,
                               xp, flag);
        } else {
            const int32_t b = T.level_ptr[L.l0], e = T.level_ptr[L.l1];
            hipLaunchKernelGGL(k_trsv_wide, dim3((e - b + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                               (const TrsvRec *)T.recs, (const int32_t *)T.pq, (const double *)T.pv, b, e, xp, flag);
        }
    }
}

int download_csr(sgm_mat A, std::vector<int32_t> &ptr1, std::vector<int32_t> &node1, std::vector<double> &val)
{
    const Part &p = A->parts[0];
    ptr1.resize((size_t)p.n + 1);
    node1.resize((size_t)p.nnz);
    val.resize((size_t)p.nnz);
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    SGM_HIP(hipMemcpy(ptr1.data(), p.rowptr, ptr1.size() * 4, hipMemcpyDeviceToHost));
    if (p.nnz) {
        SGM_HIP(hipMemcpy(node1.data(), p.col, node1.size() * 4, hipMemcpyDeviceToHost));
        SGM_HIP(hipMemcpy(val.data(), p.val, val.size() * 8, hipMemcpyDeviceToHost));
    }
    for (auto &v : ptr1) v += 1;
    for (auto &v : node1) v += 1;
    return SGM_OK;
}

}  // namespace

namespace sgm {

int pc_kind(sgm_pc pc) { return pc ? pc->kind : 0; }
const double *pc_idiag(sgm_pc pc, size_t part) { return pc->parts[part].idiag; }

int pc_apply_parts(sgm_pc pc, sgm_mat A, const double *const *r, double *const *z, const int *const *flags)
{
    hipStream_t st = g_rt.stream;
    if (pc->kind == SGM_PC_JACOBI) {
        for (size_t ip = 0; ip < A->parts.size(); ++ip) {
            const int64_t n = A->parts[ip].n;
            hipLaunchKernelGGL(k_scale_by, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, pc->parts[ip].idiag, r[ip], z[ip],
                               flags ? flags[ip] : nullptr);
        }
    } else {
        const int64_t n = pc->n;
        const int *flag = flags ? flags[0] : nullptr;
        const int g = vec_grid(n);
        hipLaunchKernelGGL(k_perm_gather, dim3(g), dim3(kBlock), 0, st, n, pc->xpL, r[0], (const int32_t *)pc->L.order, flag);
        trsv(pc->L, pc->xpL, flag);                                             // (I+L) x = b
        hipLaunchKernelGGL(k_lu_transition, dim3(g), dim3(kBlock), 0, st, n, pc->xpU, (const double *)pc->xpL,
                           (const int32_t *)pc->mapLU, (const double *)pc->Dp, flag);                       // x = x / D
        trsv(pc->U, pc->xpU, flag);                                             // (I+U) x = x
        hipLaunchKernelGGL(k_perm_scatter, dim3(g), dim3(kBlock), 0, st, n, z[0], (const double *)pc->xpU,
                           (const int32_t *)pc->U.order, flag);
    }
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

}  // namespace sgm

extern "C" {

int sgm_pc_setup(sgm_pc pc, sgm_mat A)
{
    SGM_TRY(require_init());
    if (!pc || !A) return fail(SGM_ERR_BAD_ARG, "sgm_pc_setup: null argument");
    if (A->nrow != A->ncol)      // jacobi_solvers.f90:46-50, ldu_solvers.f90:104-108
        return fail(SGM_ERR_DIMS, "Cannot make a %s solver for a non-square matrix",
                    pc->kind == SGM_PC_JACOBI ? "Jacobi" : "LDU");
    hipStream_t st = g_rt.stream;
    if (A->fmt == SGM_FMT_COMPOSITE)
        return fail(SGM_ERR_UNSUPPORTED, "preconditioners need a leaf (CSR / ELLPACK) matrix, not a composite");
    if (pc->kind == SGM_PC_JACOBI) {
        if (pc->parts.size() != A->parts.size()) {
            for (auto &pp : pc->parts) dfree(pp.idiag);
            pc->parts.assign(A->parts.size(), PartPC());
        }
        pc->n = A->nrow;
        for (size_t ip = 0; ip < A->parts.size(); ++ip) {
            const Part &p = A->parts[ip];
            if (!pc->parts[ip].idiag) SGM_TRY(dalloc(&pc->parts[ip].idiag, (size_t)p.n + 2));
            const int grid = (p.n + kBlock - 1) / kBlock;
            if (!grid) continue;
            if (A->fmt == SGM_FMT_CSR)
                hipLaunchKernelGGL(k_jacobi_setup_csr, dim3(grid), dim3(kBlock), 0, st, p.n, p.rowptr, p.col, p.val,
                                   pc->parts[ip].idiag);
            else
                hipLaunchKernelGGL(k_jacobi_setup_ell, dim3(grid), dim3(kBlock), 0, st, p.n, p.max_d, p.ecol, p.eval,
                                   pc->parts[ip].idiag);
        }
        SGM_HIP(hipGetLastError());
        return finish();
    }
    // ILDU(0)
    if (A->fmt != SGM_FMT_CSR || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "ILDU(0) needs a single-GPU CSR matrix");
    std::vector<int32_t> ptr1, node1;
    std::vector<double> val;
    SGM_TRY(download_csr(A, ptr1, node1, val));
    const int32_t n = A->nrow;
    const bool fresh = pc->n != n || pc->hLptr.empty();      // ldu_solvers.f90:117-125: pattern once
    if (fresh) ildu_pattern(pc, n, ptr1, node1);
    pc->n = n;
    ildu_factor(pc, n, ptr1, node1, val);
    SGM_TRY(upload_tri(pc->L, n, pc->hLptr, pc->hLnode, pc->hLval, true, fresh));
    SGM_TRY(upload_tri(pc->U, n, pc->hUptr, pc->hUnode, pc->hUval, false, fresh));
    if (fresh) {
        dfree(pc->D); dfree(pc->xpL); dfree(pc->xpU); dfree(pc->Dp); dfree(pc->mapLU);
        SGM_TRY(dalloc(&pc->D, (size_t)n));
        SGM_TRY(dalloc(&pc->xpL, (size_t)n));
        SGM_TRY(dalloc(&pc->xpU, (size_t)n));
        SGM_TRY(dalloc(&pc->Dp, (size_t)n));
        SGM_TRY(dalloc(&pc->mapLU, (size_t)n));
        std::vector<int32_t> map((size_t)std::max(n, 1));
        for (int32_t p = 0; p < n; ++p) map[p] = pc->L.h_pos[pc->U.h_order[p]];
        if (n) SGM_HIP(hipMemcpy(pc->mapLU, map.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    std::vector<double> dp((size_t)std::max(n, 1));
    for (int32_t p = 0; p < n; ++p) dp[p] = pc->hD[pc->U.h_order[p]];
    if (n) {
        SGM_HIP(hipMemcpy(pc->D, pc->hD.data(), (size_t)n * 8, hipMemcpyHostToDevice));
        SGM_HIP(hipMemcpy(pc->Dp, dp.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    }
    return SGM_OK;
}

int sgm_jacobi_create(sgm_pc *out, sgm_mat A)
{
    if (!out) return fail(SGM_ERR_BAD_ARG, "sgm_jacobi_create: null out pointer");
    sgm_pc pc = new sgm_pc_s;
    pc->kind = SGM_PC_JACOBI;
    int rc = sgm_pc_setup(pc, A);
    if (rc != SGM_OK) { sgm_pc_destroy(pc); return rc; }
    *out = pc;
    return SGM_OK;
}

int sgm_ildu0_create(sgm_pc *out, sgm_mat A)
{
    if (!out) return fail(SGM_ERR_BAD_ARG, "sgm_ildu0_create: null out pointer");
    sgm_pc pc = new sgm_pc_s;
    pc->kind = SGM_PC_ILDU0;
    int rc = sgm_pc_setup(pc, A);
    if (rc != SGM_OK) { sgm_pc_destroy(pc); return rc; }
    *out = pc;
    return SGM_OK;
}

int sgm_pc_apply(sgm_pc pc, const double *r, double *z, int where)
{
    SGM_TRY(require_init());
    if (!pc || !r || !z) return fail(SGM_ERR_BAD_ARG, "sgm_pc_apply: null argument");
    if (pc->kind == SGM_PC_JACOBI && pc->parts.size() != 1)
        return fail(SGM_ERR_UNSUPPORTED, "sgm_pc_apply: stand-alone apply needs a single-part matrix");
    Staged sr, sz;
    SGM_TRY(stage_in(sr, r, pc->n, where, true));
    SGM_TRY(stage_in(sz, z, pc->n, where, false));
    // a throw-away matrix view with the right part count for pc_apply_parts
    sgm_mat_s view;
    view.parts.resize(1);
    view.parts[0].n = pc->n;
    const double *rs[1] = {sr.dev};
    double *zs[1] = {sz.dev};
    SGM_TRY(pc_apply_parts(pc, &view, rs, zs, nullptr));
    SGM_TRY(stage_out(sz, z, pc->n, where));
    return finish();
}

int sgm_pc_get(sgm_pc pc, const char *name, void *out, size_t bytes, size_t *needed)
{
    if (!pc || !name) return fail(SGM_ERR_BAD_ARG, "sgm_pc_get: null argument");
    const void *src = nullptr;
    size_t sz = 0;
    std::string nm(name);
    static const char kEmpty = 0;
    if (pc->kind == SGM_PC_JACOBI && nm == "idiag") {
        if (pc->parts.size() != 1) return fail(SGM_ERR_UNSUPPORTED, "sgm_pc_get(idiag): single-part only");
        pc->hidiag.resize((size_t)pc->n);
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
        if (pc->n) SGM_HIP(hipMemcpy(pc->hidiag.data(), pc->parts[0].idiag, (size_t)pc->n * 8, hipMemcpyDeviceToHost));
        src = pc->hidiag.data(); sz = pc->hidiag.size() * 8;
    } else if (pc->kind == SGM_PC_ILDU0) {
        if (nm == "Lptr") { src = pc->hLptr.data(); sz = pc->hLptr.size() * 4; }
        else if (nm == "Lnode") { src = pc->hLnode.data(); sz = pc->hLnode.size() * 4; }
        else if (nm == "Lval") { src = pc->hLval.data(); sz = pc->hLval.size() * 8; }
        else if (nm == "Uptr") { src = pc->hUptr.data(); sz = pc->hUptr.size() * 4; }
        else if (nm == "Unode") { src = pc->hUnode.data(); sz = pc->hUnode.size() * 4; }
        else if (nm == "Uval") { src = pc->hUval.data(); sz = pc->hUval.size() * 8; }
        else if (nm == "D") { src = pc->hD.data(); sz = pc->hD.size() * 8; }
        else if (nm == "levels") {
            static int32_t lv[2];
            lv[0] = (int32_t)pc->L.level_ptr.size() - 1;
            lv[1] = (int32_t)pc->U.level_ptr.size() - 1;
            src = lv; sz = sizeof lv;
        }
    }
    const bool known = nm == "idiag" || nm == "Lptr" || nm == "Lnode" || nm == "Lval" || nm == "Uptr" ||
                       nm == "Unode" || nm == "Uval" || nm == "D" || nm == "levels";
    if (!known || (!src && sz)) return fail(SGM_ERR_BAD_ARG, "sgm_pc_get: unknown array '%s'", name);
    if (!src) src = &kEmpty;
    if (needed) *needed = sz;
    if (out && sz) {
        if (bytes < sz) return fail(SGM_ERR_BAD_ARG, "sgm_pc_get: buffer too small (%zu < %zu)", bytes, sz);
        memcpy(out, src, sz);
    }
    return SGM_OK;
}

int sgm_pc_destroy(sgm_pc pc)
{
    if (!pc) return SGM_OK;
    for (auto &pp : pc->parts) dfree(pp.idiag);
    free_tri(pc->L);
    free_tri(pc->U);
    dfree(pc->D); dfree(pc->xpL); dfree(pc->xpU); dfree(pc->Dp); dfree(pc->mapLU);
    delete pc;
    return SGM_OK;
}

}  // extern "C"
