// Device-resident Krylov loops for gfx950: CG / PCG (cg_solvers.f90:116-194), BiCGStab /
// preconditioned BiCGStab (bicgstab_solvers.f90:124-237) and GMRES(m) (no reference
// counterpart), plus the dot / axpy statements they are made of.
//
// Design: the whole `do while (dsqrt(res2) > tolerance)` loop runs on the GPU.
//  * Every dot product is produced as <= 2048 per-workgroup partial sums (fixed grid,
//    xor-butterfly wave reduction + fixed-order LDS sum).  Every CONSUMER workgroup
//    re-reduces the partials in the same order (ScalarRef), so alpha/beta/omega are
//    computed redundantly but bit-identically by all workgroups: no host round trip, no
//    atomics, no grid barrier between a dot and the update that needs it.
//  * The loop condition is a device flag written by the kernel that learns the new
//    res2; all later kernels of the batch exit at once when it is set, so the host only
//    polls once per batch and `iterations` still equals the reference's count exactly.
//  * Fusion (bytes per CG iteration: B_csr + 72 n): p.q is folded into the
//    SpMV epilogue; x/r update + r.r (or the Jacobi apply + r.z) is one pass; the p
//    update is one pass.
//  * Multi-GPU / multi-partition: the same kernels run per row block; a dot is then
//    reduced to one slot per part and summed across parts (ncclAllReduce or k_sum_parts).
// Compiled with -ffp-contract=off: a*b+c is never fused, like the reference build.
#include "sgm_krylov.hpp"

namespace sgm {

// the same for a table of partial arrays (an in-process partition's parts x dots): block b = entry b
constexpr int kReduceTabMax = 96;
struct ReduceTab { const double *part[kReduceTabMax]; double *slot[kReduceTabMax]; int count[kReduceTabMax]; };
__global__ __launch_bounds__(kBlock) void k_reduce_tab(ReduceTab rt)
{
    __shared__ double red[kBlock / 64];
    ScalarRef r{rt.part[blockIdx.x], rt.count[blockIdx.x]};
    const double d = load_scalar<kBlock>(r, red);
    if (threadIdx.x == 0) *rt.slot[blockIdx.x] = d;
}

// ---- dot_order = 1: the reference's dot_product order (the rule and seq_chain_lds: sgm_krylov.hpp) ----------

struct SeqDot {
    const double *a, *b;      // operands (this part's rows)
    const double *init;       // running sum of the parts / ranks before this one (null: the sum starts here, at +0.0)
    double *out;              // running sum after this part's rows
};
// ND (1 or 2) dot products at once, ONE workgroup: waves 2 and 3 stream the operands (16-byte coalesced loads), form the
// products and park them in LDS, chunk c + 1 while chain wave d (wave 0, wave 1) walks chunk c of dot d.
constexpr int kSeqChunk = 1024;
template <int ND>
__global__ __launch_bounds__(kBlock) void k_dot_seq(int64_t n, SeqDot d0, SeqDot d1, const int *flag, int gen)
{
    __shared__ __attribute__((aligned(16))) double buf[ND][2][kSeqChunk];
    if (flag) { const int st = *flag; if (st && gen >= st) return; }
    const int wave = threadIdx.x >> 6;
    const int64_t nchunks = (n + kSeqChunk - 1) / kSeqChunk;
    auto fill = [&](int64_t c) {               // waves 2, 3: 128 threads, 8 products per thread and dot
        const int t = threadIdx.x - 128;
        const int64_t base = c * kSeqChunk;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double *a = d ? d1.a : d0.a, *b = d ? d1.b : d0.b;
            double *dst = buf[d][c & 1];
#pragma unroll
            for (int u = 0; u < kSeqChunk / 256; ++u) {
                const int64_t e = base + 2 * (t + 128 * u);
                if (e + 1 < n) {
                    const double2 x = ld2<false>(a, e >> 1), y = ld2<false>(b, e >> 1);
                    dst[e - base] = x.x * y.x;
                    dst[e - base + 1] = x.y * y.y;
                } else if (e < n) {
                    dst[e - base] = a[e] * b[e];
                }
            }
        }
    };
    double s = 0.0;
    if (wave < ND) { const SeqDot &d = wave ? d1 : d0; if (d.init) s = *d.init; }
    if (wave >= 2 && nchunks > 0) fill(0);
    __syncthreads();
    for (int64_t c = 0; c < nchunks; ++c) {
        if (wave >= 2) { if (c + 1 < nchunks) fill(c + 1); }
        else if (wave < ND) s = seq_chain_lds(buf[wave][c & 1], (int32_t)(n - c * kSeqChunk < kSeqChunk ? n - c * kSeqChunk : kSeqChunk), s);
        __syncthreads();
    }
    if (wave < ND && (threadIdx.x & 63) == 0) *(wave ? d1.out : d0.out) = s;
}
// the total (the last part's running sum) into another part's slot
__global__ void k_copy_slot(const double *src, double *dst, const int *flag, int gen)
{
    if (flag) { const int st = *flag; if (st && gen >= st) return; }
    if (threadIdx.x == 0) *dst = *src;
}

// block c: slots[c] = sum of partial array c (count entries each, kMaxGrid apart)
// up to four dots collapsed by one launch: block b = dot b (k_reduce's sum, same order)
struct ReduceSet { const double *part[4]; int count[4]; double *slot[4]; };
__global__ __launch_bounds__(kBlock) void k_reduce_set(ReduceSet rs)
{
    __shared__ double red[kBlock / 64];
    ScalarRef r{rs.part[blockIdx.x], rs.count[blockIdx.x]};
    const double d = load_scalar<kBlock>(r, red);
    if (threadIdx.x == 0) *rs.slot[blockIdx.x] = d;
}


int num_work_vectors(int kind) { return kind == SGM_SOLVER_CG ? 4 : kind == SGM_SOLVER_BICGSTAB ? 8 : 3; }

void free_work(sgm_solver s)
{
    for (auto &w : s->work) {
        for (double *v : w.vec) dfree(v);
        dfree(w.partials); dfree(w.slots); dfree(w.flag); dfree(w.iters); dfree(w.res);
        dfree(w.history); dfree(w.gmres); dfree(w.V);
    }
    s->work.clear();
    dfree(s->x_backup);
    s->x_backup = nullptr;
    dfree(s->coop_buf);
    s->coop_buf = nullptr;
    dfree(s->perm_x); dfree(s->perm_b);
    s->perm_x = s->perm_b = nullptr;
}

// ScalarRef of partial array k on part ip
ScalarRef ref(sgm_solver s, size_t ip, int k)
{
    PartWork &w = s->work[ip];
    if (s->multi || s->seq || s->reduce_single) return ScalarRef{w.slots + k, 1};
    return ScalarRef{w.partials + (size_t)k * kMaxGrid, w.count[k]};
}
double *part(sgm_solver s, size_t ip, int k) { return s->work[ip].partials + (size_t)k * kMaxGrid; }

// after the producers of partial arrays ks[] ran on every part: make the totals visible
// dot_order = 1: `vecs[t]` names the two work vectors dot ks[t] is taken over; the partial sums the producers left are
// ignored and the products are formed again, in order (k_dot_seq).  Across in-process parts the running sum is handed
// from one part's kernel to the next one's; across ranks it travels rank 0 -> 1 -> ... (seq_chain_recv / _share), so a
// partitioned solve adds the same products in the same global order as the one-part solve.
// `halo_of` (one extended vector per part, multi-part solves only): its boundary rows travel to the neighbours' halo slots in
// the same step as the sums (halo_exchange_allreduce)
int finish_dots(sgm_solver s, sgm_mat A, const int *ks, int nk, const int (*vecs)[2], bool use_flag, int gen, double *const *halo_of)
{
    if (halo_of && s->seq) SGM_TRY(halo_exchange(A, halo_of, g_rt.stream));
    if (s->seq) {
        if (!vecs) return fail(SGM_ERR_UNSUPPORTED, "dot_order = 1: this dot product has no sequential form");
        const size_t P = s->work.size();
        const bool ranks = A->comm && A->comm->nranks > 1;
        for (int t = 0; t < nk;) {
            const int nd = (!ranks && t + 1 < nk) ? 2 : 1;
            if (ranks) SGM_TRY(seq_chain_recv(A, s->work[0].slots + ks[t]));
            for (size_t ip = 0; ip < P; ++ip) {
                PartWork &w = s->work[ip];
                SeqDot d[2];
                for (int u = 0; u < nd; ++u) {
                    const int k = ks[t + u];
                    d[u].a = w.vec[vecs[t + u][0]];
                    d[u].b = w.vec[vecs[t + u][1]];
                    d[u].init = ip ? s->work[ip - 1].slots + k : (ranks && A->comm->rank > 0 ? w.slots + k : nullptr);
                    d[u].out = w.slots + k;
                }
                if (nd == 1) {
                    d[1] = d[0];
                    hipLaunchKernelGGL((k_dot_seq<1>), dim3(1), dim3(kBlock), 0, g_rt.stream, w.n, d[0], d[1],
                                       use_flag ? (const int *)w.flag : nullptr, gen);
                } else {
                    hipLaunchKernelGGL((k_dot_seq<2>), dim3(1), dim3(kBlock), 0, g_rt.stream, w.n, d[0], d[1],
                                       use_flag ? (const int *)w.flag : nullptr, gen);
                }
            }
            for (size_t ip = 0; ip + 1 < P; ++ip)
                for (int u = 0; u < nd; ++u)
                    hipLaunchKernelGGL(k_copy_slot, dim3(1), dim3(64), 0, g_rt.stream,
                                       (const double *)(s->work[P - 1].slots + ks[t + u]), s->work[ip].slots + ks[t + u],
                                       use_flag ? (const int *)s->work[ip].flag : nullptr, gen);
            if (ranks) SGM_TRY(seq_chain_share(A, s->work[0].slots + ks[t]));
            t += nd;
        }
        return SGM_OK;
    }
    if (s->reduce_single) {
        // one part, a solver whose update kernels each read many scalars (BiCGStab: up to six), a large system: every dot is
        // collapsed ONCE by a one-block kernel instead of being re-reduced from its <= 4096 partials by each of the 2048
        // workgroups of every consumer (k_reduce IS load_scalar: same order, same bits)
        for (int t = 0; t < nk; t += 4) {
            ReduceSet rs{};
            const int m = std::min(4, nk - t);
            for (int u = 0; u < m; ++u) {
                rs.part[u] = part(s, 0, ks[t + u]);
                rs.count[u] = s->work[0].count[ks[t + u]];
                rs.slot[u] = s->work[0].slots + ks[t + u];
            }
            hipLaunchKernelGGL(k_reduce_set, dim3(m), dim3(kBlock), 0, g_rt.stream, rs);
        }
        return SGM_OK;
    }
    if (!s->multi) return SGM_OK;
    // slots ks[] must be contiguous for the all-reduce: callers pass consecutive ids
    prof_begin(PH_DOT_REDUCE, g_rt.stream);
    if (!A->comm && s->work.size() * (size_t)nk <= (size_t)kReduceTabMax) {
        // an in-process partition: every part's partial sums of every dot by one launch
        ReduceTab rt;
        int e = 0;
        for (size_t ip = 0; ip < s->work.size(); ++ip)
            for (int t = 0; t < nk; ++t, ++e) {
                rt.part[e] = part(s, ip, ks[t]);
                rt.count[e] = s->work[ip].count[ks[t]];
                rt.slot[e] = s->work[ip].slots + ks[t];
            }
        hipLaunchKernelGGL(k_reduce_tab, dim3(e), dim3(kBlock), 0, g_rt.stream, rt);
    } else {
        for (size_t ip = 0; ip < s->work.size(); ++ip)
            for (int t = 0; t < nk; ++t)
                hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, g_rt.stream, part(s, ip, ks[t]),
                                   s->work[ip].count[ks[t]], s->work[ip].slots + ks[t]);
    }
    prof_end(PH_DOT_REDUCE, g_rt.stream);
    std::vector<double *> ptrs(s->work.size());
    for (size_t ip = 0; ip < s->work.size(); ++ip) ptrs[ip] = s->work[ip].slots + ks[0];
    if (halo_of) return halo_exchange_allreduce(A, halo_of, ptrs.data(), ks[nk - 1] - ks[0] + 1, s->opt.dist_halo_fused == 1);
    return allreduce_slots(A, ptrs.data(), ks[nk - 1] - ks[0] + 1);
}

int read_state(sgm_solver s, int *flag, int64_t *iters, double *res)
{
    PartWork &w = s->work[0];
    SGM_HIP(hipMemcpyAsync(flag, w.flag, sizeof(int), hipMemcpyDeviceToHost, g_rt.stream));
    SGM_HIP(hipMemcpyAsync(iters, w.iters, sizeof(int64_t), hipMemcpyDeviceToHost, g_rt.stream));
    SGM_HIP(hipMemcpyAsync(res, w.res, sizeof(double), hipMemcpyDeviceToHost, g_rt.stream));
    if (s->abort_dev) SGM_HIP(hipMemcpyAsync(&s->aborted, s->abort_dev, sizeof(int32_t), hipMemcpyDeviceToHost, g_rt.stream));
    hb_phase(HB_SOLVER_WAIT);
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    hb_phase(HB_SOLVER_ENQUEUE);
    g_hb.iteration = *iters;
    return SGM_OK;
}

// a solve qualifies when nothing but kernel launches (and stream forks that join again) makes up an iteration: one GPU's matrix or
// the parts of an in-process partition (their gathers and sums are kernels: 73 launches per CG iteration with 8 parts, and the
// iteration is its launch path until they are replayed), plain, Jacobi, or ILDU(0) with two-level factors on every part (a colour
// ordering: the sweeps are two row-space launches; the pipelined sweeps of a natural order have an abort word the host watches)
bool graph_applies(sgm_solver s, sgm_mat A, sgm_pc pc)
{
    const int pk = pc ? pc_kind(pc) : 0;
    bool pc_ok = pk == 0 || pk == SGM_PC_JACOBI;
    if (pk == SGM_PC_ILDU0 && !s->seq && s->opt.reorder_solve >= 2) {
        pc_ok = true;
        for (size_t ip = 0; pc_ok && ip < s->work.size(); ++ip) pc_ok = pc_cg_fused_rows(pc, ip) > 0;
    }
    return s->opt.krylov_graph && !A->comm && A->fmt != SGM_FMT_COMPOSITE && pc_ok && !prof_on();
}

}  // namespace sgm

// ======================================================================================
// C ABI
// ======================================================================================
extern "C" {

static int solver_create(sgm_solver *out, int kind, double tol, int32_t restart)
{
    if (!out) return fail(SGM_ERR_BAD_ARG, "solver create: null out pointer");
    sgm_solver s = new sgm_solver_s;
    s->kind = kind;
    s->tolerance = tol;
    s->restart = restart;
    *out = s;
    return SGM_OK;
}
int sgm_cg_create(sgm_solver *out, double tolerance) { return solver_create(out, SGM_SOLVER_CG, tolerance, 0); }
int sgm_bicgstab_create(sgm_solver *out, double tolerance) { return solver_create(out, SGM_SOLVER_BICGSTAB, tolerance, 0); }
int sgm_gmres_create(sgm_solver *out, double tolerance, int32_t restart)
{
    if (restart < 1 || restart > kGmresMaxRestart)
        return fail(SGM_ERR_BAD_ARG, "sgm_gmres_create: restart must be in 1..%d", kGmresMaxRestart);
    return solver_create(out, SGM_SOLVER_GMRES, tolerance, restart);
}

// solver%tolerance is a public field the reference's loops read at every solve (`do while( dsqrt(res2)>solver%tolerance )`,
// cg_solvers.f90:133,175; bicgstab_solvers.f90:153) and set_params may be called again (cg_solvers.f90:95-111): the tolerance of
// a handle is live.  It is a launch argument of the loop kernels, read when a solve starts; work vectors, `iterations` and the
// options are untouched.
int sgm_solver_set_tolerance(sgm_solver s, double tolerance)
{
    if (!s) return fail(SGM_ERR_BAD_ARG, "sgm_solver_set_tolerance: null solver");
    if (tolerance != tolerance) return fail(SGM_ERR_BAD_ARG, "sgm_solver_set_tolerance: the tolerance is NaN");
    s->tolerance = tolerance;
    return SGM_OK;
}

int sgm_solver_setup(sgm_solver s, sgm_mat A)
{
    SGM_TRY(require_init());
    if (!s || !A) return fail(SGM_ERR_BAD_ARG, "sgm_solver_setup: null argument");
    if (A->nrow != A->ncol)     // cg_solvers.f90:61-65
        return fail(SGM_ERR_DIMS, "Cannot make a %s solver for a non-square matrix",
                    s->kind == SGM_SOLVER_CG ? "CG" : s->kind == SGM_SOLVER_BICGSTAB ? "BiCGStab" : "GMRES");
    s->nn = A->nrow;
    s->iterations = 0;          // cg_solvers.f90:72
    s->multi = A->distributed();
    {
        // (BiCGStab: six scalars per update kernel, C3 2535 -> 2859 it/s; CG: two or three, C2 4650 -> 4730)
        s->reduce_single = !s->multi && A->fmt != SGM_FMT_COMPOSITE && (s->kind == SGM_SOLVER_BICGSTAB || s->kind == SGM_SOLVER_CG) &&
                           A->nrow >= (1 << 21);
    }
    bool realloc = !s->initialized || s->work.size() != A->parts.size();
    for (size_t ip = 0; !realloc && ip < A->parts.size(); ++ip)
        realloc = s->work[ip].n != A->parts[ip].n || s->work[ip].next != A->parts[ip].xlen();
    if (realloc) {
        free_work(s);
        s->work.resize(A->parts.size());
        for (size_t ip = 0; ip < A->parts.size(); ++ip) {
            PartWork &w = s->work[ip];
            w.n = A->parts[ip].n;
            w.next = (std::max<int64_t>(A->parts[ip].xlen(), w.n) + 1) & ~(int64_t)1;    // even: 16-byte aligned basis columns
            w.vec.resize(num_work_vectors(s->kind));
            for (auto &p : w.vec) SGM_TRY(dalloc(&p, (size_t)w.next + 2));
            SGM_TRY(dalloc(&w.partials, (size_t)kNumPartials * kMaxGrid));
            SGM_TRY(dalloc(&w.slots, (size_t)kNumPartials));
            SGM_TRY(dalloc(&w.flag, 1));
            SGM_TRY(dalloc(&w.iters, 1));
            SGM_TRY(dalloc(&w.res, 1));
            if (s->kind == SGM_SOLVER_GMRES) {
                SGM_TRY(dalloc(&w.gmres, 1));
                SGM_TRY(dalloc(&w.V, (size_t)(s->restart + 1) * w.next + 2));
            }
        }
        s->initialized = true;
    }
    for (auto &w : s->work) {   // cg_solvers.f90:84-88: zero the work vectors on every setup
        for (auto &p : w.vec) SGM_HIP(hipMemsetAsync(p, 0, ((size_t)w.next + 2) * 8, g_rt.stream));
        SGM_HIP(hipMemsetAsync(w.partials, 0, (size_t)kNumPartials * kMaxGrid * 8, g_rt.stream));
        SGM_HIP(hipMemsetAsync(w.slots, 0, (size_t)kNumPartials * 8, g_rt.stream));
    }
    return finish();
}

int sgm_solver_set_max_iter(sgm_solver s, int64_t max_iter)
{
    if (!s) return fail(SGM_ERR_BAD_ARG, "null solver");
    s->max_iter = max_iter > 0 ? max_iter : 0;
    return SGM_OK;
}

/* sgm_solver_set_option: this solver's own copy of "cg_small", "bicgstab_small", "krylov_graph", "dot_order", "gmres_cgs2"
 * (sgm_set_option only changes what solvers created LATER start with); read at the next solve. */
int sgm_solver_set_option(sgm_solver s, const char *name, int value)
{
    if (!s || !name) return fail(SGM_ERR_BAD_ARG, "sgm_solver_set_option: null argument");
    int v = 0;
    SGM_TRY(normalise_option(name, value, &v));
    int *f = solver_option_field(s->opt, name);
    if (!f) return fail(SGM_ERR_BAD_ARG, "sgm_solver_set_option: '%s' is not a solver option", name);
    *f = v;
    return SGM_OK;
}

int sgm_solver_set_history(sgm_solver s, int64_t capacity)
{
    if (!s) return fail(SGM_ERR_BAD_ARG, "null solver");
    s->hist_cap = capacity > 0 ? capacity : 0;
    for (auto &w : s->work) { dfree(w.history); w.history = nullptr; }
    return SGM_OK;
}

int sgm_solver_solve(sgm_solver s, sgm_mat A, double *x, const double *b, sgm_pc pc, int where)
{
    SGM_TRY(require_init());
    if (!s || !A || !x || !b) return fail(SGM_ERR_BAD_ARG, "sgm_solver_solve: null argument");
    if (!s->initialized) return fail(SGM_ERR_BAD_ARG, "sgm_solver_solve: solver%%setup(A) has not been called");
    if (A->nrow != s->nn || s->work.size() != A->parts.size())
        return fail(SGM_ERR_DIMS, "sgm_solver_solve: matrix does not match the one given to setup");
    const size_t P = A->parts.size();
    // the caller's vectors: global length for a single / in-process-partitioned matrix,
    // owned slice for a matrix distributed over processes
    const int64_t nvec = A->comm ? A->parts[0].n : A->nrow;
    Staged sx, sb;
    SGM_TRY(stage_in(sx, x, nvec, where, true));
    SGM_TRY(stage_in(sb, b, nvec, where, true));
    std::vector<double *> xs(P);
    std::vector<const double *> bs(P);
    for (size_t ip = 0; ip < P; ++ip) {
        const int64_t off = A->comm ? 0 : A->parts[ip].row_begin;
        if (off & 1) return fail(SGM_ERR_UNSUPPORTED, "partition boundaries must be even rows (16-B vector access)");
        xs[ip] = sx.dev + off;
        bs[ip] = sb.dev + off;
        PartWork &w = s->work[ip];
        SGM_HIP(hipMemsetAsync(w.flag, 0, sizeof(int), g_rt.stream));
        SGM_HIP(hipMemsetAsync(w.iters, 0, sizeof(int64_t), g_rt.stream));
        if (s->hist_cap && !w.history && ip == 0) {
            SGM_TRY(dalloc(&w.history, (size_t)s->hist_cap));
        }
        if (w.history) SGM_HIP(hipMemsetAsync(w.history, 0, (size_t)s->hist_cap * 8, g_rt.stream));
    }
    // dot_order = 1: CG / BiCGStab add their dot products in the reference's order (GMRES has no reference counterpart
    // and keeps the tree order)
    s->seq = s->opt.dot_order == 1 && s->kind != SGM_SOLVER_GMRES;
    // A pipelined ILDU sweep has bounded waits; one that gives up leaves NaN patterns behind and raises the preconditioner's
    // sticky word.  It is read with every look at the stop flag (read_state); if it was raised the iterates are spoiled:
    // the pipelines are retired, the initial guess restored and the solve run again with the level-scheduled sweeps --
    // never `converged` on a NaN the library produced itself.
    s->abort_dev = pc ? pc_abort_word(pc) : nullptr;
    s->aborted = 0;
    if (s->abort_dev) {
        if (!s->x_backup) SGM_TRY(dalloc(&s->x_backup, (size_t)nvec + 2));
        SGM_HIP(hipMemcpyAsync(s->x_backup, sx.dev, (size_t)nvec * 8, hipMemcpyDeviceToDevice, g_rt.stream));
    }
    int rc = SGM_OK;
    struct HbScope { HbScope() { g_hb.solves = g_hb.solves + 1; g_hb.iteration = 0; hb_phase(HB_SOLVER_ENQUEUE); } ~HbScope() { hb_phase(HB_IDLE); } } hb_scope;
    // A preconditioner that factorised the colour-ordered matrix P A P^T (option ildu_reorder) brings that matrix along: the
    // whole solve runs in its order -- x' = P x, b' = P b once, the products on P A P^T, x = P^T x' at the end -- the same
    // iteration as on A with P^T M^-1 P (permutations commute with dot products up to the order of the sum), without two
    // permutations of r and z around every apply.  Only for the matrix the preconditioner was set up with, unchanged since.
    sgm_mat Arun = A;
    struct PermScope { sgm_pc pc = nullptr; ~PermScope() { if (pc) pc_in_permuted(pc, false); } } perm_scope;
    const bool perm_off = s->opt.reorder_solve == 0;      // (option reorder_solve: 0 = r and z permuted around every apply instead)
    if (sgm_mat Ap = perm_off ? nullptr : pc_permuted_matrix(pc, A); Ap && Ap->parts.size() == P && Ap->nrow == A->nrow) {
        if (!s->perm_x) SGM_TRY(dalloc(&s->perm_x, (size_t)nvec + 2));
        if (!s->perm_b) SGM_TRY(dalloc(&s->perm_b, (size_t)nvec + 2));
        for (size_t ip = 0; ip < P; ++ip) {          // every part / rank its own slice, by its own local ordering
            const int64_t off = xs[ip] - sx.dev;
            pc_permute_vec(pc, ip, sx.dev + off, s->perm_x + off, true);
            pc_permute_vec(pc, ip, sb.dev + off, s->perm_b + off, true);
            xs[ip] = s->perm_x + off; bs[ip] = s->perm_b + off;
        }
        Arun = Ap;
        perm_scope.pc = pc;
        pc_in_permuted(pc, true);
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (s->kind == SGM_SOLVER_CG) rc = run_cg(s, Arun, xs.data(), bs.data(), pc);
        else if (s->kind == SGM_SOLVER_BICGSTAB) rc = run_bicgstab(s, Arun, xs.data(), bs.data(), pc);
        else rc = run_gmres(s, Arun, xs.data(), bs.data(), pc);
        if (rc != SGM_OK) return rc;
        if (!s->aborted) break;
        if (attempt == 1) return fail(SGM_ERR_HIP, "sgm_solver_solve: a triangular sweep aborted again after the pipelines were retired");
        SGM_TRY(pc_retire_pipelines(pc));
        s->abort_dev = nullptr;
        s->aborted = 0;
        SGM_HIP(hipMemcpyAsync(sx.dev, s->x_backup, (size_t)nvec * 8, hipMemcpyDeviceToDevice, g_rt.stream));
        if (Arun != A)
            for (size_t ip = 0; ip < P; ++ip) { const int64_t off = xs[ip] - s->perm_x; pc_permute_vec(pc, ip, sx.dev + off, s->perm_x + off, true); }
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            SGM_HIP(hipMemsetAsync(w.flag, 0, sizeof(int), g_rt.stream));
            SGM_HIP(hipMemsetAsync(w.iters, 0, sizeof(int64_t), g_rt.stream));
            if (w.history) SGM_HIP(hipMemsetAsync(w.history, 0, (size_t)s->hist_cap * 8, g_rt.stream));
        }
    }
    s->abort_dev = nullptr;
    // A breakdown (BiCGStab: rho or omega reaches 0 / 0) ends the loop as it ends the reference's -- `dsqrt(res2) > tolerance` is
    // false for a NaN (bicgstab_solvers.f90:154, :214) -- with x full of NaNs.  That is not "converged".
    if (s->res2 != s->res2) s->converged = 0;
    if (Arun != A)
        for (size_t ip = 0; ip < P; ++ip) { const int64_t off = xs[ip] - s->perm_x; pc_permute_vec(pc, ip, s->perm_x + off, sx.dev + off, false); }
    s->iterations += s->last_iterations;
    if (s->hist_cap) {
        const int64_t cnt = std::min<int64_t>(s->last_iterations, s->hist_cap);
        s->history.resize((size_t)cnt);
        if (cnt) SGM_HIP(hipMemcpy(s->history.data(), s->work[0].history, (size_t)cnt * 8, hipMemcpyDeviceToHost));
    }
    SGM_TRY(stage_out(sx, x, nvec, where));
    SGM_TRY(finish());
    // SGM_TRACE: a CG solve past n / 2 iterations is in the regime where the ORDER of the dot products decides the count (the
    // short recurrence has lost its orthogonality to rounding: INTEGRATION.md, "Which parity gate each dot order meets") --
    // say which switch reproduces the CPU build's count before anyone compares the two numbers
    if (trace_on() && s->kind == SGM_SOLVER_CG && !s->seq && s->last_iterations > (int64_t)s->nn / 2)
        fprintf(stderr, "[sigma_hip] cg: %lld iterations on %d rows (> n / 2) with tree-order dot products; validation against the CPU "
                        "build's iteration count: option dot_order = 1 (the reference's summation order, bit-identical iterates)\n",
                (long long)s->last_iterations, (int)s->nn);
    if (s->max_iter > 0 && !s->converged) {
        if (s->res2 != s->res2) fail(SGM_ERR_NOT_CONVERGED, "solver broke down after %lld iterations (res2 is NaN)", (long long)s->last_iterations);
        else fail(SGM_ERR_NOT_CONVERGED, "solver stopped at max_iter=%lld with sqrt(res2)=%g > %g",
                  (long long)s->max_iter, std::sqrt(s->res2), s->tolerance);
        return SGM_ERR_NOT_CONVERGED;
    }
    return SGM_OK;
}

int sgm_solver_info(sgm_solver s, int64_t *iterations, double *res2, int32_t *converged, int64_t *last)
{
    if (!s) return fail(SGM_ERR_BAD_ARG, "null solver");
    if (iterations) *iterations = s->iterations;
    if (res2) *res2 = s->res2;
    if (converged) *converged = s->converged;
    if (last) *last = s->last_iterations;
    return SGM_OK;
}

int sgm_solver_get_history(sgm_solver s, double *out, int64_t capacity, int64_t *count)
{
    if (!s) return fail(SGM_ERR_BAD_ARG, "null solver");
    const int64_t c = std::min<int64_t>((int64_t)s->history.size(), capacity);
    if (out && c) memcpy(out, s->history.data(), (size_t)c * 8);
    if (count) *count = (int64_t)s->history.size();
    return SGM_OK;
}

int sgm_solver_destroy(sgm_solver s)
{
    if (!s) return SGM_OK;
    free_work(s);
    delete s;
    return SGM_OK;
}

int sgm_dot(int64_t n, const double *a, const double *b, double *result, int where)
{
    SGM_TRY(require_init());
    if (n < 0 || !a || !b || !result) return fail(SGM_ERR_BAD_ARG, "sgm_dot: bad argument");
    Staged sa, sb2;
    SGM_TRY(stage_in(sa, a, n, where, true));
    SGM_TRY(stage_in(sb2, b, n, where, true));
    double *partials = nullptr, *slot = nullptr;
    SGM_TRY(dalloc(&partials, (size_t)kMaxGrid));
    SGM_TRY(dalloc(&slot, 1));
    const int grid = dot_grid(n);
    launch_elem(n, FDot2{sa.dev, sb2.dev, nullptr, nullptr, partials, nullptr}, nullptr);
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, g_rt.stream, partials, grid, slot);
    SGM_HIP(hipMemcpyAsync(result, slot, 8, hipMemcpyDeviceToHost, g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    dfree(partials);
    dfree(slot);
    return SGM_OK;
}

int sgm_axpy(int64_t n, double alpha, const double *x, double *y, int where)
{
    SGM_TRY(require_init());
    if (n < 0 || !x || !y) return fail(SGM_ERR_BAD_ARG, "sgm_axpy: bad argument");
    Staged sx, sy;
    SGM_TRY(stage_in(sx, x, n, where, true));
    SGM_TRY(stage_in(sy, y, n, where, true));
    launch_elem(n, FAxpy{sy.dev, sx.dev, alpha}, nullptr);
    SGM_HIP(hipGetLastError());
    SGM_TRY(stage_out(sy, y, n, where));
    return finish();
}

}  // extern "C"
