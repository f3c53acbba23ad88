// Shared by the Krylov translation units (sgm_solvers.hip: solver object, dots, C ABI; sgm_cg.hip; sgm_bicgstab.hip;
// sgm_gmres.hip; sgm_lanczos.hip): the generic fused vector kernel and its launcher, the functors more than one loop uses,
// the solver object and the helpers every loop calls.  Kernels defined here are `static`: each unit launches its own copy.
#pragma once
#include "sgm_internal.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include <vector>

namespace sgm {

struct Staged {
    double *dev = nullptr;
    bool owned = false;
    ~Staged() { if (owned) dfree(dev); }
};
int stage_in(Staged &s, const double *v, int64_t n, int where, bool copy);
int stage_out(const Staged &s, double *v, int64_t n, int where);
int pc_apply_parts(sgm_pc pc, sgm_mat A, const double *const *r, double *const *z, const int *const *flags);
int pc_kind(sgm_pc pc);
bool pc_apply_is_short(sgm_pc pc);
const double *pc_idiag(sgm_pc pc, size_t part);
int32_t *pc_abort_word(sgm_pc pc);        // sgm_pc.hip: sticky abort word of a pipelined ILDU apply (null: nothing to watch)
int pc_retire_pipelines(sgm_pc pc);
sgm_mat pc_permuted_matrix(sgm_pc pc, sgm_mat A);    // ILDU of the colour-ordered A: P A P^T (the solve runs in its order); null otherwise
void pc_in_permuted(sgm_pc pc, bool on);
void pc_permute_vec(sgm_pc pc, size_t part, const double *src, double *dst, bool to_permuted);
int32_t pc_cg_fused_rows(sgm_pc pc, size_t part);
bool pc_cg_fused(sgm_pc pc, size_t ip, ScalarRef res2, ScalarRef dpr, const double *q, double *r, double *z, double *part, int *count, const int *flag, int gen);

// ------------------------------------------------------------------ generic fused kernel
// F provides: bool prepare(double* red) (block-uniform; false = nothing to do),
//             void pair(int64_t i2) (elements 2*i2, 2*i2+1), void single(int64_t i),
//             void finish(double* red).
// Stop flag protocol: *flag == 0: keep going.  A kernel of "generation" gen is skipped when
// *flag != 0 && gen >= *flag.  The CG p/x-update kernel of iteration k (generation k+1) sets
// flag = k+2 when the new res2 meets the tolerance: every kernel of iterations > k is skipped,
// while all workgroups of the setting kernel itself still run (they carry the last x update).
// Kernels that do not take part in this (gen = INT_MAX) stop on any nonzero flag.
template <class T, class = void> struct has_commit : std::false_type {};
template <class T> struct has_commit<T, std::void_t<decltype(std::declval<T &>().commit())>> : std::true_type {};
template <class F, bool NT>
__global__ __launch_bounds__(kBlock) void k_elem(int64_t n, F f, const int *flag, int gen)
{
    __shared__ double red[8 * (kBlock / 64)];       // (up to 8 scalars per load_scalars call)
    // The stop flag is REQUESTED first and LOOKED AT after prepare(): prepare() only loads and reduces scalars (no side
    // effects), so the flag's round trip and the partial sums' are one wait instead of two -- below n ~ 1e5 these
    // dependent round trips, not the launches, are what an iteration is made of.  Side effects (iteration count, history,
    // raising the flag) live in commit(), which a skipped kernel never reaches.
    const int st = flag ? *flag : 0;
    const bool go = f.prepare(red);
    if (st && gen >= st) return;
    if constexpr (has_commit<F>::value) f.commit();
    if (!go) return;
    const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t n2 = n >> 1;
    for (int64_t i = gtid; i < n2; i += stride) f.template pair<NT>(i);
    if ((n & 1) && gtid == 0) f.single(n - 1);
    f.finish(red);
}

// 16-byte vector access of the streaming kernels.  NT (per launch) marks the accesses
// nontemporal: measured on CG, plain accesses win while the vectors still find room in the
// 256 MiB Infinity Cache (n = 1e7: 238 vs 242 us per iteration) and lose beyond it
// (n = 2.7e7: 813 vs 766 us), so the launcher turns NT on for vectors >= 128 MiB.
typedef double f64x2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ inline double2 ld2(const double *p, int64_t i)
{
    const f64x2v *q = reinterpret_cast<const f64x2v *>(p) + i;
    const f64x2v v = NT ? __builtin_nontemporal_load(q) : *q;
    return make_double2(v.x, v.y);
}
template <bool NT>
__device__ inline void st2(double *p, int64_t i, double2 v)
{
    f64x2v w;
    w.x = v.x; w.y = v.y;
    f64x2v *q = reinterpret_cast<f64x2v *>(p) + i;
    if (NT) __builtin_nontemporal_store(w, q); else *q = w;
}

__device__ inline void put_partial(double v, double *part, double *red)
{
    const double t = block_sum<kBlock>(v, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// dst = src
struct FCopy {
    static constexpr bool kDot = false;
    double *dst; const double *src;
    __device__ bool prepare(double *) { return true; }
    template <bool NT> __device__ void pair(int64_t i) { st2<NT>(dst, i, ld2<NT>(src, i)); }
    __device__ void single(int64_t i) { dst[i] = src[i]; }
    __device__ void finish(double *) {}
};
// up to two dots: a.b -> part0, c.d -> part1 (c == nullptr: one dot)
struct FDot2 {
    const double *a, *b, *c, *d; double *part0, *part1;
    double s0 = 0.0, s1 = 0.0;
    __device__ bool prepare(double *) { return true; }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 x = ld2<NT>(a, i), y = ld2<NT>(b, i);
        s0 += x.x * y.x; s0 += x.y * y.y;
        if (c) { const double2 u = ld2<NT>(c, i), w = ld2<NT>(d, i); s1 += u.x * w.x; s1 += u.y * w.y; }
    }
    __device__ void single(int64_t i) { s0 += a[i] * b[i]; if (c) s1 += c[i] * d[i]; }
    __device__ void finish(double *red) { put_partial(s0, part0, red); if (c) put_partial(s1, part1, red); }
};
// y = y + alpha * x  (host scalar)
struct FAxpy {
    static constexpr bool kDot = false;
    double *y; const double *x; double alpha;
    __device__ bool prepare(double *) { return true; }
    template <bool NT> __device__ void pair(int64_t i)
    {
        double2 a = ld2<NT>(y, i); const double2 b = ld2<NT>(x, i);
        a.x = a.x + alpha * b.x; a.y = a.y + alpha * b.y; st2<NT>(y, i, a);
    }
    __device__ void single(int64_t i) { y[i] = y[i] + alpha * x[i]; }
    __device__ void finish(double *) {}
};

// ---- CG -----------------------------------------------------------------------------
// r = b - q ; [p = r ; partial r.r]          cg_solvers.f90:129-131
struct FCgInit {
    const double *b, *q; double *r, *p; double *part; bool with_p;
    double s = 0.0;
    __device__ bool prepare(double *) { return true; }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 bb = ld2<NT>(b, i), qq = ld2<NT>(q, i);
        double2 rr; rr.x = bb.x - qq.x; rr.y = bb.y - qq.y;
        st2<NT>(r, i, rr);
        if (with_p) { st2<NT>(p, i, rr); s += rr.x * rr.x; s += rr.y * rr.y; }
    }
    __device__ void single(int64_t i)
    {
        const double rr = b[i] - q[i]; r[i] = rr;
        if (with_p) { p[i] = rr; s += rr * rr; }
    }
    __device__ void finish(double *red) { if (with_p) put_partial(s, part, red); }
};
// p = z ; partial r.z                          cg_solvers.f90:172-173
struct FCopyDot {
    double *p; const double *z, *r; double *part; double s = 0.0;
    __device__ bool prepare(double *) { return true; }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 zz = ld2<NT>(z, i), rr = ld2<NT>(r, i);
        st2<NT>(p, i, zz); s += rr.x * zz.x; s += rr.y * zz.y;
    }
    __device__ void single(int64_t i) { p[i] = z[i]; s += r[i] * z[i]; }
    __device__ void finish(double *red) { put_partial(s, part, red); }
};
// alpha = res2/dpr ; r = r-alpha*q ; then
//   MODE 0: partial r.r   MODE 1: z = idiag*r, partial r.z   MODE 2: nothing (generic pc follows)
// cg_solvers.f90:138-140 / :181-185 with jacobi_solve jacobi_solvers.f90:77 folded in.
// (x = x+alpha*p, :137, is carried out by FCgPX: p is read there anyway, which saves one pass
// over p per iteration; the operations and their operands are the reference's.)
template <int MODE>
struct FCgR {
    static constexpr bool kDot = MODE != 2;
    ScalarRef res2, dpr;
    const double *q; double *r; const double *idiag; double *z; double *part;
    double alpha = 0.0, s = 0.0;
    __device__ bool prepare(double *red)
    {
        const ScalarRef rs[2] = {res2, dpr};
        double sc[2];
        load_scalars<kBlock, 2>(rs, sc, red);
        alpha = sc[0] / sc[1];
        return true;
    }
    __device__ void one(double qv, double &rv, double idv, double &zv)
    {
        rv = rv - alpha * qv;
        if (MODE == 0) s += rv * rv;
        if (MODE == 1) { zv = idv * rv; s += rv * zv; }
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 qq = ld2<NT>(q, i);
        double2 rr = ld2<NT>(r, i), zz = make_double2(0, 0), dd = make_double2(0, 0);
        if (MODE == 1) dd = ld2<NT>(idiag, i);
        one(qq.x, rr.x, dd.x, zz.x);
        one(qq.y, rr.y, dd.y, zz.y);
        st2<NT>(r, i, rr);
        if (MODE == 1) st2<NT>(z, i, zz);
    }
    __device__ void single(int64_t i)
    {
        double rv = r[i], zv = 0.0;
        one(q[i], rv, MODE == 1 ? idiag[i] : 0.0, zv);
        r[i] = rv;
        if (MODE == 1) z[i] = zv;
    }
    __device__ void finish(double *red) { if (MODE != 2) put_partial(s, part, red); }
};
// alpha = res2/dpr ; beta = dnew/res2 ; x = x + alpha*p ; p = z + beta*p ;
// bookkeeping: iterations++, history, loop condition        cg_solvers.f90:137,141-145
struct FCgPX {
    static constexpr bool kDot = false;
    ScalarRef res2, dpr, dnew; const double *z; double *p, *x;
    double tol; int *flag; int stop_value; int64_t *iters; double *history; int64_t hist_cap; double *res_out;
    // nx: elements that have an x (the owned rows).  The launch may run past them over the HALO slots of z and p
    // (run_cg, option dist_halo_fused): there only p = z + beta*p is formed -- the owner's statement on the owner's operands,
    // so the neighbour's copy of p's boundary rows has the owner's bits without travelling.
    int64_t nx = INT64_MAX;
    double alpha = 0.0, beta = 0.0, dnew_v = 0.0;
    __device__ bool prepare(double *red)
    {
        const ScalarRef rs[3] = {res2, dpr, dnew};
        double sc[3];
        load_scalars<kBlock, 3>(rs, sc, red);
        const double a = sc[0], b = sc[1];
        dnew_v = sc[2];
        alpha = a / b;
        beta = dnew_v / a;
        return true;
    }
    __device__ void commit()
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const double d = dnew_v;
            const int64_t it = *iters;
            if (history && it < hist_cap) history[it] = d;
            *iters = it + 1;
            *res_out = d;
            if (!(sqrt(d) > tol)) *flag = stop_value;
        }
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        if (2 * i + 1 >= nx) { single(2 * i); single(2 * i + 1); return; }
        const double2 zz = ld2<NT>(z, i); double2 pp = ld2<NT>(p, i), xx = ld2<NT>(x, i);
        xx.x = xx.x + alpha * pp.x; xx.y = xx.y + alpha * pp.y;
        pp.x = zz.x + beta * pp.x; pp.y = zz.y + beta * pp.y;
        st2<NT>(x, i, xx); st2<NT>(p, i, pp);
    }
    __device__ void single(int64_t i)
    {
        const double pv = p[i];
        if (i < nx) x[i] = x[i] + alpha * pv;
        p[i] = z[i] + beta * pv;
    }
    __device__ void finish(double *) {}
};

// one block: res = sum(ref) ; flag = !(sqrt(res) > tol)   (the loop test before iteration 1)
static __global__ __launch_bounds__(kBlock) void k_check(ScalarRef ref, double tol, int *flag, double *res_out)
{
    __shared__ double red[kBlock / 64];
    if (*flag) return;
    const double d = load_scalar<kBlock>(ref, red);
    if (threadIdx.x == 0) {
        *res_out = d;
        if (!(sqrt(d) > tol)) *flag = 1;
    }
}
// Start of a batch of iterations whose kernels carry generations RELATIVE to the batch (1, 2, ...): a stop raised in an
// earlier batch -- some generation of THAT batch -- becomes 1, which every generation of this and all later batches is >= :
// they all exit at once.  This is what lets one captured batch (a hipGraph) be replayed unchanged.
static __global__ void k_flag_norm(int *flag)
{
    if (threadIdx.x == 0 && *flag) *flag = 1;
}
// one block: slot = sum(partials)
static __global__ __launch_bounds__(kBlock) void k_reduce(const double *part, int count, double *slot)
{
    __shared__ double red[kBlock / 64];
    ScalarRef r{part, count};
    const double d = load_scalar<kBlock>(r, red);
    if (threadIdx.x == 0) *slot = d;
}

// ---- GMRES(m) --------------------------------------------------------------------------
constexpr int kGmresMaxRestart = 64;
struct GmresState {                 // lives in device memory, one per part (all parts hold the same values)
    double H[(kGmresMaxRestart + 1) * kGmresMaxRestart];   // column-major, R after rotations
    double cs[kGmresMaxRestart], sn[kGmresMaxRestart], g[kGmresMaxRestart + 1], y[kGmresMaxRestart];
    int j;                          // Arnoldi steps done in this cycle
    // low-synchronisation Gram-Schmidt (k_gsl, gmres_cgs2 = 1): the STORED columns S are projected once and never corrected;
    // R = the Cholesky factor of their Gram matrix S^T S (upper, column-major, leading dimension kGsLd) makes V = S R^-1 the
    // orthonormal basis, Gs the Hessenberg matrix of the stored basis (A S_k = S_{k+1} Gs), coef the projection the second pass
    // subtracts ([a_0 .. a_{k-1}, 1 / alpha])
    double R[33 * 33], Gs[34 * 33], coef[34];
};
constexpr int kGsLd = 33;
// w = w - h_prev*v_prev (if v_prev) ; partial w.v_cur (v_cur == nullptr: partial w.w)
struct FMgs {
    double *w; const double *v_prev, *v_cur; ScalarRef h_prev; double *part; double h = 0.0, s = 0.0;
    __device__ bool prepare(double *red)
    {
        if (v_prev) h = load_scalar<kBlock>(h_prev, red);
        return true;
    }
    __device__ void one(int64_t i)
    {
        double wv = w[i];
        if (v_prev) { wv = wv - h * v_prev[i]; w[i] = wv; }
        s += wv * (v_cur ? v_cur[i] : wv);
    }
    // 16-byte accesses (measured: 858 -> 1035 GMRES iterations/s on C3 against 8-byte ones); the
    // basis vectors are read once per pass (nontemporal), w with the launch's policy
    template <bool NT> __device__ void pair(int64_t i)
    {
        double2 wv = ld2<NT>(w, i);
        if (v_prev) {
            const double2 vp = ld2<true>(v_prev, i);
            wv.x = wv.x - h * vp.x;
            wv.y = wv.y - h * vp.y;
            st2<NT>(w, i, wv);
        }
        const double2 vc = v_cur ? ld2<true>(v_cur, i) : wv;
        s += wv.x * vc.x;
        s += wv.y * vc.y;
    }
    __device__ void single(int64_t i) { one(i); }
    __device__ void finish(double *red) { put_partial(s, part, red); }
};
// dst = src / sqrt(sum(nrm2))      (v_{j+1} = w / h_{j+1,j} ; v_1 = r / beta)
struct FScaleInv {
    static constexpr bool kDot = false;
    double *dst; const double *src; ScalarRef nrm2; double d = 1.0;
    __device__ bool prepare(double *red) { d = sqrt(load_scalar<kBlock>(nrm2, red)); return true; }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 a = ld2<NT>(src, i); double2 o; o.x = a.x / d; o.y = a.y / d; st2<NT>(dst, i, o);
    }
    __device__ void single(int64_t i) { dst[i] = src[i] / d; }
    __device__ void finish(double *) {}
};
// ---- dot_order = 1: the reference's dot_product order -----------------------------------------------------
// The pinned reference build (amdflang -O2, x86-64 without FMA) turns `dot_product(a, b)` into ONE accumulator that
// starts at +0.0 and takes the individually rounded products a(i) * b(i) first element to last
// (cg_solvers.f90:131,135,140; bicgstab_solvers.f90:152,155,160,164,169).  The tree order above is a legal
// dot_product too, but only this order makes the iterates bit-identical to the reference's.  The chain is serial by
// nature -- one dependent fp64 add per element, about 4 ns each -- so this is a VALIDATION mode (n <~ 1e5), not a
// production one: the products are formed in parallel and parked in LDS, one wave walks them in order.
//
// s + v[0] + v[1] + ... + v[cnt-1], left to right, out of LDS; every lane of the wave runs the same chain on the same
// (broadcast) addresses.  The next 16 values are requested before the current 16 are added, so the chain never waits
// for an LDS round trip.  `pr` must be 16-byte aligned.
__device__ inline double seq_chain_lds(const double *pr, int32_t cnt, double s)
{
    int32_t j = 0;
    if (cnt >= 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = pr[u];
        for (; j + 32 <= cnt; j += 16) {
            double w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = pr[j + 16 + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) s = s + v[u];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = w[u];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) s = s + v[u];
        j += 16;
    }
    for (; j < cnt; ++j) s = s + pr[j];
    return s;
}

// Grid policy of the vector kernels.  Kernels that leave partial sums need grid <= kMaxGrid;
// pure update kernels take one pass over a large grid (a copy-like stream runs ~30 % faster
// that way on MI355X than as a small persistent grid: tools/stream_bench.cpp).
struct VecCfg { int dot_grid = 1024, nodot_grid = 2048; };
static VecCfg &vec_cfg()
{
    static VecCfg c;
    static bool init = false;
    if (!init) {
        init = true;
        if (c.dot_grid > kMaxGrid) c.dot_grid = kMaxGrid;
    }
    return c;
}
inline int dot_grid(int64_t n)
{
    int64_t g = (n + 4 * kBlock - 1) / (4 * kBlock);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, vec_cfg().dot_grid));
}
static int nodot_grid(int64_t n)
{
    int64_t g = (n / 2 + kBlock - 1) / kBlock;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, vec_cfg().nodot_grid));
}

template <class T, class = void> struct has_kdot : std::false_type {};
template <class T> struct has_kdot<T, std::void_t<decltype(T::kDot)>> : std::true_type {};
template <class F> constexpr bool leaves_partials()
{
    if constexpr (has_kdot<F>::value) return F::kDot; else return true;
}

template <class F>
static inline void launch_elem(int64_t n, const F &f, const int *flag, int gen = INT32_MAX)
{
    const int grid = leaves_partials<F>() ? dot_grid(n) : nodot_grid(n);
    if (n >= (int64_t)(128 << 20) / 8)
        hipLaunchKernelGGL((k_elem<F, true>), dim3(grid), dim3(kBlock), 0, g_rt.stream, n, f, flag, gen);
    else
        hipLaunchKernelGGL((k_elem<F, false>), dim3(grid), dim3(kBlock), 0, g_rt.stream, n, f, flag, gen);
}

}  // namespace sgm

using namespace sgm;

// ======================================================================================
// solver object
// ======================================================================================
constexpr int kNumPartials = 72;     // partial arrays per part (GMRES: restart+2 with MGS; the low-synchronisation form: 0..32, 36..68, 71)

struct PartWork {
    int64_t n = 0, next = 0;         // owned length, extended (owned+halo) length
    std::vector<double *> vec;       // work vectors, each `next` long
    double *partials = nullptr;      // kNumPartials x kMaxGrid
    double *slots = nullptr;         // kNumPartials reduced scalars (multi-part only)
    int *flag = nullptr;             // device: loop finished
    int64_t *iters = nullptr;        // device: iterations of the current solve
    double *res = nullptr;           // device: last res2
    double *history = nullptr;
    GmresState *gmres = nullptr;
    double *V = nullptr;             // GMRES basis, (restart+1) x next
    int count[kNumPartials] = {0};   // producer grid of each partial array
};

struct sgm_solver_s {
    int kind = 0;
    double tolerance = 1e-16;        // cg_set_params default, cg_solvers.f90:106
    int32_t restart = 30;
    int64_t max_iter = 0;
    int64_t hist_cap = 0;
    bool initialized = false;
    int32_t nn = 0;
    int64_t iterations = 0;          // accumulates across solves (cg_solvers.f90:72,145)
    int64_t last_iterations = 0;
    double res2 = 0.0;
    int32_t converged = 0;
    bool seq = false;                // this solve runs with dot_order = 1 (set by sgm_solver_solve from the option)
    int32_t *abort_dev = nullptr;    // the preconditioner's sticky abort word while its pipelined sweeps are in use (sgm_pc.hip)
    int32_t aborted = 0;             // ... as last read by read_state: nonzero = this solve's iterates are spoiled, stop and redo
    double *x_backup = nullptr;      // the caller's initial guess, kept while abort_dev is watched
    std::vector<PartWork> work;
    std::vector<double> history;
    bool multi = false;
    bool reduce_single = false;       // one part: collapse every dot to its slot with a one-block kernel (see finish_dots)
    // cooperative CG (k_cg_coop): exchange vector + dot slots + {counter, abort}; the counter is monotonic across launches
    double *coop_buf = nullptr;
    int coop_base = 0;
    int64_t coop_iters0 = 0;
    bool coop_retired = false, coop_xl_retired = false;
    double *perm_x = nullptr, *perm_b = nullptr;       // x and b in the order of a reordering preconditioner's matrix (sgm_solver_solve)
    SolverOptions opt = g_opt.solver;   // this solver's options: the defaults at its creation, then sgm_solver_set_option
    int64_t small_chunk() const { return opt.cg_small > 1 ? opt.cg_small : 50000; }          // iterations per launch of the one-workgroup kernels
    int64_t graph_after() const { return opt.krylov_graph > 1 ? opt.krylov_graph : 64; }     // iterations before the group is captured
};

namespace sgm {

struct Views {       // per-part pointer tables for spmv_parts / pc
    std::vector<const double *> cx;
    std::vector<double *> y;
    std::vector<const double *> w;
    std::vector<double *> p0, p1;
    std::vector<const int *> flags;
};

// A group of kGraphIters Krylov iterations captured once per solve as a hipGraph and replayed: below n ~ 1e6 an iteration of
// the launch loops IS its launches (CG: three dependent ones, ~4.8 us each from the host; ~1.8 us each when replayed:
// tools/probes/graph_probe.cpp), so long solves of mid-sized systems spend two thirds of their time in the launch path.  The group
// is what the loop would launch -- same kernels, same arguments, generations relative to the group (k_flag_norm) -- captured
// on the launch stream after the solve has run long enough to pay for the capture.
constexpr int kGraphIters = 16;
struct GraphBatch {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool failed = false;
    ~GraphBatch()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }
    // body(): enqueues the group's launches on g_rt.stream
    template <class Body>
    bool ensure(Body &&body)
    {
        if (exec) return true;
        if (failed) return false;
        failed = true;                                     // (until the whole sequence below has worked)
        if (hipStreamBeginCapture(g_rt.stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return false; }
        const int rc = body();
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(g_rt.stream, &g);
        if (rc != SGM_OK || e != hipSuccess || !g) { (void)hipGetLastError(); if (g) (void)hipGraphDestroy(g); return false; }
        graph = g;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); exec = nullptr; return false; }
        failed = false;
        return true;
    }
};

int num_work_vectors(int kind);
void free_work(sgm_solver s);
// ScalarRef of partial array k on part ip / the array itself
ScalarRef ref(sgm_solver s, size_t ip, int k);
double *part(sgm_solver s, size_t ip, int k);
int finish_dots(sgm_solver s, sgm_mat A, const int *ks, int nk, const int (*vecs)[2] = nullptr, bool use_flag = false,
                int gen = INT32_MAX, double *const *halo_of = nullptr);
int read_state(sgm_solver s, int *flag, int64_t *iters, double *res);
bool graph_applies(sgm_solver s, sgm_mat A, sgm_pc pc);
int run_cg(sgm_solver s, sgm_mat A, double *const *x, const double *const *b, sgm_pc pc);
int run_bicgstab(sgm_solver s, sgm_mat A, double *const *x, const double *const *b, sgm_pc pc);
int run_gmres(sgm_solver s, sgm_mat A, double *const *x, const double *const *b, sgm_pc pc);

}  // namespace sgm
