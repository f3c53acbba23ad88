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
#include "sgm_internal.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>

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
__global__ __launch_bounds__(kBlock) void k_check(ScalarRef ref, double tol, int *flag, double *res_out)
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
__global__ void k_flag_norm(int *flag)
{
    if (threadIdx.x == 0 && *flag) *flag = 1;
}
// one block: slot = sum(partials)
__global__ __launch_bounds__(kBlock) void k_reduce(const double *part, int count, double *slot)
{
    __shared__ double red[kBlock / 64];
    ScalarRef r{part, count};
    const double d = load_scalar<kBlock>(r, red);
    if (threadIdx.x == 0) *slot = d;
}

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

// ---- BiCGStab -------------------------------------------------------------------------
struct BiScalars {         // dot results of the CURRENT (cur) and PREVIOUS (old) iteration
    ScalarRef rr, rho, rho_old, r0v_old, st_old, tt_old, r0v, st, tt;
    int first;              // iteration 1: rho_old = alpha = omega = 1 (bicgstab_solvers.f90:144-147)
    int nan_guard;          // plain variant only (:165)
};
__device__ inline double bi_omega(double st, double tt, int guard)
{
    double om = st / tt;
    if (guard && isnan(om)) om = 0.0;
    return om;
}
// loop test + rho/beta + p = r + beta*(p - omega*v)     bicgstab_solvers.f90:154-157
struct FBiP {
    static constexpr bool kDot = false;
    BiScalars S; const double *r, *v; double *p;
    double tol; int *flag; int64_t *iters; double *history; int64_t hist_cap; double *res_out;
    double beta = 0.0, omega = 1.0, res2_v = 0.0; bool stop_v = false;
    __device__ bool prepare(double *red)
    {
        // (iteration 1 has no previous dots: their slots are read all the same -- zero-filled at setup -- and not used)
        const ScalarRef rs[6] = {S.rr, S.rho, S.rho_old, S.r0v_old, S.st_old, S.tt_old};
        double sc[6];
        load_scalars<kBlock, 6>(rs, sc, red);
        res2_v = sc[0];
        stop_v = !(sqrt(res2_v) > tol);
        if (stop_v) return false;
        const double rho = sc[1];
        double rho_old = 1.0, alpha = 1.0;
        omega = 1.0;
        if (!S.first) {
            rho_old = sc[2];
            alpha = rho_old / sc[3];
            omega = bi_omega(sc[4], sc[5], S.nan_guard);
        }
        beta = rho / rho_old * alpha / omega;
        return true;
    }
    __device__ void commit()
    {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const int64_t it = *iters;
            *res_out = res2_v;
            if (!S.first && history && it - 1 < hist_cap && it >= 1) history[it - 1] = res2_v;
            if (stop_v) *flag = 1; else *iters = it + 1;
        }
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 rr = ld2<NT>(r, i), vv = ld2<NT>(v, i); double2 pp = ld2<NT>(p, i);
        pp.x = rr.x + beta * (pp.x - omega * vv.x);
        pp.y = rr.y + beta * (pp.y - omega * vv.y);
        st2<NT>(p, i, pp);
    }
    __device__ void single(int64_t i) { p[i] = r[i] + beta * (p[i] - omega * v[i]); }
    __device__ void finish(double *) {}
};
// alpha = rho / (r0.v) ; s = r - alpha*v                 bicgstab_solvers.f90:160-161
struct FBiS {
    static constexpr bool kDot = false;
    ScalarRef rho, r0v; const double *r, *v; double *s; double alpha = 0.0;
    __device__ bool prepare(double *red)
    {
        const ScalarRef rs[2] = {rho, r0v};
        double sc[2];
        load_scalars<kBlock, 2>(rs, sc, red);
        alpha = sc[0] / sc[1];
        return true;
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 rr = ld2<NT>(r, i), vv = ld2<NT>(v, i); double2 ss;
        ss.x = rr.x - alpha * vv.x; ss.y = rr.y - alpha * vv.y; st2<NT>(s, i, ss);
    }
    __device__ void single(int64_t i) { s[i] = r[i] - alpha * v[i]; }
    __device__ void finish(double *) {}
};
// omega ; x = x + alpha*p + omega*s ; r = s - omega*t ; partial r.r and r0.r
// bicgstab_solvers.f90:164-169 (+ rho of the next iteration, :155)
struct FBiXR {
    ScalarRef rho, r0v, st, tt; int nan_guard;
    const double *p, *s, *t, *r0; double *x, *r; double *part_rr, *part_rho;
    double alpha = 0.0, omega = 0.0, srr = 0.0, srho = 0.0;
    __device__ bool prepare(double *red)
    {
        const ScalarRef rs[4] = {rho, r0v, st, tt};
        double sc[4];
        load_scalars<kBlock, 4>(rs, sc, red);
        alpha = sc[0] / sc[1];
        omega = bi_omega(sc[2], sc[3], nan_guard);
        return true;
    }
    __device__ void one(double pv, double sv, double tv, double r0v_, double &xv, double &rv)
    {
        xv = xv + alpha * pv + omega * sv;
        rv = sv - omega * tv;
        srr += rv * rv;
        srho += r0v_ * rv;
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        const double2 pp = ld2<NT>(p, i), ss = ld2<NT>(s, i), tt_ = ld2<NT>(t, i), r00 = ld2<NT>(r0, i);
        double2 xx = ld2<NT>(x, i), rr;
        one(pp.x, ss.x, tt_.x, r00.x, xx.x, rr.x);
        one(pp.y, ss.y, tt_.y, r00.y, xx.y, rr.y);
        st2<NT>(x, i, xx); st2<NT>(r, i, rr);
    }
    __device__ void single(int64_t i)
    {
        double xv = x[i], rv;
        one(p[i], s[i], t[i], r0[i], xv, rv);
        x[i] = xv; r[i] = rv;
    }
    __device__ void finish(double *red) { put_partial(srr, part_rr, red); put_partial(srho, part_rho, red); }
};
// r0 = src ; r = r0 ; v = 0 ; p = 0 ; partial r.r (twice: res2 and rho)   :140-152
struct FBiInit {
    const double *b, *q; bool sub; double *r0, *r, *v, *p; double *part_rr, *part_rho; double s = 0.0;
    __device__ bool prepare(double *) { return true; }
    __device__ void one(int64_t i)
    {
        const double w = sub ? b[i] - q[i] : b[i];
        r0[i] = w; r[i] = w; v[i] = 0.0; p[i] = 0.0; s += w * w;
    }
    template <bool NT> __device__ void pair(int64_t i) { one(2 * i); one(2 * i + 1); }
    __device__ void single(int64_t i) { one(i); }
    __device__ void finish(double *red)
    {
        const double t = block_sum<kBlock>(s, red);
        if (threadIdx.x == 0) { part_rr[blockIdx.x] = t; part_rho[blockIdx.x] = t; }
    }
};

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
// start of a cycle: beta = sqrt(sum) ; g = (beta,0,...) ; j = 0 ; loop test
__global__ __launch_bounds__(kBlock) void k_gmres_start(ScalarRef nrm2, GmresState *G, double tol, int *flag,
                                                        double *res_out)
{
    __shared__ double red[kBlock / 64];
    const double d = load_scalar<kBlock>(nrm2, red);
    if (threadIdx.x == 0) {
        const double beta = sqrt(d);
        G->j = 0;
        G->g[0] = beta;
        G->R[0] = 1.0;                       // (k_gsl: the Gram matrix of the one stored column s_0 = r / beta)
        *res_out = beta * beta;
        if (!(beta > tol)) *flag = 1;
    }
}
// after the Gram-Schmidt sweep of step j: column j of H from the partial arrays, previous
// rotations, new rotation, residual estimate, loop test
__global__ __launch_bounds__(kBlock) void k_gmres_givens(const double *parts, int stride, int count,
                                                         int in_slots, const double *slots, int m,
                                                         GmresState *G, double tol, int *flag, int64_t *iters,
                                                         double *history, int64_t hist_cap, double *res_out)
{
    __shared__ double red[kBlock / 64];
    if (*flag) return;
    const int j = G->j;
    __shared__ double hcol[kGmresMaxRestart + 2];
    for (int i = 0; i <= j + 1; ++i) {          // h_0..h_j and the squared norm at j+1
        ScalarRef r = in_slots ? ScalarRef{slots + i, 1} : ScalarRef{parts + (size_t)i * stride, count};
        const double d = load_scalar<kBlock>(r, red);
        if (threadIdx.x == 0) hcol[i] = d;
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    double *H = G->H + (size_t)j * (m + 1);
    for (int i = 0; i <= j; ++i) H[i] = hcol[i];
    H[j + 1] = sqrt(hcol[j + 1]);
    for (int i = 0; i < j; ++i) {
        const double h0 = H[i], h1 = H[i + 1];
        H[i] = G->cs[i] * h0 + G->sn[i] * h1;
        H[i + 1] = -G->sn[i] * h0 + G->cs[i] * h1;
    }
    const double h0 = H[j], h1 = H[j + 1];
    const double d = sqrt(h0 * h0 + h1 * h1);
    G->cs[j] = h0 / d;
    G->sn[j] = h1 / d;
    H[j] = d;
    H[j + 1] = 0.0;
    G->g[j + 1] = -G->sn[j] * G->g[j];
    G->g[j] = G->cs[j] * G->g[j];
    const double res = fabs(G->g[j + 1]);
    const int64_t it = *iters;
    if (history && it < hist_cap) history[it] = res * res;
    *iters = it + 1;
    *res_out = res * res;
    G->j = j + 1;
    if (!(res > tol)) *flag = 1;
}
// end of a cycle: back substitution for y (k = G->j columns)
__global__ void k_gmres_solve_y(GmresState *G, int m)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int k = G->j;
    for (int i = k - 1; i >= 0; --i) {
        double s = G->g[i];
        for (int l = i + 1; l < k; ++l) s = s - G->H[i + (size_t)l * (m + 1)] * G->y[l];
        G->y[i] = s / G->H[i + (size_t)i * (m + 1)];
    }
}
// x = x + sum_i y_i v_i   (one pass over x, k passes over V)
struct FGmresUpdate {
    static constexpr bool kDot = false;
    double *x; const double *V; int64_t ldv; const GmresState *G; int k = 0;
    __device__ bool prepare(double *) { k = G->j; return k > 0; }
    __device__ void one(int64_t i)
    {
        double xv = x[i];
        for (int c = 0; c < k; ++c) xv = xv + G->y[c] * V[(size_t)c * ldv + i];
        x[i] = xv;
    }
    template <bool NT> __device__ void pair(int64_t i)        // (ldv is even: the columns are 16-byte aligned)
    {
        double2 xv = ld2<NT>(x, i);
        for (int c = 0; c < k; ++c) {
            const double yc = G->y[c];
            const double2 vv = ld2<true>(V + (size_t)c * ldv, i);
            xv.x = xv.x + yc * vv.x;
            xv.y = xv.y + yc * vv.y;
        }
        st2<NT>(x, i, xv);
    }
    __device__ void single(int64_t i) { one(i); }
    __device__ void finish(double *) {}
};

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
__global__ __launch_bounds__(kBlock) void k_reduce_many(const double *parts, int count, double *slots)
{
    __shared__ double red[kBlock / 64];
    ScalarRef r{parts + (size_t)blockIdx.x * kMaxGrid, count};
    const double d = load_scalar<kBlock>(r, red);
    if (threadIdx.x == 0) slots[blockIdx.x] = d;
}
// ---- low-synchronisation Gram-Schmidt: the basis read TWICE per step, two reductions -------------------------------------
// Classical Gram-Schmidt applied twice reads the basis three times (h1 = V^T w | w -= V h1, h2 = V^T w | w -= V h2, norm): the
// second correction cannot start before h2 has been summed.  Here it is never applied to the vector: the stored column
// s_{k} = (z - S a) / alpha is the ONCE-projected vector, and what the second projection would have removed is kept as numbers --
// the new column (S^T s_k, s_k . s_k) of the Gram matrix of the stored columns, measured by the same pass that forms s_k.  With
// R = chol(S^T S) the orthonormal basis is V = S R^-1 (never formed), the projection of the next z is the exact one,
// a = (S^T S)^-1 S^T z = R^-1 R^-T (S^T z), and Arnoldi's relation in the orthonormal basis is A V_k = V_{k+1} (R Gs R^-1).
// (The inverse-compact-WY / "low-synch" Gram-Schmidt of the GMRES literature, written with the full Gram factor.)
//   pass 1 (MODE 0)  g = S^T z, t = z.z                                   k + 1 reads
//   small            a = R^-1 R^-T g ; alpha = sqrt(t - |R^-T g|^2)         (k_gmres_ls1: one workgroup)
//   pass 2 (MODE 1)  s_k = (z - S a) / alpha ; c = S^T s_k, d = s_k.s_k    k + 1 reads, 1 write
//   small            R grows by (R^-T c, sqrt(d - |R^-T c|^2)) ; H(:, j) = R Gs R^-1 e_j ; rotations      (k_gmres_ls2)
// 2 k + 3 vector passes per step where blocked CGS-2 + the scaling pass took 3 k + 8; two all-reduces across ranks, not three.
template <int KB, int MODE>
__global__ __launch_bounds__(kBlock) void k_gsl(int64_t n, int kk, const double *__restrict__ z, double *V, int64_t ldv,
                                                const double *__restrict__ coef, double *__restrict__ part_out, const int *flag)
{
    __shared__ double red[kBlock / 64];
    if (flag && *flag) return;
    double a[KB], acc[KB];
#pragma unroll
    for (int c = 0; c < KB; ++c) {
        a[c] = (MODE == 1 && c < kk) ? coef[c] : 0.0;
        acc[c] = 0.0;
    }
    const double inv_alpha = MODE == 1 ? coef[kk] : 0.0;
    double own = 0.0;                                   // z.z (pass 1) / s_k.s_k (pass 2)
    const int64_t gtid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t n2 = n >> 1;
    for (int64_t i = gtid; i < n2; i += stride) {
        double2 wv = ld2<false>(z, i);
        double2 vv[KB];
#pragma unroll
        for (int c = 0; c < KB; ++c)
            if (c < kk) vv[c] = ld2<true>(V + (size_t)c * ldv, i);
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < KB; ++c)
                if (c < kk) { wv.x = wv.x - a[c] * vv[c].x; wv.y = wv.y - a[c] * vv[c].y; }
            wv.x = wv.x * inv_alpha; wv.y = wv.y * inv_alpha;
            st2<false>(V + (size_t)kk * ldv, i, wv);
        }
#pragma unroll
        for (int c = 0; c < KB; ++c)
            if (c < kk) { acc[c] += vv[c].x * wv.x; acc[c] += vv[c].y * wv.y; }
        own += wv.x * wv.x; own += wv.y * wv.y;
    }
    if ((n & 1) && gtid == 0) {
        const int64_t i = n - 1;
        double wv = z[i];
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < KB; ++c)
                if (c < kk) wv = wv - a[c] * V[(size_t)c * ldv + i];
            wv = wv * inv_alpha;
            V[(size_t)kk * ldv + i] = wv;
        }
#pragma unroll
        for (int c = 0; c < KB; ++c)
            if (c < kk) acc[c] += V[(size_t)c * ldv + i] * wv;
        own += wv * wv;
    }
#pragma unroll
    for (int c = 0; c < KB; ++c)
        if (c < kk) {                        // kk is uniform: every thread takes the same branches
            const double t = block_sum<kBlock>(acc[c], red);
            if (threadIdx.x == 0) part_out[(size_t)c * kMaxGrid + blockIdx.x] = t;
        }
    const double t = block_sum<kBlock>(own, red);
    if (threadIdx.x == 0) part_out[(size_t)kk * kMaxGrid + blockIdx.x] = t;
}

// The small dense steps run on ONE WAVE out of LDS: lane i holds entry i of each vector, the triangular solves sweep by
// columns (the pivot lane's value goes round by __shfl, the other lanes update their own entry), the matrix-vector products are
// a row per lane.  kGsLd = 33 makes both R(i, lane) and R(lane, i) conflict-free LDS reads.
__device__ inline void gsl_load_R(const GmresState *G, int kk, double *Rl)
{
    for (int e = threadIdx.x; e < kk * kGsLd; e += blockDim.x) Rl[e] = G->R[e];
    __syncthreads();
}
// entry `from` of a lane-held vector in every lane (`from` is uniform: v_readlane, no trip through the LDS crossbar)
__device__ inline double lane_value(double v, int from)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), from), __builtin_amdgcn_readlane(__double2loint(v), from));
}
// lane i: 1 / R(i, i) -- the pivots are divided out once, in parallel, and not on the solves' dependent chain
__device__ inline double wave_inv_diag(const double *Rl, int kk)
{
    const int lane = threadIdx.x;
    return lane < kk ? 1.0 / Rl[lane + lane * kGsLd] : 0.0;
}
// R^T u = b (R upper, kk x kk): lane i passes b_i in and gets u_i back.  Row i's sum runs over l = 0..i-1 ascending.
__device__ inline double wave_solve_Rt(const double *Rl, int kk, double dinv, double b)
{
    const int lane = threadIdx.x;
    b = b * dinv;                                        // (row i scaled by 1 / R(i,i): b_i is u_i once its sum is complete)
    for (int i = 0; i + 1 < kk; ++i) {
        const double ui = lane_value(b, i);
        if (lane > i && lane < kk) b = b - (Rl[i + lane * kGsLd] * dinv) * ui;
    }
    return b;
}
// R a = b
__device__ inline double wave_solve_R(const double *Rl, int kk, double dinv, double b)
{
    const int lane = threadIdx.x;
    b = b * dinv;
    for (int i = kk - 1; i > 0; --i) {
        const double ai = lane_value(b, i);
        if (lane < i) b = b - (Rl[lane + i * kGsLd] * dinv) * ai;
    }
    return b;
}
// after pass 1 (slots g[0..kk-1], t at [kk]): a = (S^T S)^-1 g through R, alpha from Pythagoras (a scale only: what it misses
// ends up in R), column j of Gs
__global__ __launch_bounds__(64) void k_gmres_ls1(const double *gt, GmresState *G, const int *flag)
{
    __shared__ double Rl[33 * kGsLd];
    if (*flag) return;
    const int j = G->j, kk = j + 1, lane = threadIdx.x;
    const double g = lane <= kk ? gt[lane] : 0.0;        // (issued before R's load: one round trip for both)
    gsl_load_R(G, kk, Rl);
    const double dinv = wave_inv_diag(Rl, kk);
    const double u = wave_solve_Rt(Rl, kk, dinv, lane < kk ? g : 0.0);
    const double uu = wave_sum(u * u);
    const double a = wave_solve_R(Rl, kk, dinv, u);
    const double t = __shfl(g, kk), est = t - uu;
    const double alpha = est > 1e-24 * t ? sqrt(est) : (t > 0.0 ? 1e-12 * sqrt(t) : 1.0);
    if (lane < kk) { G->coef[lane] = a; G->Gs[lane + j * 34] = a; }
    if (lane == kk) { G->coef[kk] = 1.0 / alpha; G->Gs[kk + j * 34] = alpha; }
}
// after pass 2 (slots c[0..kk-1], d at [kk]): R grows by a column, column j of H = R Gs R^-1 e_j, then the rotations, the residual
// estimate and the loop test exactly as k_gmres_givens
__global__ __launch_bounds__(64) void k_gmres_ls2(const double *cd, int m, GmresState *G, double tol, int *flag, int64_t *iters,
                                                  double *history, int64_t hist_cap, double *res_out)
{
    __shared__ double Rl[34 * kGsLd];
    __shared__ double Gl[34 * 33];
    __shared__ double q[64], pv[64], hv[64], csl[32], snl[32];
    if (*flag) return;
    const int j = G->j, kk = j + 1, lane = threadIdx.x;
    const double c = lane <= kk ? cd[lane] : 0.0;
    if (lane < j) { csl[lane] = G->cs[lane]; snl[lane] = G->sn[lane]; }
    for (int e = lane; e < kk * 34; e += 64) Gl[e] = G->Gs[e];          // columns 0..j of Gs
    gsl_load_R(G, kk, Rl);
    const double dinv = wave_inv_diag(Rl, kk);
    const double r = wave_solve_Rt(Rl, kk, dinv, lane < kk ? c : 0.0); // R^T r = c : the new column of R
    const double rr = wave_sum(r * r);
    const double d = __shfl(c, kk), rho2 = d - rr;
    const double rho = rho2 > 1e-24 * d ? sqrt(rho2) : (d > 0.0 ? 1e-12 * sqrt(d) : 1.0);
    if (lane <= kk) {
        const double v = lane < kk ? r : rho;
        Rl[lane + kk * kGsLd] = v;
        G->R[lane + kk * kGsLd] = v;
    }
    q[lane] = wave_solve_R(Rl, kk, dinv, lane == j ? 1.0 : 0.0);             // R_kk q = e_j : the last column of R_kk^-1
    __syncthreads();
    double sacc = 0.0;                                                  // p = Gs(:, 0..j) q   (upper Hessenberg: row i has columns >= i - 1)
    if (lane <= kk)
        for (int l = lane > 0 ? lane - 1 : 0; l <= j; ++l) sacc += Gl[lane + l * 34] * q[l];
    pv[lane] = sacc;
    __syncthreads();
    sacc = 0.0;                                                         // H(:, j) = R_{kk+1} p
    if (lane <= kk)
        for (int l = lane; l <= kk; ++l) sacc += Rl[lane + l * kGsLd] * pv[l];
    hv[lane] = sacc;
    __syncthreads();
    double *H = G->H + (size_t)j * (m + 1);
    if (lane == 0) {
        for (int i = 0; i < j; ++i) {
            const double h0 = hv[i], h1_ = hv[i + 1];
            hv[i] = csl[i] * h0 + snl[i] * h1_;
            hv[i + 1] = -snl[i] * h0 + csl[i] * h1_;
        }
        const double h0 = hv[j], hn = hv[j + 1];
        const double dd = sqrt(h0 * h0 + hn * hn);
        const double cj = h0 / dd, sj = hn / dd;
        G->cs[j] = cj;
        G->sn[j] = sj;
        hv[j] = dd;
        hv[j + 1] = 0.0;
        const double gj = G->g[j];
        G->g[j + 1] = -sj * gj;
        G->g[j] = cj * gj;
        const double res = fabs(sj * gj);
        const int64_t it = *iters;
        if (history && it < hist_cap) history[it] = res * res;
        *iters = it + 1;
        *res_out = res * res;
        G->j = j + 1;
        if (!(res > tol)) *flag = 1;
    }
    __syncthreads();
    if (lane <= kk) H[lane] = hv[lane];
}
// end of a cycle: the coefficients of the orthonormal basis (k_gmres_solve_y) become those of the stored columns, y <- R^-1 y
__global__ __launch_bounds__(64) void k_gmres_ls_y(GmresState *G)
{
    __shared__ double Rl[33 * kGsLd];
    const int k = G->j, lane = threadIdx.x;
    if (k <= 0) return;
    const double y = lane < k ? G->y[lane] : 0.0;
    gsl_load_R(G, k, Rl);
    const double a = wave_solve_R(Rl, k, wave_inv_diag(Rl, k), y);
    if (lane < k) G->y[lane] = a;
}

// ---- Lanczos (src/eigensolver.f90:27-90) -------------------------------------------------
// w = w - alpha*q_i - beta*q_{i-1}   (eigensolver.f90:69; beta = sqrt(sum(nrm2)), q_prev may be null)
struct FLanczosW {
    static constexpr bool kDot = false;
    double *w; const double *qi, *qprev; ScalarRef alpha, nrm2; double a = 0.0, b = 0.0;
    __device__ bool prepare(double *red)
    {
        a = load_scalar<kBlock>(alpha, red);
        if (qprev) b = sqrt(load_scalar<kBlock>(nrm2, red));
        return true;
    }
    __device__ void one(int64_t i)
    {
        double wv = w[i] - a * qi[i];
        if (qprev) wv = wv - b * qprev[i];
        w[i] = wv;
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        double2 wv = ld2<NT>(w, i);
        const double2 q = ld2<NT>(qi, i);
        wv.x = wv.x - a * q.x;
        wv.y = wv.y - a * q.y;
        if (qprev) {
            const double2 p = ld2<NT>(qprev, i);
            wv.x = wv.x - b * p.x;
            wv.y = wv.y - b * p.y;
        }
        st2<NT>(w, i, wv);
    }
    __device__ void single(int64_t i) { one(i); }
    __device__ void finish(double *) {}
};
// T(2,i) = alpha ; T(3,i) = T(1,i) = beta      (eigensolver.f90:78-80)
__global__ __launch_bounds__(kBlock) void k_lanczos_record(ScalarRef alpha, ScalarRef nrm2, int has_beta,
                                                           double *T3 /* 3 x n, column-major */, int col)
{
    __shared__ double red[kBlock / 64];
    const double a = load_scalar<kBlock>(alpha, red);
    double b = 0.0;
    if (has_beta) b = sqrt(load_scalar<kBlock>(nrm2, red));
    if (threadIdx.x == 0) {
        T3[3 * col + 1] = a;
        if (has_beta) { T3[3 * col + 2] = b; T3[3 * col + 0] = b; }
    }
}

// ---- generalized Lanczos (src/eigensolver.f90:95-155) ---------------------------------------
// v = w - beta*z_prev (beta = sqrt(sum(nrm2)); z_prev may be null) ; partial sum(v*q)   (:133-134, :150-151)
struct FGlV {
    const double *w, *zprev, *q; double *v; ScalarRef nrm2; double *part; double b = 0.0, s = 0.0;
    __device__ bool prepare(double *red)
    {
        if (zprev) b = sqrt(load_scalar<kBlock>(nrm2, red));
        return true;
    }
    __device__ void one(int64_t i)
    {
        double vv = w[i];
        if (zprev) vv = vv - b * zprev[i];
        v[i] = vv;
        s += vv * q[i];
    }
    template <bool NT> __device__ void pair(int64_t i)
    {
        double2 vv = ld2<NT>(w, i);
        if (zprev) {
            const double2 zz = ld2<NT>(zprev, i);
            vv.x = vv.x - b * zz.x;
            vv.y = vv.y - b * zz.y;
        }
        st2<NT>(v, i, vv);
        const double2 qq = ld2<NT>(q, i);
        s += vv.x * qq.x;
        s += vv.y * qq.y;
    }
    __device__ void single(int64_t i) { one(i); }
    __device__ void finish(double *red) { put_partial(s, part, red); }
};
// y = y - a*x   (a = a device scalar)                                                   (:135)
struct FSubScaled {
    static constexpr bool kDot = false;
    double *y; const double *x; ScalarRef a; double av = 0.0;
    __device__ bool prepare(double *red) { av = load_scalar<kBlock>(a, red); return true; }
    template <bool NT> __device__ void pair(int64_t i)
    {
        double2 yy = ld2<NT>(y, i); const double2 xx = ld2<NT>(x, i);
        yy.x = yy.x - av * xx.x; yy.y = yy.y - av * xx.y; st2<NT>(y, i, yy);
    }
    __device__ void single(int64_t i) { y[i] = y[i] - av * x[i]; }
    __device__ void finish(double *) {}
};

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
int dot_grid(int64_t n)
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
namespace {
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
}  // namespace

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

namespace {

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
int finish_dots(sgm_solver s, sgm_mat A, const int *ks, int nk, const int (*vecs)[2] = nullptr, bool use_flag = false,
                int gen = INT32_MAX, double *const *halo_of = nullptr)
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

struct Views {       // per-part pointer tables for spmv_parts / pc
    std::vector<const double *> cx;
    std::vector<double *> y;
    std::vector<const double *> w;
    std::vector<double *> p0, p1;
    std::vector<const int *> flags;
};

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

// ---------------------------------------------------------------------------------- CG
enum { C_PQ = 0, C_RR0 = 1, C_RR1 = 2 };
enum { V_P = 0, V_Q = 1, V_R = 2, V_Z = 3 };

// The RMAX row sums of one thread of a single-workgroup solver (rows tid, tid + 1024, ...), x gathered out of LDS
// (`pl`), side by side: slot e of every row is requested before any of them is used, so a row's entries are still
// added left to right but the thread waits for one round trip per SLOT, not per entry (rows one after the other:
// 16 us per CG iteration at n = 1e4, five entries per row).  SL: sliced form (a0 = code words, a1 = offset
// dictionary, val = sval); otherwise CSR (a0 = rowptr, a1 = col).
template <int RMAX, bool SL>
__device__ inline void small_row_sums(double (&q)[RMAX], const double *pl, int32_t n, int32_t sw,
                                      const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                      const double *__restrict__ val)
{
    constexpr int BLOCK = 1024;
    const int tid = threadIdx.x;
    if (SL) {
        uint32_t cw[RMAX];
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            q[u] = 0.0;
            cw[u] = i < n ? (uint32_t)rowptr[i] : 0xffffffffu;
        }
        constexpr int H = RMAX > 5 ? (RMAX + 1) / 2 : RMAX;        // rows side by side (all ten: 71-89 registers spilled)
#pragma unroll
        for (int h0 = 0; h0 < RMAX; h0 += H)
            for (int32_t e = 0; e < sw; ++e) {
                double v[H];
#pragma unroll
                for (int u = h0; u < h0 + H && u < RMAX; ++u) {
                    const int32_t i = tid + u * BLOCK;
                    if (((cw[u] >> (4 * e)) & 15u) != 15u) v[u - h0] = val[((i >> 9) * sw + e) * 512 + (i & 511)];
                }
#pragma unroll
                for (int u = h0; u < h0 + H && u < RMAX; ++u) {
                    const int32_t i = tid + u * BLOCK;
                    const uint32_t cd = (cw[u] >> (4 * e)) & 15u;
                    if (cd != 15u) q[u] = q[u] + v[u - h0] * pl[i + col[cd]];
                }
            }
#pragma unroll
        for (int u = 0; u < RMAX; ++u) q[u] = 0.0 + q[u];
        return;
    }
    int32_t k0[RMAX], len[RMAX];
    int32_t longest = 0;
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = tid + u * BLOCK;
        k0[u] = 0; len[u] = 0; q[u] = 0.0;
        if (i < n) { k0[u] = rowptr[i]; len[u] = rowptr[i + 1] - k0[u]; }
        longest = max(longest, len[u]);
    }
    for (int32_t e = 0; e < longest; ++e) {
        double v[RMAX]; int32_t c[RMAX];
#pragma unroll
        for (int u = 0; u < RMAX; ++u)
            if (e < len[u]) { v[u] = val[k0[u] + e]; c[u] = col[k0[u] + e]; }
#pragma unroll
        for (int u = 0; u < RMAX; ++u)
            if (e < len[u]) q[u] = q[u] + v[u] * pl[c[u]];
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) q[u] = 0.0 + q[u];          // A%matvec: y = 0 ; y(i) = y(i) + z
}

// A dot product inside a single-workgroup solver: thread t holds the products of its rows t, t + BLOCK, ... (0.0 beyond n).
// Tree order (dot_order = 0): the thread's own rows first, then the block sum.  SEQ (dot_order = 1): the products are
// parked in LDS by row and ONE wave adds them row 0 to row n-1 -- the reference's dot_product order.
template <int BLOCK, int RMAX, bool SEQ>
__device__ inline double small_dot(const double (&prod)[RMAX], int32_t n, double *pr, double *red)
{
    if (!SEQ) {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < RMAX; ++u)
            if ((int32_t)threadIdx.x + u * BLOCK < n) s += prod[u];
        return block_sum<BLOCK>(s, red);
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = threadIdx.x + u * BLOCK;
        if (i < n) pr[i] = prod[u];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const double s = seq_chain_lds(pr, n, 0.0);
        if (threadIdx.x == 0) red[0] = s;
    }
    __syncthreads();
    return red[0];
}
// two at once (BiCGStab's s.t / t.t and r.r / r0.r): SEQ walks them side by side, waves 0 and 1
template <int BLOCK, int RMAX, bool SEQ>
__device__ inline void small_dot2(const double (&prod0)[RMAX], const double (&prod1)[RMAX], int32_t n, double *pr0, double *pr1,
                                  double *red, double &out0, double &out1)
{
    if (!SEQ) {                                             // (each summed as small_dot sums it; the two share their barriers)
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int u = 0; u < RMAX; ++u)
            if ((int32_t)threadIdx.x + u * BLOCK < n) { a += prod0[u]; b += prod1[u]; }
        block_sum2<BLOCK>(a, b, red, out0, out1);
        return;
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = threadIdx.x + u * BLOCK;
        if (i < n) { pr0[i] = prod0[u]; pr1[i] = prod1[u]; }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int w = threadIdx.x >> 6;
        const double s = seq_chain_lds(w ? pr1 : pr0, n, 0.0);
        if ((threadIdx.x & 63) == 0) red[w] = s;
    }
    __syncthreads();
    out0 = red[0];
    out1 = red[1];
}

// ---- CG on a small system: the whole solve in ONE workgroup -----------------------------------------
// Below n ~ 1e5 an iteration of the loop above IS its three launches (about 15 us whatever n is).  A system of up to
// 10240 rows fits one workgroup: p lives in LDS (what the row sums gather from), x and r (and 1 / diag for Jacobi) in the
// registers of the row's thread (rows t, t + 1024, ...), q is consumed where it is formed, the two dot products are
// block sums -- no launch, no grid-wide hand-off inside the loop.  Same statements and operands as FCgR / FCgPX above
// (cg_solvers.f90:129-145, :170-190 with jacobi_solve folded in): row sums left to right in stored order with
// individually rounded products, alpha = res2 / dpr, beta = dnew / res2, the loop test `sqrt(res2) > tolerance` before
// every iteration.  Only the summation order of the dot products differs (compiler-defined in the reference).
// SL: the matrix is read from its sliced form (512-row slices, slot-major values, one word of 4-bit offset codes per
// row: sgm_spmv.hip, k_csr_sl) -- coalesced for rows t, t + 1024, ...; `rowptr` then carries the code words, `col` the
// offset dictionary, `sw` the slots per row.  Otherwise plain CSR arrays (every lane its own row: one CU's address
// pipe limits that to about 4096 rows).
// SEQ (dot_order = 1): both dot products in the reference's order -- the products parked in LDS, one wave adds them first
// row to last (small_dot) -- which makes the whole solve bit-identical to cg_solve / cg_solve_pc.
// (x and b carry no __restrict__: sgm_solver_solve hands caller pointers through, and they may alias.)
template <int RMAX, bool JAC, bool SL, bool SEQ>
__global__ __launch_bounds__(1024) void k_cg_small(
    int32_t n, int32_t sw, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
    double *x, const double *b, const double *__restrict__ idiag, double tol, int64_t it_end,
    int resume, double *__restrict__ wr, double *__restrict__ wp,
    int *flag, int64_t *iters, double *res_out, double *history, int64_t hist_cap)
{
    constexpr int BLOCK = 1024;
    extern __shared__ double pl[];             // p (n entries), the block-sum scratch, then (SEQ) the parked products
    double *red = pl + ((n + 1) & ~1);
    double *pr = red + 16;
    const int tid = threadIdx.x;
    double xr[RMAX], rr[RMAX];             // (row pointers and 1 / diag are re-read where needed: L1 / L2 hits, not registers)
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = tid + u * BLOCK;
        xr[u] = 0.0; rr[u] = 0.0;
        if (i < n) { xr[u] = x[i]; pl[i] = xr[u]; }
    }
    __syncthreads();
    auto row_sums = [&](double (&q)[RMAX]) { small_row_sums<RMAX, SL>(q, pl, n, sw, rowptr, col, val); };
    double res2;
    double prod[RMAX];
    int64_t it = 0;
    if (!resume) {
        // r = b - A x ; z = M^-1 r ; p = z ; res2 = r.z
        double zr[RMAX];
        row_sums(zr);
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            const double q = zr[u];
            zr[u] = 0.0;
            prod[u] = 0.0;
            if (i < n) {
                rr[u] = b[i] - q;
                zr[u] = JAC ? idiag[i] * rr[u] : rr[u];
                prod[u] = rr[u] * zr[u];
            }
        }
        res2 = small_dot<BLOCK, RMAX, SEQ>(prod, n, pr, red);      // (its barriers: every row sum has read x out of LDS)
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            if (i < n) pl[i] = zr[u];
        }
    } else {                                  // a solve that outlives one launch: r, p, res2 and the count come back from memory
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            if (i < n) { rr[u] = wr[i]; pl[i] = wp[i]; }
        }
        res2 = *res_out;
        it = *iters;
    }
    __syncthreads();
    bool conv = !(sqrt(res2) > tol);
    while (!conv && it < it_end) {
        double qv[RMAX];
        row_sums(qv);
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            prod[u] = i < n ? pl[i] * qv[u] : 0.0;
        }
        const double dpr = small_dot<BLOCK, RMAX, SEQ>(prod, n, pr, red);
        const double alpha = res2 / dpr;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            prod[u] = 0.0;
            if (i < n) {
                rr[u] = rr[u] - alpha * qv[u];
                const double zv = JAC ? idiag[i] * rr[u] : rr[u];
                prod[u] = rr[u] * zv;
            }
        }
        const double dnew = small_dot<BLOCK, RMAX, SEQ>(prod, n, pr, red);    // (its barriers: every row sum of this iteration has read p)
        const double beta = dnew / res2;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            if (i < n) {
                const double pv = pl[i];
                const double zv = JAC ? idiag[i] * rr[u] : rr[u];
                xr[u] = xr[u] + alpha * pv;
                pl[i] = zv + beta * pv;
            }
        }
        __syncthreads();
        if (tid == 0 && history && it < hist_cap) history[it] = dnew;
        ++it;
        res2 = dnew;
        conv = !(sqrt(res2) > tol);
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = tid + u * BLOCK;
        if (i < n) {
            x[i] = xr[u];
            if (!conv) { wr[i] = rr[u]; wp[i] = pl[i]; }
        }
    }
    if (tid == 0) { *iters = it; *res_out = res2; *flag = conv ? 1 : 0; }
}

constexpr int kCgSmallMax = 10240;          // 10 rows per thread (sliced matrices): x, r, q in registers (16 rows: 53-168 spilled)
constexpr int kCgSmallMaxCsr = 4096;        // plain CSR arrays: 4 rows per thread (10 rows, n = 1e4: 19 us per iteration, one CU's address pipe)
constexpr int kCgSmallMaxSeq = 10200;       // dot_order = 1: p AND the parked products live in LDS (2 x 10200 doubles + scratch <= 160 KiB)
constexpr int kBiSmallMax = 4096;           // BiCGStab: seven vectors in registers, 4 rows per thread
// (a structured ELLPACK matrix with max_d <= 8 keeps the same sliced form, every slot an entry: its padding slots'
// 0.0 * x(last neighbour) terms are added like the reference's ellpack_matvec_add does)
static bool cg_small_sliced(const Part &p)
{
    return p.scode && p.sval && p.dict && p.opt.csr_sliced && (p.ecol ? p.opt.ell_offset_dict : p.opt.csr_offset_dict) && p.sw <= 8;
}
static bool small_applies(sgm_solver s, sgm_mat A, sgm_pc pc, bool bicg)
{
    if (!(bicg ? s->opt.bicgstab_small : s->opt.cg_small) || s->multi || A->parts.size() != 1 ||
        (A->fmt != SGM_FMT_CSR && A->fmt != SGM_FMT_ELL))
        return false;
    const Part &p = A->parts[0];
    if (p.n < 1 || p.n_halo != 0) return false;
    if (!cg_small_sliced(p) && (A->fmt != SGM_FMT_CSR || !p.rowptr || !p.col || !p.val)) return false;
    int32_t nmax = cg_small_sliced(p) ? kCgSmallMax : kCgSmallMaxCsr;
    if (s->seq) nmax = std::min(nmax, kCgSmallMaxSeq);
    if (bicg) nmax = std::min(nmax, kBiSmallMax);
    if (p.n > nmax) return false;
    // one CU takes about 2.5 us + 0.22 us per 1000 stored slots per iteration (5-point 80^2: 11.3 us, 100^2: 14.9; tridiagonal
    // n = 1e4: 11.0; 7-point 20^3: 16.2); the launch loop 14.3-14.7 whatever the size: beyond ~49k slots the loop it is
    // (BiCGStab: two products per iteration against five launches -- the same break-even)
    if ((cg_small_sliced(p) ? (int64_t)p.n * p.sw : p.nnz) > 49152) return false;
    const int pk = pc ? pc_kind(pc) : 0;
    return pk == 0 || pk == SGM_PC_JACOBI;
}
// more than 64 KiB of dynamic LDS needs the attribute, once per kernel; false = the runtime refused (the caller takes the
// launch loop instead)
static bool allow_lds(const void *fn, size_t bytes)
{
    static std::vector<std::pair<const void *, size_t>> done;
    for (auto &d : done)
        if (d.first == fn && d.second >= bytes) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    done.emplace_back(fn, bytes);
    return true;
}
// returns SGM_OK with *ran = false when the kernel cannot be launched here (LDS request refused)
static int run_cg_small(sgm_solver s, sgm_mat A, double *x, const double *b, sgm_pc pc, bool *ran)
{
    const Part &p = A->parts[0];
    PartWork &w = s->work[0];
    const bool jac = pc && pc_kind(pc) == SGM_PC_JACOBI;
    const bool sliced = cg_small_sliced(p);
    const size_t npad = (size_t)((p.n + 1) & ~1);
    const size_t lds = ((s->seq ? 2 : 1) * npad + 16) * sizeof(double);
    // the reference's loop has no iteration cap; a launch has one (kCgSmallChunk iterations), after which the solve
    // continues in the next launch from r, p and res2 parked in the solver's work vectors -- the host stays in control
    const int64_t kCgSmallChunk = s->small_chunk();
    int flag = 0; int64_t iters = 0; double res = 0.0;
    *ran = true;
    for (int resume = 0;; resume = 1) {
        int64_t it_end = iters + kCgSmallChunk;
        if (s->max_iter > 0) it_end = std::min<int64_t>(it_end, s->max_iter);
#define LS(R, J, S, Q)                                                                                               \
    do {                                                                                                             \
        if (!allow_lds((const void *)k_cg_small<R, J, S, Q>, lds)) { *ran = false; return SGM_OK; }                   \
        hipLaunchKernelGGL((k_cg_small<R, J, S, Q>), dim3(1), dim3(1024), lds, g_rt.stream, p.n, p.sw,                \
                           S ? reinterpret_cast<const int32_t *>(p.scode) : (const int32_t *)p.rowptr,                \
                           S ? (const int32_t *)p.dict : (const int32_t *)p.col, S ? (const double *)p.sval : (const double *)p.val, \
                           x, b, jac ? pc_idiag(pc, 0) : nullptr,                                                     \
                           s->tolerance, it_end, resume, w.vec[V_R], w.vec[V_P], w.flag, w.iters, w.res, w.history,   \
                           s->hist_cap);                                                                              \
    } while (0)
#define LSQ(R, J, S) do { if (s->seq) LS(R, J, S, true); else LS(R, J, S, false); } while (0)
        if (sliced) {
            if (p.n <= 4096) { if (jac) LSQ(4, true, true); else LSQ(4, false, true); }
            else { if (jac) LSQ(10, true, true); else LSQ(10, false, true); }
        } else { if (jac) LSQ(4, true, false); else LSQ(4, false, false); }
#undef LSQ
#undef LS
        SGM_HIP(hipGetLastError());
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (flag || (s->max_iter > 0 && iters >= s->max_iter)) break;
    }
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    return SGM_OK;
}

// ---- CG on a mid-sized system: the whole solve in ONE launch of up to 256 co-resident workgroups ------------------------
// Between k_cg_small's reach (one workgroup, <= 10240 rows) and n ~ 1e6 (256 workgroups x 4096 rows) an iteration of the launch loop is three dependent
// kernels of ~4.4 us each, whatever the traffic (n = 1e5: 13.3 us for 1.4 us of bytes; n = 1e6: 36 us for 15).  Here the
// k_cg_small scheme is spread over G workgroups, one per CU: workgroup b owns RMAX * 1024 consecutive rows -- x and r in the
// registers of the row's thread, its part of p plus a halo of `H` rows either side in LDS, the matrix re-read from its sliced
// form every iteration (L2 / Infinity Cache hits at these sizes) -- and an iteration needs TWO grid-wide hand-offs:
//   (1) q = A p on the own rows, partial p.q -> slot[b]                     | arrive / wait | every workgroup adds the G partials
//   (2) r -= alpha q, z = M^-1 r, partial r.z -> slot[G + b]; the z of its     | arrive / wait | in the same order: same bits
//       first and last H rows -> a global vector                              |               | everywhere, no broadcast
//   (3) x += alpha p, p = z + beta p on the own rows AND on the halo (the neighbours' z from the global vector: the p halo
//       is kept up to date locally, no third hand-off)
// Hand-offs follow the guide's counter recipe (cdna_hip_programming.md section 6, Guideline 16): published doubles leave as sc1
// (agent-scope, write-through) stores, every wave drains its stores, the workgroup joins, lane 0 adds to ONE monotonic
// counter; waiters poll it with relaxed sc1 loads and read the published data with sc1 loads only.  Every wait is bounded:
// a workgroup that gives up raises `abort` -- nothing has been written to x, r, p by then -- and the host runs the launch
// loop instead (grids of <= 256 single-workgroup-per-CU blocks are co-resident on an otherwise idle GPU, but nothing
// promises it).  Same statements and operands as the launch loop / cg_solve (cg_solvers.f90:129-145); only the dot
// products' summation order differs (per-workgroup block sums, then the G partials in index order).
__device__ inline void st_sc1(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_sc1(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a store that stays in the storing XCD's L2 (write-through from the CU's L1 like every store): what the XCD-local variant
// publishes with once every participant has proved to sit on ONE XCD, whose L2 is then the coherence point; readers keep
// their sc1 loads (L1 bypassed, served by that L2)
__device__ inline void st_l2(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void st_pub(double *p, double v, bool l2) { if (l2) st_l2(p, v); else st_sc1(p, v); }
__device__ inline int xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 15;
}

// One grid-wide hand-off = one all-reduced scalar.  No counter and no atomic: workgroup b publishes its partial sum in
// slot[h % 4][b] (an sc1 store, after every wave has drained the sc1 stores of whatever else it publishes with this hand-off);
// thread t < G of EVERY workgroup polls slot[h % 4][t] with sc1 loads until it no longer holds the "not yet written"
// pattern (a NaN with a payload no arithmetic produces), and the block sum of the G values -- same order everywhere -- is the
// scalar.  A slot set is re-armed two hand-offs ahead of its reuse by its owners (slot[(h + 2) % 4][b] when b has passed
// h): every reader of that set's previous use arrived at hand-off h - 1 before anyone could pass it.  Bounded: a poll that
// gives up raises `abort`, and every poll loop looks at it.
constexpr unsigned long long kCoopPoison = 0x7ff8c0de5a5a0001ull;
// The slot sets exist in kCoopReplicas copies, kCoopRepStride doubles apart (lines -- and memory channels -- of their own): a
// publisher writes all of them with ONE wave instruction (lane k stores copy k), a workgroup polls the copy of its XCD.  With
// one copy, 256 workgroups re-reading the same 16 lines made a poll round a queue at one or two channels (2.3 us per
// all-CU hand-off against 0.4 inside one XCD).  Which copy a workgroup polls is a matter of speed only.
constexpr int kCoopReplicas = 8, kCoopRepStride = 4 * 256 + 32, kCoopSecond = kCoopReplicas * kCoopRepStride;      // (second scalar of a hand-off: a region of its own)
constexpr int kCoopSlotDoubles = 2 * kCoopSecond;
#ifdef SGM_COOP_PROBE
// tuning aid (-DSGM_COOP_PROBE builds only): where an iteration's time goes, in 10 ns ticks summed over the launch, as seen by
// thread 0 of workgroup 0.  [0..7] the phases of the iteration, [8..11] inside a hand-off, [15] iterations
__device__ long long g_coop_probe[16];
#define PROBE_T(k) do { if (probing) { const long long t_ = wall_clock64(); pacc[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define PROBE_T(k) do { } while (0)
#endif
__device__ inline bool coop_handoff(double *slots /* replicas x 4 x 256 */, int h, double mine, int wg, int G, bool l2, int *abort, int spin_limit, double *red,
                                    int *lds_ok, double *sum_out, int reps, long long *pacc = nullptr, bool second_region = false)
{
    // (reps = 1, the one-XCD variant: copy 0 only, for its proof of co-location too -- at most 32 pollers)
    const int my_rep = reps == 1 ? 0 : (xcc_id() & (kCoopReplicas - 1));
#ifdef SGM_COOP_PROBE
    const bool probing = pacc != nullptr;
    long long tlast = probing ? wall_clock64() : 0;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's sc1 stores (boundary rows) have left
    __syncthreads();
    PROBE_T(8);
    const int tid = threadIdx.x;
    double *set = slots + (h & 3) * 256;
    if (tid == 0) *lds_ok = 1;
    if (tid < reps) st_pub(set + tid * kCoopRepStride + wg, mine, l2);
    double v = 0.0;
    int ok = 1;
    if (tid < G) {
        int spins = 0;
        for (;;) {
            v = ld_sc1(set + my_rep * kCoopRepStride + tid);
            if (__double_as_longlong(v) != (long long)kCoopPoison) break;
            if (++spins > spin_limit || ((spins & 31) == 0 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0; v = 0.0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    PROBE_T(9);
    __syncthreads();                                         // (lds_ok = 1 is visible before anyone clears it)
    PROBE_T(10);
    if (!ok) *lds_ok = 0;
    const double ssum = block_sum<1024>(v, red);             // (its barriers publish lds_ok)
    PROBE_T(11);
    if (tid < reps) {
        st_pub(slots + ((h + 2) & 3) * 256 + tid * kCoopRepStride + wg, __longlong_as_double((long long)kCoopPoison), l2);
        if (second_region) st_pub(slots + kCoopSecond + ((h + 2) & 3) * 256 + tid * kCoopRepStride + wg, __longlong_as_double((long long)kCoopPoison), l2);
    }
    *sum_out = ssum;
    return *lds_ok != 0;
}
// The same hand-off carrying TWO scalars (BiCGStab's t.s and t.t, r.r and r0.r): the second one through a slot region of its
// own, kCoopSecond doubles further on.  A kernel that uses it passes second_region = true to EVERY hand-off it makes, so that
// both regions' sets are re-armed two hand-offs ahead whichever kind those hand-offs are.
__device__ inline bool coop_handoff2(double *slots, int h, double mine_a, double mine_b, int wg, int G, bool l2, int *abort, int spin_limit, double *red,
                                     int *lds_ok, double *sum_a, double *sum_b, int reps)
{
    const int my_rep = reps == 1 ? 0 : (xcc_id() & (kCoopReplicas - 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int tid = threadIdx.x;
    double *set = slots + (h & 3) * 256;
    if (tid == 0) *lds_ok = 1;
    if (tid < reps) {
        st_pub(set + tid * kCoopRepStride + wg, mine_a, l2);
        st_pub(set + kCoopSecond + tid * kCoopRepStride + wg, mine_b, l2);
    }
    double va = 0.0, vb = 0.0;
    int ok = 1;
    if (tid < G) {
        int spins = 0;
        for (;;) {
            va = ld_sc1(set + my_rep * kCoopRepStride + tid);
            vb = ld_sc1(set + kCoopSecond + my_rep * kCoopRepStride + tid);
            if (__double_as_longlong(va) != (long long)kCoopPoison && __double_as_longlong(vb) != (long long)kCoopPoison) break;
            if (++spins > spin_limit || ((spins & 31) == 0 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0; va = 0.0; vb = 0.0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    if (!ok) *lds_ok = 0;
    double sa, sb;
    block_sum2<1024>(va, vb, red, sa, sb);                   // (red: 32 doubles here; its barriers publish lds_ok)
    if (tid < reps) {
        st_pub(slots + ((h + 2) & 3) * 256 + tid * kCoopRepStride + wg, __longlong_as_double((long long)kCoopPoison), l2);
        st_pub(slots + kCoopSecond + ((h + 2) & 3) * 256 + tid * kCoopRepStride + wg, __longlong_as_double((long long)kCoopPoison), l2);
    }
    *sum_a = sa;
    *sum_b = sb;
    return *lds_ok != 0;
}
// SW > 0: the own rows' matrix entries (SW slots each) are loaded ONCE and live in registers for the whole launch (RMAX * SW
// doubles per thread; what an iteration then reads from memory is the hand-offs); SW = 0: streamed every iteration.
// XL (XCD-local; systems of up to 32 workgroups' rows): the grid is 8 x G workgroups, of which those with blockIdx % 8 == 0 --
// dealt to ONE XCD by the round-robin dispatch -- take part and the others leave at once.  The participants first PROVE the
// co-location: a hand-off of the general (sc1, placement-independent) kind carries 64^(own XCC id), and only a sum of
// G x 64^(own id) -- every participant on this XCD -- lets the launch continue; anything else raises `abort` like a poll that gave up.
// From then on the published doubles leave as stores that STAY in that XCD's L2 and the sc1 polls are L2 hits: a hand-off
// costs a few hundred cycles instead of two trips over the fabric.
// LS (streamed form only): the first LS slots of the own rows are copied into LDS once (beside p) and only the others are
// re-read every iteration -- at 4 rows per thread 3 of a 5-point matrix's 5 slots fit (96 KiB), and what is left of a
// 1e6-row matrix (16 MB) stays in the L2s instead of streaming 40 MB from the Infinity Cache per iteration.
// RL: r lives in LDS beside p instead of in registers (eight rows per thread: systems of up to 256 x 8192 rows, where the
// launch loop is traffic-bound at 57 us per iteration and everything but the matrix fits the chip).
template <int RMAX, bool JAC, int SW, bool XL, int LS = 0, bool RL = false>
__global__ __launch_bounds__(1024) void k_cg_coop(
    int32_t n, int32_t sw, int32_t H, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict, const double *__restrict__ sval,
    double *x, const double *b, const double *__restrict__ idiag, double tol, int64_t it_end, int resume,
    double *__restrict__ wr, double *__restrict__ wp, double *gz /* n: the exchanged z rows */, double *slots /* 4 x 256 */,
    int *abort, int h0 /* number of the first hand-off of this launch */, int spin_limit,
    int *flag, int64_t *iters, double *res_out, double *history, int64_t hist_cap)
{
    constexpr int BLOCK = 1024, RPW = RMAX * BLOCK;
    extern __shared__ double lds[];
    double *pl = lds;                                   // p of rows r0 - H .. r0 + RPW + H - 1
    double *red = pl + RPW + 2 * H;                      // 16 doubles of block-sum scratch
    int *lds_ok = reinterpret_cast<int *>(red + 16);
    double *ml = red + 32;                               // LS x RPW matrix entries (LS > 0) -- or, RL, the own rows of r
    double *rls = red + 32;
    __shared__ int32_t dl[16];
    if (XL && (blockIdx.x & 7) != 0) return;
    const int tid = threadIdx.x, G = XL ? (int)(gridDim.x >> 3) : (int)gridDim.x, wg = XL ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int32_t r0 = wg * RPW, r1 = min(n, r0 + RPW);
    if (tid < 16) dl[tid] = dict[tid];
    int h = h0;
    // all-reduce of one partial sum per workgroup (+ whatever this workgroup published before the call)
#ifdef SGM_COOP_PROBE
    long long pacc_[16] = {0}, *pacc = pacc_, tlast = 0;
    const bool probing = wg == 0 && tid == 0;
    auto handoff = [&](double mine, double &total) { const bool ok_ = coop_handoff(slots, h, mine, wg, G, XL, abort, spin_limit, red, lds_ok, &total, XL ? 1 : kCoopReplicas, probing ? pacc : nullptr); ++h; return ok_; };
#else
    auto handoff = [&](double mine, double &total) { const bool ok_ = coop_handoff(slots, h, mine, wg, G, XL, abort, spin_limit, red, lds_ok, &total, XL ? 1 : kCoopReplicas); ++h; return ok_; };
#endif
    auto own_dot = [&](const double (&prod)[RMAX]) {
        double sacc = 0.0;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) sacc += prod[u];
        return block_sum<BLOCK>(sacc, red);
    };
    // the halo of a vector that lives in global memory (published with sc1 stores by its owners) into pl
    uint32_t cwr[RMAX];
    double mv[RMAX][SW > 0 ? SW : 1];
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = r0 + tid + u * BLOCK;
        cwr[u] = i < r1 ? scode[i] : 0xffffffffu;
        if (SW > 0) {
#pragma unroll
            for (int e = 0; e < SW; ++e)
                mv[u][e] = ((cwr[u] >> (4 * e)) & 15u) != 15u ? sval[((int64_t)(i >> 9) * sw + e) * 512 + (i & 511)] : 0.0;
        }
        if (SW == 0 && LS > 0) {
#pragma unroll
            for (int e = 0; e < LS; ++e)
                ml[e * RPW + tid + u * BLOCK] = (e < sw && ((cwr[u] >> (4 * e)) & 15u) != 15u) ? sval[((int64_t)(i >> 9) * sw + e) * 512 + (i & 511)] : 0.0;
        }
    }
    auto row_sums = [&](double (&q)[RMAX]) {
        if (SW > 0) {
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                double z = 0.0;
#pragma unroll
                for (int e = 0; e < SW; ++e) {
                    const uint32_t cd = (cwr[u] >> (4 * e)) & 15u;
                    if (cd != 15u) z = z + mv[u][e] * pl[H + tid + u * BLOCK + dl[cd]];
                }
                q[u] = 0.0 + z;
            }
            return;
        }
        uint32_t cw[RMAX];
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            q[u] = 0.0;
            cw[u] = cwr[u];
        }
        for (int32_t e = 0; e < sw; ++e) {
            double v[RMAX];
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                const int32_t i = r0 + tid + u * BLOCK;
                if (LS > 0 && e < LS) v[u] = ml[e * RPW + tid + u * BLOCK];       // (its own thread wrote it: no barrier needed)
                else if (((cw[u] >> (4 * e)) & 15u) != 15u) v[u] = sval[((int64_t)(i >> 9) * sw + e) * 512 + (i & 511)];
            }
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                const uint32_t cd = (cw[u] >> (4 * e)) & 15u;
                if (cd != 15u) q[u] = q[u] + v[u] * pl[H + tid + u * BLOCK + dl[cd]];
            }
        }
#pragma unroll
        for (int u = 0; u < RMAX; ++u) q[u] = 0.0 + q[u];
    };
    // publish the first / last H own rows of z; (after the hand-off) halo rows of pl <- f(neighbour's z, old halo p)
    auto publish = [&](const double (&zr)[RMAX]) {
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t li = tid + u * BLOCK, i = r0 + li;
            if (i < r1 && (li < H || i >= r1 - H)) st_pub(gz + i, zr[u], XL);
        }
    };
    double xr[RMAX], rr[RL ? 1 : RMAX], prod[RMAX];
    auto R = [&](int u) -> double & { return RL ? rls[tid + u * BLOCK] : rr[RL ? 0 : u]; };
    // ---- start: p = x in LDS (own rows + halo) for r = b - A x
    for (int32_t li = tid; li < RPW + 2 * H; li += BLOCK) {
        const int32_t i = r0 - H + li;
        pl[li] = (i >= 0 && i < n) ? (resume ? wp[i] : x[i]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = r0 + tid + u * BLOCK;
        xr[u] = i < r1 ? x[i] : 0.0;
        R(u) = 0.0;
    }
    __syncthreads();
    double res2;
    int64_t it = 0;
    bool ok = true;
    if (XL) {                                              // the proof of co-location (placement-independent hand-off)
        double total;
        const double mark = __longlong_as_double((long long)(1023 + 6 * xcc_id()) << 52);       // 64^id
        ok = coop_handoff(slots, h, mark, wg, G, false, abort, spin_limit, red, lds_ok, &total, 1);
        ++h;
        if (ok && total != mark * (double)G) {
            if (tid == 0) __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
        }
        if (!ok) return;
    }
    if (!resume) {
        double zr[RMAX];
        row_sums(zr);
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0 + tid + u * BLOCK;
            const double q = zr[u];
            zr[u] = 0.0; prod[u] = 0.0;
            if (i < r1) {
                R(u) = b[i] - q;
                zr[u] = JAC ? idiag[i] * R(u) : R(u);
                prod[u] = R(u) * zr[u];
            }
        }
        const double mine = own_dot(prod);                 // (its barriers: every row sum has read x out of LDS)
        publish(zr);
        ok = handoff(mine, res2);
        if (ok) {
            // p = z: own rows from registers, halo rows from the neighbours' published z
#pragma unroll
            for (int u = 0; u < RMAX; ++u) pl[H + tid + u * BLOCK] = zr[u];
            for (int32_t li = tid; li < 2 * H; li += BLOCK) {
                const int32_t l2 = li < H ? li : RPW + li, i = r0 - H + l2;        // left halo, then right halo
                pl[l2] = (i >= 0 && i < n && (i < r0 || i >= r1)) ? ld_sc1(gz + i) : 0.0;
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0 + tid + u * BLOCK;
            if (i < r1) R(u) = wr[i];
        }
        res2 = *res_out;
        it = *iters;
    }
    __syncthreads();
    bool conv = ok && !(sqrt(res2) > tol);
    while (ok && !conv && it < it_end) {
        double qv[RMAX], zr[RMAX];
#ifdef SGM_COOP_PROBE
        if (probing) tlast = wall_clock64();
#endif
        row_sums(qv);
        PROBE_T(0);
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0 + tid + u * BLOCK;
            prod[u] = i < r1 ? pl[H + tid + u * BLOCK] * qv[u] : 0.0;
        }
        double mine = own_dot(prod), dpr, dnew;
        PROBE_T(1);
        ok = handoff(mine, dpr);                            // ---- hand-off 1: p.q
        if (!ok) break;
#ifdef SGM_COOP_PROBE
        if (probing) tlast = wall_clock64();
#endif
        PROBE_T(2);
        const double alpha = res2 / dpr;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0 + tid + u * BLOCK;
            prod[u] = 0.0; zr[u] = 0.0;
            if (i < r1) {
                R(u) = R(u) - alpha * qv[u];
                zr[u] = JAC ? idiag[i] * R(u) : R(u);
                prod[u] = R(u) * zr[u];
            }
        }
        mine = own_dot(prod);
        PROBE_T(3);
        publish(zr);
        PROBE_T(4);
        ok = handoff(mine, dnew);                           // ---- hand-off 2: r.z and the boundary rows of z
        if (!ok) break;
#ifdef SGM_COOP_PROBE
        if (probing) tlast = wall_clock64();
#endif
        const double beta = dnew / res2;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0 + tid + u * BLOCK;
            if (i < r1) {
                const double pv = pl[H + tid + u * BLOCK];
                xr[u] = xr[u] + alpha * pv;
                pl[H + tid + u * BLOCK] = zr[u] + beta * pv;
            }
        }
        for (int32_t li = tid; li < 2 * H; li += BLOCK) {
            const int32_t l2 = li < H ? li : RPW + li, i = r0 - H + l2;
            if (i >= 0 && i < n && (i < r0 || i >= r1)) pl[l2] = ld_sc1(gz + i) + beta * pl[l2];
        }
        PROBE_T(5);
        __syncthreads();
        PROBE_T(6);
#ifdef SGM_COOP_PROBE
        if (probing) pacc[15] += 1;
#endif
        if (wg == 0 && tid == 0 && history && it < hist_cap) history[it] = dnew;
        ++it;
        res2 = dnew;
        conv = !(sqrt(res2) > tol);
    }
    if (!ok) return;                                        // (the host puts the caller's x back and takes the launch loop)
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = r0 + tid + u * BLOCK;
        if (i < r1) {
            x[i] = xr[u];
            if (!conv) { wr[i] = R(u); wp[i] = pl[H + tid + u * BLOCK]; }
        }
    }
    if (wg == 0 && tid == 0) { *iters = it; *res_out = res2; *flag = conv ? 1 : 0; }
#ifdef SGM_COOP_PROBE
    if (probing)
        for (int k = 0; k < 16; ++k) g_coop_probe[k] = pacc[k];
#endif
}
#ifdef SGM_COOP_PROBE
extern "C" int sgm_debug_coop_probe(long long out[16])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_coop_probe), sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

constexpr int kCoopSpinLimit = 1 << 19;       // polls (about a microsecond each) before a hand-off gives up
// sliced stencil matrix on one GPU, plain or Jacobi, tree-order dots, beyond the one-workgroup kernel and up to 256 workgroups
static bool coop_applies(sgm_solver s, sgm_mat A, sgm_pc pc, int *rmax_out, int *halo_out, bool *xl_out, bool bicg = false)
{
    if (!(bicg ? s->opt.bicgstab_small : s->opt.cg_small) || s->multi || s->seq || A->parts.size() != 1 || A->comm ||
        (A->fmt != SGM_FMT_CSR && A->fmt != SGM_FMT_ELL) || prof_on())
        return false;
    const Part &p = A->parts[0];
    // (an ELLPACK matrix in its sliced form is the same arrays: every slot an entry, padding = 0.0 x the last neighbour;
    //  its dictionary's unused entries are 0)
    const int ndict = p.ecol ? 15 : p.ndict;
    if (p.n_halo != 0 || !cg_small_sliced(p) || p.n < 2048 || ndict < 1 || ndict > 15) return false;        // (k_cg_small had its turn already)
    const int pk = pc ? pc_kind(pc) : 0;
    if (pk != 0 && pk != SGM_PC_JACOBI) return false;
    // the stencil's reach in rows (the largest |offset| of THIS matrix's dictionary, kept on the part where the dictionary is
    // built -- never cached on the solver: one handle may serve matrices of the same size and different stencils)
    const int H = (p.dict_reach + 1) & ~1;
    // option cg_coop_variant: low four bits = rows per thread pinned (1, 2, 4, 8; 0 = chosen by size), 16 = never the one-XCD variant
    const int force_rmax = s->opt.cg_coop_variant & 15;
    // XCD-local variant: the whole system on the <= 32 CUs of one XCD, 1, 2 or 3 rows per thread with the matrix in registers
    // (4 rows per thread stream the matrix through one XCD's L2 / fabric port: 8.6-9.8 us per iteration at n = 1e5 .. 1.3e5
    // where the all-CU variant with one row per thread takes ~8.5)
    const bool xl_off = (s->opt.cg_coop_variant & 16) != 0;
    *xl_out = false;
    if (!xl_off && !s->coop_xl_retired && g_rt.num_cu >= 64) {
        for (int rmax : {1, 2, 3, 4}) {
            if (force_rmax ? rmax != force_rmax : rmax == 4) continue;
            // (k_bicg_coop: 1, 2 or 4 rows per thread, and on one XCD only one -- two there take 17.6 us per iteration at
            //  n = 65536 where 64 workgroups of one row per thread on all CUs take 15.0)
            if (bicg && rmax != 1 && !force_rmax) continue;
            if (bicg && rmax == 3) continue;
            const int64_t rpw = (int64_t)rmax * 1024, G = (p.n + rpw - 1) / rpw;
            if (G > std::min(32, g_rt.num_cu / 8) || (rpw + (bicg ? 4 : 2) * H + 48 + (bicg && rmax >= 2 ? 2 * rpw : 0)) * 8 > 160 * 1024) continue;
            *rmax_out = rmax; *halo_out = H; *xl_out = true;
            return true;
        }
    }
    for (int rmax : {1, 2, 4, 8}) {
        if (force_rmax && rmax != force_rmax) continue;
        if (rmax == 8 && bicg) continue;                      // (CG only: r moves into LDS to make room)
        const int64_t rpw = (int64_t)rmax * 1024, G = (p.n + rpw - 1) / rpw;
        // one workgroup per CU (co-residency), LDS: p + halo + scratch <= 160 KiB.  (A halo wider than a workgroup's rows -- the
        // planes of a 3-D grid -- is fine: then every row is published, and the halo is read from several owners' rows.)
        if (G > std::min(256, g_rt.num_cu) || (rpw + (bicg ? 4 : 2) * H + 48 + ((bicg && rmax >= 2) ? 2 * rpw : rmax == 8 ? rpw : 0)) * 8 > 160 * 1024) continue;
        *rmax_out = rmax; *halo_out = H;
        return true;
    }
    return false;
}

// every slot of the cooperative kernels' exchange buffer "not yet written", abort word clear, hand-offs counted from 0
static int coop_arm(sgm_solver s, size_t n /* doubles of exchange vectors in front of the slots */)
{
    std::vector<unsigned long long> pat((size_t)kCoopSlotDoubles, kCoopPoison);
    SGM_HIP(hipMemcpyAsync(s->coop_buf + n, pat.data(), pat.size() * 8, hipMemcpyHostToDevice, g_rt.stream));
    SGM_HIP(hipMemsetAsync(s->coop_buf + n + (size_t)kCoopSlotDoubles, 0, 64, g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    s->coop_base = 0;
    return SGM_OK;
}

// *ran = false: the kernel could not be launched here, or a hand-off gave up -- the caller runs the launch loop from the caller's x
static int run_cg_coop(sgm_solver s, sgm_mat A, double *x, const double *b, sgm_pc pc, int rmax, int H, bool xl, bool *ran)
{
    const Part &p = A->parts[0];
    PartWork &w = s->work[0];
    const bool jac = pc && pc_kind(pc) == SGM_PC_JACOBI;
    const int64_t rpw = (int64_t)rmax * 1024;
    const int G = (int)((p.n + rpw - 1) / rpw);
    // 4 rows per thread (streamed matrix): three slots of it in LDS where they fit beside p and its halo
    const bool ls3 = !xl && rmax == 4 && (size_t)(rpw + 2 * H + 32 + 3 * rpw) * sizeof(double) <= 160 * 1024;
    const size_t lds = (size_t)(rpw + 2 * H + 32 + (ls3 ? 3 * rpw : rmax == 8 ? rpw : 0)) * sizeof(double);
    *ran = false;
    auto arm = [&]() -> int { return coop_arm(s, p.n); };
    if (!s->coop_buf) {
        if (dalloc(&s->coop_buf, (size_t)p.n + (size_t)kCoopSlotDoubles + 64) != SGM_OK) return SGM_OK;
        SGM_TRY(arm());
    }
    double *gz = s->coop_buf, *slots = gz + p.n;
    int *abortw = reinterpret_cast<int *>(slots + (size_t)kCoopSlotDoubles);
    int flag = 0; int64_t iters = 0; double res = 0.0;
    // the caller's x, kept until the first launch has ended: a hand-off that gives up at the LAST join of a launch can leave
    // workgroups that passed it storing their rows of x while the others leave theirs -- an aborted first launch hands the
    // launch loop the caller's x again, not that mixture
    if (!s->x_backup) SGM_TRY(dalloc(&s->x_backup, (size_t)p.n + 2));
    SGM_HIP(hipMemcpyAsync(s->x_backup, x, (size_t)p.n * 8, hipMemcpyDeviceToDevice, g_rt.stream));
    for (int resume = 0;; resume = 1) {
        int64_t it_end = iters + s->small_chunk();
        if (s->max_iter > 0) it_end = std::min<int64_t>(it_end, s->max_iter);
        const int spin = s->opt.coop_spin_limit > 0 ? s->opt.coop_spin_limit : kCoopSpinLimit;
#define LC(R, J, W, X)                                                                                                 \
    do {                                                                                                             \
        if (!allow_lds((const void *)k_cg_coop<R, J, W, X>, lds)) return SGM_OK;                                      \
        hipLaunchKernelGGL((k_cg_coop<R, J, W, X>), dim3(X ? 8 * G : G), dim3(1024), lds, g_rt.stream, p.n, p.sw, H, (const uint32_t *)p.scode, \
                           (const int32_t *)p.dict, (const double *)p.sval, x, b, jac ? pc_idiag(pc, 0) : nullptr,   \
                           s->tolerance, it_end, resume, w.vec[V_R], w.vec[V_P], gz, slots, abortw, s->coop_base & 3, spin, \
                           w.flag, w.iters, w.res, w.history, s->hist_cap);                                           \
    } while (0)
#define LC5(R, J, W, X, L)                                                                                              \
    do {                                                                                                             \
        if (!allow_lds((const void *)k_cg_coop<R, J, W, X, L>, lds)) return SGM_OK;                                   \
        hipLaunchKernelGGL((k_cg_coop<R, J, W, X, L>), dim3(X ? 8 * G : G), dim3(1024), lds, g_rt.stream, p.n, p.sw, H, (const uint32_t *)p.scode, \
                           (const int32_t *)p.dict, (const double *)p.sval, x, b, jac ? pc_idiag(pc, 0) : nullptr,   \
                           s->tolerance, it_end, resume, w.vec[V_R], w.vec[V_P], gz, slots, abortw, s->coop_base & 3, spin, \
                           w.flag, w.iters, w.res, w.history, s->hist_cap);                                           \
    } while (0)
#define LC6(R, J)                                                                                                      \
    do {                                                                                                             \
        if (!allow_lds((const void *)k_cg_coop<R, J, 0, false, 0, true>, lds)) return SGM_OK;                         \
        hipLaunchKernelGGL((k_cg_coop<R, J, 0, false, 0, true>), dim3(G), dim3(1024), lds, g_rt.stream, p.n, p.sw, H, (const uint32_t *)p.scode, \
                           (const int32_t *)p.dict, (const double *)p.sval, x, b, jac ? pc_idiag(pc, 0) : nullptr,   \
                           s->tolerance, it_end, resume, w.vec[V_R], w.vec[V_P], gz, slots, abortw, s->coop_base & 3, spin, \
                           w.flag, w.iters, w.res, w.history, s->hist_cap);                                           \
    } while (0)
#define LCJ(R, W, X) do { if (jac) LC(R, true, W, X); else LC(R, false, W, X); } while (0)
        // the matrix in registers where RMAX * sw doubles fit beside x, r and the temporaries
#define LCW(R, X) do { if (p.sw == 3) LCJ(R, 3, X); else if (p.sw == 5) LCJ(R, 5, X); else if (p.sw == 7) LCJ(R, 7, X); else LCJ(R, 8, X); } while (0)
        constexpr bool stream_env = false;          // (true: never keep the matrix in registers -- measured slower wherever the registers hold it)
        // (RMAX = 4 with the matrix in registers spills 14-76 VGPRs, RMAX = 10 streamed 99-157: not instantiated)
        if (xl) {
            if (rmax == 1 && !stream_env) LCW(1, true);
            else if (rmax == 2 && !stream_env) LCW(2, true);
            else if (rmax == 3 && !stream_env && p.sw <= 5) { if (p.sw == 3) LCJ(3, 3, true); else LCJ(3, 5, true); }
            else if (rmax == 1) LCJ(1, 0, true);
            else if (rmax == 2) LCJ(2, 0, true);
            else if (rmax == 3) LCJ(3, 0, true);
            else LCJ(4, 0, true);
        }
        else if (rmax == 1 && !stream_env) LCW(1, false);
        else if (rmax == 2 && !stream_env) LCW(2, false);
        else if (rmax == 1) LCJ(1, 0, false);
        else if (rmax == 2) LCJ(2, 0, false);
        else if (rmax == 8) { if (jac) LC6(8, true); else LC6(8, false); }
        else if (ls3) { if (jac) LC5(4, true, 0, false, 3); else LC5(4, false, 0, false, 3); }
        else LCJ(4, 0, false);
#undef LCW
#undef LCJ
#undef LC6
#undef LC5
#undef LC
        SGM_HIP(hipGetLastError());
        int habort = 0;
        SGM_HIP(hipMemcpyAsync(&habort, abortw, sizeof(int), hipMemcpyDeviceToHost, g_rt.stream));
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (habort) {
            // a hand-off gave up (the grid was not co-resident, or the GPU is shared).  A resumed solve has moved x already:
            // restart is only exact from the caller's x, which is put back after a first launch
            SGM_TRY(arm());
            if (!resume) SGM_HIP(hipMemcpyAsync(x, s->x_backup, (size_t)p.n * 8, hipMemcpyDeviceToDevice, g_rt.stream));
            if (xl) {
                // (the participants were not dealt to one XCD, or one of them never started: the all-CU variant has its turn)
                s->coop_xl_retired = true;
            } else {
                fprintf(stderr, "[sigma_hip] cooperative CG gave up waiting for a workgroup (grid not co-resident / shared GPU?): "
                                "this solver takes the launch loop from now on\n");
                s->coop_retired = true;
            }
            if (resume) return fail(SGM_ERR_HIP, "cooperative CG aborted in a continued launch");
            return SGM_OK;
        }
        // hand-offs this launch made: 2 per iteration (+ 1 at the start of a fresh solve, + 1 for the XCD-local variant's proof
        // of co-location); only their count mod 4 matters
        s->coop_base = (int)((s->coop_base + 2 * (iters - (resume ? s->coop_iters0 : 0)) + (resume ? 0 : 1) + (xl ? 1 : 0)) & 3);
        s->coop_iters0 = iters;
        if (flag || (s->max_iter > 0 && iters >= s->max_iter)) break;
    }
    *ran = true;
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    if (trace_on())
        fprintf(stderr, "[sigma_hip] cg: one cooperative launch per %lld iterations, %s, %d workgroups x %lld rows\n", (long long)s->small_chunk(),
                xl ? "on one XCD" : "all CUs", G, (long long)rpw);
    return SGM_OK;
}

int run_cg(sgm_solver s, sgm_mat A, double *const *x, const double *const *b, sgm_pc pc)
{
    auto coop = [&](bool *ran) -> int {
        int rmax = 0, H = 0;
        bool xl = false;
        *ran = false;
        for (int attempt = 0; attempt < 2 && !*ran && !s->coop_retired && coop_applies(s, A, pc, &rmax, &H, &xl); ++attempt) {
            SGM_TRY(run_cg_coop(s, A, x[0], b[0], pc, rmax, H, xl, ran));
            if (!*ran && xl) s->coop_xl_retired = true;
            if (!xl) break;                                  // (the XCD-local variant stood down: once more with all CUs)
        }
        return SGM_OK;
    };
    bool ran = false, coop_tried = false;
    if (small_applies(s, A, pc, false)) {
        // one workgroup takes ~1.5 us + 0.22 us per 1000 stored slots per iteration, a few workgroups of one XCD ~5 us whatever
        // the size (round 4: tridiagonal n = 1e4 9.9 vs 5.0 us): the cooperative kernel first where it applies (>= 2048 rows)
        // and the system has more than 12288 slots
        const Part &p0 = A->parts[0];
        if ((cg_small_sliced(p0) ? (int64_t)p0.n * p0.sw : p0.nnz) > 12288) {
            coop_tried = true;
            SGM_TRY(coop(&ran));
            if (ran) return SGM_OK;
        }
        SGM_TRY(run_cg_small(s, A, x[0], b[0], pc, &ran));
        if (ran) return SGM_OK;
    }
    if (!coop_tried) {
        SGM_TRY(coop(&ran));
        if (ran) return SGM_OK;
    }
    const size_t P = s->work.size();
    const int pk = pc ? pc_kind(pc) : 0;
    Views v;
    v.cx.resize(P); v.y.resize(P); v.w.resize(P); v.p0.resize(P); v.flags.resize(P);
    auto W = [&](size_t ip, int k) { return s->work[ip].vec[k]; };
    int grid = 0;

    // q = A x  (x staged into p: a distributed matvec needs the halo slots)
    for (size_t ip = 0; ip < P; ++ip) {
        launch_elem(s->work[ip].n, FCopy{W(ip, V_P), x[ip]}, nullptr);
        v.cx[ip] = W(ip, V_P); v.y[ip] = W(ip, V_Q); v.flags[ip] = s->work[ip].flag;
    }
    SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, nullptr, nullptr, &grid));
    if (pk == 0) {
        for (size_t ip = 0; ip < P; ++ip) {
            const int64_t n = s->work[ip].n;
            s->work[ip].count[C_RR0] = dot_grid(n);
            launch_elem(n, FCgInit{b[ip], W(ip, V_Q), W(ip, V_R), W(ip, V_P), part(s, ip, C_RR0), true}, nullptr);
        }
    } else {
        std::vector<const double *> rr(P); std::vector<double *> zz(P);
        for (size_t ip = 0; ip < P; ++ip) {
            launch_elem(s->work[ip].n, FCgInit{b[ip], W(ip, V_Q), W(ip, V_R), nullptr, nullptr, false}, nullptr);
            rr[ip] = W(ip, V_R); zz[ip] = W(ip, V_Z);
        }
        SGM_TRY(pc_apply_parts(pc, A, rr.data(), zz.data(), nullptr));
        for (size_t ip = 0; ip < P; ++ip) {
            const int64_t n = s->work[ip].n;
            s->work[ip].count[C_RR0] = dot_grid(n);
            launch_elem(n, FCopyDot{W(ip, V_P), W(ip, V_Z), W(ip, V_R), part(s, ip, C_RR0)}, nullptr);
        }
    }
    const int vz[1][2] = {{V_R, pk == 0 ? V_R : V_Z}}, vpq[1][2] = {{V_P, V_Q}};     // operands of r.r / r.z and p.q
    // Row partitions (option dist_halo_fused): p's halo is FORMED where it is used.  The boundary rows of u (r, or z with a
    // preconditioner) travel beside the all-reduce of r.u -- one communication step between "r -= alpha q" and the p update
    // (cg_solvers.f90:138-142) instead of an all-reduce there and an exchange of p in front of the next product -- and the p
    // update runs over the halo slots too: p_halo = u_halo + beta * p_halo, the owner's statement on the owner's operands,
    // hence the owner's bits.  The product then starts with a complete p: no exchange, no wait for one.
    bool fuse = s->multi && s->opt.dist_halo_fused && A->fmt != SGM_FMT_COMPOSITE;
    for (size_t ip = 0; fuse && ip < P; ++ip) fuse = A->parts[ip].ncol_own == A->parts[ip].n;
    std::vector<double *> uext(P);
    for (size_t ip = 0; ip < P; ++ip) uext[ip] = W(ip, pk == 0 ? V_R : V_Z);
    { const int ks[1] = {C_RR0}; SGM_TRY(finish_dots(s, A, ks, 1, vz, false, INT32_MAX, fuse ? uext.data() : nullptr)); }
    if (fuse)
        for (size_t ip = 0; ip < P; ++ip)          // p = u on the halo slots as on the owned rows (cg_solvers.f90:130 / :172)
            if (const int32_t nh = A->parts[ip].n_halo)
                SGM_HIP(hipMemcpyAsync(W(ip, V_P) + s->work[ip].n, uext[ip] + s->work[ip].n, (size_t)nh * 8, hipMemcpyDeviceToDevice, g_rt.stream));
    for (size_t ip = 0; ip < P; ++ip)
        hipLaunchKernelGGL(k_check, dim3(1), dim3(kBlock), 0, g_rt.stream, ref(s, ip, C_RR0), s->tolerance,
                           s->work[ip].flag, s->work[ip].res);

    int64_t k = 0;
    int flag = 0; int64_t iters = 0; double res = 0.0;
    const int64_t batch_max = pc_apply_is_short(pc) ? 16 : 1;
    // one iteration (number k: it picks the parity of the r.r slots) with generation `gen`, relative to its batch
    auto enqueue_iter = [&](int64_t k, int gen) -> int {
        const int cur = (k & 1) ? C_RR1 : C_RR0, nxt = (k & 1) ? C_RR0 : C_RR1;
        // q = A p, partial p.q
        SpmvDots dots;
        for (size_t ip = 0; ip < P; ++ip) {
            v.cx[ip] = W(ip, V_P); v.y[ip] = W(ip, V_Q); v.w[ip] = W(ip, V_P); v.p0[ip] = part(s, ip, C_PQ);
        }
        dots.w = v.w.data(); dots.part_wy = v.p0.data();
        // all parts share one flag value; spmv takes part 0's flag for every launch on
        // this device (identical contents)
        SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, &dots, s->work[0].flag, &grid, gen, false, /*halo_ready=*/fuse));
        for (size_t ip = 0; ip < P; ++ip) s->work[ip].count[C_PQ] = spmv_grid(A->parts[ip]);
        { const int ks[1] = {C_PQ}; SGM_TRY(finish_dots(s, A, ks, 1, vpq, true, gen)); }
        bool fused_pc = false;
        // two-level factors on EVERY part (colour orderings): the r update, both sweeps and the partial sums of r.z in the sweeps' launches
        bool all_fused = pk == SGM_PC_ILDU0 && !s->seq && s->opt.reorder_solve >= 2;
        for (size_t ip = 0; all_fused && ip < P; ++ip) all_fused = pc_cg_fused_rows(pc, ip) > 0;
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            w.count[nxt] = dot_grid(w.n);
            if (pk == 0)
                launch_elem(w.n, FCgR<0>{ref(s, ip, cur), ref(s, ip, C_PQ), W(ip, V_Q), W(ip, V_R), nullptr, nullptr,
                                         part(s, ip, nxt)}, w.flag, gen);
            else if (pk == SGM_PC_JACOBI)
                launch_elem(w.n, FCgR<1>{ref(s, ip, cur), ref(s, ip, C_PQ), W(ip, V_Q), W(ip, V_R), pc_idiag(pc, ip),
                                         W(ip, V_Z), part(s, ip, nxt)}, w.flag, gen);
            else if (const int32_t n0 = all_fused ? pc_cg_fused_rows(pc, ip) : 0) {
                // two-level factors: r -= alpha q on the rows without L entries, then that update for the other rows, both sweeps
                // and the partial sums of r.z in the sweeps' two launches
                launch_elem((int64_t)n0, FCgR<2>{ref(s, ip, cur), ref(s, ip, C_PQ), W(ip, V_Q), W(ip, V_R), nullptr, nullptr, nullptr}, w.flag, gen);
                fused_pc = pc_cg_fused(pc, ip, ref(s, ip, cur), ref(s, ip, C_PQ), W(ip, V_Q), W(ip, V_R), W(ip, V_Z), part(s, ip, nxt), &w.count[nxt],
                                       w.flag, gen);
                if (!fused_pc) return fail(SGM_ERR_HIP, "run_cg: the fused sweeps withdrew after their first step");
            }
            else
                launch_elem(w.n, FCgR<2>{ref(s, ip, cur), ref(s, ip, C_PQ), W(ip, V_Q), W(ip, V_R), nullptr, nullptr,
                                         nullptr}, w.flag, gen);
        }
        if (pk != 0 && pk != SGM_PC_JACOBI && !fused_pc) {
            std::vector<const double *> rr(P); std::vector<double *> zz(P);
            for (size_t ip = 0; ip < P; ++ip) { rr[ip] = W(ip, V_R); zz[ip] = W(ip, V_Z); }
            SGM_TRY(pc_apply_parts(pc, A, rr.data(), zz.data(), v.flags.data()));
            for (size_t ip = 0; ip < P; ++ip) {
                PartWork &w = s->work[ip];
                launch_elem(w.n, FDot2{W(ip, V_R), W(ip, V_Z), nullptr, nullptr, part(s, ip, nxt), nullptr}, w.flag, gen);
            }
        }
        { const int ks[1] = {nxt}; SGM_TRY(finish_dots(s, A, ks, 1, vz, true, gen, fuse ? uext.data() : nullptr)); }
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            launch_elem(w.n + (fuse ? A->parts[ip].n_halo : 0),
                        FCgPX{ref(s, ip, cur), ref(s, ip, C_PQ), ref(s, ip, nxt), pk == 0 ? W(ip, V_R) : W(ip, V_Z),
                              W(ip, V_P), x[ip], s->tolerance, w.flag, gen + 1, w.iters,
                              ip == 0 ? w.history : nullptr, s->hist_cap, w.res, fuse ? w.n : INT64_MAX}, w.flag, gen);
        }
        return SGM_OK;
    };
    // `count` iterations from iteration k0 on, generations 1 .. count
    auto enqueue_group = [&](int64_t k0, int count) -> int {
        for (size_t ip = 0; ip < P; ++ip) hipLaunchKernelGGL(k_flag_norm, dim3(1), dim3(64), 0, g_rt.stream, s->work[ip].flag);
        for (int j = 0; j < count; ++j) SGM_TRY(enqueue_iter(k0 + j, j + 1));
        return SGM_OK;
    };
    const bool graphs = graph_applies(s, A, pc);
    GraphBatch gb;
    for (;;) {
        // the host looks at the stop flag once per batch (a stream synchronisation + three small copies, ~20 us): batches
        // grow with the iterations already done -- at most an eighth of them run past the stop as early-exit kernels
        int64_t batch = batch_max > 1 ? std::min<int64_t>(128, std::max<int64_t>(batch_max, k / 8)) : batch_max;
        if (graphs && batch > kGraphIters) batch -= batch % kGraphIters;     // k stays on the replay grid whatever krylov_graph_after is
        if (s->max_iter > 0) batch = std::min<int64_t>(batch, s->max_iter - k);
        // a solve that has run graph_after() iterations goes on as replays of one captured group of kGraphIters (k is
        // a multiple of it here: the parity of the r.r slots repeats)
        if (graphs && k >= s->graph_after() && k % kGraphIters == 0 && batch >= kGraphIters &&
            gb.ensure([&]() { return enqueue_group(k, kGraphIters); })) {
            const int64_t groups = batch / kGraphIters;
            for (int64_t g = 0; g < groups; ++g) SGM_HIP(hipGraphLaunch(gb.exec, g_rt.stream));
            k += groups * kGraphIters;
        } else {
            SGM_TRY(enqueue_group(k, (int)batch));
            k += batch;
        }
        SGM_HIP(hipGetLastError());
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (flag || s->aborted || (s->max_iter > 0 && k >= s->max_iter)) break;
    }
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    return SGM_OK;
}

// ---------------------------------------------------------------------------- BiCGStab
// partial arrays: parity-indexed dot results
enum { B_RR = 0, B_RHO = 2, B_R0V = 4, B_ST = 6, B_TT = 8 };     // +parity
enum { W_P = 0, W_Q = 1, W_R = 2, W_R0 = 3, W_V = 4, W_S = 5, W_T = 6, W_Z = 7 };

// ---- BiCGStab on a small system: the whole solve in ONE workgroup ---------------------------------------
// The single-workgroup twin of the launch loop below for the reference's own test sizes
// (test/solver_test_advection_diffusion_1d.f90:58-122, n = 1024): the vector a product gathers from (p, then s) lives
// in LDS, x, r, r0, p, v, s and t in the registers of the row's thread (rows t, t + 1024, ...).  Statements and operands
// are bicgstab_solve's / bicgstab_solve_pc's (bicgstab_solvers.f90:140-173, :199-233, jacobi_solve folded in):
// beta = rho / rho_old * alpha / omega, p = r + beta * (p - omega * v), alpha = rho / (r0 . v), s = r - alpha * v,
// omega = (s . t) / (t . t) with the NaN guard of the plain variant, x = x + alpha * p + omega * s, r = s - omega * t.
// With SEQ every dot product adds its products first row to last -- the solve is then bit-identical to the
// reference's; in tree order the iteration count may differ by a few (BiCGStab's residual is not monotone).
template <int RMAX, bool JAC, bool SL, bool SEQ>
__global__ __launch_bounds__(1024) void k_bicgstab_small(
    int32_t n, int32_t sw, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const double *__restrict__ val,
    double *x, const double *b, const double *__restrict__ idiag, double tol, int64_t it_end, int resume,
    double *__restrict__ wr, double *__restrict__ wr0, double *__restrict__ wp, double *__restrict__ wv,
    double *__restrict__ scal /* alpha, omega, rho_old, rho across launches */,
    int *flag, int64_t *iters, double *res_out, double *history, int64_t hist_cap)
{
    constexpr int BLOCK = 1024;
    extern __shared__ double pl[];             // the vector being multiplied (n entries), scratch, then (SEQ) two product arrays
    const int32_t npad = (n + 1) & ~1;
    double *red = pl + npad, *pr0 = red + 32, *pr1 = pr0 + npad;          // (red: two block sums side by side)
    const int tid = threadIdx.x;
    double xr[RMAX], rr[RMAX], r0[RMAX], pp[RMAX], vv[RMAX], prod0[RMAX], prod1[RMAX];
    auto row_sums = [&](double (&q)[RMAX]) { small_row_sums<RMAX, SL>(q, pl, n, sw, rowptr, col, val); };
    double alpha = 1.0, omega = 1.0, rho_old = 1.0, rho = 1.0, res2;       // bicgstab_solvers.f90:144-147
    int64_t it = 0;
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = tid + u * BLOCK;
        xr[u] = 0.0; rr[u] = 0.0; r0[u] = 0.0; pp[u] = 0.0; vv[u] = 0.0;
        if (i < n) { xr[u] = x[i]; pl[i] = xr[u]; }
    }
    __syncthreads();
    if (!resume) {
        // r0 = [M^-1] (b - A x) ; r = r0 ; v = p = 0 ; res2 = r.r ; (rho of the first iteration = r0.r: the same products)
        double q[RMAX];
        row_sums(q);
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            prod0[u] = 0.0;
            if (i < n) {
                const double w = b[i] - q[u];
                r0[u] = JAC ? idiag[i] * w : w;
                rr[u] = r0[u];
                prod0[u] = rr[u] * rr[u];
            }
        }
        res2 = small_dot<BLOCK, RMAX, SEQ>(prod0, n, pr0, red);
        rho = res2;
    } else {
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            if (i < n) { rr[u] = wr[i]; r0[u] = wr0[i]; pp[u] = wp[i]; vv[u] = wv[i]; }
        }
        alpha = scal[0]; omega = scal[1]; rho_old = scal[2]; rho = scal[3];
        res2 = *res_out;
        it = *iters;
    }
    bool conv = !(sqrt(res2) > tol);
    while (!conv && it < it_end) {
        const double beta = rho / rho_old * alpha / omega;
        __syncthreads();                       // every row sum of the previous product has read pl
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            if (i < n) { pp[u] = rr[u] + beta * (pp[u] - omega * vv[u]); pl[i] = pp[u]; }
        }
        __syncthreads();
        double q[RMAX], ss[RMAX];
        row_sums(q);                           // v = [M^-1] A p
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            prod0[u] = 0.0;
            if (i < n) { vv[u] = JAC ? idiag[i] * q[u] : q[u]; prod0[u] = r0[u] * vv[u]; }
        }
        const double r0v = small_dot<BLOCK, RMAX, SEQ>(prod0, n, pr0, red);      // (its barriers: the product has read p)
        alpha = rho / r0v;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            ss[u] = 0.0;
            if (i < n) { ss[u] = rr[u] - alpha * vv[u]; pl[i] = ss[u]; }
        }
        __syncthreads();
        row_sums(q);                           // t = [M^-1] A s
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            prod0[u] = 0.0; prod1[u] = 0.0;
            if (i < n) {
                if (JAC) q[u] = idiag[i] * q[u];
                prod0[u] = ss[u] * q[u];
                prod1[u] = q[u] * q[u];
            }
        }
        double st, tt;
        small_dot2<BLOCK, RMAX, SEQ>(prod0, prod1, n, pr0, pr1, red, st, tt);
        omega = st / tt;
        if (!JAC && isnan(omega)) omega = 0.0;                                   // bicgstab_solvers.f90:165 (plain variant only)
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = tid + u * BLOCK;
            prod0[u] = 0.0; prod1[u] = 0.0;
            if (i < n) {
                xr[u] = xr[u] + alpha * pp[u] + omega * ss[u];
                rr[u] = ss[u] - omega * q[u];
                prod0[u] = rr[u] * rr[u];
                prod1[u] = r0[u] * rr[u];
            }
        }
        rho_old = rho;
        small_dot2<BLOCK, RMAX, SEQ>(prod0, prod1, n, pr0, pr1, red, res2, rho);  // res2 = r.r ; rho of the next iteration = r0.r
        if (tid == 0 && history && it < hist_cap) history[it] = res2;
        ++it;
        conv = !(sqrt(res2) > tol);
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = tid + u * BLOCK;
        if (i < n) {
            x[i] = xr[u];
            if (!conv) { wr[i] = rr[u]; wr0[i] = r0[u]; wp[i] = pp[u]; wv[i] = vv[u]; }
        }
    }
    if (tid == 0) {
        scal[0] = alpha; scal[1] = omega; scal[2] = rho_old; scal[3] = rho;
        *iters = it; *res_out = res2; *flag = conv ? 1 : 0;
    }
}

static int run_bicgstab_small(sgm_solver s, sgm_mat A, double *x, const double *b, sgm_pc pc, bool *ran)
{
    const Part &p = A->parts[0];
    PartWork &w = s->work[0];
    const bool jac = pc && pc_kind(pc) == SGM_PC_JACOBI;
    const bool sliced = cg_small_sliced(p);
    const size_t npad = (size_t)((p.n + 1) & ~1);
    const size_t lds = ((s->seq ? 3 : 1) * npad + 32) * sizeof(double);
    const int64_t chunk = s->small_chunk();
    int flag = 0; int64_t iters = 0; double res = 0.0;
    *ran = true;
    for (int resume = 0;; resume = 1) {
        int64_t it_end = iters + chunk;
        if (s->max_iter > 0) it_end = std::min<int64_t>(it_end, s->max_iter);
#define LS(J, S, Q)                                                                                                  \
    do {                                                                                                             \
        if (!allow_lds((const void *)k_bicgstab_small<4, J, S, Q>, lds)) { *ran = false; return SGM_OK; }             \
        hipLaunchKernelGGL((k_bicgstab_small<4, J, S, Q>), dim3(1), dim3(1024), lds, g_rt.stream, p.n, p.sw,          \
                           S ? reinterpret_cast<const int32_t *>(p.scode) : (const int32_t *)p.rowptr,                \
                           S ? (const int32_t *)p.dict : (const int32_t *)p.col, S ? (const double *)p.sval : (const double *)p.val, \
                           x, b, jac ? pc_idiag(pc, 0) : nullptr, s->tolerance, it_end, resume, w.vec[W_R], w.vec[W_R0], \
                           w.vec[W_P], w.vec[W_V], w.slots, w.flag, w.iters, w.res, w.history, s->hist_cap);          \
    } while (0)
#define LSQ(J, S) do { if (s->seq) LS(J, S, true); else LS(J, S, false); } while (0)
        if (sliced) { if (jac) LSQ(true, true); else LSQ(false, true); }
        else { if (jac) LSQ(true, false); else LSQ(false, false); }
#undef LSQ
#undef LS
        SGM_HIP(hipGetLastError());
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (flag || (s->max_iter > 0 && iters >= s->max_iter)) break;
    }
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    return SGM_OK;
}


// ---- BiCGStab on a mid-sized system: the whole solve in ONE cooperative launch ------------------------------------------
// k_bicgstab_small's statements (bicgstab_solvers.f90:124-177 / :182-237 with a diagonal M) spread over G workgroups the way
// k_cg_coop spreads CG: workgroup b owns RMAX * 1024 rows -- r, p, v (and, one row per thread, x and r0; else those two in
// LDS) in the registers of the row's thread, the vector being multiplied (p, then s) + halo in LDS -- and an iteration needs
// THREE grid-wide hand-offs:
//   v = [M^-1] A p, partial r0.v; boundary rows of v published                      | (b) r0.v        -> alpha
//   s = r - alpha v on the own rows AND on the halo (r's halo is kept, v's just arrived)
//   t = [M^-1] A s, partials t.s, t.t                                               | (d) two scalars -> omega
//   x += alpha p + omega s ; r = s - omega t ; w = p - omega v; partials r.r, r0.r;
//   boundary rows of r and of w published                                           | (e) two scalars -> res2, rho -> beta
//   p = r + beta w on the own rows AND on the halo
// Neither halo needs a hand-off of its own: what a neighbour lacks for p and for s is ONE SCALAR (beta, alpha), which the
// hand-off that carries the vectors' ingredients delivers anyway.  The halo values are formed by the same two statements
// as the owner's (w = p - omega v; p = r + beta w; s = r - alpha v): same bits.  Hand-offs, bounds, abort and fall-back as
// in k_cg_coop (every hand-off re-arms both scalar regions of the slot sets); three exchange vectors (r, w, v).
// The launch loop takes 19.1 / 21.2 / 72.7 us per iteration at n = 1e4 / 1e5 / 1e6.
template <int RMAX, bool JAC, int SW, bool XL>
__global__ __launch_bounds__(1024) void k_bicg_coop(
    int32_t n, int32_t sw, int32_t H, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict, const double *__restrict__ sval,
    double *x, const double *b, const double *__restrict__ idiag, double tol, int64_t it_end, int resume,
    double *__restrict__ wr, double *__restrict__ wr0, double *__restrict__ wp, double *__restrict__ wv, double *__restrict__ scal,
    double *gz /* 3 n: boundary rows of r, w, v */, double *slots, int *abort, int h0, int spin_limit, int *flag, int64_t *iters,
    double *res_out, double *history, int64_t hist_cap)
{
    constexpr int BLOCK = 1024, RPW = RMAX * BLOCK;
    extern __shared__ double lds[];
    double *pl = lds;                                   // the vector being multiplied: rows r0 - H .. r0 + RPW + H - 1
    double *red = pl + RPW + 2 * H;                      // 32 doubles: two block sums side by side
    int *lds_ok = reinterpret_cast<int *>(red + 32);
    // two or more rows per thread: x and r0 (touched once and twice per iteration) live in LDS, not in registers -- with all
    // seven vectors in registers two rows per thread spill 18-80 VGPRs, four ~100
    constexpr bool LDSV = RMAX >= 2;
    double *rh = red + 48;                               // r on the halo rows (2 H)
    double *xs = rh + 2 * H, *r0s = xs + RPW;
    __shared__ int32_t dl[16];
    if (XL && (blockIdx.x & 7) != 0) return;
    const int tid = threadIdx.x, G = XL ? (int)(gridDim.x >> 3) : (int)gridDim.x, wg = XL ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int reps = XL ? 1 : kCoopReplicas;
    const int32_t r0w = wg * RPW, r1w = min(n, r0w + RPW);
    double *gz_r = gz, *gz_w = gz + n, *gz_v = gz + 2 * (size_t)n;
    if (tid < 16) dl[tid] = dict[tid];
    int h = h0;
    auto handoff1 = [&](double mine, double &total) {
        const bool ok_ = coop_handoff(slots, h, mine, wg, G, XL, abort, spin_limit, red, lds_ok, &total, reps, nullptr, true);
        ++h;
        return ok_;
    };
    auto handoff2 = [&](double ma, double mb, double &ta, double &tb) {
        const bool ok_ = coop_handoff2(slots, h, ma, mb, wg, G, XL, abort, spin_limit, red, lds_ok, &ta, &tb, reps);
        ++h;
        return ok_;
    };
    uint32_t cwr[RMAX];
    double mv[RMAX][SW > 0 ? SW : 1];
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = r0w + tid + u * BLOCK;
        cwr[u] = i < r1w ? scode[i] : 0xffffffffu;
        if (SW > 0) {
#pragma unroll
            for (int e = 0; e < SW; ++e)
                mv[u][e] = ((cwr[u] >> (4 * e)) & 15u) != 15u ? sval[((int64_t)(i >> 9) * sw + e) * 512 + (i & 511)] : 0.0;
        }
    }
    // the rows' sums over pl (k_cg_coop's row_sums: same order of additions as every other kernel of the library)
    auto row_sums = [&](double (&q)[RMAX]) {
        if (SW > 0) {
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                double z = 0.0;
#pragma unroll
                for (int e = 0; e < SW; ++e) {
                    const uint32_t cd = (cwr[u] >> (4 * e)) & 15u;
                    if (cd != 15u) z = z + mv[u][e] * pl[H + tid + u * BLOCK + dl[cd]];
                }
                q[u] = 0.0 + z;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < RMAX; ++u) q[u] = 0.0;
        for (int32_t e = 0; e < sw; ++e) {
            double v[RMAX];
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                const int32_t i = r0w + tid + u * BLOCK;
                if (((cwr[u] >> (4 * e)) & 15u) != 15u) v[u] = sval[((int64_t)(i >> 9) * sw + e) * 512 + (i & 511)];
            }
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                const uint32_t cd = (cwr[u] >> (4 * e)) & 15u;
                if (cd != 15u) q[u] = q[u] + v[u] * pl[H + tid + u * BLOCK + dl[cd]];
            }
        }
#pragma unroll
        for (int u = 0; u < RMAX; ++u) q[u] = 0.0 + q[u];
    };
    // the boundary rows of an own-row vector to an exchange vector
    auto publish = [&](double *dst, const double (&w)[RMAX]) {
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t li = tid + u * BLOCK, i = r0w + li;
            if (i < r1w && (li < H || i >= r1w - H)) st_pub(dst + i, w[u], XL);
        }
    };
    auto block_dot = [&](const double (&prod)[RMAX]) {
        double sacc = 0.0;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) sacc += prod[u];
        return block_sum<BLOCK>(sacc, red);
    };
    auto block_dot2 = [&](const double (&pa)[RMAX], const double (&pb)[RMAX], double &sa, double &sb) {
        double a = 0.0, c = 0.0;
#pragma unroll
        for (int u = 0; u < RMAX; ++u) { a += pa[u]; c += pb[u]; }
        block_sum2<BLOCK>(a, c, red, sa, sb);
    };
    double xr[LDSV ? 1 : RMAX], rr[RMAX], r0[LDSV ? 1 : RMAX], pp[RMAX], vv[RMAX], prod0[RMAX], prod1[RMAX];
    auto X = [&](int u) -> double & { return LDSV ? xs[tid + u * BLOCK] : xr[LDSV ? 0 : u]; };
    auto R0 = [&](int u) -> double & { return LDSV ? r0s[tid + u * BLOCK] : r0[LDSV ? 0 : u]; };
    double alpha = 1.0, omega = 1.0, rho_old = 1.0, rho = 1.0, res2 = 0.0;       // bicgstab_solvers.f90:144-147
    int64_t it = 0;
    for (int32_t li = tid; li < RPW + 2 * H; li += BLOCK) {
        const int32_t i = r0w - H + li;
        pl[li] = (i >= 0 && i < n) ? x[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = r0w + tid + u * BLOCK;
        X(u) = i < r1w ? x[i] : 0.0;
        rr[u] = 0.0; R0(u) = 0.0; pp[u] = 0.0; vv[u] = 0.0;
    }
    __syncthreads();
    bool ok = true;
    if (XL) {                                              // the proof of co-location (k_cg_coop)
        double total;
        const double mark = __longlong_as_double((long long)(1023 + 6 * xcc_id()) << 52);
        ok = coop_handoff(slots, h, mark, wg, G, false, abort, spin_limit, red, lds_ok, &total, 1, nullptr, true);
        ++h;
        if (ok && total != mark * (double)G) {
            if (tid == 0) __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = false;
        }
        if (!ok) return;
    }
    bool from_work = resume != 0;                          // first pass of a continued launch: the halos come from the work vectors
    if (!resume) {
        // r0 = [M^-1] (b - A x) ; r = r0 ; v = p = 0 ; res2 = r.r ; rho of the first iteration = r0.r: the same products
        double q[RMAX];
        row_sums(q);
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0w + tid + u * BLOCK;
            prod0[u] = 0.0;
            if (i < r1w) {
                const double w = b[i] - q[u];
                const double w0 = JAC ? idiag[i] * w : w;
                R0(u) = w0;
                rr[u] = w0;
                prod0[u] = rr[u] * rr[u];
            }
        }
        const double mine = block_dot(prod0);
        publish(gz_r, rr);
        publish(gz_w, pp);                                 // (w = p - omega v = 0)
        ok = handoff1(mine, res2);
        rho = res2;
    } else {
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0w + tid + u * BLOCK;
            if (i < r1w) { rr[u] = wr[i]; R0(u) = wr0[i]; pp[u] = wp[i]; vv[u] = wv[i]; }
        }
        alpha = scal[0]; omega = scal[1]; rho_old = scal[2]; rho = scal[3];
        res2 = *res_out;
        it = *iters;
    }
    bool conv = ok && !(sqrt(res2) > tol);
    while (ok && !conv && it < it_end) {
        const double beta = rho / rho_old * alpha / omega;
        __syncthreads();                                   // every row sum of the previous product has read pl
        // p = r + beta (p - omega v): own rows, and the halo rows from their owners' r and w = p - omega v
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t li = tid + u * BLOCK, i = r0w + li;
            if (i < r1w) {
                const double w = pp[u] - omega * vv[u];
                pp[u] = rr[u] + beta * w;
                pl[H + li] = pp[u];
            }
        }
        for (int32_t li = tid; li < 2 * H; li += BLOCK) {
            const int32_t l2 = li < H ? li : RPW + li, i = r0w - H + l2;
            double rv = 0.0, wv_ = 0.0;
            if (i >= 0 && i < n && (i < r0w || i >= r1w)) {
                if (from_work) { rv = wr[i]; wv_ = wp[i] - omega * wv[i]; }
                else { rv = ld_sc1(gz_r + i); wv_ = ld_sc1(gz_w + i); }
            }
            rh[li] = rv;
            pl[l2] = rv + beta * wv_;
        }
        from_work = false;
        __syncthreads();
        double q[RMAX], ss[RMAX];
        row_sums(q);                                       // v = [M^-1] A p
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0w + tid + u * BLOCK;
            prod0[u] = 0.0;
            if (i < r1w) { vv[u] = JAC ? idiag[i] * q[u] : q[u]; prod0[u] = R0(u) * vv[u]; }
        }
        double mine = block_dot(prod0), r0v;               // (its barriers: the product has read p)
        publish(gz_v, vv);
        ok = handoff1(mine, r0v);                           // ---- (b) r0.v ; v's boundary rows
        if (!ok) break;
        alpha = rho / r0v;
        // s = r - alpha v: own rows and halo
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t li = tid + u * BLOCK, i = r0w + li;
            ss[u] = 0.0;
            if (i < r1w) { ss[u] = rr[u] - alpha * vv[u]; pl[H + li] = ss[u]; }
        }
        for (int32_t li = tid; li < 2 * H; li += BLOCK) {
            const int32_t l2 = li < H ? li : RPW + li, i = r0w - H + l2;
            pl[l2] = (i >= 0 && i < n && (i < r0w || i >= r1w)) ? rh[li] - alpha * ld_sc1(gz_v + i) : 0.0;
        }
        __syncthreads();
        row_sums(q);                                       // t = [M^-1] A s
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0w + tid + u * BLOCK;
            prod0[u] = 0.0; prod1[u] = 0.0;
            if (i < r1w) {
                if (JAC) q[u] = idiag[i] * q[u];
                prod0[u] = ss[u] * q[u];
                prod1[u] = q[u] * q[u];
            }
        }
        double ma, mb, st, tt;
        block_dot2(prod0, prod1, ma, mb);
        ok = handoff2(ma, mb, st, tt);                      // ---- (d) t.s, t.t
        if (!ok) break;
        omega = st / tt;
        if (!JAC && isnan(omega)) omega = 0.0;             // bicgstab_solvers.f90:165 (plain variant only)
        double wn[RMAX];
#pragma unroll
        for (int u = 0; u < RMAX; ++u) {
            const int32_t i = r0w + tid + u * BLOCK;
            prod0[u] = 0.0; prod1[u] = 0.0; wn[u] = 0.0;
            if (i < r1w) {
                X(u) = X(u) + alpha * pp[u] + omega * ss[u];
                rr[u] = ss[u] - omega * q[u];
                wn[u] = pp[u] - omega * vv[u];             // (what the next iteration's p update starts from: the neighbours' copy)
                prod0[u] = rr[u] * rr[u];
                prod1[u] = R0(u) * rr[u];
            }
        }
        rho_old = rho;
        block_dot2(prod0, prod1, ma, mb);
        publish(gz_r, rr);
        publish(gz_w, wn);
        ok = handoff2(ma, mb, res2, rho);                   // ---- (e) r.r ; rho of the next iteration = r0.r ; boundary rows of r, w
        if (!ok) break;
        if (wg == 0 && tid == 0 && history && it < hist_cap) history[it] = res2;
        ++it;
        conv = !(sqrt(res2) > tol);
    }
    if (!ok) return;                                        // (the host puts the caller's x back and takes the launch loop)
#pragma unroll
    for (int u = 0; u < RMAX; ++u) {
        const int32_t i = r0w + tid + u * BLOCK;
        if (i < r1w) {
            x[i] = X(u);
            if (!conv) { wr[i] = rr[u]; wr0[i] = R0(u); wp[i] = pp[u]; wv[i] = vv[u]; }
        }
    }
    if (wg == 0 && tid == 0) {
        scal[0] = alpha; scal[1] = omega; scal[2] = rho_old; scal[3] = rho;
        *iters = it; *res_out = res2; *flag = conv ? 1 : 0;
    }
}

// hand-offs of one launch: 3 per iteration, + 1 at the start of a fresh solve, + 1 for the one-XCD variant's proof
static int run_bicg_coop(sgm_solver s, sgm_mat A, double *x, const double *b, sgm_pc pc, int rmax, int H, bool xl, bool *ran)
{
    const Part &p = A->parts[0];
    PartWork &w = s->work[0];
    const bool jac = pc && pc_kind(pc) == SGM_PC_JACOBI;
    const int64_t rpw = (int64_t)rmax * 1024;
    const int G = (int)((p.n + rpw - 1) / rpw);
    const size_t lds = (size_t)(rpw + 4 * H + 48 + (rmax >= 2 ? 2 * rpw : 0)) * sizeof(double);      // p / s + halo, scratch, r's halo, (x, r0)
    const size_t nx = 3 * (size_t)p.n;                                                                // exchange vectors: r, w, v
    *ran = false;
    if (!s->coop_buf) {
        if (dalloc(&s->coop_buf, nx + (size_t)kCoopSlotDoubles + 64) != SGM_OK) return SGM_OK;
        SGM_TRY(coop_arm(s, nx));
    }
    double *gz = s->coop_buf, *slots = gz + nx;
    int *abortw = reinterpret_cast<int *>(slots + (size_t)kCoopSlotDoubles);
    int flag = 0; int64_t iters = 0; double res = 0.0;
    if (!s->x_backup) SGM_TRY(dalloc(&s->x_backup, (size_t)p.n + 2));            // (as in run_cg_coop: the caller's x, for an aborted first launch)
    SGM_HIP(hipMemcpyAsync(s->x_backup, x, (size_t)p.n * 8, hipMemcpyDeviceToDevice, g_rt.stream));
    for (int resume = 0;; resume = 1) {
        int64_t it_end = iters + s->small_chunk();
        if (s->max_iter > 0) it_end = std::min<int64_t>(it_end, s->max_iter);
        const int spin = s->opt.coop_spin_limit > 0 ? s->opt.coop_spin_limit : kCoopSpinLimit;
#define LB(R, J, W, X)                                                                                                 \
    do {                                                                                                             \
        if (!allow_lds((const void *)k_bicg_coop<R, J, W, X>, lds)) return SGM_OK;                                    \
        hipLaunchKernelGGL((k_bicg_coop<R, J, W, X>), dim3(X ? 8 * G : G), dim3(1024), lds, g_rt.stream, p.n, p.sw, H, (const uint32_t *)p.scode, \
                           (const int32_t *)p.dict, (const double *)p.sval, x, b, jac ? pc_idiag(pc, 0) : nullptr, s->tolerance, it_end,  \
                           resume, w.vec[W_R], w.vec[W_R0], w.vec[W_P], w.vec[W_V], w.slots, gz, slots, abortw, s->coop_base & 3, spin,  \
                           w.flag, w.iters, w.res, w.history, s->hist_cap);                                           \
    } while (0)
#define LBJ(R, W, X) do { if (jac) LB(R, true, W, X); else LB(R, false, W, X); } while (0)
#define LBX(R, W) do { if (xl) LBJ(R, W, true); else LBJ(R, W, false); } while (0)
        // the matrix in registers where it fits beside the five vectors (one row per thread: any slice width; two: <= 5 slots)
        constexpr bool stream_env = false;
        if (rmax == 1 && !stream_env) { if (p.sw == 3) LBX(1, 3); else if (p.sw == 5) LBX(1, 5); else if (p.sw == 7) LBX(1, 7); else LBX(1, 8); }
        else if (rmax == 2 && !stream_env && p.sw <= 5) { if (p.sw == 3) LBX(2, 3); else LBX(2, 5); }
        else if (rmax == 1) LBX(1, 0);
        else if (rmax == 2) LBX(2, 0);
        else LBX(4, 0);
#undef LBX
#undef LBJ
#undef LB
        SGM_HIP(hipGetLastError());
        int habort = 0;
        SGM_HIP(hipMemcpyAsync(&habort, abortw, sizeof(int), hipMemcpyDeviceToHost, g_rt.stream));
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (habort) {
            SGM_TRY(coop_arm(s, nx));
            if (!resume) SGM_HIP(hipMemcpyAsync(x, s->x_backup, (size_t)p.n * 8, hipMemcpyDeviceToDevice, g_rt.stream));
            if (xl) s->coop_xl_retired = true;
            else {
                fprintf(stderr, "[sigma_hip] cooperative BiCGStab gave up waiting for a workgroup (grid not co-resident / shared GPU?): "
                                "this solver takes the launch loop from now on\n");
                s->coop_retired = true;
            }
            if (resume) return fail(SGM_ERR_HIP, "cooperative BiCGStab aborted in a continued launch");
            return SGM_OK;
        }
        s->coop_base = (int)((s->coop_base + 3 * (iters - (resume ? s->coop_iters0 : 0)) + (resume ? 0 : 1) + (xl ? 1 : 0)) & 3);
        s->coop_iters0 = iters;
        if (flag || (s->max_iter > 0 && iters >= s->max_iter)) break;
    }
    *ran = true;
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    if (trace_on())
        fprintf(stderr, "[sigma_hip] bicgstab: one cooperative launch per %lld iterations, %s, %d workgroups x %lld rows\n", (long long)s->small_chunk(),
                xl ? "on one XCD" : "all CUs", G, (long long)rpw);
    return SGM_OK;
}

int run_bicgstab(sgm_solver s, sgm_mat A, double *const *x, const double *const *b, sgm_pc pc)
{
    auto coop = [&](bool *ran) -> int {
        int rmax = 0, H = 0;
        bool xl = false;
        *ran = false;
        for (int attempt = 0; attempt < 2 && !*ran && !s->coop_retired && coop_applies(s, A, pc, &rmax, &H, &xl, true); ++attempt) {
            SGM_TRY(run_bicg_coop(s, A, x[0], b[0], pc, rmax, H, xl, ran));
            if (!*ran && xl) s->coop_xl_retired = true;
            if (!xl) break;
        }
        return SGM_OK;
    };
    bool ran = false, coop_tried = false;
    if (small_applies(s, A, pc, true)) {
        // (one workgroup: 8.2 us per iteration at 5k stored slots, 12.9 at 20k; the cooperative kernel ~10.7 whatever the size)
        const Part &p0 = A->parts[0];
        if ((cg_small_sliced(p0) ? (int64_t)p0.n * p0.sw : p0.nnz) > 12288) {
            coop_tried = true;
            SGM_TRY(coop(&ran));
            if (ran) return SGM_OK;
        }
        SGM_TRY(run_bicgstab_small(s, A, x[0], b[0], pc, &ran));
        if (ran) return SGM_OK;
    }
    if (!coop_tried) {
        SGM_TRY(coop(&ran));
        if (ran) return SGM_OK;
    }
    const size_t P = s->work.size();
    const int pk = pc ? pc_kind(pc) : 0;
    Views v;
    v.cx.resize(P); v.y.resize(P); v.w.resize(P); v.p0.resize(P); v.p1.resize(P); v.flags.resize(P);
    auto W = [&](size_t ip, int k) { return s->work[ip].vec[k]; };
    int grid = 0;
    for (size_t ip = 0; ip < P; ++ip) v.flags[ip] = s->work[ip].flag;

    for (size_t ip = 0; ip < P; ++ip) {
        launch_elem(s->work[ip].n, FCopy{W(ip, W_P), x[ip]}, nullptr);
        v.cx[ip] = W(ip, W_P); v.y[ip] = W(ip, W_Q);
    }
    SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, nullptr, nullptr, &grid));
    if (pk) {   // z = b - q ; r0 = M^-1 z
        std::vector<const double *> zz(P); std::vector<double *> r0(P);
        for (size_t ip = 0; ip < P; ++ip) {
            launch_elem(s->work[ip].n, FCgInit{b[ip], W(ip, W_Q), W(ip, W_Z), nullptr, nullptr, false}, nullptr);
            zz[ip] = W(ip, W_Z); r0[ip] = W(ip, W_R0);
        }
        SGM_TRY(pc_apply_parts(pc, A, zz.data(), r0.data(), nullptr));
    }
    for (size_t ip = 0; ip < P; ++ip) {
        const int64_t n = s->work[ip].n;
        s->work[ip].count[B_RR] = s->work[ip].count[B_RHO] = dot_grid(n);
        launch_elem(n, FBiInit{pk ? W(ip, W_R0) : b[ip], W(ip, W_Q), pk == 0, W(ip, W_R0), W(ip, W_R), W(ip, W_V),
                               W(ip, W_P), part(s, ip, B_RR), part(s, ip, B_RHO)}, nullptr);
    }
    const int v_rr_rho[2][2] = {{W_R, W_R}, {W_R0, W_R}}, v_r0v[1][2] = {{W_R0, W_V}}, v_st_tt[2][2] = {{W_S, W_T}, {W_T, W_T}};
    if (s->seq) { const int ks[2] = {B_RR, B_RHO}; SGM_TRY(finish_dots(s, A, ks, 2, v_rr_rho)); }
    else { const int ks[3] = {B_RR, B_RR + 1, B_RHO}; SGM_TRY(finish_dots(s, A, ks, 3)); }

    int64_t k = 0;
    int flag = 0; int64_t iters = 0; double res = 0.0;
    const int64_t batch_max = pc_apply_is_short(pc) ? 16 : 1;
    auto enqueue_test = [&](int cur) {   // loop test only (no p update): used after the last batch
        for (size_t ip = 0; ip < P; ++ip)
            hipLaunchKernelGGL(k_check, dim3(1), dim3(kBlock), 0, g_rt.stream, ref(s, ip, B_RR + cur), s->tolerance,
                               s->work[ip].flag, s->work[ip].res);
    };
    auto enqueue_iter = [&](int64_t k) -> int {
        const int c = (int)(k & 1), o = c ^ 1;
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            BiScalars S{ref(s, ip, B_RR + c), ref(s, ip, B_RHO + c), ref(s, ip, B_RHO + o), ref(s, ip, B_R0V + o),
                        ref(s, ip, B_ST + o), ref(s, ip, B_TT + o), ref(s, ip, B_R0V + c), ref(s, ip, B_ST + c),
                        ref(s, ip, B_TT + c), k == 0, pk == 0};
            launch_elem(w.n, FBiP{S, W(ip, W_R), W(ip, W_V), W(ip, W_P), s->tolerance, w.flag, w.iters,
                                  ip == 0 ? w.history : nullptr, s->hist_cap, w.res}, w.flag);
        }
        // v = [M^-1] A p ; r0.v
        SpmvDots dots;
        for (size_t ip = 0; ip < P; ++ip) {
            v.cx[ip] = W(ip, W_P); v.y[ip] = pk ? W(ip, W_Z) : W(ip, W_V);
            v.w[ip] = W(ip, W_R0); v.p0[ip] = part(s, ip, B_R0V + c);
        }
        dots.w = v.w.data(); dots.part_wy = v.p0.data();
        SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, pk ? nullptr : &dots, s->work[0].flag, &grid));
        for (size_t ip = 0; ip < P; ++ip) s->work[ip].count[B_R0V + c] = spmv_grid(A->parts[ip]);
        if (pk) {
            std::vector<const double *> zz(P); std::vector<double *> vv(P);
            for (size_t ip = 0; ip < P; ++ip) { zz[ip] = W(ip, W_Z); vv[ip] = W(ip, W_V); }
            SGM_TRY(pc_apply_parts(pc, A, zz.data(), vv.data(), v.flags.data()));
            for (size_t ip = 0; ip < P; ++ip) {
                PartWork &w = s->work[ip];
                w.count[B_R0V + c] = dot_grid(w.n);
                launch_elem(w.n, FDot2{W(ip, W_R0), W(ip, W_V), nullptr, nullptr, part(s, ip, B_R0V + c), nullptr},
                            w.flag);
            }
        }
        { const int ks[1] = {B_R0V + c}; SGM_TRY(finish_dots(s, A, ks, 1, v_r0v, true)); }
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            launch_elem(w.n, FBiS{ref(s, ip, B_RHO + c), ref(s, ip, B_R0V + c), W(ip, W_R), W(ip, W_V), W(ip, W_S)},
                        w.flag);
        }
        // t = [M^-1] A s ; s.t , t.t
        for (size_t ip = 0; ip < P; ++ip) {
            v.cx[ip] = W(ip, W_S); v.y[ip] = pk ? W(ip, W_Z) : W(ip, W_T);
            v.w[ip] = W(ip, W_S); v.p0[ip] = part(s, ip, B_ST + c); v.p1[ip] = part(s, ip, B_TT + c);
        }
        dots.w = v.w.data(); dots.part_wy = v.p0.data(); dots.part_yy = v.p1.data();
        SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, pk ? nullptr : &dots, s->work[0].flag, &grid));
        for (size_t ip = 0; ip < P; ++ip) s->work[ip].count[B_ST + c] = s->work[ip].count[B_TT + c] = spmv_grid(A->parts[ip]);
        if (pk) {
            std::vector<const double *> zz(P); std::vector<double *> tt(P);
            for (size_t ip = 0; ip < P; ++ip) { zz[ip] = W(ip, W_Z); tt[ip] = W(ip, W_T); }
            SGM_TRY(pc_apply_parts(pc, A, zz.data(), tt.data(), v.flags.data()));
            for (size_t ip = 0; ip < P; ++ip) {
                PartWork &w = s->work[ip];
                w.count[B_ST + c] = w.count[B_TT + c] = dot_grid(w.n);
                launch_elem(w.n, FDot2{W(ip, W_S), W(ip, W_T), W(ip, W_T), W(ip, W_T), part(s, ip, B_ST + c),
                                       part(s, ip, B_TT + c)}, w.flag);
            }
        }
        // ST/TT ids are not adjacent for one parity: two calls keep slots contiguous
        if (s->seq) { const int ks[2] = {B_ST + c, B_TT + c}; SGM_TRY(finish_dots(s, A, ks, 2, v_st_tt, true)); }
        else if (s->reduce_single) { const int ks[2] = {B_ST + c, B_TT + c}; SGM_TRY(finish_dots(s, A, ks, 2)); }      // (one launch collapses both)
        else {
            { const int ks[1] = {B_ST + c}; SGM_TRY(finish_dots(s, A, ks, 1)); }
            { const int ks[1] = {B_TT + c}; SGM_TRY(finish_dots(s, A, ks, 1)); }
        }
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            w.count[B_RR + o] = w.count[B_RHO + o] = dot_grid(w.n);
            launch_elem(w.n, FBiXR{ref(s, ip, B_RHO + c), ref(s, ip, B_R0V + c), ref(s, ip, B_ST + c),
                                   ref(s, ip, B_TT + c), pk == 0, W(ip, W_P), W(ip, W_S), W(ip, W_T), W(ip, W_R0),
                                   x[ip], W(ip, W_R), part(s, ip, B_RR + o), part(s, ip, B_RHO + o)}, w.flag);
        }
        if (s->seq) { const int ks[2] = {B_RR + o, B_RHO + o}; SGM_TRY(finish_dots(s, A, ks, 2, v_rr_rho, true)); }
        else if (s->reduce_single) { const int ks[2] = {B_RR + o, B_RHO + o}; SGM_TRY(finish_dots(s, A, ks, 2)); }
        else {
            { const int ks[1] = {B_RR + o}; SGM_TRY(finish_dots(s, A, ks, 1)); }
            { const int ks[1] = {B_RHO + o}; SGM_TRY(finish_dots(s, A, ks, 1)); }
        }
        return SGM_OK;
    };
    const bool graphs = graph_applies(s, A, pc);
    GraphBatch gb;
    for (;;) {
        // the host looks at the stop flag once per batch (a stream synchronisation + three small copies, ~20 us): batches
        // grow with the iterations already done -- at most an eighth of them run past the stop as early-exit kernels
        int64_t batch = batch_max > 1 ? std::min<int64_t>(128, std::max<int64_t>(batch_max, k / 8)) : batch_max;
        if (graphs && batch > kGraphIters) batch -= batch % kGraphIters;     // k stays on the replay grid whatever krylov_graph_after is
        if (s->max_iter > 0) batch = std::min<int64_t>(batch, s->max_iter - k);
        // (replays of one captured group of kGraphIters iterations once the solve has run long enough: see GraphBatch; the
        //  kernels stop on any nonzero flag, so a group needs no generations)
        if (graphs && k >= s->graph_after() && k % kGraphIters == 0 && batch >= kGraphIters &&
            gb.ensure([&]() { for (int j = 0; j < kGraphIters; ++j) SGM_TRY(enqueue_iter(k + j)); return (int)SGM_OK; })) {
            const int64_t groups = batch / kGraphIters;
            for (int64_t g = 0; g < groups; ++g) SGM_HIP(hipGraphLaunch(gb.exec, g_rt.stream));
            k += groups * kGraphIters;
        } else {
            for (int64_t bi = 0; bi < batch; ++bi, ++k) SGM_TRY(enqueue_iter(k));
        }
        // the loop test of the NEXT iteration decides whether we are done (k_check only ever
        // sets the flag, so an earlier in-batch stop is kept)
        enqueue_test((int)(k & 1));
        SGM_HIP(hipGetLastError());
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (flag || s->aborted || (s->max_iter > 0 && k >= s->max_iter)) break;
    }
    if (s->hist_cap && iters >= 1 && iters <= s->hist_cap)    // res2 after the last iteration
        SGM_HIP(hipMemcpy(s->work[0].history + (iters - 1), s->work[0].res, 8, hipMemcpyDeviceToDevice));
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    return SGM_OK;
}

// ------------------------------------------------------------------------------- GMRES
enum { G_W = 0, G_T = 1, G_X = 2 };     // work vectors: w, tmp, (x staging unused)

int run_gmres(sgm_solver s, sgm_mat A, double *const *x, const double *const *b, sgm_pc pc)
{
    const size_t P = s->work.size();
    const int m = s->restart;
    Views v;
    v.cx.resize(P); v.y.resize(P); v.flags.resize(P);
    auto W = [&](size_t ip, int k) { return s->work[ip].vec[k]; };
    auto Vc = [&](size_t ip, int c) { return s->work[ip].V + (size_t)c * s->work[ip].next; };
    int grid = 0;
    for (size_t ip = 0; ip < P; ++ip) v.flags[ip] = s->work[ip].flag;
    // Gram-Schmidt variant: low-synchronisation CGS-2 (k_gsl: two passes, two reductions per step) unless the option is off or
    // the restart length exceeds its 32-vector kernels; modified Gram-Schmidt (j+2 fused passes) otherwise
    const bool lowsync = s->opt.gmres_cgs2 != 0 && m <= 32;
    // partial array ids.  MGS: 0..m = h column (h_0..h_j, norm at j+1), NRM = m+1 the start norm.
    // low-sync: 0..k = g and t of pass 1, LS2.. = c and d of pass 2, NRM = the start norm
    const int LS2 = 36;
    const int NRM = lowsync ? 71 : m + 1;
    int64_t done_steps = 0;
    int flag = 0; int64_t iters = 0; double res = 0.0;

    auto apply_A = [&](int srcV, int src_col, double *const *dst_w) -> int {
        // dst_w = [M^-1] A src   (src is a column of V or the x staging in W)
        for (size_t ip = 0; ip < P; ++ip) {
            v.cx[ip] = srcV ? Vc(ip, src_col) : W(ip, G_X);
            v.y[ip] = pc ? W(ip, G_T) : dst_w[ip];
        }
        SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, nullptr, s->work[0].flag, &grid));
        if (pc) {
            std::vector<const double *> tt(P);
            for (size_t ip = 0; ip < P; ++ip) tt[ip] = W(ip, G_T);
            SGM_TRY(pc_apply_parts(pc, A, tt.data(), dst_w, v.flags.data()));
        }
        return SGM_OK;
    };
    std::vector<double *> wv(P);
    for (size_t ip = 0; ip < P; ++ip) wv[ip] = W(ip, G_W);

    for (;;) {
        // r = M^-1 (b - A x) ; beta ; v_0 = r / beta
        for (size_t ip = 0; ip < P; ++ip) launch_elem(s->work[ip].n, FCopy{W(ip, G_X), x[ip]}, s->work[ip].flag);
        if (pc) {
            for (size_t ip = 0; ip < P; ++ip) { v.cx[ip] = W(ip, G_X); v.y[ip] = W(ip, G_W); }
            SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, nullptr, s->work[0].flag, &grid));
            std::vector<const double *> tt(P);
            for (size_t ip = 0; ip < P; ++ip) {
                launch_elem(s->work[ip].n, FCgInit{b[ip], W(ip, G_W), W(ip, G_T), nullptr, nullptr, false}, s->work[ip].flag);
                tt[ip] = W(ip, G_T);
            }
            SGM_TRY(pc_apply_parts(pc, A, tt.data(), wv.data(), v.flags.data()));
        } else {
            for (size_t ip = 0; ip < P; ++ip) { v.cx[ip] = W(ip, G_X); v.y[ip] = W(ip, G_T); }
            SGM_TRY(spmv_parts(A, v.cx.data(), v.y.data(), false, nullptr, s->work[0].flag, &grid));
            for (size_t ip = 0; ip < P; ++ip)
                launch_elem(s->work[ip].n, FCgInit{b[ip], W(ip, G_T), W(ip, G_W), nullptr, nullptr, false}, s->work[ip].flag);
        }
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            w.count[NRM] = dot_grid(w.n);
            launch_elem(w.n, FMgs{W(ip, G_W), nullptr, nullptr, ScalarRef{nullptr, 0}, part(s, ip, NRM)}, w.flag);
        }
        { const int ks[1] = {NRM}; SGM_TRY(finish_dots(s, A, ks, 1)); }
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            hipLaunchKernelGGL(k_gmres_start, dim3(1), dim3(kBlock), 0, g_rt.stream, ref(s, ip, NRM), w.gmres,
                               s->tolerance, w.flag, w.res);
            launch_elem(w.n, FScaleInv{Vc(ip, 0), W(ip, G_W), ref(s, ip, NRM)}, w.flag);
        }
        int steps = m;
        if (s->max_iter > 0) steps = (int)std::min<int64_t>(m, s->max_iter - done_steps);
        for (int j = 0; j < steps; ++j) {
            SGM_TRY(apply_A(1, j, wv.data()));
            if (lowsync) {
                const int k = j + 1;                 // stored columns s_0 .. s_j; the step writes s_k
                auto pass = [&](int mode, PartWork &w, double *out) {
#define SGM_GSL(KB)                                                                                             \
    do {                                                                                                        \
        if (mode == 0) hipLaunchKernelGGL((k_gsl<KB, 0>), dim3(dot_grid(w.n)), dim3(kBlock), 0, g_rt.stream, w.n, k, (const double *)w.vec[G_W], \
                                          w.V, w.next, (const double *)w.gmres->coef, out, (const int *)w.flag); \
        else hipLaunchKernelGGL((k_gsl<KB, 1>), dim3(dot_grid(w.n)), dim3(kBlock), 0, g_rt.stream, w.n, k, (const double *)w.vec[G_W], \
                                w.V, w.next, (const double *)w.gmres->coef, out, (const int *)w.flag);          \
    } while (0)
                    if (k <= 4) SGM_GSL(4); else if (k <= 8) SGM_GSL(8); else if (k <= 16) SGM_GSL(16); else SGM_GSL(32);
#undef SGM_GSL
                };
                auto reduce_sum = [&](int id0, int cnt) -> int {
                    for (size_t ip = 0; ip < P; ++ip) {
                        PartWork &w = s->work[ip];
                        hipLaunchKernelGGL(k_reduce_many, dim3(cnt), dim3(kBlock), 0, g_rt.stream, part(s, ip, id0),
                                           dot_grid(w.n), w.slots + id0);
                    }
                    if (!s->multi) return SGM_OK;
                    std::vector<double *> ptrs(P);
                    for (size_t ip = 0; ip < P; ++ip) ptrs[ip] = s->work[ip].slots + id0;
                    return allreduce_slots(A, ptrs.data(), cnt);
                };
                for (size_t ip = 0; ip < P; ++ip) pass(0, s->work[ip], part(s, ip, 0));
                SGM_TRY(reduce_sum(0, k + 1));
                for (size_t ip = 0; ip < P; ++ip)
                    hipLaunchKernelGGL(k_gmres_ls1, dim3(1), dim3(64), 0, g_rt.stream, (const double *)s->work[ip].slots, s->work[ip].gmres,
                                       (const int *)s->work[ip].flag);
                for (size_t ip = 0; ip < P; ++ip) pass(1, s->work[ip], part(s, ip, LS2));
                SGM_TRY(reduce_sum(LS2, k + 1));
                for (size_t ip = 0; ip < P; ++ip) {
                    PartWork &w = s->work[ip];
                    hipLaunchKernelGGL(k_gmres_ls2, dim3(1), dim3(64), 0, g_rt.stream, (const double *)(w.slots + LS2), m, w.gmres,
                                       s->tolerance, w.flag, w.iters, ip == 0 ? w.history : nullptr, s->hist_cap, w.res);
                }
                continue;
            }
            // modified Gram-Schmidt: h_i = w.v_i ; w -= h_i v_i, fused as
            //   pass i: [w -= h_{i-1} v_{i-1}] ; partial w.v_i        (i = 0..j)
            //   pass j+1: w -= h_j v_j ; partial w.w
            for (int i = 0; i <= j + 1; ++i) {
                for (size_t ip = 0; ip < P; ++ip) {
                    PartWork &w = s->work[ip];
                    w.count[i] = dot_grid(w.n);
                    launch_elem(w.n, FMgs{W(ip, G_W), i ? Vc(ip, i - 1) : nullptr, i <= j ? Vc(ip, i) : nullptr,
                                          i ? ref(s, ip, i - 1) : ScalarRef{nullptr, 0}, part(s, ip, i)}, w.flag);
                }
                { const int ks[1] = {i}; SGM_TRY(finish_dots(s, A, ks, 1)); }
            }
            for (size_t ip = 0; ip < P; ++ip) {
                PartWork &w = s->work[ip];
                // v_{j+1} = w / h_{j+1,j} must use the norm BEFORE the rotation -> scale first
                launch_elem(w.n, FScaleInv{Vc(ip, j + 1), W(ip, G_W), ref(s, ip, j + 1)}, w.flag);
                hipLaunchKernelGGL(k_gmres_givens, dim3(1), dim3(kBlock), 0, g_rt.stream, w.partials, kMaxGrid,
                                   w.count[0], s->multi ? 1 : 0, w.slots, m, w.gmres, s->tolerance, w.flag, w.iters,
                                   ip == 0 ? w.history : nullptr, s->hist_cap, w.res);
            }
        }
        done_steps += steps;
        // x = x + V y  (always: also when the loop test fired mid-cycle)
        for (size_t ip = 0; ip < P; ++ip) {
            PartWork &w = s->work[ip];
            hipLaunchKernelGGL(k_gmres_solve_y, dim3(1), dim3(64), 0, g_rt.stream, w.gmres, m);
            if (lowsync) hipLaunchKernelGGL(k_gmres_ls_y, dim3(1), dim3(64), 0, g_rt.stream, w.gmres);      // (x += V y = S (R^-1 y))
            launch_elem(w.n, FGmresUpdate{x[ip], w.V, w.next, w.gmres}, nullptr);
        }
        SGM_HIP(hipGetLastError());
        SGM_TRY(read_state(s, &flag, &iters, &res));
        if (flag || s->aborted || (s->max_iter > 0 && done_steps >= s->max_iter)) break;
    }
    s->last_iterations = iters;
    s->res2 = res;
    s->converged = flag;
    return SGM_OK;
}

}  // namespace

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
    if (s->max_iter > 0 && !s->converged) {
        fail(SGM_ERR_NOT_CONVERGED, "solver stopped at max_iter=%lld with sqrt(res2)=%g > %g",
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

// ---- exported vector statements ------------------------------------------------------
// sgm_lanczos <- lanczos(A, T, Q)  src/eigensolver.f90:27-90: n = nsteps Lanczos steps with full
// re-orthogonalisation against q_1..q_{i-2}; T is the 3 x n band (T(2,:) diagonal, T(1,:)=T(3,:)
// off-diagonal), Q the n_rows x n Lanczos vectors.  The reference draws q_1 from a time-seeded
// RNG (util.f90:72-102); here the caller supplies it (it is normalised like eigensolver.f90:49).
// Vector layout of both Lanczos routines: a single matrix or an in-process partition works on plain global vectors; one
// rank of a matrix distributed over processes on its owned slice, every Lanczos vector with room for the halo behind it
// (it is an SpMV input).  Dot products: per-workgroup partial sums, re-reduced by their consumers (one GPU) or reduced
// to a slot and all-reduced (ranks) -- T is then the same on every rank.
namespace {
struct LzCtx {
    sgm_mat A;
    int64_t nloc = 0, ld = 0;
    int gd = 0;
    bool ranks = false;
    double *slots = nullptr;       // 8 reduced scalars (ranks only)
    int init(sgm_mat A_)
    {
        A = A_;
        ranks = A->comm != nullptr;
        nloc = ranks ? A->parts[0].n : A->nrow;
        const int64_t xl = ranks ? A->parts[0].xlen() : A->nrow;
        ld = (std::max(nloc, xl) + 1) & ~(int64_t)1;          // even leading dimension: 16-B aligned columns
        gd = dot_grid(nloc);
        if (ranks) SGM_TRY(dalloc(&slots, 8));
        return SGM_OK;
    }
    ~LzCtx() { dfree(slots); }
    // the scalar a producer left as `count` partial sums in `part`
    int fin(double *part, int count, int slot, ScalarRef *out)
    {
        if (!ranks) { *out = ScalarRef{part, count}; return SGM_OK; }
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kBlock), 0, g_rt.stream, (const double *)part, count, slots + slot);
        double *ptrs[1] = {slots + slot};
        SGM_TRY(allreduce_slots(A, ptrs, 1));
        *out = ScalarRef{slots + slot, 1};
        return SGM_OK;
    }
    int dot(const double *a, const double *b, double *part, int slot, ScalarRef *out)
    {
        launch_elem(nloc, FDot2{a, b, nullptr, nullptr, part, nullptr}, nullptr);
        return fin(part, gd, slot, out);
    }
    // y = M x (+ partial sums of w . y into part_wy when one leaf kernel can carry them)
    int apply(sgm_mat M, const double *x, double *y) { return matvec_plain(M, x, y); }
};
}  // namespace

int sgm_lanczos(sgm_mat A, int32_t nsteps, const double *q1, double *T_host, double *Q_out, int where)
{
    SGM_TRY(require_init());
    if (!A || nsteps < 2 || !q1 || !T_host) return fail(SGM_ERR_BAD_ARG, "sgm_lanczos: bad argument");
    if (A->nrow != A->ncol) return fail(SGM_ERR_DIMS, "sgm_lanczos: square matrices only");
    LzCtx L;
    SGM_TRY(L.init(A));
    const int64_t n = L.nloc, ld = L.ld;
    struct Bufs {
        double *Q = nullptr, *w = nullptr, *parts = nullptr, *T3 = nullptr;
        ~Bufs() { dfree(Q); dfree(w); dfree(parts); dfree(T3); }
    } m;
    SGM_TRY(dalloc(&m.Q, (size_t)ld * nsteps + 2));
    SGM_TRY(dalloc(&m.w, (size_t)ld + 2));
    SGM_TRY(dalloc(&m.parts, (size_t)4 * kMaxGrid));
    SGM_TRY(dalloc(&m.T3, (size_t)3 * nsteps));
    hipStream_t st = g_rt.stream;
    SGM_HIP(hipMemsetAsync(m.T3, 0, (size_t)3 * nsteps * 8, st));
    SGM_HIP(hipMemsetAsync(m.Q, 0, ((size_t)ld * nsteps + 2) * 8, st));
    double *P_ALPHA = m.parts, *P_NRM = m.parts + kMaxGrid, *P_H[2] = {m.parts + 2 * kMaxGrid, m.parts + 3 * kMaxGrid};
    enum { S_ALPHA = 0, S_NRM = 1, S_H0 = 2 };                    // slots (ranks): S_H0, S_H0 + 1 alternate like P_H
    auto q = [&](int i) { return m.Q + (size_t)(i - 1) * ld; };          // 1-based like the reference
    ScalarRef nrm{nullptr, 0}, alpha{nullptr, 0};
    {   // q_1 = q1 / sqrt(sum(q1*q1))
        Staged s1;
        SGM_TRY(stage_in(s1, q1, n, where, true));
        SGM_TRY(L.dot(s1.dev, s1.dev, P_NRM, S_NRM, &nrm));
        launch_elem(n, FScaleInv{q(1), s1.dev, nrm}, nullptr);
        SGM_HIP(hipStreamSynchronize(st));
    }
    const bool fused = !A->distributed() && A->fmt != SGM_FMT_COMPOSITE;      // one leaf kernel carries q_i . w in its epilogue
    for (int i = 1; i <= nsteps; ++i) {
        // w = A q_i ; alpha = sum(q_i * w)
        if (fused) {
            const double *xs[1] = {q(i)};
            double *ys[1] = {m.w};
            const double *ws[1] = {q(i)};
            double *pw[1] = {P_ALPHA};
            SpmvDots dots;
            dots.w = ws; dots.part_wy = pw;
            SGM_TRY(spmv_parts(A, xs, ys, false, &dots, nullptr, nullptr));
            alpha = ScalarRef{P_ALPHA, spmv_grid(A->parts[0])};
        } else {
            SGM_TRY(L.apply(A, q(i), m.w));
            SGM_TRY(L.dot(q(i), m.w, P_ALPHA, S_ALPHA, &alpha));
        }
        if (i == nsteps) {                                   // eigensolver.f90:87-88
            hipLaunchKernelGGL(k_lanczos_record, dim3(1), dim3(kBlock), 0, st, alpha, alpha, 0, m.T3, i - 1);
            break;
        }
        // w = w - alpha q_i - beta q_{i-1}   (beta of the previous step = sqrt(nrm))
        launch_elem(n, FLanczosW{m.w, q(i), i > 1 ? q(i - 1) : nullptr, alpha, nrm}, nullptr);
        // full re-orthogonalisation: for k = 1..i-2: w = w - sum(q_k*w) q_k  (fused like the GMRES MGS
        // sweep), then beta^2 = sum(w*w)
        const int nre = i - 2 > 0 ? i - 2 : 0;
        ScalarRef hprev{nullptr, 0};
        for (int k = 1; k <= nre + 1; ++k) {
            const double *vprev = k > 1 ? q(k - 1) : nullptr;
            const double *vcur = k <= nre ? q(k) : nullptr;
            double *out = k <= nre ? P_H[k & 1] : P_NRM;
            launch_elem(n, FMgs{m.w, vprev, vcur, hprev, out}, nullptr);
            if (k <= nre) SGM_TRY(L.fin(out, L.gd, S_H0 + (k & 1), &hprev));
            else SGM_TRY(L.fin(out, L.gd, S_NRM, &nrm));
        }
        launch_elem(n, FScaleInv{q(i + 1), m.w, nrm}, nullptr);
        hipLaunchKernelGGL(k_lanczos_record, dim3(1), dim3(kBlock), 0, st, alpha, nrm, 1, m.T3, i - 1);
    }
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipMemcpyAsync(T_host, m.T3, (size_t)3 * nsteps * 8, hipMemcpyDeviceToHost, st));
    if (Q_out)
        SGM_HIP(hipMemcpy2DAsync(Q_out, (size_t)n * 8, m.Q, (size_t)ld * 8, (size_t)n * 8, nsteps,
                                 where == SGM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));
    return SGM_OK;
}

// sgm_generalized_lanczos <- generalized_lanczos(A, B, T, Q)  src/eigensolver.f90:95-155: Lanczos for
// A x = lambda B x.  Every step solves B w = v with the solver the caller set up for B (the reference
// reads B%solver / B%pc, :140), started from the current w = A q_i exactly like `call B%solve(w, v)`.
// No re-orthogonalisation (the reference has none here).  q1 replaces the time-seeded start vector
// and is normalised in the B-norm (:123-124).  A and B may be row-partitioned (the same way).
int sgm_generalized_lanczos(sgm_mat A, sgm_mat B, sgm_solver solver, sgm_pc pc, int32_t nsteps, const double *q1,
                            double *T_host, double *Q_out, int where)
{
    SGM_TRY(require_init());
    if (!A || !B || !solver || nsteps < 2 || !q1 || !T_host) return fail(SGM_ERR_BAD_ARG, "sgm_generalized_lanczos: bad argument");
    if (A->nrow != A->ncol || B->nrow != B->ncol || A->nrow != B->nrow)
        return fail(SGM_ERR_DIMS, "sgm_generalized_lanczos: A and B must be square and of one size");
    // (composites -- the reference's own test runs on one, eigensolver_test_generalized_lanczos.f90:150 -- work on their local
    //  vector layout: the concatenation of this rank's slices of the block vectors, the same for A and B)
    if ((A->comm != B->comm) || A->parts.size() != B->parts.size() || (A->fmt == SGM_FMT_COMPOSITE) != (B->fmt == SGM_FMT_COMPOSITE) ||
        (A->fmt == SGM_FMT_COMPOSITE && (A->blk_row_ptr != B->blk_row_ptr || A->blk_col_ptr != B->blk_col_ptr)))
        return fail(SGM_ERR_UNSUPPORTED, "sgm_generalized_lanczos: A and B must be partitioned (and, composites, blocked) the same way");
    for (size_t ip = 0; ip < A->parts.size(); ++ip)
        if (A->parts[ip].n != B->parts[ip].n || A->parts[ip].row_begin != B->parts[ip].row_begin)
            return fail(SGM_ERR_UNSUPPORTED, "sgm_generalized_lanczos: A and B must be partitioned the same way");
    if (!solver->initialized || solver->nn != B->nrow)
        return fail(SGM_ERR_BAD_ARG, "sgm_generalized_lanczos: the solver has not been set up for B (B%%set_solver)");
    LzCtx L, LB;
    SGM_TRY(L.init(A));
    SGM_TRY(LB.init(B));
    const int64_t n = L.nloc, ld = std::max(L.ld, LB.ld);         // (every vector may be an input of either product)
    struct Bufs {
        double *Q = nullptr, *Z = nullptr, *w = nullptr, *v = nullptr, *parts = nullptr, *T3 = nullptr;
        ~Bufs() { dfree(Q); dfree(Z); dfree(w); dfree(v); dfree(parts); dfree(T3); }
    } m;
    SGM_TRY(dalloc(&m.Q, (size_t)ld * nsteps + 2));
    SGM_TRY(dalloc(&m.Z, (size_t)ld * (nsteps + 1) + 2));        // z(:, 0:n), column 0 stays zero
    SGM_TRY(dalloc(&m.w, (size_t)ld + 2));
    SGM_TRY(dalloc(&m.v, (size_t)ld + 2));
    SGM_TRY(dalloc(&m.parts, (size_t)2 * kMaxGrid));
    SGM_TRY(dalloc(&m.T3, (size_t)3 * nsteps));
    hipStream_t st = g_rt.stream;
    SGM_HIP(hipMemsetAsync(m.T3, 0, (size_t)3 * nsteps * 8, st));
    SGM_HIP(hipMemsetAsync(m.Q, 0, ((size_t)ld * nsteps + 2) * 8, st));
    SGM_HIP(hipMemsetAsync(m.Z, 0, ((size_t)ld * (nsteps + 1) + 2) * 8, st));
    SGM_HIP(hipMemsetAsync(m.w, 0, ((size_t)ld + 2) * 8, st));
    SGM_HIP(hipMemsetAsync(m.v, 0, ((size_t)ld + 2) * 8, st));
    double *P_ALPHA = m.parts, *P_B2 = m.parts + kMaxGrid;
    enum { S_ALPHA = 0, S_B2 = 1 };
    auto q = [&](int i) { return m.Q + (size_t)(i - 1) * ld; };      // 1-based like the reference
    auto z = [&](int i) { return m.Z + (size_t)i * ld; };            // 0-based: z(:, 0:n)
    ScalarRef b2{nullptr, 0}, alpha{nullptr, 0};
    Staged s1;
    {   // q_1 = q1 / sqrt(sum((B q1) * q1)) ; z_1 = B q_1      (q1 staged with halo room: it is multiplied by B)
        double *q1d = nullptr;
        SGM_TRY(dalloc(&q1d, (size_t)ld + 2));
        s1.dev = q1d; s1.owned = true;
        SGM_HIP(hipMemsetAsync(q1d, 0, ((size_t)ld + 2) * 8, st));
        SGM_HIP(hipMemcpyAsync(q1d, q1, (size_t)n * 8, where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
        SGM_TRY(L.apply(B, q1d, m.w));
        SGM_TRY(L.dot(m.w, q1d, P_B2, S_B2, &b2));
        launch_elem(n, FScaleInv{q(1), q1d, b2}, nullptr);
        SGM_TRY(L.apply(B, q(1), z(1)));
        SGM_HIP(hipStreamSynchronize(st));
    }
    for (int i = 1; i <= nsteps - 1; ++i) {
        SGM_TRY(L.apply(A, q(i), m.w));                                                 // w = A q_i
        launch_elem(n, FGlV{m.w, i > 1 ? z(i - 1) : nullptr, q(i), m.v, b2, P_ALPHA}, nullptr);
        SGM_TRY(L.fin(P_ALPHA, L.gd, S_ALPHA, &alpha));
        launch_elem(n, FSubScaled{m.v, z(i), alpha}, nullptr);                          // v = v - alpha z_i
        SGM_HIP(hipGetLastError());
        // call B%solve(w, v): the solver's own loop, x = w in place (initial guess A q_i), b = v
        const int rc = sgm_solver_solve(solver, B, m.w, m.v, pc, SGM_DEVICE);
        if (rc != SGM_OK) return rc;
        // alpha was consumed before the solve (its slot is reused only after this step's record); the NEW beta = sqrt(sum(w*v))
        ScalarRef b2n{nullptr, 0};
        SGM_TRY(L.dot(m.w, m.v, P_B2, S_B2, &b2n));
        b2 = b2n;
        hipLaunchKernelGGL(k_lanczos_record, dim3(1), dim3(kBlock), 0, st, alpha, b2, 1, m.T3, i - 1);
        launch_elem(n, FScaleInv{q(i + 1), m.w, b2}, nullptr);
        launch_elem(n, FScaleInv{z(i + 1), m.v, b2}, nullptr);
    }
    // v = A q_n - beta z_n ; T(2,n) = sum(q_n * v)
    SGM_TRY(L.apply(A, q(nsteps), m.w));
    launch_elem(n, FGlV{m.w, z(nsteps), q(nsteps), m.v, b2, P_ALPHA}, nullptr);
    SGM_TRY(L.fin(P_ALPHA, L.gd, S_ALPHA, &alpha));
    hipLaunchKernelGGL(k_lanczos_record, dim3(1), dim3(kBlock), 0, st, alpha, alpha, 0, m.T3, nsteps - 1);
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipMemcpyAsync(T_host, m.T3, (size_t)3 * nsteps * 8, hipMemcpyDeviceToHost, st));
    if (Q_out)
        SGM_HIP(hipMemcpy2DAsync(Q_out, (size_t)n * 8, m.Q, (size_t)ld * 8, (size_t)n * 8, nsteps,
                                 where == SGM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
    SGM_HIP(hipStreamSynchronize(st));
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
