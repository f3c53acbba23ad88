// BiCGStab / preconditioned BiCGStab (bicgstab_solvers.f90:124-237) on the device: functors of the launch loop, the
// one-workgroup kernel, the cooperative one-launch kernel, the loop (design notes: sgm_solvers.hip).  -ffp-contract=off.
#include "sgm_coop.hpp"

namespace sgm {

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

}  // namespace sgm
