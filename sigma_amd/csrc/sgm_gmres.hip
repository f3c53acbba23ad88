// GMRES(m) on the device (no reference counterpart: parity unpinned by the reference): the cycle bookkeeping kernels, modified
// Gram-Schmidt and the low-synchronisation CGS-2 passes, the loop (design notes: sgm_solvers.hip).  -ffp-contract=off.
#include "sgm_krylov.hpp"

namespace sgm {

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

}  // namespace sgm
