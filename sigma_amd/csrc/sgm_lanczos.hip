// Lanczos and generalized Lanczos (src/eigensolver.f90:27-155) on the device vectors.  -ffp-contract=off.
#include "sgm_krylov.hpp"

namespace sgm {

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

}  // namespace sgm

extern "C" {

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

}  // extern "C"
