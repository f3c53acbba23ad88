// CG / PCG (cg_solvers.f90:116-194) on the device: the one-workgroup kernel, the cooperative one-launch kernel and the launch
// loop (design notes: sgm_solvers.hip).  Compiled with -ffp-contract=off.
#include "sgm_coop.hpp"

namespace sgm {

// ---------------------------------------------------------------------------------- CG
enum { C_PQ = 0, C_RR0 = 1, C_RR1 = 2 };
enum { V_P = 0, V_Q = 1, V_R = 2, V_Z = 3 };

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

}  // namespace sgm
