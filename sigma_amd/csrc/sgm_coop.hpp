// Shared by sgm_cg.hip and sgm_bicgstab.hip: what the one-workgroup and the cooperative (one launch, up to 256 workgroups)
// solvers of small and mid-sized systems are made of -- the row sums out of LDS, the block dots in both orders, the grid-wide
// hand-offs, and the rules that decide when these kernels apply.
#pragma once
#include "sgm_krylov.hpp"

namespace sgm {

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
static __device__ long long g_coop_probe[16];
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

}  // namespace sgm
