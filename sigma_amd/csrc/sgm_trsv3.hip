// Slab-pipelined triangular solves for ILDU(0) factors of 3-D grid-like matrices (gfx950).
//
// The factor rows depend only on r-1, r-w and r-w*h (7-point stencils in natural order:
// ldu_solvers.f90:208-265 walks them one row after the other).  The level-scheduled walkers of
// sgm_pc.hip need one launch per hyperplane i+j+k (298 at 100^3): launch-bound.  Here ONE launch
// does a whole sweep:
//   * the grid is cut into STRIPS of 64 columns (i) and GROUPS of HB lines (j); workgroup b owns
//     line group b for all strips and all planes k; chain wave a of the workgroup owns strip a;
//   * lane l of a chain wave handles column 64a+l; its rows are visited plane by plane, line by line
//     (u = k*HB + jl), skewed by one step per lane (step t = u + l).  The (i-1,j,k) neighbour is then
//     lane l-1's previous result (DPP wave_shr:1; lane 0 takes lane 63 of the wave to its left out of
//     an LDS ring), the (i,j-1,k) neighbour the lane's own previous result (or, for the first line of
//     the group, a value of the workgroup b-1), the (i,j,k-1) neighbour the lane's own result HB
//     steps ago (the wave's LDS result ring);
//   * a step subtracts the three products in the row's STORED order, each rounded on its own; an absent
//     term contributes an exact 0.0 whatever its operand holds  =>  bit-identical to the sequential sweep;
//   * workgroup b only waits for workgroup b-1: two helper waves per workgroup forward the chain waves'
//     result rings to memory / fetch the upstream ones, with sc1 (agent-scope) accesses, valid across
//     XCDs.  The results are their own flags: the result arrays hold a signalling-NaN pattern until
//     written (no subtraction can produce one; the gather / hand-over kernels reset them before every
//     sweep), so a hand-off costs ONE memory round trip.  Every wait loop is bounded and raises an
//     abort word instead of hanging.
// The path is only trusted with a pattern after it reproduced the row-by-row sweeps bit for bit on a
// test vector at setup (sgm_pc.hip).
#include "sgm_internal.hpp"

#include <algorithm>
#include <cstdlib>

namespace sgm {

namespace {

constexpr int kPadPos = 32 * 64;           // positions of padding behind the record arrays (the deepest look-ahead)
constexpr int kEdgeRing = 512;            // lane-63 results a chain wave keeps for the wave to its right
constexpr int kBig = 1 << 29;
// "not yet written": a SIGNALLING NaN no subtraction can produce (arithmetic quiets NaNs) -- the forwarded results are their own flags
constexpr unsigned long long kEmpty = 0x7FF4A5A5A5A5A5A5ull;
typedef double f64x2s __attribute__((ext_vector_type(2)));

struct SlabTri {
    bool on = false;
    int32_t w = 0, h = 0, nk = 0;          // grid: columns, lines per plane, planes
    int32_t NI = 0, NB = 0, HB = 0, S = 0; // strips (chain waves per workgroup), line groups (workgroups), lines per group, steps
    int order = 2;                         // 0: every row subtracts back, up, left; 1: left, up, back; 2: per-row code
    bool regular = false;                  // every row has exactly the dependencies its grid position implies
    int64_t NP = 0;                        // positions = NB * NI * S * 64
    f64x2s *rec = nullptr;                 // device: 2 per position {c_back, c_up}, {c_left, rhs}
    int32_t *code = nullptr;               // device: bit0 has back, bit1 has up, bit2 has left; bits 3.. three 2-bit source ids in stored order
    int32_t *row = nullptr;                // device: position -> row (-1 = padding)
    int32_t *progress = nullptr;           // device: NB*NI forwarded steps (diagnostics), [NB*NI] = abort word
    long long *clk = nullptr;              // device: 2 per (b, a): chain start / end
    int32_t *pos = nullptr;                // device: row -> position (index work only; freed after it)
    int32_t *src[3] = {nullptr, nullptr, nullptr};   // device: position -> entry of the factor's val array per source (-1 = none)
};

}  // namespace

struct Slab3 {
    int32_t n = 0;
    SlabTri L, U;
    double *xL = nullptr, *xU = nullptr, *Dp = nullptr;
    int32_t *mapLU = nullptr;
};

namespace {

__device__ inline double dpp_up(double v, double lane0)
{
    // lane l receives lane l-1's v (wave_shr:1 crosses the 16-lane DPP rows on gfx9); lane 0 receives lane0
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_update_dpp(__double2loint(lane0), lo, 0x138, 0xf, 0xf, false);
    const int shi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(shi, slo);
}
__device__ inline int lds_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void lds_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// LDS (dynamic), R = 2 * DEPTH: vring[NI][R][64] results of the last R steps per chain wave; inb[NI][R][64] upstream results;
// ering[NI+1][512] lane-63 results (ring 0 stays zero: the wave left of strip 0); scratch[NI][64+CH];
// ints: oc[NI+2] steps completed per chain wave (oc[0], oc[NI+1] = sentinels), os[NI] steps forwarded, ia[NI] upstream steps in inb, abort.
// The rings are two halves of DEPTH steps and the step loop is unrolled DEPTH times, so every ring access of a step has a
// static offset from one of two base registers that swap once per DEPTH steps.
// The chain waves only WRITE LDS: the forwarder wave stores their results to memory (the solution in position space,
// which is also what the next workgroup's fetcher reads).
// REG: every row has exactly the dependencies its grid position implies -- then an absent term's coefficient (+0.0)
// always meets an operand that is +0.0 (zeroed rings, padding rows), the product is +0.0, and no presence codes are read.
constexpr int kSlabMaxNI = 4;
template <int DEPTH, int CH, int HB, int ORDER, bool REG, int TPB>
__global__ __launch_bounds__(TPB) void k_trsv_slab(int32_t NI, int32_t NB, int32_t S, const f64x2s *__restrict__ rec,
                                                   const int32_t *__restrict__ code, double *xp, int32_t *progress, long long *clk,
                                                   const int *flag, int spin_limit, int32_t *sticky)
{
    constexpr int R = 2 * DEPTH;
    static_assert(DEPTH % CH == 0 && DEPTH % HB == 0 && HB >= 2 && HB + CH < R, "ring geometry");
    extern __shared__ double smem[];
    if (flag && *flag) return;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int32_t b = blockIdx.x;
    double *vring = smem;
    double *inb = vring + (size_t)NI * R * 64;
    double *ering = inb + (size_t)NI * R * 64;
    double *scratch = ering + (size_t)(NI + 1) * kEdgeRing;
    int *oc = reinterpret_cast<int *>(scratch + (size_t)NI * (64 + CH));
    int *os = oc + NI + 2;
    int *ia = os + NI;
    int *lds_abort = ia + NI;
    int32_t *abort_word = progress + (int64_t)NB * NI;
    const int nthreads = blockDim.x;
    for (int q = threadIdx.x; q < (NI + 1) * kEdgeRing; q += nthreads) ering[q] = 0.0;
    for (int q = threadIdx.x; q < 2 * NI * R * 64; q += nthreads) vring[q] = 0.0;         // (vring and inb are contiguous)
    if (threadIdx.x < NI + 2) oc[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == NI + 1) ? kBig : 0;
    if (threadIdx.x < NI) { os[threadIdx.x] = 0; ia[threadIdx.x] = b == 0 ? kBig : 0; }
    if (threadIdx.x == 0) *lds_abort = 0;
    __syncthreads();
    auto ld_relaxed = [](const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (wv < NI) {
        // ------------------------------------------------------------------ chain wave of strip a
        const int a = wv;
        const int64_t base = ((int64_t)b * NI + a) * S * 64;
        const f64x2s *Rb = rec + 2 * base;              // wave-uniform; the arrays are padded by DEPTH steps: no clamping
        const int32_t *Cb = code + base;
        double *myv = vring + (size_t)a * R * 64 + lane;
        const double *myin = inb + (size_t)a * R * 64 + lane;
        const double *lefte = ering + (size_t)a * kEdgeRing;           // ring of the wave to the left (ring 0: zeros)
        double *mye = ering + (size_t)(a + 1) * kEdgeRing;
        double *myscr = scratch + (size_t)a * (64 + CH) + lane;
        f64x2s ra[DEPTH], rb[DEPTH];
        int32_t rc[REG ? 1 : DEPTH];
        auto fetch = [&](int slot, int32_t t) {
            const f64x2s *q = Rb + (int64_t)t * 128;
            ra[slot] = q[2 * lane];
            rb[slot] = q[2 * lane + 1];
            if (!REG) rc[slot] = (Cb + (int64_t)t * 64)[lane];
        };
        // look-ahead: DEPTH steps, but never more than 60 loads in flight (vmcnt counts to 63; beyond that the compiler can
        // only wait for ALL of them, once per trip of the unrolled loop)
        constexpr int LA = (REG ? 2 : 3) * DEPTH > 60 ? 60 / (REG ? 2 : 3) : DEPTH;
#pragma unroll
        for (int j = 0; j < LA; ++j) fetch(j, j);
        const long long clk0 = wall_clock64();
        const int lmod = lane & (HB - 1);
        double prev = 0.0;
        double back = 0.0, inv = 0.0;            // operands of the current step, read out of LDS one step ahead
        double eE[CH];
        double *ow = myscr;
        long long *trace = clk + 2 * (int64_t)NB * NI + ((int64_t)b * NI + a) * (S / 16);      // diagnostics: a clock every 64 steps (slots of 16)
        for (int32_t t0 = 0; t0 < S; t0 += DEPTH) {
            const int half = (t0 / DEPTH) & 1;
            double *cur = myv + (size_t)half * DEPTH * 64, *prv = myv + (size_t)(half ^ 1) * DEPTH * 64;
            const double *cin = myin + (size_t)half * DEPTH * 64, *pin = myin + (size_t)(half ^ 1) * DEPTH * 64;
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int32_t t = t0 + j;
                if (j == 0 && (t0 & 63) == 0 && lane == 0) trace[t / 16] = wall_clock64();
                if (j % CH == 0) {
                    // this chunk needs: the left wave 63 + CH steps ahead, the upstream results HB + CH steps ahead,
                    // the result ring's slots forwarded, the edge ring's slots consumed by the wave to the right
                    int spins = 0;
                    for (;;) {
                        const int c0 = ld_relaxed(&oc[a]);        // one LDS round trip for the four
                        const int c1 = ld_relaxed(&ia[a]);
                        const int c2 = ld_relaxed(&os[a]);
                        const int c3 = lds_ld(&oc[a + 2]);        // (acquire on LDS = s_waitcnt lgkmcnt(0): orders the ring reads below after all four)
                        if ((c0 >= t + CH + 63) & (c1 >= t + CH + HB) & (c2 >= t + CH - R) & (c3 >= t + CH - kEdgeRing - 63)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > spin_limit || ld_relaxed(lds_abort)) {
                            if (lane == 0) {
                                { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (sticky) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                                __hip_atomic_store(lds_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            return;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < CH; ++u) eE[u] = lefte[(t + u + 63) & (kEdgeRing - 1)];
                    ow = lane == 63 ? &mye[t & (kEdgeRing - 1)] : myscr;
                    if (j == 0 && t0 == 0) {       // operands of step 0 (nothing is present there: the rings hold zeros)
                        back = prv[(size_t)(DEPTH - HB) * 64];
                        inv = cin[(size_t)(HB - 1) * 64];
                    }
                }
                // next step's LDS operands: the own result HB-1 steps back from here (written: HB >= 2) and the upstream entry t + HB
                constexpr int kNone = 0;
                (void)kNone;
                const int jb = j + 1 - HB, ji = j + HB;
                const double back_n = jb >= 0 ? cur[(size_t)jb * 64] : prv[(size_t)(DEPTH + jb) * 64];
                const double inv_n = ji < DEPTH ? cin[(size_t)ji * 64] : pin[(size_t)(ji - DEPTH) * 64];
                const double left = dpp_up(prev, eE[j % CH]);
                const bool first_line = lmod == (j & (HB - 1));         // (t - lane) % HB == 0: t0 is a multiple of HB
                const double up = first_line ? inv : prev;
                const uint32_t cc = REG ? 7u : (uint32_t)rc[REG ? 0 : j];
                const double pB = REG || (cc & 1u) ? ra[j].x * back : 0.0;
                const double pU = REG || (cc & 2u) ? ra[j].y * up : 0.0;
                const double pL = REG || (cc & 4u) ? rb[j].x * left : 0.0;
                double z = rb[j].y;
                if (ORDER == 0) { z = z - pB; z = z - pU; z = z - pL; }
                else if (ORDER == 1) { z = z - pL; z = z - pU; z = z - pB; }
                else {
                    const uint32_t s1 = (cc >> 3) & 3u, s2 = (cc >> 5) & 3u, s3 = (cc >> 7) & 3u;
                    z = z - (s1 == 0 ? pB : s1 == 1 ? pU : s1 == 2 ? pL : 0.0);
                    z = z - (s2 == 0 ? pB : s2 == 1 ? pU : s2 == 2 ? pL : 0.0);
                    z = z - (s3 == 0 ? pB : s3 == 1 ? pU : s3 == 2 ? pL : 0.0);
                }
                cur[(size_t)j * 64] = z;
                ow[j % CH] = z;                         // lane 63: the edge ring; the other lanes: scratch
                prev = z;
                back = back_n;
                inv = inv_n;
                fetch((j + LA) % DEPTH, t + LA);
                if (j % CH == CH - 1 && lane == 0) lds_st(&oc[a + 1], t + 1);
            }
        }
        if (lane == 0) {
            lds_st(&oc[a + 1], S + 2 * kEdgeRing);       // lets the wave to the right run out its last chunks
            clk[2 * ((int64_t)b * NI + a)] = clk0;
            clk[2 * ((int64_t)b * NI + a) + 1] = wall_clock64();
        }
        return;
    }
    int spins = 0;
    if (wv == NI) {
        // ------------------------------------------------------------------ forwarder: result rings -> memory
        int32_t sent[kSlabMaxNI];
        long long *htrace = clk + (int64_t)NB * NI * (2 + S / 16) + (int64_t)b * 512;       // diagnostics: (clock, steps forwarded) per pass
        int npass = 0;
#pragma unroll
        for (int a = 0; a < kSlabMaxNI; ++a) sent[a] = 0;
        for (;;) {
            int32_t made[kSlabMaxNI];
            bool all = true, any = false;
#pragma unroll
            for (int a = 0; a < kSlabMaxNI; ++a) made[a] = a < NI ? min(lds_ld(&oc[a + 1]), S) : 0;
#pragma unroll
            for (int a = 0; a < kSlabMaxNI; ++a) {
                if (a >= NI) continue;
                if (sent[a] < S) all = false;
                if (made[a] <= sent[a]) continue;
                any = true;
                const double *src = vring + (size_t)a * R * 64 + lane;
                double *dst = xp + (((int64_t)b * NI + a) * S) * 64 + lane;
                for (int32_t q = sent[a]; q < made[a]; q += 8) {          // 8 entries per round trip through LDS
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(min(q + u, made[a] - 1) & (R - 1)) * 64];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (q + u < made[a]) __hip_atomic_store(dst + (int64_t)(q + u) * 64, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (all) return;
            if (any) {
                // (no wait for the stores to be acknowledged, no flag: a result is kEmpty in memory until it has landed)
                if (lane == 0 && npass < 128) { htrace[2 * npass] = wall_clock64(); htrace[2 * npass + 1] = made[0]; ++npass; }
#pragma unroll
                for (int a = 0; a < kSlabMaxNI; ++a) {
                    if (a >= NI || made[a] <= sent[a]) continue;
                    if (lane == 0) {
                        __hip_atomic_store(progress + (int64_t)b * NI + a, made[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (diagnostics only)
                        lds_st(&os[a], made[a]);
                    }
                    sent[a] = made[a];
                }
                spins = 0;
                continue;
            }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_limit || ld_relaxed(lds_abort)) {
                if (lane == 0) { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (sticky) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                return;
            }
        }
    }
    // ---------------------------------------------------------------------- fetcher: upstream results -> inb
    if (b == 0) return;
    long long *htrace = clk + (int64_t)NB * NI * (2 + S / 16) + (int64_t)b * 512 + 256;
    // One memory round trip per pass: load the next kBatch entries of every strip (all loads in flight together) and keep
    // the leading ones whose 64 values have all landed (kEmpty = not yet; the gather / hand-over kernels reset the arrays).
    // (Tried: laying consecutive groups onto ONE XCD and reading the upstream results through its L2 with sc0 loads --
    // the producer's sc1 write-through stores do not refresh the L2 line, the reader sees stale data for ~100 us; and a
    // whole sweep's record traffic through one XCD's fabric port halves the chain rate.  Memory (sc1) it is.)
    {
        constexpr int kScope = __HIP_MEMORY_SCOPE_AGENT;
        constexpr int NIM = TPB == 256 ? 2 : kSlabMaxNI;       // strips per workgroup of this variant
        constexpr int kBatch = 16;                             // entries per strip per request (32: slower -- the pass itself gets longer)
        constexpr bool DBL = false;                            // (a second request in flight per strip: slower -- 614 vs 530 us at 100^3 -- the speculative loads come back empty and are asked again)
        int32_t got[NIM], next[NIM];                           // entries in inb / first entry not yet requested
        int32_t baseA[NIM], uptoA[NIM], baseB[NIM], uptoB[NIM];
        double vA[NIM][kBatch], vB[DBL ? NIM : 1][kBatch];
        int npass = 0;
#pragma unroll
        for (int a = 0; a < NIM; ++a) { got[a] = 0; next[a] = 0; baseA[a] = baseB[a] = -1; uptoA[a] = uptoB[a] = 0; }
        auto issue = [&](auto &v, int32_t (&base)[NIM], int32_t (&upto)[NIM]) {
#pragma unroll
            for (int a = 0; a < NIM; ++a) {
                base[a] = -1;
                if (a >= NI || next[a] >= S) continue;
                // entry q overwrites q - R, consumed once the chain has completed step q - R - HB + 1; stay clear of it
                const int32_t room = min(ld_relaxed(&oc[a + 1]), S) + R - 1;
                const int32_t up = min(min(S, room), next[a] + kBatch);
                if (up <= next[a]) continue;
                base[a] = next[a];
                upto[a] = up;
                const double *src = xp + (((int64_t)(b - 1) * NI + a) * S) * 64 + lane;
#pragma unroll
                for (int u = 0; u < kBatch; ++u)
                    v[a][u] = __hip_atomic_load(src + (int64_t)min(base[a] + u, up - 1) * 64, __ATOMIC_RELAXED, kScope);
                next[a] = up;
            }
        };
        auto consume = [&](auto &v, int32_t (&base)[NIM], int32_t (&upto)[NIM]) {
            bool any = false;
#pragma unroll
            for (int a = 0; a < NIM; ++a) {
                if (a >= NI || base[a] < 0) continue;
                if (base[a] != got[a]) { next[a] = got[a]; continue; }       // behind a request that landed only partly: dropped
                double *dst = inb + (size_t)a * R * 64 + lane;
                int32_t nvalid = 0;
                bool open = true;
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const bool landed = __ballot((unsigned long long)__double_as_longlong(v[a][u]) != kEmpty) == ~0ull;
                    open = open && landed && base[a] + u < upto[a];
                    if (open) { dst[(size_t)((base[a] + u) & (R - 1)) * 64] = v[a][u]; ++nvalid; }
                }
                if (base[a] + nvalid < upto[a]) next[a] = got[a] + nvalid;    // ask again from the first entry that had not landed
                if (!nvalid) continue;
                got[a] += nvalid;
                if (lane == 0) lds_st(&ia[a], got[a] >= S ? kBig : got[a]);
                if (a == 0 && lane == 0 && npass < 128) { htrace[2 * npass] = wall_clock64(); htrace[2 * npass + 1] = got[0]; ++npass; }
                any = true;
            }
            return any;
        };
        auto done = [&]() {
            bool all = true;
#pragma unroll
            for (int a = 0; a < NIM; ++a) if (a < NI && got[a] < S) all = false;
            return all;
        };
        auto stalled = [&](bool any) {
            if (any) { spins = 0; return false; }
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spin_limit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || ld_relaxed(lds_abort)) {
                if (lane == 0) {
                    { __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (sticky) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                    __hip_atomic_store(lds_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                return true;
            }
            return false;
        };
        issue(vA, baseA, uptoA);
        for (;;) {
            if constexpr (DBL) issue(vB, baseB, uptoB);
            bool any = consume(vA, baseA, uptoA);
            if (done()) return;
            issue(vA, baseA, uptoA);
            if constexpr (DBL) {
                any = consume(vB, baseB, uptoB) || any;
                if (done()) return;
            }
            if (stalled(any)) return;
        }
    }
}

// position-space gather / hand-over / scatter (padding positions hold 0); rhs = second half of a record's second pair
// (both also clear the progress words of the sweep that follows)
// and mark the sweep's result array "not yet written"
__global__ void k_slab_gather(int64_t np, f64x2s *__restrict__ rec, const double *__restrict__ src, const int32_t *__restrict__ row,
                              int32_t *__restrict__ progress, int32_t nprog, unsigned long long *__restrict__ xres, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = p; q < nprog; q += stride) progress[q] = 0;
    for (; p < np; p += stride) { const int32_t r = row[p]; rec[2 * p + 1].y = r >= 0 ? src[r] : 0.0; xres[p] = kEmpty; }
}
__global__ void k_slab_transition(int64_t np, f64x2s *__restrict__ recU, const double *__restrict__ xL, const int32_t *__restrict__ mapLU,
                                  const double *__restrict__ Dp, int32_t *__restrict__ progress, int32_t nprog,
                                  unsigned long long *__restrict__ xres, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = p; q < nprog; q += stride) progress[q] = 0;
    for (; p < np; p += stride) { const int32_t q = mapLU[p]; recU[2 * p + 1].y = q >= 0 ? xL[q] / Dp[p] : 0.0; xres[p] = kEmpty; }   // x = x / D
}
__global__ void k_slab_scatter(int64_t np, double *__restrict__ dst, const double *__restrict__ xp, const int32_t *__restrict__ row, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < np; p += stride) { const int32_t r = row[p]; if (r >= 0) dst[r] = xp[p]; }
}

void free_tri(SlabTri &G)
{
    dfree(G.rec); dfree(G.code); dfree(G.row); dfree(G.progress); dfree(G.clk);
    for (auto &v : G.src) dfree(v);
    dfree(G.pos);
    G = SlabTri();
}

// Is the factor a 3-D grid's?  lower: deps of row r within {r-1, r-w, r-w*h}, r-1 never across a line, r-w never
// across a plane; upper: the mirror image.  Sets w, h (0 = no).
void slab_dims(int32_t n, const std::vector<int32_t> &ptr1, const std::vector<int32_t> &node1, bool lower, int32_t &w, int32_t &h)
{
    w = h = 0;
    int32_t d1 = 0, d2 = 0;              // the two distances > 1, ascending
    for (int32_t r = 0; r < n; ++r) {
        if (ptr1[r + 1] - ptr1[r] > 3) return;
        for (int32_t k = ptr1[r] - 1; k < ptr1[r + 1] - 1; ++k) {
            const int32_t dlt = lower ? r - (node1[k] - 1) : (node1[k] - 1) - r;
            if (dlt <= 0) return;
            if (dlt == 1 || dlt == d1 || dlt == d2) continue;
            if (!d1) d1 = dlt;
            else if (!d2) { d2 = dlt; if (d2 < d1) std::swap(d1, d2); }
            else return;
        }
    }
    if (d1 < 2 || !d2 || d2 % d1) return;
    const int32_t ww = d1, hh = d2 / d1;
    if (hh < 2) return;
    for (int32_t r = 0; r < n; ++r) {
        int32_t seen[3] = {0, 0, 0};
        for (int32_t k = ptr1[r] - 1; k < ptr1[r + 1] - 1; ++k) {
            const int32_t c = node1[k] - 1;
            const int32_t dlt = lower ? r - c : c - r;
            const int32_t hi = lower ? r : c;                         // the later of the two rows
            const int id = dlt == 1 ? 2 : dlt == ww ? 1 : 0;
            if (seen[id]++) return;                                   // duplicate entries: the general walkers
            if (dlt == 1 && hi % ww == 0) return;
            if (dlt == ww && (hi / ww) % hh == 0) return;
        }
    }
    w = ww; h = hh;
}

int choose_hb(int32_t nk, int32_t h, int NI, int R)
{
    int best = 0;
    double best_cost = 0;
    for (int hb = 2; hb <= 16; hb *= 2) {
        if (hb + 8 >= R) continue;
        const int nb = (h + hb - 1) / hb;
        const double cost = (double)nk * hb + 63.0 + 64.0 * (NI - 1) + (double)(nb - 1) * (hb + 56.0);     // a hop between workgroups costs ~7 us, a step ~0.11
        if (!best || cost < best_cost) { best = hb; best_cost = cost; }
    }
    return best;
}

// index work of the slab layout, one lane per row (device pattern, 0-based): position of the row, its entries' places in
// the factor's val array per source (0 back r-wh, 1 up r-w, 2 left r-1), presence bits and the three 2-bit source ids in
// stored order; flags: [0] some row's order is back-up-left, [1] left-up-back, [2] anything else, [3] some row lacks a
// dependency its grid position implies (or has one it does not).
__global__ void k_slab_build(int32_t n, int32_t w, int32_t h, int32_t nk, int32_t NI, int32_t HB, int32_t S, int lower,
                             const int32_t *__restrict__ ptr, const int32_t *__restrict__ node, int32_t *__restrict__ row,
                             int32_t *__restrict__ s0, int32_t *__restrict__ s1, int32_t *__restrict__ s2, int32_t *__restrict__ code,
                             int32_t *__restrict__ pos, int32_t *flags)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int64_t wh = (int64_t)w * h;
    int32_t i = r % w, j = (int32_t)((r / w) % h), k = (int32_t)(r / wh);
    if (!lower) { i = w - 1 - i; j = h - 1 - j; k = nk - 1 - k; }
    const int32_t a = i / 64, l = i % 64, b = j / HB, jl = j % HB;
    const int64_t t = (int64_t)k * HB + jl + l;
    const int64_t p = (((int64_t)b * NI + a) * S + t) * 64 + l;
    pos[r] = (int32_t)p;
    row[p] = r;
    int32_t c = 0, ids[3] = {3, 3, 3};
    int cnt = 0;
    for (int32_t e = ptr[r]; e < ptr[r + 1]; ++e, ++cnt) {
        const int32_t dlt = lower ? r - node[e] : node[e] - r;
        const int id = dlt == 1 ? 2 : dlt == w ? 1 : 0;
        c |= 1 << id;
        if (id == 0) s0[p] = e; else if (id == 1) s1[p] = e; else s2[p] = e;
        if (cnt < 3) ids[cnt] = id;
    }
    if ((c & 7) != ((k > 0 ? 1 : 0) | (j > 0 ? 2 : 0) | (i > 0 ? 4 : 0))) flags[3] = 1;
    c |= ids[0] << 3 | ids[1] << 5 | ids[2] << 7;
    code[p] = c;
    if (cnt >= 2) {
        bool asc = true, desc = true;
        for (int q = 1; q < cnt && q < 3; ++q) { if (ids[q] < ids[q - 1]) asc = false; if (ids[q] > ids[q - 1]) desc = false; }
        flags[asc ? 0 : desc ? 1 : 2] = 1;
    }
}
__global__ void k_slab_map(int32_t n, const int32_t *__restrict__ posU, const int32_t *__restrict__ posL, int32_t *__restrict__ map)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) map[posU[r]] = posL[r];
}

// (dptr / dnode: the factor's pattern on the device, 0-based; slab_dims has checked it on the host copy)
int build_tri(SlabTri &G, int32_t n, int32_t w, int32_t h, const int32_t *dptr, const int32_t *dnode, bool lower)
{
    free_tri(G);
    const int64_t wh = (int64_t)w * h;
    G.w = w; G.h = h;
    G.nk = (int32_t)((n + wh - 1) / wh);
    G.NI = (w + 63) / 64;
    G.HB = choose_hb(G.nk, h, G.NI, 32);
    G.NB = (h + G.HB - 1) / G.HB;
    G.S = (int32_t)(((int64_t)G.nk * G.HB + 63 + 31) / 32 * 32);      // a multiple of every look-ahead depth
    G.NP = (int64_t)G.NB * G.NI * G.S * 64;
    if (G.NP >= INT32_MAX) return SGM_OK;                     // (positions are int32)
    hipStream_t st = g_rt.stream;
    int32_t *flags = nullptr;
    SGM_TRY(dalloc(&G.rec, ((size_t)G.NP + kPadPos) * 2));            // + one look-ahead of padding: the chain never clamps
    SGM_TRY(dalloc(&G.code, (size_t)G.NP + kPadPos));
    SGM_TRY(dalloc(&G.row, (size_t)G.NP));
    SGM_TRY(dalloc(&G.progress, (size_t)G.NB * G.NI + 1));
    SGM_TRY(dalloc(&G.clk, ((size_t)G.NB * G.NI * (2 + G.S / 16) + (size_t)G.NB * 512)));
    SGM_TRY(dalloc(&G.pos, (size_t)std::max(n, 1)));
    for (int id = 0; id < 3; ++id) {
        SGM_TRY(dalloc(&G.src[id], (size_t)G.NP));
        SGM_HIP(hipMemsetAsync(G.src[id], 0xff, (size_t)G.NP * 4, st));      // -1 = no such term
    }
    SGM_TRY(dalloc(&flags, 4));
    SGM_HIP(hipMemsetAsync(G.rec, 0, ((size_t)G.NP + kPadPos) * 16, st));
    SGM_HIP(hipMemsetAsync(G.code, 0, ((size_t)G.NP + kPadPos) * 4, st));
    SGM_HIP(hipMemsetAsync(G.row, 0xff, (size_t)G.NP * 4, st));             // -1 = padding
    SGM_HIP(hipMemsetAsync(G.clk, 0, ((size_t)G.NB * G.NI * (2 + G.S / 16) + (size_t)G.NB * 512) * 8, st));
    SGM_HIP(hipMemsetAsync(flags, 0, 16, st));
    if (n) hipLaunchKernelGGL(k_slab_build, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, w, h, G.nk, G.NI, G.HB, G.S, lower ? 1 : 0,
                              dptr, dnode, G.row, G.src[0], G.src[1], G.src[2], G.code, G.pos, flags);
    int32_t hf[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(hf, flags, 16, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    dfree(flags);
    SGM_HIP(e);
    G.regular = n == wh * G.nk && !hf[3];
    G.order = hf[2] || (hf[0] && hf[1]) ? 2 : hf[1] ? 1 : 0;
    G.on = true;
    return SGM_OK;
}

// records (every setup) from the factor's values on the device: {c_back, c_up}, {c_left, rhs = 0} per position
__global__ void k_slab_records(int64_t np, const int32_t *__restrict__ s0, const int32_t *__restrict__ s1, const int32_t *__restrict__ s2,
                               const double *__restrict__ val, f64x2s *__restrict__ rec)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < np; p += stride) {
        f64x2s a, b;
        a.x = s0[p] >= 0 ? val[s0[p]] : 0.0;
        a.y = s1[p] >= 0 ? val[s1[p]] : 0.0;
        b.x = s2[p] >= 0 ? val[s2[p]] : 0.0;
        b.y = 0.0;
        rec[2 * p] = a;
        rec[2 * p + 1] = b;
    }
}
__global__ void k_slab_diag(int64_t np, const int32_t *__restrict__ row, const double *__restrict__ D, double *__restrict__ Dp)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < np; p += stride) Dp[p] = row[p] >= 0 ? D[row[p]] : 1.0;
}
int refresh_tri(SlabTri &G, const double *val)
{
    hipLaunchKernelGGL(k_slab_records, dim3(vec_grid(G.NP)), dim3(kBlock), 0, g_rt.stream, G.NP, (const int32_t *)G.src[0],
                       (const int32_t *)G.src[1], (const int32_t *)G.src[2], val, G.rec);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

size_t slab_lds(int NI, int R, int CH)
{
    return ((size_t)2 * NI * R * 64 + (size_t)(NI + 1) * kEdgeRing + (size_t)NI * (64 + CH)) * 8 + (size_t)(3 * NI + 3 + 1) * 4 + 16;
}

template <int DEPTH, int HB, int ORDER, bool REG, int TPB>
void launch_slab(const SlabTri &G, double *xp, const int *flag, int spin, int32_t *sticky)
{
    constexpr int CH = 8;
    constexpr size_t pad = (size_t)96 * 1024;
    // at least 96 KiB per workgroup: ONE workgroup per CU, so that chain waves of two workgroups never share a SIMD
    const size_t lds = std::max(slab_lds(G.NI, 2 * DEPTH, CH), pad);
    static size_t attr = 0;
    if (attr < lds) {
        (void)hipFuncSetAttribute((const void *)k_trsv_slab<DEPTH, CH, HB, ORDER, REG, TPB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((k_trsv_slab<DEPTH, CH, HB, ORDER, REG, TPB>), dim3(G.NB), dim3(64 * (G.NI + 2)), lds, g_rt.stream, G.NI, G.NB, G.S,
                       (const f64x2s *)G.rec, (const int32_t *)G.code, xp, G.progress, G.clk, flag, spin, sticky);
}

// look-ahead 32 (and 64-step rings) for one or two strips per workgroup -- one wave per SIMD, 512 registers each --
// when no presence codes are read (two loads per step: 64 in flight is what vmcnt can count); 16 otherwise
template <int HB>
void launch_slab_o(const SlabTri &G, double *xp, const int *flag, int spin, int32_t *sticky)
{
    const bool deep = G.NI <= 2 && G.regular && G.order != 2;
    if (G.order == 0) {
        if (deep) launch_slab<32, HB, 0, true, 256>(G, xp, flag, spin, sticky);
        else if (G.regular) launch_slab<16, HB, 0, true, 384>(G, xp, flag, spin, sticky);
        else launch_slab<16, HB, 0, false, 384>(G, xp, flag, spin, sticky);
    } else if (G.order == 1) {
        if (deep) launch_slab<32, HB, 1, true, 256>(G, xp, flag, spin, sticky);
        else if (G.regular) launch_slab<16, HB, 1, true, 384>(G, xp, flag, spin, sticky);
        else launch_slab<16, HB, 1, false, 384>(G, xp, flag, spin, sticky);
    } else launch_slab<16, HB, 2, false, 384>(G, xp, flag, spin, sticky);
}

void trsv_slab(const SlabTri &G, double *xp, const int *flag, int spin, int32_t *sticky)
{
    switch (G.HB) {
    case 2: launch_slab_o<2>(G, xp, flag, spin, sticky); break;
    case 4: launch_slab_o<4>(G, xp, flag, spin, sticky); break;
    case 8: launch_slab_o<8>(G, xp, flag, spin, sticky); break;
    default: launch_slab_o<16>(G, xp, flag, spin, sticky); break;
    }
}

}  // namespace

void slab3_free(Slab3 *S)
{
    if (!S) return;
    free_tri(S->L); free_tri(S->U);
    dfree(S->xL); dfree(S->xU); dfree(S->Dp); dfree(S->mapLU);
    delete S;
}

// slab_dims on the device (no host copy of the pattern).  Pass 1: info[0] / info[1] = smallest / largest dependency distance
// > 1 (a 3-D grid's factor has exactly two: w and w*h), info[2] = a row breaks the shape (more than three entries, a
// dependency on the wrong side).  Pass 2, with w and h: info[3] = a distance that is none of 1, w, w*h, the same distance
// twice in a row, an r-1 dependency across a line, an r-w one across a plane.
__global__ void k_slab_detect1(int32_t n, int lower, const int32_t *__restrict__ ptr, const int32_t *__restrict__ node, int32_t *info)
{
    __shared__ int32_t lo, hi, bad;
    if (threadIdx.x == 0) { lo = INT32_MAX; hi = 0; bad = 0; }
    __syncthreads();
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const int32_t b = ptr[r], cnt = ptr[r + 1] - b;
        if (cnt > 3) bad = 1;
        for (int32_t k = b; k < b + cnt; ++k) {
            const int32_t dlt = lower ? r - node[k] : node[k] - r;
            if (dlt <= 0) bad = 1;
            else if (dlt > 1) { atomicMin(&lo, dlt); atomicMax(&hi, dlt); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (lo != INT32_MAX) { atomicMin(info, lo); atomicMax(info + 1, hi); }
        if (bad) info[2] = 1;
    }
}
__global__ void k_slab_detect2(int32_t n, int lower, int32_t ww, int32_t hh, const int32_t *__restrict__ ptr,
                               const int32_t *__restrict__ node, int32_t *info)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int seen[3] = {0, 0, 0};
    const int32_t wh = ww * hh;
    for (int32_t k = ptr[r]; k < ptr[r + 1]; ++k) {
        const int32_t c = node[k];
        const int32_t dlt = lower ? r - c : c - r;
        const int32_t hi = lower ? r : c;                             // the later of the two rows
        if (dlt != 1 && dlt != ww && dlt != wh) { info[3] = 1; return; }
        const int id = dlt == 1 ? 2 : dlt == ww ? 1 : 0;
        if (seen[id]++) info[3] = 1;                                  // duplicate entries: the general walkers
        if (dlt == 1 && hi % ww == 0) info[3] = 1;
        if (dlt == ww && (hi / ww) % hh == 0) info[3] = 1;
    }
}
// w = h = 0: not a 3-D grid's factor
int slab_dims_device(int32_t n, const int32_t *dptr, const int32_t *dnode, bool lower, int32_t *w, int32_t *h)
{
    *w = *h = 0;
    if (n < 1) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *info = nullptr;
    SGM_TRY(dalloc(&info, 4));
    struct Tmp { int32_t *&a; ~Tmp() { dfree(a); } } guard{info};
    const int32_t init[4] = {INT32_MAX, 0, 0, 0};
    int32_t hv[4];
    SGM_HIP(hipMemcpyAsync(info, init, 16, hipMemcpyHostToDevice, st));
    const int grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_slab_detect1, dim3(grid), dim3(kBlock), 0, st, n, lower ? 1 : 0, dptr, dnode, info);
    SGM_HIP(hipMemcpyAsync(hv, info, 16, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    const int32_t d1 = hv[0], d2 = hv[1];
    if (hv[2] || d1 == INT32_MAX || d1 < 2 || d2 <= d1 || d2 % d1 || d2 / d1 < 2) return SGM_OK;
    hipLaunchKernelGGL(k_slab_detect2, dim3(grid), dim3(kBlock), 0, st, n, lower ? 1 : 0, d1, d2 / d1, dptr, dnode, info);
    SGM_HIP(hipMemcpyAsync(hv, info, 16, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    if (!hv[3]) { *w = d1; *h = d2 / d1; }
    return SGM_OK;
}

// *out stays null when the factors are not a 3-D grid's (or too small / too wide for this path).  Host copies of the
// patterns (1-based) for the detection, device copies (0-based) for the index work.
int slab3_build(Slab3 **out, int32_t n, const std::vector<int32_t> &Lptr, const std::vector<int32_t> &Lnode,
                const std::vector<int32_t> &Uptr, const std::vector<int32_t> &Unode, const int32_t *dLptr, const int32_t *dLnode,
                const int32_t *dUptr, const int32_t *dUnode)
{
    *out = nullptr;
    int32_t wl, hl, wu, hu;
    if (Lptr.empty()) {                  // (no host copy of the pattern: the same questions asked on the device)
        SGM_TRY(slab_dims_device(n, dLptr, dLnode, true, &wl, &hl));
        if (!wl) return SGM_OK;
        SGM_TRY(slab_dims_device(n, dUptr, dUnode, false, &wu, &hu));
    } else {
        slab_dims(n, Lptr, Lnode, true, wl, hl);
        if (!wl) return SGM_OK;
        slab_dims(n, Uptr, Unode, false, wu, hu);
    }
    if (wl != wu || hl != hu) return SGM_OK;
    const int64_t wh = (int64_t)wl * hl;
    if (wl < 32 || wl > 256 || hl < 8 || (n + wh - 1) / wh < 8) return SGM_OK;
    Slab3 *S = new Slab3;
    S->n = n;
    int rc = build_tri(S->L, n, wl, hl, dLptr, dLnode, true);
    if (rc == SGM_OK) rc = build_tri(S->U, n, wl, hl, dUptr, dUnode, false);
    if (rc != SGM_OK || !S->L.on || !S->U.on) { slab3_free(S); return rc; }
    rc = dalloc(&S->xL, (size_t)S->L.NP);
    if (rc == SGM_OK) rc = dalloc(&S->xU, (size_t)S->U.NP);
    if (rc == SGM_OK) rc = dalloc(&S->Dp, (size_t)S->U.NP);
    if (rc == SGM_OK) rc = dalloc(&S->mapLU, (size_t)S->U.NP);
    if (rc != SGM_OK) { slab3_free(S); return rc; }
    hipStream_t st = g_rt.stream;
    hipError_t e = hipMemsetAsync(S->mapLU, 0xff, (size_t)S->U.NP * 4, st);
    if (e == hipSuccess && n)
        hipLaunchKernelGGL(k_slab_map, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)S->U.pos,
                           (const int32_t *)S->L.pos, S->mapLU);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    dfree(S->L.pos); dfree(S->U.pos);
    S->L.pos = S->U.pos = nullptr;
    if (e != hipSuccess) {
        slab3_free(S);
        return fail(SGM_ERR_HIP, "slab3_build: index work failed");
    }
    *out = S;
    return SGM_OK;
}

// (Lval, Uval, D: the factors on the device)
int slab3_refresh(Slab3 *S, const double *Lval, const double *Uval, const double *D)
{
    SGM_TRY(refresh_tri(S->L, Lval));
    SGM_TRY(refresh_tri(S->U, Uval));
    hipLaunchKernelGGL(k_slab_diag, dim3(vec_grid(S->U.NP)), dim3(kBlock), 0, g_rt.stream, S->U.NP, (const int32_t *)S->U.row, D, S->Dp);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// z = (I+U)^-1 D^-1 (I+L)^-1 r
// (spin_limit: polls before a wait gives up; sticky: the preconditioner's abort word, set -- never cleared -- by a sweep that did)
void slab3_apply(const Slab3 *S, const double *r, double *z, const int *flag, int spin_limit, int32_t *sticky)
{
    hipStream_t st = g_rt.stream;
    const int gl = vec_grid(S->L.NP), gu = vec_grid(S->U.NP);
    hipLaunchKernelGGL(k_slab_gather, dim3(gl), dim3(kBlock), 0, st, S->L.NP, S->L.rec, r, (const int32_t *)S->L.row, S->L.progress,
                       S->L.NB * S->L.NI + 1, reinterpret_cast<unsigned long long *>(S->xL), flag);
    trsv_slab(S->L, S->xL, flag, spin_limit, sticky);
    hipLaunchKernelGGL(k_slab_transition, dim3(gu), dim3(kBlock), 0, st, S->U.NP, S->U.rec, (const double *)S->xL,
                       (const int32_t *)S->mapLU, (const double *)S->Dp, S->U.progress, S->U.NB * S->U.NI + 1,
                       reinterpret_cast<unsigned long long *>(S->xU), flag);
    trsv_slab(S->U, S->xU, flag, spin_limit, sticky);
    hipLaunchKernelGGL(k_slab_scatter, dim3(gu), dim3(kBlock), 0, st, S->U.NP, z, (const double *)S->xU, (const int32_t *)S->U.row, flag);
}

// abort words of the last sweeps (stream must be idle)
void slab3_dims(const Slab3 *S, int32_t *w, int32_t *h) { *w = S->L.w; *h = S->L.h; }

// the result of the last L sweep in row order (setup self-check)
void slab3_lower_result(const Slab3 *S, double *dst)
{
    hipLaunchKernelGGL(k_slab_scatter, dim3(vec_grid(S->L.NP)), dim3(kBlock), 0, g_rt.stream, S->L.NP, dst, (const double *)S->xL,
                       (const int32_t *)S->L.row, (const int *)nullptr);
}

int slab3_aborted(const Slab3 *S, int32_t *abL, int32_t *abU)
{
    SGM_HIP(hipMemcpy(abL, S->L.progress + (int64_t)S->L.NB * S->L.NI, 4, hipMemcpyDeviceToHost));
    SGM_HIP(hipMemcpy(abU, S->U.progress + (int64_t)S->U.NB * S->U.NI, 4, hipMemcpyDeviceToHost));
    return SGM_OK;
}

// {strips, line groups, lines per group, steps, order of L, order of U}
void slab3_info(const Slab3 *S, int32_t out[6])
{
    out[0] = S->L.NI; out[1] = S->L.NB; out[2] = S->L.HB; out[3] = S->L.S;
    out[4] = S->L.order + (S->L.regular && S->L.order != 2 ? 10 : 0); out[5] = S->U.order + (S->U.regular && S->U.order != 2 ? 10 : 0);       // +10: no presence codes read
}

// chain start / end clocks (100 MHz) of the L sweep, 2 per (group, strip), then S / 16 samples per (group, strip)
int slab3_clocks(const Slab3 *S, std::vector<long long> &out)
{
    out.assign(((size_t)S->L.NB * S->L.NI * (2 + S->L.S / 16) + (size_t)S->L.NB * 512), 0);
    SGM_HIP(hipMemcpy(out.data(), S->L.clk, out.size() * 8, hipMemcpyDeviceToHost));
    return SGM_OK;
}

}  // namespace sgm
