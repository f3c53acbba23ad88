// Column-blocked two-phase product for ELLPACK matrices whose columns have no locality
// (BASELINE config C4: 32-regular random digraph, n = 5e6).
//
// Why: k_ell_spmv on C4 is bound by the rate at which random 128-byte lines leave the Infinity
// Cache, not by bytes (profiles/r02/pmc_c4_k_ell_spmv.json: 156 M fabric read requests for 160 M
// gathers, L2 hit rate 8.6 % -- x is 40 MB, an XCD's L2 4 MB --, 2.72 ms = 59 G lines/s against
// the 67 G lines/s this chip delivers for random lines out of the Infinity Cache).  The row sum's
// ORDER is fixed by parity, but only the additions are ordered: the products val(k,i)*x(node(k,i))
// can be formed in any order.  So:
//
//   phase 1  k_ellcb_mul   entries sorted once by (column block, row, slot); a workgroup loads ONE
//                          block of x (16384 columns = 128 KiB) into LDS and streams the block's
//                          entries -- value 8 B + column-inside-block 2 B, coalesced 16-byte loads --,
//                          gathers from LDS and writes the products P in sorted order (8 B, coalesced)
//   phase 2  k_ellcb_sum   a workgroup owns a tile of R rows; the tile's products are nb short runs
//                          of P (one per column block), copied into an LDS image with coalesced
//                          loads; lane i then adds row i's products IN SLOT ORDER (2-byte LDS
//                          position per entry), exactly the reference's rounding sequence
//
// Every random access is an LDS access; HBM sees streams only: 18 B + 10 B per entry (+ run
// tables, x blocks) instead of 12 B per entry plus a 128-byte line per gather.  Results are
// bit-identical to k_ell_spmv (products rounded individually, added left to right; a padding
// slot's 0 * x(last) term is kept).  Built at sgm_ell_create when the matrix looks random
// (option "ell_colblock": 0 never, 1 automatic, 2 always).
//
// (Round 3 also ran the two phases band by band over one product buffer sized for the Infinity Cache -- measured slower,
// profiles/r03/c4_row_bands_kernel_stats.txt, and removed in round 4; the products make their round trip through HBM.)
#include "sgm_internal.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>

namespace sgm {

typedef double f64x2c __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------ setup kernels
// Where the entries come from: an ELLPACK matrix (every one of the max_d slots of a row is an entry: a padding slot keeps its
// 0 * x(last neighbour) term, like ellpack_matvec_add) or a CSR matrix with scattered columns and rows of <= 128 entries (slot k
// of row i = its k-th stored entry; slots beyond the row's length do not exist: they take no part in either phase, so the row
// sum is csr_matvec_add's, cs_matrices.f90:611-620, term for term).
struct EllSrc {
    const int32_t *ecol; const double *eval; int32_t n, max_d;
    __device__ bool has(int32_t, int32_t) const { return true; }
    __device__ int32_t col(int32_t i, int32_t k) const { return ecol[(int64_t)k * n + i]; }
    __device__ double val(int32_t i, int32_t k) const { return eval[(int64_t)k * n + i]; }
};
struct CsrSrc {
    const int32_t *rowptr, *ccol; const double *cval; int32_t n, max_d;
    __device__ bool has(int32_t i, int32_t k) const { return k < rowptr[i + 1] - rowptr[i]; }
    __device__ int32_t col(int32_t i, int32_t k) const { return ccol[rowptr[i] + k]; }
    __device__ double val(int32_t i, int32_t k) const { return cval[rowptr[i] + k]; }
};
// key of entry e = i*max_d + k (row-major: a stable sort by key leaves (row, slot) order inside a block)
// key = column block; a slot that holds no entry gets key nb: it sorts behind every block and nobody looks at it again
template <class Src>
__global__ void k_ellcb_keys(Src S, int32_t cb, int32_t nb, uint16_t *__restrict__ key, int32_t *__restrict__ ent)
{
    const int64_t total = (int64_t)S.n * S.max_d;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int32_t i = (int32_t)(e / S.max_d), k = (int32_t)(e % S.max_d);
        key[e] = S.has(i, k) ? (uint16_t)(S.col(i, k) / cb) : (uint16_t)nb;
        ent[e] = (int32_t)e;
    }
}
// bstart[b] = first sorted position whose key is >= b   (b = 0..nb)
__global__ void k_ellcb_bstart(int64_t total, int32_t nb, const uint16_t *__restrict__ skey, int32_t *__restrict__ bstart)
{
    const int32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > nb) return;
    int64_t lo = 0, hi = total;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int32_t)skey[mid] < b) lo = mid + 1; else hi = mid;
    }
    bstart[b] = (int32_t)lo;
}
// values (and, with lcol != null, block-local columns) in sorted order; `count` = the sorted positions that hold entries
template <class Src>
__global__ void k_ellcb_gather(int64_t count, Src S, int32_t cb, const int32_t *__restrict__ perm,
                               double *__restrict__ sval, uint16_t *__restrict__ lcol)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
        const int32_t e = perm[j], i = e / S.max_d, k = e % S.max_d;
        sval[j] = S.val(i, k);
        if (lcol) lcol[j] = (uint16_t)(S.col(i, k) % cb);
    }
}
// run (t, b): the entries of column block b whose rows lie in tile t = sorted positions
// [start, start + len); rows ascend inside a block, so both ends are binary searches.
// Descriptor = {start, len | base << 16}: base = where the run sits in the tile's LDS image.
__global__ void k_ellcb_runs(int32_t ntiles, int32_t nb, int32_t R, int32_t max_d, const int32_t *__restrict__ bstart,
                             const int32_t *__restrict__ perm, int2 *__restrict__ fdesc)
{
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (int64_t)ntiles * nb) return;
    const int32_t t = (int32_t)(id / nb), b = (int32_t)(id % nb);
    const int32_t kb = b;
    auto first_row_at_least = [&](int64_t row) {
        int32_t lo = bstart[kb], hi = bstart[kb + 1];
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if ((int64_t)(perm[mid] / max_d) < row) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int32_t a = first_row_at_least((int64_t)t * R), z = first_row_at_least((int64_t)(t + 1) * R);
    fdesc[id] = make_int2(a, z - a);       // len <= R * max_d <= 16384; the base is filled in by k_ellcb_bases
}
// LDS base of every run of a tile (exclusive prefix sum over the column blocks)
__global__ void k_ellcb_bases(int32_t ntiles, int32_t nb, int2 *__restrict__ fdesc)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    uint32_t acc = 0;
    for (int32_t b = 0; b < nb; ++b) {
        int2 d = fdesc[(int64_t)t * nb + b];
        const uint32_t len = (uint32_t)d.y;
        d.y = (int32_t)(len | (acc << 16));
        fdesc[(int64_t)t * nb + b] = d;
        acc += len;
    }
}
// lpos (slot-major like eval): where entry (k, i) sits in its tile's LDS image
template <class Src>
__global__ void k_ellcb_lpos(int64_t count, Src S, int32_t cb, int32_t nb, int32_t R,
                             const int32_t *__restrict__ perm, const int2 *__restrict__ fdesc, uint16_t *__restrict__ lpos)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += stride) {
        const int32_t e = perm[j], i = e / S.max_d, k = e % S.max_d;
        const int2 d = fdesc[(int64_t)(i / R) * nb + S.col(i, k) / cb];
        lpos[(int64_t)k * S.n + i] = (uint16_t)(((uint32_t)d.y >> 16) + (uint32_t)(j - d.x));
    }
}
// mean |column - row| over a sample of rows (is the matrix "random"?); cnt = the entries looked at
template <class Src>
__global__ void k_ellcb_sample(Src S, int32_t step, unsigned long long *sum, unsigned long long *cnt)
{
    const int32_t r = (blockIdx.x * blockDim.x + threadIdx.x) * step;
    if (r >= S.n) return;
    unsigned long long s = 0, c = 0;
    for (int32_t k = 0; k < S.max_d; ++k) {
        if (!S.has(r, k)) break;
        const int32_t cc = S.col(r, k);
        s += (unsigned long long)(cc > r ? cc - r : r - cc);
        ++c;
    }
    atomicAdd(sum, s);
    atomicAdd(cnt, c);
}

// ------------------------------------------------------------------------------ phase 1
// Workgroup (column block b, chunk c of it): the x block goes to LDS, then the chunk's entries are streamed (value 8 B +
// column-inside-block 2 B, U x 16-byte loads per lane in flight), multiplied and their products stored (nontemporal:
// nobody reads them before the next launch).  grid = nb * chunks: every workgroup loads exactly one x block.
template <int TPB, int U>
__global__ __launch_bounds__(TPB) void k_ellcb_mul(int32_t ncol, int32_t cb, int32_t nb, int32_t chunks, const int32_t *__restrict__ bstart,
                                                   const double *__restrict__ sval, const uint16_t *__restrict__ lcol,
                                                   const double *__restrict__ x, double *__restrict__ P,
                                                   const int *__restrict__ flag_done, int gen)
{
    extern __shared__ double xs[];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    const int32_t lo = blockIdx.x / chunks, c = blockIdx.x % chunks;
    const int64_t j0b = bstart[lo], j1b = bstart[lo + 1], len = j1b - j0b;
    int64_t j0 = j0b + (len * c / chunks), j1 = j0b + (len * (c + 1) / chunks);
    if (c > 0) j0 = (j0 + 1) & ~(int64_t)1;
    if (c + 1 < chunks) j1 = (j1 + 1) & ~(int64_t)1;
    if (j1 > j1b) j1 = j1b;
    if (j0 >= j1) return;
    for (int32_t b = lo; j0 < j1 && b < nb; ++b) {
        const int64_t jz = min(j1, (int64_t)bstart[b + 1]);
        if (jz <= j0) continue;
        const int32_t col0 = b * cb, cnt = min(cb, ncol - col0);
        __syncthreads();                     // the previous block's gathers are done with xs
        {   // x block -> LDS (col0 is even: cb is; 16-byte loads, odd tail alone)
            const f64x2c *src = reinterpret_cast<const f64x2c *>(x + col0);
            f64x2c *dst = reinterpret_cast<f64x2c *>(xs);
            for (int32_t t = threadIdx.x; t < (cnt >> 1); t += TPB) dst[t] = src[t];
            if ((cnt & 1) && threadIdx.x == 0) xs[cnt - 1] = x[col0 + cnt - 1];
        }
        __syncthreads();
        int64_t ja = j0;
        if (ja & 1) {                        // odd head: one entry alone
            if (threadIdx.x == 0) P[ja] = sval[ja] * xs[lcol[ja]];
            ++ja;
        }
        const int64_t npair = (jz - ja) >> 1;
        const f64x2c *v2 = reinterpret_cast<const f64x2c *>(sval + ja);
        const uint32_t *c2 = reinterpret_cast<const uint32_t *>(lcol + ja);
        f64x2c *p2 = reinterpret_cast<f64x2c *>(P + ja);
        int64_t q = threadIdx.x;
        for (; q + (int64_t)(U - 1) * TPB < npair; q += (int64_t)U * TPB) {
            f64x2c v[U];
            uint32_t cc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = __builtin_nontemporal_load(v2 + q + (int64_t)u * TPB);
                cc[u] = __builtin_nontemporal_load(c2 + q + (int64_t)u * TPB);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                f64x2c o;
                o.x = v[u].x * xs[cc[u] & 0xffffu];
                o.y = v[u].y * xs[cc[u] >> 16];
                __builtin_nontemporal_store(o, p2 + q + (int64_t)u * TPB);
            }
        }
        for (; q < npair; q += TPB) {
            const f64x2c v = v2[q];
            const uint32_t cc = c2[q];
            f64x2c o;
            o.x = v.x * xs[cc & 0xffffu];
            o.y = v.y * xs[cc >> 16];
            p2[q] = o;
        }
        if (((jz - ja) & 1) && threadIdx.x == 0) P[jz - 1] = sval[jz - 1] * xs[lcol[jz - 1]];
        j0 = jz;
    }
}

// ------------------------------------------------------------------------------ phase 2
// A workgroup walks tiles t = blockIdx.x, + gridDim.x, ...
// Copy of the tile's nb runs into the LDS image: every wave takes a contiguous share of the runs,
// fetches their descriptors 64 at a time with ONE coalesced load (a lane per run) and then moves two
// runs per wave instruction (a half-wave each; runs average 27 entries), eight instructions in flight.
// The 2-byte image positions of the lane's own row are requested before the copy, so their latency
// hides behind it.
// R rows per tile, TPB = CM * R threads: all TPB / 64 waves copy runs, the first R threads own the rows.
// FULLW: one run per wave instruction (64 lanes; tiles of 512 rows, runs average 54 entries) instead of two half-wave runs.
template <int R, int TPB, int MAXD, bool ADD, bool DOT_W, bool DOT_YY, bool FULLW = false>
__global__ __launch_bounds__(TPB) void k_ellcb_sum(int32_t n, int32_t max_d, int32_t nb, int32_t ntiles,
                                                   const int2 *__restrict__ fdesc, const uint16_t *__restrict__ lpos,
                                                   const double *__restrict__ P, double *__restrict__ y,
                                                   const double *__restrict__ w, double *__restrict__ part_wy,
                                                   double *__restrict__ part_yy, const int *__restrict__ flag_done, int gen,
                                                   int chain, const int32_t *__restrict__ rowptr /* CSR origin: row lengths; null = all max_d slots */)
{
    extern __shared__ double img[];
    __shared__ double red[TPB / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    constexpr int NW = TPB / 64;
    const int lane = threadIdx.x & 63, half = FULLW ? 0 : lane >> 5, q = FULLW ? lane : lane & 31;
    constexpr int RW = FULLW ? 64 : 32;          // lanes per run
    constexpr int RPI = FULLW ? 1 : 2;           // runs per wave instruction
    const int wave = threadIdx.x >> 6;
    const int32_t per = (nb + NW - 1) / NW, bw0 = wave * per, bw1 = min(nb, bw0 + per);
    double dwy = 0.0, dyy = 0.0;
    for (int32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int32_t i = t * R + (int32_t)threadIdx.x;
        const bool live = (int)threadIdx.x < R && i < n;
        const int32_t deg = !live ? 0 : rowptr ? rowptr[i + 1] - rowptr[i] : max_d;       // slots of this row that hold entries
        // positions of this lane's row (slots 0..MAXD-1 in registers; longer rows re-read them later)
        uint16_t pz[MAXD];
#pragma unroll
        for (int k = 0; k < MAXD; ++k)
            pz[k] = k < deg ? __builtin_nontemporal_load(lpos + (int64_t)k * n + i) : (uint16_t)0;
        const int2 *D = fdesc + (int64_t)t * nb;
        for (int32_t b0 = bw0; b0 < bw1; b0 += 64) {
            const int32_t cnt = min(64, bw1 - b0);
            int2 d = make_int2(0, 0);
            if (lane < cnt) d = D[b0 + lane];
            for (int32_t f0 = 0; f0 < cnt; f0 += 8 * RPI) {     // 8 instructions x RPI runs in flight
                double v[8];
                int32_t o[8], l[8], g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int src = f0 + RPI * u + half;        // (runs beyond cnt carry len 0)
                    g[u] = __shfl(d.x, src, 64);
                    const uint32_t lb = (uint32_t)__shfl(d.y, src, 64);
                    l[u] = (int32_t)(lb & 0xffffu);
                    o[u] = (int32_t)(lb >> 16);
                    v[u] = q < l[u] ? __builtin_nontemporal_load(P + g[u] + q) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (q < l[u]) img[o[u] + q] = v[u];
                    for (int32_t r = q + RW; r < l[u]; r += RW) img[o[u] + r] = P[g[u] + r];    // runs longer than RW lanes
                }
            }
        }
        __syncthreads();
        if (live) {
            double y0 = 0.0;
            if (ADD) y0 = y[i];
            double z = (ADD && chain) ? y0 : 0.0;
#pragma unroll
            for (int k = 0; k < MAXD; ++k)
                if (k < deg) z = z + img[pz[k]];
            for (int32_t k = MAXD; k < deg; ++k) z = z + img[lpos[(int64_t)k * n + i]];
            const double yi = ADD ? (chain ? z : y0 + z) : 0.0 + z;
            y[i] = yi;
            if (DOT_W) dwy += w[i] * yi;
            if (DOT_YY) dyy += yi * yi;
        }
        __syncthreads();
    }
    if (DOT_W) {
        const double s = block_sum<TPB>(dwy, red);
        if (threadIdx.x == 0) part_wy[blockIdx.x] = s;
    }
    if (DOT_YY) {
        const double s = block_sum<TPB>(dyy, red);
        if (threadIdx.x == 0) part_yy[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------ host side
// matrix options "ell_colblock_cols" / "ell_colblock_rows"; tuning aids read from the environment when a form is built:
// workgroups per column block in the multiply phase: 16 (C4 sweep 4 / 8 / 16: 1.32 / 1.31 / 1.27 ms); phase-2 grid cap: 2048
static int cb_grid2() { return 2048; }

void free_ell_colblock(Part &p)
{
    dfree(p.cb_perm); dfree(p.cb_sval); dfree(p.cb_lcol); dfree(p.cb_bstart); dfree(p.cb_lpos);
    dfree(p.cb_fdesc); dfree(p.cb_P);
    p.cb_perm = nullptr; p.cb_sval = nullptr; p.cb_lcol = nullptr; p.cb_bstart = nullptr; p.cb_lpos = nullptr;
    p.cb_fdesc = nullptr; p.cb_P = nullptr;
    p.cb_cols = p.cb_nb = p.cb_R = p.cb_ntiles = p.cb_chunks = 0;
    p.cb_count = 0;
}

bool use_ell_colblock(const Part &p) { return p.cb_P != nullptr && p.opt.ell_colblock != 0; }
static bool cb_from_csr(const Part &p) { return !p.ecol; }
static EllSrc ell_src(const Part &p) { return EllSrc{p.ecol, p.eval, p.n, p.cb_maxd}; }
static CsrSrc csr_src(const Part &p) { return CsrSrc{p.rowptr, p.col, p.val, p.n, p.cb_maxd}; }

// workgroups of the sum launch = partial sums the fused dots leave
int ell_colblock_grid(const Part &p) { return std::max(1, std::min(p.cb_ntiles, cb_grid2())); }

// does the matrix qualify, and do its columns look random?
static int wants_colblock(Part &p, bool *yes)
{
    *yes = false;
    const bool csr = cb_from_csr(p);
    p.cb_maxd = csr ? p.max_row : p.max_d;
    if (!p.opt.ell_colblock || p.n <= 0 || p.cb_maxd < 1 || p.n_halo != 0) return SGM_OK;
    if (csr && (!p.rowptr || !p.col || !p.val || p.lean)) return SGM_OK;
    if ((int64_t)p.n * p.cb_maxd >= INT32_MAX || p.cb_maxd > 128) return SGM_OK;
    // a CSR matrix: rows of similar length only (the 2-byte positions and the sort keys are per SLOT: n * max_row of them)
    const double slots_per_entry = (double)p.n * p.cb_maxd / (double)std::max<int64_t>(p.nnz, 1);
    if (csr && slots_per_entry > (p.opt.ell_colblock >= 2 ? 16.0 : 2.0)) return SGM_OK;
    if (p.opt.ell_colblock >= 2) { *yes = true; return SGM_OK; }
    if (p.ecode || p.scode || p.code || p.sbcode) return SGM_OK;  // structured: the dictionary kernels serve it
    // x within reach of the L2s / too few gathers.  (7 MiB: with uniformly random columns the two-phase form overtakes the row
    // kernels between x = 4 and 8 MB -- 16 / 32 per row at 7.6 MB: 1.06 / 1.25 x, 11.4 MB: 1.39 / 1.64, 15.3 MB: 1.55 / 1.84;
    // tools/probes/colblock_threshold.py.  The limit was 16 MiB until that sweep.)
    if ((int64_t)p.ncol_own * 8 < (int64_t)7 << 20 || p.cb_maxd < 8) return SGM_OK;
    unsigned long long *dsum = nullptr, hsum[2] = {0, 0};
    SGM_TRY(dalloc(&dsum, 2));
    hipStream_t st = g_rt.stream;
    SGM_HIP(hipMemsetAsync(dsum, 0, 16, st));
    const int32_t step = std::max(1, p.n / 4096), rows = (p.n + step - 1) / step;
    if (csr) hipLaunchKernelGGL((k_ellcb_sample<CsrSrc>), dim3((rows + 255) / 256), dim3(256), 0, st, csr_src(p), step, dsum, dsum + 1);
    else hipLaunchKernelGGL((k_ellcb_sample<EllSrc>), dim3((rows + 255) / 256), dim3(256), 0, st, ell_src(p), step, dsum, dsum + 1);
    SGM_HIP(hipMemcpyAsync(hsum, dsum, 16, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    dfree(dsum);
    const double mean = (double)hsum[0] / (double)std::max<unsigned long long>(hsum[1], 1);
    p.col_spread = mean;
    *yes = mean > (double)p.ncol_own / 16.0 && (!csr || (double)hsum[1] >= 8.0 * rows);
    return SGM_OK;
}

// values in sorted order (at creation and after every value update)
int refresh_ell_colblock_values(Part &p)
{
    if (!p.cb_P) return SGM_OK;
    const int64_t count = p.cb_count;
    if (cb_from_csr(p)) {
        SGM_TRY(csr_need_arrays(p));
        hipLaunchKernelGGL((k_ellcb_gather<CsrSrc>), dim3(vec_grid(count)), dim3(kBlock), 0, g_rt.stream, count, csr_src(p), p.cb_cols,
                           (const int32_t *)p.cb_perm, p.cb_sval, (uint16_t *)nullptr);
    } else
        hipLaunchKernelGGL((k_ellcb_gather<EllSrc>), dim3(vec_grid(count)), dim3(kBlock), 0, g_rt.stream, count, ell_src(p), p.cb_cols,
                           (const int32_t *)p.cb_perm, p.cb_sval, (uint16_t *)nullptr);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// index work of the column-blocked form (once per structure; the matrix's options at that moment are what it is built with)
int build_ell_colblock(Part &p)
{
    free_ell_colblock(p);
    bool yes = false;
    SGM_TRY(wants_colblock(p, &yes));
    if (trace_on() && (p.opt.ell_colblock >= 2 || yes))
        fprintf(stderr, "[sigma_hip] column-blocked form (%s, n = %d, slots per row %d, option %d): %s\n", cb_from_csr(p) ? "csr" : "ellpack", p.n,
                p.cb_maxd, p.opt.ell_colblock, yes ? "building" : "declined");
    if (!yes) return SGM_OK;
    const bool csr = cb_from_csr(p);
    hipStream_t st = g_rt.stream;
    const int32_t md = p.cb_maxd;
    const int64_t total = (int64_t)p.n * md;
    const int32_t cb = p.opt.ell_colblock_cols, nb = (p.ncol_own + cb - 1) / cb;
    if (nb > 65534) return SGM_OK;
    // tile image <= 64 KiB (two workgroups per CU) -- or, option ell_colblock_rows = 512 (automatic for rows <= 32 slots:
    // runs twice as long, one 1024-thread workgroup per CU with a 128 KiB image) -- whole waves
    const int want_rows = p.opt.ell_colblock_rows;
    int32_t R = std::min(256, 8192 / md) / 64 * 64;
    if ((want_rows == 512 || (want_rows == 0 && md >= 16)) && md <= 32) R = 512;
    // (round 4: 20480-column blocks and a 152 KiB image of 608 rows -- runs 48 % longer -- measured SLOWER, 1222 / 1171 / 1321 us
    //  against 1134: profiles/r04/c4_cols_rows_sweep.jsonl; the run length is not what bounds the second phase)
    if (R < 64) return SGM_OK;
    const int32_t ntiles = (p.n + R - 1) / R;
    p.cb_cols = cb; p.cb_nb = nb; p.cb_R = R; p.cb_ntiles = ntiles; p.cb_chunks = 16;

    uint16_t *key = nullptr, *skey = nullptr;
    int32_t *ent = nullptr;
    void *tmp = nullptr;
    struct Scratch { uint16_t **a, **b; int32_t **c; void **d; ~Scratch() { dfree(*a); dfree(*b); dfree(*c); dfree(*d); } } guard{&key, &skey, &ent, &tmp};
    SGM_TRY(dalloc(&key, (size_t)total));
    SGM_TRY(dalloc(&skey, (size_t)total));
    SGM_TRY(dalloc(&ent, (size_t)total));
    SGM_TRY(dalloc(&p.cb_perm, (size_t)total + 2));
    if (csr) hipLaunchKernelGGL((k_ellcb_keys<CsrSrc>), dim3(vec_grid(total)), dim3(kBlock), 0, st, csr_src(p), cb, nb, key, ent);
    else hipLaunchKernelGGL((k_ellcb_keys<EllSrc>), dim3(vec_grid(total)), dim3(kBlock), 0, st, ell_src(p), cb, nb, key, ent);
    int bits = 1;
    while ((1 << bits) < nb + 1) ++bits;               // (keys 0 .. nb: nb = "no entry")
    size_t tmp_bytes = 0;
    SGM_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, skey, ent, p.cb_perm, (int)total, 0, bits, st));
    char *tmpc = nullptr;
    SGM_TRY(dalloc(&tmpc, tmp_bytes));
    tmp = tmpc;
    SGM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, skey, ent, p.cb_perm, (int)total, 0, bits, st));   // stable

    SGM_TRY(dalloc(&p.cb_bstart, (size_t)nb + 1));
    hipLaunchKernelGGL(k_ellcb_bstart, dim3((nb + 1 + 255) / 256), dim3(256), 0, st, total, nb, (const uint16_t *)skey, p.cb_bstart);
    int32_t count32 = 0;                                // sorted positions that hold entries (= total for an ELLPACK matrix)
    SGM_HIP(hipMemcpyAsync(&count32, p.cb_bstart + nb, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    const int64_t count = p.cb_count = count32;
    SGM_TRY(dalloc(&p.cb_sval, (size_t)count + 2));
    SGM_TRY(dalloc(&p.cb_lcol, (size_t)count + 2));
    if (csr) hipLaunchKernelGGL((k_ellcb_gather<CsrSrc>), dim3(vec_grid(count)), dim3(kBlock), 0, st, count, csr_src(p), cb,
                                (const int32_t *)p.cb_perm, p.cb_sval, p.cb_lcol);
    else hipLaunchKernelGGL((k_ellcb_gather<EllSrc>), dim3(vec_grid(count)), dim3(kBlock), 0, st, count, ell_src(p), cb,
                            (const int32_t *)p.cb_perm, p.cb_sval, p.cb_lcol);
    const int64_t nruns = (int64_t)ntiles * nb;
    int2 *fdesc = nullptr;
    SGM_TRY(dalloc(&fdesc, (size_t)nruns));
    p.cb_fdesc = fdesc;
    hipLaunchKernelGGL(k_ellcb_runs, dim3((unsigned)((nruns + 255) / 256)), dim3(256), 0, st, ntiles, nb, R, md,
                       (const int32_t *)p.cb_bstart, (const int32_t *)p.cb_perm, fdesc);
    hipLaunchKernelGGL(k_ellcb_bases, dim3((ntiles + 255) / 256), dim3(256), 0, st, ntiles, nb, fdesc);
    SGM_TRY(dalloc(&p.cb_lpos, (size_t)total + 2));
    if (csr) hipLaunchKernelGGL((k_ellcb_lpos<CsrSrc>), dim3(vec_grid(count)), dim3(kBlock), 0, st, count, csr_src(p), cb, nb, R,
                                (const int32_t *)p.cb_perm, (const int2 *)fdesc, p.cb_lpos);
    else hipLaunchKernelGGL((k_ellcb_lpos<EllSrc>), dim3(vec_grid(count)), dim3(kBlock), 0, st, count, ell_src(p), cb, nb, R,
                            (const int32_t *)p.cb_perm, (const int2 *)fdesc, p.cb_lpos);
    SGM_TRY(dalloc(&p.cb_P, (size_t)count + 2));
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));
    return SGM_OK;
}

template <int R, bool ADD>
static void launch_sum(const Part &p, int grid, double *y, const double *w, double *pwy, double *pyy, const int *flag, int gen, int chain)
{
    hipStream_t st = g_rt.stream;
    const size_t lds = (size_t)p.cb_R * p.cb_maxd * 8;
    constexpr bool BIG = R == 512;                   // 128 KiB image, 1024 threads, whole-wave runs
    constexpr int MAXD = (BIG ? 16384 : 8192) / R;   // the longest row a tile of R rows allows
    constexpr int TPB = BIG ? 1024 : 2 * R;          // twice as many waves copy runs as there are rows (C4: 696 -> see DESIGN)
#define L(DW, DY)                                                                                                     \
    do {                                                                                                              \
        static bool attr = false;                                                                                     \
        if (BIG && !attr) { (void)hipFuncSetAttribute((const void *)k_ellcb_sum<R, TPB, MAXD, ADD, DW, DY, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, R * MAXD * 8); attr = true; } \
        hipLaunchKernelGGL((k_ellcb_sum<R, TPB, MAXD, ADD, DW, DY, BIG>), dim3(grid), dim3(TPB), lds, st, p.n, p.cb_maxd, p.cb_nb, p.cb_ntiles, \
                           (const int2 *)p.cb_fdesc, (const uint16_t *)p.cb_lpos, (const double *)p.cb_P, y, w, pwy, pyy, flag, gen, chain, \
                           cb_from_csr(p) ? (const int32_t *)p.rowptr : (const int32_t *)nullptr); \
    } while (0)
    if (w && pyy) L(true, true);
    else if (w) L(true, false);
    else if (pyy) L(false, true);
    else L(false, false);
#undef L
}

int launch_ell_colblock(const Part &p, int grid, const double *x, double *y, bool add, bool chain, const double *w,
                        double *pwy, double *pyy, const int *flag, int gen)
{
    hipStream_t st = g_rt.stream;
    constexpr int TPB1 = 512, U1 = 4;      // (1024 threads x 4: 571 us on C4 against 559: the stream is not short of loads in flight)
    static bool attr_set = false;
    if (!attr_set) {       // more than 64 KiB of dynamic LDS needs the attribute (160 KiB per CU on gfx950)
        SGM_HIP(hipFuncSetAttribute((const void *)k_ellcb_mul<TPB1, U1>, hipFuncAttributeMaxDynamicSharedMemorySize, kEllcbMaxCols * 8));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_ellcb_mul<TPB1, U1>), dim3(p.cb_nb * p.cb_chunks), dim3(TPB1), (size_t)p.cb_cols * 8, st, p.ncol_own, p.cb_cols,
                       p.cb_nb, p.cb_chunks, (const int32_t *)p.cb_bstart, (const double *)p.cb_sval, (const uint16_t *)p.cb_lcol, x, p.cb_P,
                       flag, gen);
#define R_CASE(RR)                                                                              \
    if (p.cb_R == RR) {                                                                         \
        if (add) launch_sum<RR, true>(p, grid, y, w, pwy, pyy, flag, gen, chain ? 1 : 0);        \
        else launch_sum<RR, false>(p, grid, y, w, pwy, pyy, flag, gen, 0);                       \
    }
    R_CASE(64) R_CASE(128) R_CASE(192) R_CASE(256) R_CASE(512)
#undef R_CASE
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// bytes by construction (sgm_mat_footprint)
int64_t ell_colblock_resident_bytes(const Part &p)
{
    if (!p.cb_P) return 0;
    const int64_t total = (int64_t)p.n * p.cb_maxd, count = p.cb_count, nruns = (int64_t)p.cb_ntiles * p.cb_nb;
    return total * (4 + 2) + count * (8 + 2) + count * 8 + nruns * 8 + 4 * ((int64_t)p.cb_nb + 1);
}
int64_t ell_colblock_matvec_bytes(const Part &p)
{
    const int64_t count = p.cb_count, nruns = (int64_t)p.cb_ntiles * p.cb_nb;
    const int64_t xloads = (int64_t)p.cb_nb * p.cb_chunks;
    return count * (8 + 2 + 8) + xloads * p.cb_cols * 8                                  // phase 1: values, columns, products, x blocks
         + count * (8 + 2) + nruns * 8 + 8 * (int64_t)p.n + (cb_from_csr(p) ? 4 * (int64_t)p.n : 0);   // phase 2: products, positions, run tables, y (+ row lengths)
}

}  // namespace sgm
