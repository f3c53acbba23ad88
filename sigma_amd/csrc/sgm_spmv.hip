// Sparse matrix-vector products for gfx950: CSR (cs_matrices.f90:600-622) and ELLPACK
// (ellpack_matrices.f90:640-665), plus the matrix handles of the C ABI.
//
// Numerics contract (bit-identical to the reference loops): for every row the products
// val(k)*x(node(k)) are rounded individually (this file is compiled with
// -ffp-contract=off, so no v_fma_f64 is formed) and added to a scalar LEFT TO RIGHT in
// stored order; `matvec` returns 0.0 + z like `y = 0; y(i) = y(i) + z`
// (linear_operator_interface.f90:191-192).
//
// CSR kernels, picked per matrix at upload (sgm_mat_kernel names the one in use):
//   k_csr_sl    rows <= 8 entries from <= 15 (column - row) offsets: values re-laid slot-major in
//               512-row slices + one word of 4-bit dictionary codes per row; a lane owns two rows,
//               no LDS, no row pointers.                                     8 W + 4 B / row
//   k_csr_slb   rows of 9..32 entries, <= 255 offsets: the same with 1-byte codes.     9 B / slot
//   k_csr_sl32  rows <= 32 entries of similar length at arbitrary columns: the same with int32
//               columns.                                                              12 B / slot
//   k_csr_do    a 256-thread workgroup owns 256 consecutive rows; the contiguous val / code range of
//               those rows is streamed into LDS in tiles by ALL lanes, then the row's owner lane
//               gathers x and adds in stored order: 1-byte codes (other stencil-like matrices,
//               9 B / entry) or int32 columns (rows <= 64 entries, 12 B / entry)
//   k_csr_rl    int32 columns, longer rows: one 128-byte line of val per row and pass, all 256 rows
//               walked by their owner lanes at once.                                 12 B / entry
//   k_csr_spmv  int32 columns, any row length: the x gather happens while streaming (two entries
//               per lane), products are parked in LDS, the owner adds them.          12 B / entry
// The x gather is served by L2 (the workgroup -> row-block / slice maps keep neighbouring rows on one
// XCD).  Algorithmic bytes per SpMV on the reference layout: 12*nnz + 4*(n+1) + 8*m + 8*n.
#include "sgm_internal.hpp"

#include <type_traits>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

namespace sgm {

// ---------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------
__global__ void k_dec1(int32_t *a, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] -= 1;
}

// Validation of the index arrays a caller hands to sgm_csr_create (the reference prints and exits on a bad
// index only where it happens to look, sparse_matrix_interfaces.f90:663-687; a wild `node` here would be a
// GPU memory fault inside the product).  bad[0] = first row i (0-based) whose pointers are malformed --
// ptr(1) /= 1, ptr(i+1) < ptr(i), ptr(n+1) - 1 /= nnz (reported as row n) --, bad[1] = first entry k (0-based) whose
// 1-based column lies outside 1..ncols.  Both start at INT64_MAX; the create reads them at the synchronisation
// it makes anyway.  The pointer pass runs on the 1-based upload BEFORE k_dec1 (it reads a neighbour); the
// column pass is the decrement itself.
__global__ void k_check_ptr1(const int32_t *__restrict__ ptr1, int64_t n, int64_t nnz, unsigned long long *bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i <= n; i += stride) {
        const int32_t a = ptr1[i];
        bool ok = true;
        if (i == 0) ok = a == 1;
        if (i < n) ok = ok && ptr1[i + 1] >= a;
        else ok = ok && (int64_t)a - 1 == nnz;
        if (!ok) atomicMin(bad, (unsigned long long)i);
    }
}
__global__ void k_dec1_check_cols(int32_t *a, int64_t nnz, int64_t ncols, unsigned long long *bad)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) {
        const int32_t c = a[k];
        if (c < 1 || c > ncols) atomicMin(bad + 1, (unsigned long long)k);
        a[k] = c - 1;
    }
}

__global__ void k_ell_transpose(const int32_t *__restrict__ node, const double *__restrict__ val,
                                int32_t *__restrict__ ecol, double *__restrict__ eval,
                                int32_t n, int32_t max_d, int32_t ncol = 0, unsigned long long *bad = nullptr)
{
    // in: (max_d, n) column-major = row i contiguous; out: slot-major [k*n + i]
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n * max_d;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < total; t += stride) {
        const int32_t k = (int32_t)(t / n), i = (int32_t)(t % n);
        // an empty row keeps node = 0 in the reference (it then reads x(0): README.md:71-73);
        // clamp so that the padding product 0.0 * x(1) stays inside the vector
        if (node) {
            const int32_t c = node[(int64_t)i * max_d + k];
            // (validation of sgm_ell_create's input: 0 is the reference's empty-row marker, anything else must be a column)
            if (bad && (c < 0 || c > ncol)) atomicMin(bad, (unsigned long long)((int64_t)i * max_d + k));
            ecol[t] = max(c - 1, 0);
        }
        if (val) eval[t] = val[(int64_t)i * max_d + k];
    }
}

// Work-group -> row-block map.  Blocks b and b+8 share an XCD (round-robin dispatch), so
// the blocks of one XCD take CONSECUTIVE row blocks inside each sweep of the grid: the x
// entries they gather (own rows +- the stencil reach) stay in that XCD's 4 MiB L2.
// Bijective because the grid is a multiple of 8.  Placement only changes speed.
__device__ inline int64_t rowblock_of(int it, int b, int grid, int64_t nrb = 0, int mode = 1)
{
    const int per = grid >> 3;
    if (mode == 2) {        // XCD-major: every XCD walks its own contiguous eighth of the rows
        const int64_t chunk = (nrb + 7) / 8;
        const int64_t local = (int64_t)it * per + (b >> 3);
        return local < chunk ? (int64_t)(b & 7) * chunk + local : nrb;
    }
    return (int64_t)it * grid + (int64_t)(b & 7) * per + (b >> 3);
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ inline T ld_stream(const T *p, bool nt)
{
    return nt ? __builtin_nontemporal_load(p) : *p;
}

// BLOCK threads own BLOCK consecutive rows; TILE = 2*BLOCK*VPT products are staged per pass.
template <int BLOCK, int VPT, bool NT, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(BLOCK) void k_csr_spmv(
    int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ w, double *__restrict__ part_wy, double *__restrict__ part_yy,
    const int *__restrict__ flag_done, int gen, int remap)
{
    constexpr int TILE = 2 * BLOCK * VPT;
    __shared__ double prod[TILE];
    __shared__ double red[BLOCK / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }

    const int tid = threadIdx.x;
    const bool chain = (remap & 256) != 0;      // launch flag bits above the XCD-map mode
    const int rmode = remap & 255;
    const int64_t nrb = ((int64_t)n + BLOCK - 1) / BLOCK;
    double dwy = 0.0, dyy = 0.0;

    for (int it = 0;; ++it) {
        if ((int64_t)it * gridDim.x >= nrb) break;
        const int64_t rb = rmode ? rowblock_of(it, blockIdx.x, gridDim.x, nrb, rmode) : (int64_t)it * gridDim.x + blockIdx.x;
        if (rb >= nrb) continue;          // uniform per block
        const int32_t r0 = (int32_t)(rb * BLOCK);
        const int32_t r1 = min(r0 + BLOCK, n);
        const int32_t row = r0 + tid;
        int32_t k = 0, ke = 0;
        double wv = 0.0, y0 = 0.0;          // requested now, consumed after the row sum
        if (row < n) {
            k = rowptr[row];
            ke = rowptr[row + 1];
            if (DOT_W) wv = w[row];
            if (ADD) y0 = y[row];
        }
        const int32_t s = rowptr[r0] & ~1;    // tile starts are even: 16-B aligned val loads
        const int32_t e = rowptr[r1];
        double z = (ADD && chain) ? y0 : 0.0;   // chain: the row sum continues from y(i) (transpose products)

        for (int32_t ts = s; ts < e; ts += TILE) {
            const int32_t te = min(ts + TILE, e);
            // ---- phase 1: all lanes stream val/col and gather x (2 entries per lane)
            f64x2 v[VPT];
            i32x2 c[VPT];
#pragma unroll
            for (int m = 0; m < VPT; ++m) {
                const int32_t j = ts + 2 * tid + 2 * BLOCK * m;
                if (j < te) {     // arrays are padded by 2 entries: j+1 is always readable
                    v[m] = ld_stream(reinterpret_cast<const f64x2 *>(val + j), NT);
                    c[m] = ld_stream(reinterpret_cast<const i32x2 *>(col + j), NT);
                }
            }
#pragma unroll
            for (int m = 0; m < VPT; ++m) {
                const int32_t j = ts + 2 * tid + 2 * BLOCK * m;
                if (j < te) {
                    const double x0 = x[c[m].x], x1 = x[c[m].y];
                    f64x2 p;
                    p.x = v[m].x * x0;
                    p.y = v[m].y * x1;
                    *reinterpret_cast<f64x2 *>(prod + (j - ts)) = p;
                }
            }
            __syncthreads();
            // ---- phase 2: lane i adds row i's products left to right
            // (eight independent LDS reads, then the adds in stored order: a long row is a serial chain of adds, but it
            // need not be a serial chain of LDS round trips as well)
            const int32_t kend = min(ke, te);
            while (k < kend) {
                const int cnt = min(kend - k, 8);
                double pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < cnt) pv[u] = prod[k + u - ts];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (u < cnt) z = z + pv[u];
                k += cnt;
            }
            __syncthreads();
        }
        if (row < n) {
            const double yi = ADD ? (chain ? z : y0 + z) : 0.0 + z;
            if (NT) __builtin_nontemporal_store(yi, y + row); else y[row] = yi;
            if (DOT_W) dwy += wv * yi;
            if (DOT_YY) dyy += yi * yi;
        }
    }
    if (DOT_W) {
        const double t = block_sum<BLOCK>(dwy, red);
        if (tid == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<BLOCK>(dyy, red);
        if (tid == 0) part_yy[blockIdx.x] = t;
    }
}

// General CSR with LONG rows (no dictionary, rows beyond the row-owner kernel's 64 entries: finite-element matrices of
// higher order, 3-D unstructured meshes).  k_csr_spmv gathers entry-parallel: the 64 lanes of one gather instruction
// hold 64 consecutive ENTRIES -- of one or two rows, so 64 different x lines, each moved L2 -> L1 for 8 of its 128 bytes
// (measured on banded rows of 33..300 entries: 1.6-2.6 TB/s of moved bytes, the L2 -> L1 path carrying 16 x as much).
// k_csr_do lets the lane that owns a row gather for it, so one instruction holds the k-th entries of consecutive rows
// (neighbouring columns in any matrix with a banded / mesh-local numbering: a handful of lines) -- but a tile of T
// staged entries holds only T / len rows, and each of those few lanes walks len entries: time grows with the row length.
// Here a lane owns a row too, and ALL 256 rows of the block are walked at once; what makes that fit in LDS is the
// staging unit: not "every entry of the row block" but ONE 128-BYTE LINE of `val` (16 entries) per row and pass.  In
// pass c, row j's entries that lie in line (first line of row j) + c are staged -- eight lanes per row fetch the line's
// 16-byte pieces, only pieces that hold an entry of the row; `col` comes in whole 128-byte lines (32 entries) on every
// other pass, see `fetch` -- and lane j then walks its up-to-16 entries: columns out of LDS, eight x requests in flight, products rounded one by one and
// added in stored order (bit-identical to csr_matvec_add).  The loads of pass c + 1 are in flight while pass c is
// summed.  A block takes as many passes as its longest row has lines.  LDS: 256 rows x 17 (16 + 1 against bank
// conflicts) x 12 B = 52 KiB, three workgroups per CU.  Price: a line that two rows share is requested by both, in
// different passes, and L2 keeps none of the stream (PMC, rows of 33..64 entries, when `col` still came in 8-byte
// pieces beside `val`: 4.04 GB fetched for 2.36 GB needed; plain instead of nontemporal loads changed nothing) -- the
// longer the rows, the smaller that share.
template <bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(256, 3) void k_csr_rl(
    int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ w, double *__restrict__ part_wy, double *__restrict__ part_yy,
    const int *__restrict__ flag_done, int gen, int remap)
{
    constexpr int BLOCK = 256, ST = 17, RPI = BLOCK / 8, NI = BLOCK / RPI;
    __shared__ double vl[BLOCK * ST];
    __shared__ int32_t cl[BLOCK * ST];
    __shared__ int32_t rp[BLOCK + 1];
    __shared__ int wmax[BLOCK / 64];
    __shared__ double red[BLOCK / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    const int tid = threadIdx.x;
    const bool chain = (remap & 256) != 0;
    const int rmode = remap & 255;
    const int64_t nrb = ((int64_t)n + BLOCK - 1) / BLOCK;
    const int sj = tid >> 3, sp = 2 * (tid & 7);
    double dwy = 0.0, dyy = 0.0;
    for (int it = 0;; ++it) {
        if ((int64_t)it * gridDim.x >= nrb) break;
        const int64_t rb = rmode ? rowblock_of(it, blockIdx.x, gridDim.x, nrb, rmode) : (int64_t)it * gridDim.x + blockIdx.x;
        if (rb >= nrb) continue;
        const int32_t r0 = (int32_t)(rb * BLOCK);
        const int32_t row = r0 + tid;
        rp[tid] = rowptr[min(row, n)];
        if (tid == 0) rp[BLOCK] = rowptr[min(r0 + BLOCK, n)];
        double wv = 0.0, y0 = 0.0;
        if (row < n) {
            if (DOT_W) wv = w[row];
            if (ADD) y0 = y[row];
        }
        __syncthreads();
        const int32_t s = rp[tid], e = rp[tid + 1];
        int np = e > s ? ((e - 1) >> 4) - (s >> 4) + 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) np = max(np, __shfl_xor(np, off, 64));
        if ((tid & 63) == 0) wmax[tid >> 6] = np;
        __syncthreads();
        int npass = wmax[0];
#pragma unroll
        for (int t = 1; t < BLOCK / 64; ++t) npass = max(npass, wmax[t]);
        f64x2 v[NI];
        i32x4 cq[NI];
        // `col` comes in whole 128-byte lines (32 entries = two lines of val): the eight lanes of a row fetch the line's
        // 16-byte pieces when the pass starts an even val line (or the row); lanes 0-3 hold the columns of that pass,
        // lanes 4-7 keep theirs in registers for the next one
        auto fetch = [&](int c) {
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                const int32_t ss = rp[sj + RPI * m], se = rp[sj + RPI * m + 1];       // the row this lane stages for
                const int32_t vline = (ss >> 4) + c;
                const int32_t base = (vline << 4) + sp;
                if (base < se && base + 2 > ss)           // the pair holds an entry of the row (arrays are padded)
                    v[m] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + base));
                const int32_t base4 = ((vline >> 1) << 5) + 2 * sp;
                if ((c == 0 || !(vline & 1)) && base4 < se && base4 + 4 > ss)
                    cq[m] = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(col + base4));
            }
        };
        double z = (ADD && chain) ? y0 : 0.0;
        if (npass > 0) fetch(0);
        for (int c = 0; c < npass; ++c) {
#pragma unroll
            for (int m = 0; m < NI; ++m) {
                const int32_t ss = rp[sj + RPI * m], se = rp[sj + RPI * m + 1];
                const int32_t vline = (ss >> 4) + c;
                const int32_t base = (vline << 4) + sp;
                if (base < se && base + 2 > ss) {
                    const int o = (sj + RPI * m) * ST + sp;
                    vl[o] = v[m].x; vl[o + 1] = v[m].y;
                }
                const int32_t base4 = ((vline >> 1) << 5) + 2 * sp;
                if (((tid >> 2) & 1) == (vline & 1) && base4 < se && base4 + 4 > ss) {
                    const int o = (sj + RPI * m) * ST + 4 * (tid & 3);
                    cl[o] = cq[m].x; cl[o + 1] = cq[m].y; cl[o + 2] = cq[m].z; cl[o + 3] = cq[m].w;
                }
            }
            __syncthreads();
            if (c + 1 < npass) fetch(c + 1);
            const int32_t line0 = ((s >> 4) + c) << 4;
            const int qlo = max(s - line0, 0), qhi = min(e - line0, 16);
#pragma unroll
            for (int h = 0; h < 16; h += 8) {         // (all 16 requests at once: 866 -> 1123 us -- the empty half is skipped here)
                if (qhi > h && qlo < h + 8) {
                    double xv[8], vv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (h + u >= qlo && h + u < qhi) {
                            xv[u] = x[cl[tid * ST + h + u]];
                            vv[u] = vl[tid * ST + h + u];
                        }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (h + u >= qlo && h + u < qhi) z = z + vv[u] * xv[u];
                }
            }
            __syncthreads();
        }
        if (row < n) {
            const double yi = ADD ? (chain ? z : y0 + z) : 0.0 + z;
            __builtin_nontemporal_store(yi, y + row);
            if (DOT_W) dwy += wv * yi;
            if (DOT_YY) dyy += yi * yi;
        }
    }
    if (DOT_W) { const double t = block_sum<BLOCK>(dwy, red); if (tid == 0) part_wy[blockIdx.x] = t; }
    if (DOT_YY) { const double t = block_sum<BLOCK>(dyy, red); if (tid == 0) part_yy[blockIdx.x] = t; }
}

// CSR with dictionary-coded column offsets ("offset-dict" form).  Matrices from structured
// grids have very few distinct (column - row) offsets (5 for the 5-point, 7 for the 7-point
// stencil, also after the [owned | halo] renumbering of a slab partition), so the column of
// entry k is stored as a 1-byte code: col(k) = row + dict[code(k)].  HBM traffic drops from
// 12 to 9 bytes per stored entry.  Because the offset is relative to the ROW, the gather is
// done by the lane that owns the row: val and codes are streamed into LDS with wide
// coalesced loads by all lanes, then lane i walks row i left to right -- up to 8 entries'
// x values are requested at once, the adds stay in stored order (bit-identical results).
// For stencil rows lane l and lane l+1 gather neighbouring x entries: coalesced 512-B reads.
// CW = bytes per stored column: 1 = dictionary code (col = row + dict[code]); 4 = the int32
// column itself ("row-owner" form of the general kernel, used when rows are short: the lane
// that owns a row gathers for it, which keeps stencil-like gathers coalesced; 12 B / entry).
template <int BLOCK, int TILE, int CW, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(BLOCK) void k_csr_do(
    int32_t n, const int32_t *__restrict__ rowptr, const uint8_t *__restrict__ code,
    const int32_t *__restrict__ dict, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, const double *__restrict__ w, double *__restrict__ part_wy,
    double *__restrict__ part_yy, const int *__restrict__ flag_done, int gen, int remap)
{
    constexpr int VPT = (TILE + 2 * BLOCK - 1) / (2 * BLOCK);  // 16-byte val loads per lane per tile
    // x requests in flight per lane = the row length the tile was chosen for (TILE / BLOCK entries per
    // row): a longer unroll only costs registers, and the fused-dot variants must stay within 64 VGPRs
    // to keep 8 waves per SIMD like the plain kernel the persistent grid is sized for
    constexpr int U = TILE <= 4 * BLOCK ? 4 : TILE <= 6 * BLOCK ? 6 : 8;
    constexpr int CPT = (TILE / 4 + BLOCK - 1) / BLOCK;      // 4-byte code words per lane per tile
    static_assert(TILE % 4 == 0, "tiles are whole 4-byte code words");
    __shared__ double vl[TILE];
    __shared__ uint32_t cl4[TILE * CW / 4];
    __shared__ int32_t dl[CW == 1 ? 256 : 1];
    __shared__ double red[BLOCK / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    const uint8_t *cl = reinterpret_cast<const uint8_t *>(cl4);

    const int tid = threadIdx.x;
    const bool chain = (remap & 256) != 0;      // launch flag bits above the XCD-map mode
    const int rmode = remap & 255;
    if (CW == 1)
        for (int t = tid; t < 256; t += BLOCK) dl[t] = dict[t];
    const int32_t *col32 = reinterpret_cast<const int32_t *>(code);
    const int64_t nrb = ((int64_t)n + BLOCK - 1) / BLOCK;
    double dwy = 0.0, dyy = 0.0;

    for (int it = 0;; ++it) {
        if ((int64_t)it * gridDim.x >= nrb) break;
        const int64_t rb = rmode ? rowblock_of(it, blockIdx.x, gridDim.x, nrb, rmode) : (int64_t)it * gridDim.x + blockIdx.x;
        if (rb >= nrb) continue;
        const int32_t r0 = (int32_t)(rb * BLOCK);
        const int32_t r1 = min(r0 + BLOCK, n);
        const int32_t row = r0 + tid;
        int32_t k = 0, ke = 0;
        double y0 = 0.0;                    // requested now, consumed after the row sum
        if (row < n) {
            k = rowptr[row];
            ke = rowptr[row + 1];
            if (ADD) y0 = y[row];
        }
        const int32_t s = rowptr[r0] & ~3;    // tiles start at multiples of 4 entries: aligned 4-B code loads
        const int32_t e = rowptr[r1];
        double z = (ADD && chain) ? y0 : 0.0;   // chain: the row sum continues from y(i) (transpose products)

        // int32 columns on the largest tiles (long rows: several tiles per row block, phase 2 walks for microseconds): the
        // next tile's loads are requested before this tile is walked (24 more registers; the short-row shapes that must
        // stay within 64 VGPRs keep the loads where they were)
        constexpr bool PF = CW == 4 && TILE >= 2048;
        f64x2 v[VPT];
        uint32_t c4[CPT];
        i32x2 c8[VPT];
        auto fetch = [&](int32_t ts, int32_t te) {
#pragma unroll
            for (int m = 0; m < VPT; ++m) {
                const int32_t j = ts + 2 * tid + 2 * BLOCK * m;
                if (j < te) v[m] = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val + j));
            }
            if (CW == 1) {
#pragma unroll
                for (int m = 0; m < CPT; ++m) {
                    const int32_t q = tid + BLOCK * m;
                    if (q < TILE / 4 && ts + 4 * q < te)
                        c4[m] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(code + ts) + q);
                }
            } else {
#pragma unroll
                for (int m = 0; m < VPT; ++m) {
                    const int32_t j = ts + 2 * tid + 2 * BLOCK * m;
                    if (j < te) c8[m] = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(col32 + j));
                }
            }
        };
        if (PF && s < e) fetch(s, min(s + TILE, e));
        for (int32_t ts = s; ts < e; ts += TILE) {
            const int32_t te = min(ts + TILE, e);
            // ---- phase 1: stream val (16 B / lane) and codes (4 B / lane) into LDS
            if (!PF) fetch(ts, te);
            __syncthreads();       // the previous tile's phase 2 is done with the LDS buffers
#pragma unroll
            for (int m = 0; m < VPT; ++m) {
                const int32_t j = ts + 2 * tid + 2 * BLOCK * m;
                if (j < te) *reinterpret_cast<f64x2 *>(vl + (j - ts)) = v[m];
            }
            if (CW == 1) {
#pragma unroll
                for (int m = 0; m < CPT; ++m) {
                    const int32_t q = tid + BLOCK * m;
                    if (q < TILE / 4 && ts + 4 * q < te) cl4[q] = c4[m];
                }
            } else {
#pragma unroll
                for (int m = 0; m < VPT; ++m) {
                    const int32_t j = ts + 2 * tid + 2 * BLOCK * m;
                    if (j < te) *reinterpret_cast<i32x2 *>(cl4 + (j - ts)) = c8[m];
                }
            }
            __syncthreads();
            if (PF && ts + TILE < e) fetch(ts + TILE, min(ts + 2 * TILE, e));
            // ---- phase 2: lane i gathers for row i (8 requests in flight), adds in order
            const int32_t kend = min(ke, te);
            while (k < kend) {
                const int cnt = min(kend - k, U);
                double xv[U], vv[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < cnt) {
                        const int o = k + u - ts;
                        xv[u] = CW == 1 ? x[row + dl[cl[o]]] : x[(int32_t)cl4[o]];
                        vv[u] = vl[o];
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < cnt) z = z + vv[u] * xv[u];
                k += cnt;
            }
        }
        if (row < n) {
            const double yi = ADD ? (chain ? z : y0 + z) : 0.0 + z;
            __builtin_nontemporal_store(yi, y + row);
            // w(row) is read here, not with the row pointers: held across the row sum it costs the two
            // registers that push the fused variants over 64 VGPRs (7 instead of 8 waves per SIMD, and
            // a persistent grid sized for 8 then runs a second round); w is x or was just gathered: an L2 hit
            if (DOT_W) dwy += w[row] * yi;
            if (DOT_YY) dyy += yi * yi;
        }
    }
    if (DOT_W) {
        const double t = block_sum<BLOCK>(dwy, red);
        if (tid == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<BLOCK>(dyy, red);
        if (tid == 0) part_yy[blockIdx.x] = t;
    }
}

// Sliced form of the offset-dict kernel, for matrices whose rows hold <= 8 entries drawn from
// <= 15 distinct (column - row) offsets (1-D/2-D/3-D stencils).  At upload the values are re-laid
// in slices of 512 rows, slot-major inside a slice, and a row's column offsets become eight 4-bit
// dictionary codes in ONE 32-bit word (code 15 = no entry; a row's entries fill slots 0.. in
// stored order).  A lane owns the two adjacent rows 2t, 2t+1 of a slice and reads everything
// they need with independent, fully coalesced 16-byte loads (W value pairs, one pair of code
// words) plus 2 W gathers of x -- no LDS staging, no barrier, no row pointers; y leaves as one
// 16-byte store.  HBM bytes per row: 8 W + 4 (+ x, y) instead of 9 nnz_row + 4.  Row blocks go
// round-robin over the workgroups (it * grid + block), grid = min(slices, 4096): measured against
// the LDS-staged kernel (tools/probes/sl_ablate.cpp and bench): the 16-byte accesses and the plain map are
// worth 10-17 % each way of the comparison.  Products are rounded one by one and added in stored
// order per row, exactly like the other kernels (bit-identical results).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr int kSlRows = 512;       // rows per slice = 2 x workgroup size
template <int W, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(256) void k_csr_sl(
    int32_t n, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict,
    const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ w, double *__restrict__ part_wy, double *__restrict__ part_yy,
    const int *__restrict__ flag_done, int gen, int remap, const int32_t *__restrict__ sched, int sched_iters)
{
    constexpr int BLOCK = 256;
    __shared__ int32_t dl[16];
    __shared__ double red[BLOCK / 64];
    // The stop flag, the offset dictionary and the first slice's code words and values are all REQUESTED before any of them
    // is waited for (the flag is looked at before the first store): one memory round trip where flag -> dictionary -> slice
    // would be three -- which is what a product on a small matrix (a slice or two per workgroup) consists of.
    const int st = flag_done ? *flag_done : 0;
    const int tid = threadIdx.x;
    const bool chain = (remap & 256) != 0;
    const int32_t dv = tid < 16 ? dict[tid] : 0;
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    double dwy = 0.0, dyy = 0.0;

    // block map: round-robin, or XCD-block-cyclic (remap mode 3/4/5 = groups of G = 8/2/32): inside every
    // window of 8 G consecutive slices the workgroups of one XCD (blockIdx % 8) take G consecutive
    // slices, so the x entries a slice shares with its neighbours (2-D stencils: +-nx rows = a few
    // slices away) are fetched into ONE XCD's L2 instead of several (C2: 103.8 -> 99-100 us).  The
    // launcher only asks for it when the grid is a multiple of 8 G (the map is then a permutation).
    int64_t first = blockIdx.x;
    if ((remap & 255) >= 3) {
        const int G = (remap & 255) == 3 ? 8 : (remap & 255) == 4 ? 2 : (remap & 255) == 6 ? 64 : (remap & 255) == 7 ? 128 : 32;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        first = (int64_t)(loc / G) * (8 * G) + xcd * G + loc % G;
    }
    // ... or a slice schedule (slice_sched below; matrices with a far offset, 3-D grids): entry it * grid + workgroup of
    // a table, -1 = nothing left; the next entry is requested (a scalar load) before this slice's work
    int64_t sl = sched ? sched[blockIdx.x] : first;
    int sit = 0;
    u32x2 cw = {0xffffffffu, 0xffffffffu};
    f64x2 v[W];
    auto load_slice = [&](int64_t s_) {
        const int32_t row_ = (int32_t)(s_ * kSlRows) + 2 * tid;          // even: 16-byte aligned pairs
        // (the code array is padded to whole slices with "no entry" words: rows >= n do nothing)
        cw = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(scode + row_));
        const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + s_ * (int64_t)(W * kSlRows)) + tid;
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = __builtin_nontemporal_load(vb + u * BLOCK);
    };
    bool have = sl >= 0 && sl < nsl;
    if (have) load_slice(sl);
    if (st && gen >= st) return;
    if (tid < 16) dl[tid] = dv;
    __syncthreads();
    while (have) {
        int64_t nxt = sl + gridDim.x;
        if (sched) { ++sit; nxt = sit < sched_iters ? sched[(int64_t)sit * gridDim.x + blockIdx.x] : -1; }
        const int32_t row = (int32_t)(sl * kSlRows) + 2 * tid;
        f64x2 y0 = {0.0, 0.0};
        if (ADD) {
            if (row + 1 < n) y0 = *reinterpret_cast<const f64x2 *>(y + row);
            else if (row < n) y0.x = y[row];
        }
        double xa[W], xb[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const uint32_t ca = (cw.x >> (4 * u)) & 15u, cb = (cw.y >> (4 * u)) & 15u;
            xa[u] = ca != 15u ? x[row + dl[ca]] : 0.0;
            xb[u] = cb != 15u ? x[row + 1 + dl[cb]] : 0.0;
        }
        f64x2 z;
        z.x = (ADD && chain) ? y0.x : 0.0;
        z.y = (ADD && chain) ? y0.y : 0.0;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (((cw.x >> (4 * u)) & 15u) != 15u) z.x = z.x + v[u].x * xa[u];
            if (((cw.y >> (4 * u)) & 15u) != 15u) z.y = z.y + v[u].y * xb[u];
        }
        f64x2 yi;
        yi.x = ADD ? (chain ? z.x : y0.x + z.x) : 0.0 + z.x;
        yi.y = ADD ? (chain ? z.y : y0.y + z.y) : 0.0 + z.y;
        if (row + 1 < n) {
            __builtin_nontemporal_store(yi, reinterpret_cast<f64x2 *>(y + row));
            if (DOT_W) { const f64x2 wv = *reinterpret_cast<const f64x2 *>(w + row); dwy += wv.x * yi.x; dwy += wv.y * yi.y; }
            if (DOT_YY) { dyy += yi.x * yi.x; dyy += yi.y * yi.y; }
        } else if (row < n) {
            __builtin_nontemporal_store(yi.x, y + row);
            if (DOT_W) dwy += w[row] * yi.x;
            if (DOT_YY) dyy += yi.x * yi.x;
        }
        sl = nxt;
        have = sl >= 0 && sl < nsl;
        if (have) load_slice(sl);
    }
    if (DOT_W) {
        const double t = block_sum<BLOCK>(dwy, red);
        if (tid == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<BLOCK>(dyy, red);
        if (tid == 0) part_yy[blockIdx.x] = t;
    }
}

// SELL-128-512: general matrices whose rows are too long (> 32 entries) or too uneven for one slice width.
// The uniform sliced form pads every row to the longest one; rows of 20..40 or 33..64 entries then waste a third of the
// stream, and the kernels that avoid padding (k_csr_do / k_csr_rl: tiles staged through LDS, a row walked by its owner
// lane) run at 0.36-0.53 of the HBM peak, bound by the per-CU address / LDS pipes.  Here the rows
// of every window of 512 rows are SORTED by length (a permutation inside the window: y and w are still touched within one
// 4 KB window, and a chunk's x gathers stay as local as the rows' neighbourhood) and stored in chunks of 128 sorted rows, slot-major, each chunk with its own width = its longest row (rounded
// up to 2): padding is what neighbours in the sorted order differ by -- a few per cent.  One wave owns one chunk (a lane two
// adjacent positions): every matrix load is a coalesced 16 / 8 bytes per lane at a scalar base, no row pointer, no LDS, no
// barrier; a row's entries keep their stored order, products are rounded one by one and added left to right: the
// reference's row sum, bit for bit.
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
constexpr int kSellChunk = 128;
constexpr int kSellSigma = 512;       // rows sorted together (a multiple of the 512-row slices).  Measured on the banded test matrices: windows of
                                      // 2048 rows cut the padding from 10-12 % to 2-4 % and were 35-40 % SLOWER -- a chunk's 128 rows then come from a
                                      // 2048-row neighbourhood and their x gathers no longer fit the CU's L1 (33..64 entries per row: 656 -> 920 us)
// XW (x window): on banded matrices the gathers of a 512-row slice fall into a window of a few thousand columns, and what
// bounds the kernel without it is the L2 -> L1 line rate of those gathers (one 128-byte line moved per 8-byte gather: 0.47 of
// the HBM roofline on CSR bytes with rows of 33..300 entries).  With XW the slice's window of x -- win0[slice] .. + span,
// found at build (k_sell_window) -- is loaded into LDS with coalesced 16-byte loads first and every gather is an LDS read.
// Same products, same order of additions: bit-identical.  Taken when every slice's window fits 144 KiB of LDS.
// GS = 2 (wide windows: one workgroup per CU either way): a 512-thread workgroup takes TWO adjacent slices behind one window --
// 512 more columns for twice the rows, and eight waves' loads in flight instead of four.
template <bool ADD, bool DOT_W, bool DOT_YY, bool XW = false, int GS = 1>
__global__ __launch_bounds__(256 * GS) void k_csr_sell(
    int32_t n, const int64_t *__restrict__ off, const uint16_t *__restrict__ perm, const int32_t *__restrict__ scol,
    const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ w,
    double *__restrict__ part_wy, double *__restrict__ part_yy, const int *__restrict__ flag_done, int gen, int remap,
    const int32_t *__restrict__ win0 = nullptr, int32_t span = 0, int32_t xlen = 0)
{
    constexpr int BLOCK = 256 * GS;
    extern __shared__ double xs[];               // XW: the window of x of the workgroup's slice(s)
    __shared__ double red[BLOCK / 64];
    const int st = flag_done ? *flag_done : 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool chain = (remap & 256) != 0;
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    double dwy = 0.0, dyy = 0.0;
    if (st && gen >= st) return;

    int64_t first = blockIdx.x;             // XCD-block-cyclic slices, see k_csr_sl
    if ((remap & 255) >= 3) {
        const int G = (remap & 255) == 3 ? 8 : (remap & 255) == 4 ? 2 : (remap & 255) == 6 ? 64 : (remap & 255) == 7 ? 128 : 32;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        first = (int64_t)(loc / G) * (8 * G) + xcd * G + loc % G;
    }
    const int64_t ngr = (nsl + GS - 1) / GS;
    for (int64_t gr = first; gr < ngr; gr += gridDim.x) {
        const int64_t sl = gr * GS + (GS == 2 ? (wave >> 2) : 0);
        const bool has = sl < nsl;                                                  // (GS = 2: the last group may hold one slice)
        const int64_t chunk = (has ? sl : nsl - 1) * (kSlRows / kSellChunk) + (GS == 2 ? (wave & 3) : wave);
        const int64_t o0 = off[chunk];
        const int32_t W = has ? (int32_t)((off[chunk + 1] - o0) / kSellChunk) : 0;  // a multiple of 2
        int32_t w0 = 0;
        if (XW) {
            w0 = win0[gr];                                                          // (even: 16-byte loads)
            const int32_t cnt = min(span, xlen - w0);
            __syncthreads();                                                        // the previous slice's gathers are done with xs
            const f64x2 *src = reinterpret_cast<const f64x2 *>(x + w0);
            f64x2 *dst = reinterpret_cast<f64x2 *>(xs);
            for (int32_t t = threadIdx.x; t < (cnt >> 1); t += BLOCK) dst[t] = src[t];
            if ((cnt & 1) && threadIdx.x == 0) xs[cnt - 1] = x[w0 + cnt - 1];
            __syncthreads();
        }
        const u16x2 pr = *reinterpret_cast<const u16x2 *>(perm + chunk * kSellChunk + 2 * lane);
        const int32_t base = (int32_t)(sl / (kSellSigma / kSlRows)) * kSellSigma;   // the sort window's first row
        const bool va = has && pr.x != 0xffffu, vb_ = has && pr.y != 0xffffu;
        const int32_t ra = base + pr.x, rb = base + pr.y;
        double ya = 0.0, yb = 0.0;
        if (ADD) { if (va) ya = y[ra]; if (vb_) yb = y[rb]; }
        const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + o0) + lane;
        const i32x2 *cb = reinterpret_cast<const i32x2 *>(scol + o0) + lane;
        double za = (ADD && chain) ? ya : 0.0, zb = (ADD && chain) ? yb : 0.0;
        auto slots = [&](int32_t c0, auto CHc) {
            constexpr int CH = decltype(CHc)::value;
            f64x2 v[CH];
            i32x2 cc[CH];
            double xa[CH], xb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                v[u] = __builtin_nontemporal_load(vb + (int64_t)(c0 + u) * (kSellChunk / 2));
                cc[u] = __builtin_nontemporal_load(cb + (int64_t)(c0 + u) * (kSellChunk / 2));
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                xa[u] = cc[u].x >= 0 ? (XW ? xs[cc[u].x - w0] : x[cc[u].x]) : 0.0;
                xb[u] = cc[u].y >= 0 ? (XW ? xs[cc[u].y - w0] : x[cc[u].y]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (cc[u].x >= 0) za = za + v[u].x * xa[u];
                if (cc[u].y >= 0) zb = zb + v[u].y * xb[u];
            }
        };
        int32_t c0 = 0;
        for (; c0 + 8 <= W; c0 += 8) slots(c0, std::integral_constant<int, 8>());
        if (c0 + 4 <= W) { slots(c0, std::integral_constant<int, 4>()); c0 += 4; }
        if (c0 < W) slots(c0, std::integral_constant<int, 2>());
        const double yia = ADD ? (chain ? za : ya + za) : 0.0 + za;
        const double yib = ADD ? (chain ? zb : yb + zb) : 0.0 + zb;
        if (va) {
            y[ra] = yia;
            if (DOT_W) dwy += w[ra] * yia;
            if (DOT_YY) dyy += yia * yia;
        }
        if (vb_) {
            y[rb] = yib;
            if (DOT_W) dwy += w[rb] * yib;
            if (DOT_YY) dyy += yib * yib;
        }
    }
    if (DOT_W) {
        const double t = block_sum<BLOCK>(dwy, red);
        if (threadIdx.x == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<BLOCK>(dyy, red);
        if (threadIdx.x == 0) part_yy[blockIdx.x] = t;
    }
}
// setup: the window of columns every 512-row slice gathers from (its first column rounded down to even, and the span to the
// last one); the longest span of the part by atomicMax
__global__ __launch_bounds__(256) void k_sell_window(int64_t nsl, int gs, const int64_t *__restrict__ off, const int32_t *__restrict__ scol,
                                                     int32_t *__restrict__ win0, int32_t *__restrict__ max_span)
{
    __shared__ int32_t lo_s[4], hi_s[4];
    const int64_t ngr = (nsl + gs - 1) / gs;
    for (int64_t sl = blockIdx.x; sl < ngr; sl += gridDim.x) {            // (sl: group of gs slices)
        const int64_t a = off[sl * gs * (kSlRows / kSellChunk)], b = off[min((sl + 1) * gs, nsl) * (kSlRows / kSellChunk)];
        int32_t lo = INT32_MAX, hi = -1;
        for (int64_t k = a + threadIdx.x; k < b; k += 256) {
            const int32_t c = scol[k];
            if (c >= 0) { lo = min(lo, c); hi = max(hi, c); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { lo_s[threadIdx.x >> 6] = lo; hi_s[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 1; t < 4; ++t) { lo_s[0] = min(lo_s[0], lo_s[t]); hi_s[0] = max(hi_s[0], hi_s[t]); }
            const int32_t l = hi_s[0] < 0 ? 0 : lo_s[0] & ~1;
            win0[sl] = l;
            if (hi_s[0] >= 0) atomicMax(max_span, hi_s[0] - l + 1);
        }
    }
}
// setup: positions of a window's rows sorted by length (longest first, ties by row: the sort is a pure function of the row
// lengths), the chunks' widths (as entry counts, to be prefix-summed), ...
__global__ __launch_bounds__(256) void k_sell_sort(int32_t n, const int32_t *__restrict__ rowptr, uint16_t *__restrict__ perm,
                                                   int64_t *__restrict__ wid)
{
    __shared__ int32_t len[kSellSigma];
    __shared__ int32_t first_len[kSellSigma / kSellChunk];
    const int64_t nwin = ((int64_t)n + kSellSigma - 1) / kSellSigma;
    const int64_t nch = (((int64_t)n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);       // chunks that exist (whole slices)
    for (int64_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        __syncthreads();
        for (int r = threadIdx.x; r < kSellSigma; r += blockDim.x) {
            const int64_t row = win * kSellSigma + r;
            len[r] = row < n ? rowptr[row + 1] - rowptr[row] : -1;        // (rows past the end sort last)
        }
        if (threadIdx.x < kSellSigma / kSellChunk) first_len[threadIdx.x] = 0;
        __syncthreads();
        for (int r = threadIdx.x; r < kSellSigma; r += blockDim.x) {
            const int32_t l = len[r];
            int rank = 0;
            for (int j = 0; j < kSellSigma; ++j) rank += (len[j] > l || (len[j] == l && j < r)) ? 1 : 0;
            const int64_t pos = win * kSellSigma + rank;
            if (pos / kSellChunk < nch) perm[pos] = l >= 0 ? (uint16_t)r : (uint16_t)0xffffu;
            if (rank % kSellChunk == 0) first_len[rank / kSellChunk] = l > 0 ? l : 0;
        }
        __syncthreads();
        if (threadIdx.x < kSellSigma / kSellChunk) {
            const int64_t c = win * (kSellSigma / kSellChunk) + threadIdx.x;
            if (c < nch) wid[c] = (int64_t)((first_len[threadIdx.x] + 1) / 2 * 2) * kSellChunk;
        }
    }
}
// ... and the chunks' slots filled from the CSR arrays (val only: a value update)
__global__ __launch_bounds__(256) void k_sell_fill(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, const int64_t *__restrict__ off,
                                                   const uint16_t *__restrict__ perm, int32_t *__restrict__ scol, double *__restrict__ sval)
{
    const int64_t nch = (((int64_t)n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nch; chunk += (int64_t)gridDim.x * 4) {
        const int64_t o0 = off[chunk];
        const int32_t W = (int32_t)((off[chunk + 1] - o0) / kSellChunk);
        for (int h = 0; h < 2; ++h) {
            const int q = lane + 64 * h;
            const uint16_t pr = perm[chunk * kSellChunk + q];
            int32_t k = 0, ke = 0;
            if (pr != 0xffffu) {
                const int64_t row = (chunk / (kSellSigma / kSellChunk)) * kSellSigma + pr;
                k = rowptr[row]; ke = rowptr[row + 1];
            }
            for (int32_t u = 0; u < W; ++u) {
                const bool has = k + u < ke;
                sval[o0 + (int64_t)u * kSellChunk + q] = has ? val[k + u] : 0.0;
                if (scol) scol[o0 + (int64_t)u * kSellChunk + q] = has ? col[k + u] : -1;
            }
        }
    }
}

// The sliced form for matrices WITHOUT an offset dictionary (arbitrary columns) whose rows are short
// (<= 32 entries) and of similar length: the int32 column of every slot is stored beside the value,
// slot-major in the same 512-row slices (-1 = no entry); 12 bytes per slot like plain CSR, but every
// load is a coalesced 8/16 bytes per lane and there is no row pointer, no LDS, no barrier.  Slots are
// walked in chunks of 8 (registers).
template <int W, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(256) void k_csr_sl32(
    int32_t n, const int32_t *__restrict__ scol, const double *__restrict__ sval,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ w,
    double *__restrict__ part_wy, double *__restrict__ part_yy, const int *__restrict__ flag_done, int gen, int remap)
{
    constexpr int BLOCK = 256;
    __shared__ double red[BLOCK / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    const int tid = threadIdx.x;
    const bool chain = (remap & 256) != 0;
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    double dwy = 0.0, dyy = 0.0;

    int64_t first = blockIdx.x;             // XCD-block-cyclic slices, see k_csr_sl
    if ((remap & 255) >= 3) {
        const int G = (remap & 255) == 3 ? 8 : (remap & 255) == 4 ? 2 : (remap & 255) == 6 ? 64 : (remap & 255) == 7 ? 128 : 32;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        first = (int64_t)(loc / G) * (8 * G) + xcd * G + loc % G;
    }
    for (int64_t sl = first; sl < nsl; sl += gridDim.x) {
        const int32_t row = (int32_t)(sl * kSlRows) + 2 * tid;
        const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + sl * (int64_t)(W * kSlRows)) + tid;
        const i32x2 *cb = reinterpret_cast<const i32x2 *>(scol + sl * (int64_t)(W * kSlRows)) + tid;
        f64x2 y0 = {0.0, 0.0};
        if (ADD) {
            if (row + 1 < n) y0 = *reinterpret_cast<const f64x2 *>(y + row);
            else if (row < n) y0.x = y[row];
        }
        f64x2 z;
        z.x = (ADD && chain) ? y0.x : 0.0;
        z.y = (ADD && chain) ? y0.y : 0.0;
#pragma unroll
        for (int c0 = 0; c0 < W; c0 += 8) {
            constexpr int CH = 8;
            f64x2 v[CH];
            i32x2 cc[CH];
            double xa[CH], xb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (c0 + u < W) {
                    v[u] = __builtin_nontemporal_load(vb + (c0 + u) * BLOCK);
                    cc[u] = __builtin_nontemporal_load(cb + (c0 + u) * BLOCK);
                }
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (c0 + u < W) {
                    xa[u] = cc[u].x >= 0 ? x[cc[u].x] : 0.0;
                    xb[u] = cc[u].y >= 0 ? x[cc[u].y] : 0.0;
                }
#pragma unroll
            for (int u = 0; u < CH; ++u)
                if (c0 + u < W) {
                    if (cc[u].x >= 0) z.x = z.x + v[u].x * xa[u];
                    if (cc[u].y >= 0) z.y = z.y + v[u].y * xb[u];
                }
            if (W > 8) __builtin_amdgcn_sched_barrier(0);       // one chunk's registers at a time (all chunks at once, W = 28: 256 VGPRs, 302 -> 320 us)
        }
        f64x2 yi;
        yi.x = ADD ? (chain ? z.x : y0.x + z.x) : 0.0 + z.x;
        yi.y = ADD ? (chain ? z.y : y0.y + z.y) : 0.0 + z.y;
        if (row + 1 < n) {
            __builtin_nontemporal_store(yi, reinterpret_cast<f64x2 *>(y + row));
            if (DOT_W) { const f64x2 wv = *reinterpret_cast<const f64x2 *>(w + row); dwy += wv.x * yi.x; dwy += wv.y * yi.y; }
            if (DOT_YY) { dyy += yi.x * yi.x; dyy += yi.y * yi.y; }
        } else if (row < n) {
            __builtin_nontemporal_store(yi.x, y + row);
            if (DOT_W) dwy += w[row] * yi.x;
            if (DOT_YY) dyy += yi.x * yi.x;
        }
    }
    if (DOT_W) {
        const double t = block_sum<BLOCK>(dwy, red);
        if (tid == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<BLOCK>(dyy, red);
        if (tid == 0) part_yy[blockIdx.x] = t;
    }
}

// The sliced form for rows of 9..32 entries drawn from <= 255 distinct (column - row) offsets (27- / 19-point stencils,
// 2-D 9-point, block-structured FEM): values slot-major in the same 512-row slices, W = the longest row rounded up to 8,
// and ONE BYTE per slot for the column -- per chunk of 8 slots the 8 codes of a row sit together, so a lane reads the
// codes of its two rows with one 16-byte load per chunk.  9 bytes per slot instead of CSR's 12, every load a coalesced
// 16 bytes per lane, no row pointer, no barrier; the dictionary (<= 255 offsets) is looked up in LDS.  Slots are walked
// in stored order: bit-identical to csr_matvec_add.
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
// W = value slots per row (compile time: a run-time W costs 10 %): the instantiated widths below, the smallest one that
// holds the longest row; code bytes come in chunks of 8 per row.
#define SGM_SLB_WIDTHS(X) X(9) X(12) X(15) X(16) X(19) X(20) X(24) X(25) X(27) X(28) X(32)
template <int W, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(256) void k_csr_slb(
    int32_t n, const uint8_t *__restrict__ sbcode, const int32_t *__restrict__ dict, const double *__restrict__ sval,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ w,
    double *__restrict__ part_wy, double *__restrict__ part_yy, const int *__restrict__ flag_done, int gen, int remap,
    const int32_t *__restrict__ sched, int sched_iters)
{
    constexpr int BLOCK = 256;
    constexpr int NCH = (W + 7) / 8;
    static_assert(W >= 9 && W <= 32, "rows of 9..32 entries");
    __shared__ int32_t dl[256];
    __shared__ double red[BLOCK / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    const int tid = threadIdx.x;
    dl[tid] = dict[tid];
    __syncthreads();
    const bool chain = (remap & 256) != 0;
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    double dwy = 0.0, dyy = 0.0;

    int64_t first = blockIdx.x;             // XCD-block-cyclic slices, see k_csr_sl
    if ((remap & 255) >= 3) {
        const int G = (remap & 255) == 3 ? 8 : (remap & 255) == 4 ? 2 : (remap & 255) == 6 ? 64 : (remap & 255) == 7 ? 128 : 32;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        first = (int64_t)(loc / G) * (8 * G) + xcd * G + loc % G;
    }
    int64_t sl = sched ? sched[blockIdx.x] : first;        // slice schedule, see k_csr_sl
    int sit = 0;
    while (sl >= 0 && sl < nsl) {
        int64_t nxt = sl + gridDim.x;
        if (sched) { ++sit; nxt = sit < sched_iters ? sched[(int64_t)sit * gridDim.x + blockIdx.x] : -1; }
        const int32_t row = (int32_t)(sl * kSlRows) + 2 * tid;
        const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + sl * (int64_t)W * kSlRows) + tid;
        const u32x4s *cb = reinterpret_cast<const u32x4s *>(sbcode + sl * (int64_t)(NCH * 8 * kSlRows)) + tid;      // 16 bytes: rows 2t, 2t+1
        f64x2 y0 = {0.0, 0.0};
        if (ADD) {
            if (row + 1 < n) y0 = *reinterpret_cast<const f64x2 *>(y + row);
            else if (row < n) y0.x = y[row];
        }
        f64x2 z;
        z.x = (ADD && chain) ? y0.x : 0.0;
        z.y = (ADD && chain) ? y0.y : 0.0;
        // chunks of 8 slots.  Unrolled over all chunks the compiler keeps every slot live (237 VGPRs at W = 27, two waves
        // per SIMD) -- which still wins up to W = 28: all of a wave's loads are in flight at once; beyond that a rolled loop
        auto chunk = [&](int c, auto cnt_tag) {
            constexpr int CNT = decltype(cnt_tag)::value;
            f64x2 v[CNT];
            double xa[CNT], xb[CNT];
            const u32x4s cw = __builtin_nontemporal_load(cb + c * BLOCK);
#pragma unroll
            for (int u = 0; u < CNT; ++u) v[u] = __builtin_nontemporal_load(vb + (c * 8 + u) * BLOCK);
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                const uint32_t ca = ((u < 4 ? cw.x : cw.y) >> (8 * (u & 3))) & 255u;
                const uint32_t cbv = ((u < 4 ? cw.z : cw.w) >> (8 * (u & 3))) & 255u;
                xa[u] = ca != 255u ? x[row + dl[ca]] : 0.0;
                xb[u] = cbv != 255u ? x[row + 1 + dl[cbv]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                const uint32_t ca = ((u < 4 ? cw.x : cw.y) >> (8 * (u & 3))) & 255u;
                const uint32_t cbv = ((u < 4 ? cw.z : cw.w) >> (8 * (u & 3))) & 255u;
                if (ca != 255u) z.x = z.x + v[u].x * xa[u];
                if (cbv != 255u) z.y = z.y + v[u].y * xb[u];
            }
        };
        if (W > 28) {       // (measured on the 160^3 27-point matrix, rolled vs unrolled: W = 32: 249 vs 281 us; 28: 237 vs 224; 27: 229 vs 218)
#pragma unroll 1
            for (int c = 0; c < W / 8; ++c) chunk(c, std::integral_constant<int, 8>{});
        } else {
#pragma unroll
            for (int c = 0; c < W / 8; ++c) chunk(c, std::integral_constant<int, 8>{});
        }
        if (W % 8) chunk(W / 8, std::integral_constant<int, (W % 8 ? W % 8 : 8)>{});
        f64x2 yi;
        yi.x = ADD ? (chain ? z.x : y0.x + z.x) : 0.0 + z.x;
        yi.y = ADD ? (chain ? z.y : y0.y + z.y) : 0.0 + z.y;
        if (row + 1 < n) {
            __builtin_nontemporal_store(yi, reinterpret_cast<f64x2 *>(y + row));
            if (DOT_W) { const f64x2 wv = *reinterpret_cast<const f64x2 *>(w + row); dwy += wv.x * yi.x; dwy += wv.y * yi.y; }
            if (DOT_YY) { dyy += yi.x * yi.x; dyy += yi.y * yi.y; }
        } else if (row < n) {
            __builtin_nontemporal_store(yi.x, y + row);
            if (DOT_W) dwy += w[row] * yi.x;
            if (DOT_YY) dyy += yi.x * yi.x;
        }
        sl = nxt;
    }
    if (DOT_W) {
        const double t = block_sum<BLOCK>(dwy, red);
        if (tid == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<BLOCK>(dyy, red);
        if (tid == 0) part_yy[blockIdx.x] = t;
    }
}

// 1-byte codes in CSR order -> the chunked sliced layout of k_csr_slb (255 where a row has no entry in the slot)
__global__ __launch_bounds__(256) void k_slb_pack_codes(int32_t n, int32_t W, const int32_t *__restrict__ rowptr,
                                                        const uint8_t *__restrict__ code, uint8_t *__restrict__ sbcode)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            int32_t k = 0, ke = 0;
            if (row < n) { k = rowptr[row]; ke = rowptr[row + 1]; }
            const int WC = (W + 7) / 8 * 8;          // code bytes per row
            uint8_t *dst = sbcode + sl * (int64_t)WC * kSlRows + (int64_t)r * 8;
            for (int u = 0; u < WC; ++u) dst[(int64_t)(u >> 3) * kSlRows * 8 + (u & 7)] = (u < W && k + u < ke) ? code[k + u] : (uint8_t)255;
        }
}

// columns in CSR order -> sliced layout (-1 where a row has no entry in the slot; whole slices)
__global__ __launch_bounds__(256) void k_sl_pack_cols(int32_t n, int32_t W, const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, int32_t *__restrict__ scol)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            int32_t k = 0, ke = 0;
            if (row < n) { k = rowptr[row]; ke = rowptr[row + 1]; }
            int32_t *dst = scol + sl * (int64_t)W * kSlRows + r;
            for (int u = 0; u < W; ++u) dst[(int64_t)u * kSlRows] = k + u < ke ? col[k + u] : -1;
        }
}

// values in CSR order -> sliced layout (at upload and after every value update)
__global__ __launch_bounds__(256) void k_sl_pack(int32_t n, int32_t W, const int32_t *__restrict__ rowptr,
                                                 const double *__restrict__ val, double *__restrict__ sval)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            int32_t k = 0, ke = 0;
            if (row < n) { k = rowptr[row]; ke = rowptr[row + 1]; }
            double *dst = sval + sl * (int64_t)W * kSlRows + r;
            for (int u = 0; u < W; ++u) dst[(int64_t)u * kSlRows] = k + u < ke ? val[k + u] : 0.0;
        }
}

// the same for an ELLPACK matrix (slot-major eval, stride n): all max_d slots of a row are entries
__global__ __launch_bounds__(256) void k_sl_pack_ell(int32_t n, int32_t W, int32_t max_d, const double *__restrict__ eval,
                                                     double *__restrict__ sval)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            double *dst = sval + sl * (int64_t)W * kSlRows + r;
            for (int u = 0; u < W; ++u) dst[(int64_t)u * kSlRows] = (row < n && u < max_d) ? eval[(int64_t)u * n + row] : 0.0;
        }
}

// ELLPACK, slot-major device layout: lane i owns row i and walks ALL max_d slots in
// order (padding slots multiply 0.0 by x(last neighbour), exactly like the reference,
// so a non-finite x entry propagates the same way).
template <int U, bool NT, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(kBlock) void k_ell_spmv(
    int32_t n, int32_t max_d, const int32_t *__restrict__ ecol, const double *__restrict__ eval,
    const double *__restrict__ x, double *__restrict__ y, const double *__restrict__ w,
    double *__restrict__ part_wy, double *__restrict__ part_yy, const int *__restrict__ flag_done, int gen,
    int chain)
{
    __shared__ double red[kBlock / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    double dwy = 0.0, dyy = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double wv = 0.0, y0 = 0.0;
        if (DOT_W) wv = w[i];
        if (ADD) y0 = y[i];
        double z = (ADD && chain) ? y0 : 0.0;
        int32_t k = 0;
        for (; k + U <= max_d; k += U) {
            int32_t c[U];
            double v[U], xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                c[u] = ld_stream(ecol + (int64_t)(k + u) * n + i, NT);
                v[u] = ld_stream(eval + (int64_t)(k + u) * n + i, NT);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) z = z + v[u] * xv[u];
        }
        for (; k < max_d; ++k) z = z + eval[(int64_t)k * n + i] * x[ecol[(int64_t)k * n + i]];
        const double yi = ADD ? (chain ? z : y0 + z) : 0.0 + z;
        y[i] = yi;
        if (DOT_W) dwy += wv * yi;
        if (DOT_YY) dyy += yi * yi;
    }
    if (DOT_W) {
        const double t = block_sum<kBlock>(dwy, red);
        if (threadIdx.x == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<kBlock>(dyy, red);
        if (threadIdx.x == 0) part_yy[blockIdx.x] = t;
    }
}

// ELLPACK with dictionary-coded column offsets (max_d <= 16, <= 255 distinct col-row offsets):
// the MDP code bytes of a row are one 4/8/16-byte load, val stays slot-major (8-B coalesced
// loads), lane i gathers x(i + dict[code]) for its own row -- adjacent lanes read adjacent x
// entries for stencil-like matrices.  9 instead of 12 bytes per slot; all max_d slots are
// walked in order (padding included), so the result is bit-identical to k_ell_spmv.
template <int MDP, bool ADD, bool DOT_W, bool DOT_YY>
__global__ __launch_bounds__(kBlock) void k_ell_do(
    int32_t n, int32_t max_d, const uint8_t *__restrict__ ecode, const int32_t *__restrict__ dict,
    const double *__restrict__ eval, const double *__restrict__ x, double *__restrict__ y,
    const double *__restrict__ w, double *__restrict__ part_wy, double *__restrict__ part_yy,
    const int *__restrict__ flag_done, int gen, int chain)
{
    __shared__ double red[kBlock / 64];
    __shared__ int32_t dl[256];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    for (int t = threadIdx.x; t < 256; t += kBlock) dl[t] = dict[t];
    __syncthreads();
    double dwy = 0.0, dyy = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        uint32_t cw[MDP / 4];
        const uint32_t *cp = reinterpret_cast<const uint32_t *>(ecode + i * MDP);
#pragma unroll
        for (int q = 0; q < MDP / 4; ++q) cw[q] = __builtin_nontemporal_load(cp + q);
        double wv = 0.0, y0 = 0.0;
        if (DOT_W) wv = w[i];
        if (ADD) y0 = y[i];
        double v[MDP], xv[MDP];
#pragma unroll
        for (int k = 0; k < MDP; ++k)
            if (k < max_d) {
                v[k] = __builtin_nontemporal_load(eval + (int64_t)k * n + i);
                xv[k] = x[i + dl[(cw[k >> 2] >> (8 * (k & 3))) & 255]];
            }
        double z = (ADD && chain) ? y0 : 0.0;
#pragma unroll
        for (int k = 0; k < MDP; ++k)
            if (k < max_d) z = z + v[k] * xv[k];
        const double yi = ADD ? (chain ? z : y0 + z) : 0.0 + z;
        y[i] = yi;
        if (DOT_W) dwy += wv * yi;
        if (DOT_YY) dyy += yi * yi;
    }
    if (DOT_W) {
        const double t = block_sum<kBlock>(dwy, red);
        if (threadIdx.x == 0) part_wy[blockIdx.x] = t;
    }
    if (DOT_YY) {
        const double t = block_sum<kBlock>(dyy, red);
        if (threadIdx.x == 0) part_yy[blockIdx.x] = t;
    }
}

__global__ void k_gather(double *__restrict__ dst, const double *__restrict__ src,
                         const int32_t *__restrict__ idx, int32_t count)
{
    int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) dst[t] = src[idx[t]];
}

// ---------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------
static int g_launch_flags = 0;        // 256: chained accumulation (see k_csr_* `chain`)

// Launch configuration of the CSR kernel (block, vpt, nt, maxgrid, remap: fixed; the sweeps that chose them are in CHANGELOG.md)
struct SpmvCfg { int block = 256, vpt = 2, nt = 1, maxgrid = 0, remap = 1, do_vpt = 0; };   // 0 = automatic
static SpmvCfg &spmv_cfg()
{
    static SpmvCfg c;
    static bool init = false;
    if (!init) {
        init = true;
        if (c.maxgrid > kMaxGrid) c.maxgrid = kMaxGrid;
        // only instantiated (block, vpt) pairs: anything else would launch a kernel of another shape
        if (c.block != 256 && c.block != 512 && c.block != 1024) c.block = 256;
        if (c.vpt != 2 && c.vpt != 4 && c.vpt != 8) c.vpt = 2;
        if (c.block == 1024 && c.vpt == 8) c.vpt = 4;
    }
    return c;
}

static int resident_per_cu(bool dict, int block, int v, int cw = 4);
static const SliceSched *slice_sched(const Part &p, int32_t lo, int32_t hi, int grid);
static void free_slice_sched(Part &p);
int ell_grid(const Part &p);
// sgm_ellcb.hip: column-blocked two-phase product for ELLPACK matrices with random columns
bool use_ell_colblock(const Part &p);
int ell_colblock_grid(const Part &p);
int build_ell_colblock(Part &p);
int refresh_ell_colblock_values(Part &p);
void free_ell_colblock(Part &p);
int launch_ell_colblock(const Part &p, int grid, const double *x, double *y, bool add, bool chain, const double *w,
                        double *pwy, double *pyy, const int *flag, int gen);
int64_t ell_colblock_resident_bytes(const Part &p);
int64_t ell_colblock_matvec_bytes(const Part &p);
// k_csr_do exists for 256- and 512-thread workgroups only; with any other block size the
// matrices it would serve take the streaming kernel (which has the 1024-thread variants) instead
static bool do_block_ok() { const int b = spmv_cfg().block; return b == 256 || b == 512; }
static bool use_offset_dict(const Part &p) { return (p.code || (p.lean && p.dict)) && p.opt.csr_offset_dict && do_block_ok(); }
static bool use_sliced(const Part &p) { return p.scode && p.opt.csr_sliced && p.opt.csr_offset_dict; }
static bool use_sliced32(const Part &p) { return p.scol && p.opt.csr_sliced && !p.ecol; }
static bool use_slicedb(const Part &p) { return p.sbcode && p.opt.csr_sliced && p.opt.csr_offset_dict; }
static bool use_sell(const Part &p) { return p.sl_val && p.opt.csr_sliced && p.opt.csr_sell && !p.ecol; }
static bool lean_sell(const Part &p);      // (the SELL form is the part's only resident layout: see csr_lean below)
static bool any_sliced(const Part &p) { return use_sliced(p) || use_sliced32(p) || use_slicedb(p) || use_sell(p); }
// k_csr_do serves both the dictionary form and, for short rows, plain int32 columns
static bool use_row_owner(const Part &p)
{
    // (rows of 33..64 entries, banded: 809-822 us against 906 with k_csr_rl and 1100-1150 with k_csr_spmv; beyond 64 the
    // few lanes that own a tile's rows walk too long: 64..128 entries 1160 us against 890 with k_csr_rl)
    return use_offset_dict(p) || (do_block_ok() && p.opt.csr_row_owner && p.max_row > 0 && p.max_row <= 64);
}

// long rows without a dictionary: the line-staged row-owner kernel (k_csr_rl).  A block takes as many passes as its
// longest row has lines, so a matrix with a row beyond 4096 entries (an arrow matrix's dense row) stays with the
// streaming kernel, whose gathers do not wait for one lane.
// (rows of 64..128 entries, banded: 890 us against 1020 with a row-grouped gather variant -- contiguous tiles, Q gather
// lanes per row, sums by the owner; in history -- 1160 with k_csr_do and 1180 with k_csr_spmv; 150..300: 1070 against 1670)
static bool use_row_lines(const Part &p)
{
    // ... and rows of SIMILAR length, at least a line long on average: every row of a block waits for the block's longest one
    // (the 5-point matrix forced through it: 491 us against 153 with k_csr_spmv), so a few long rows among short ones
    // (max > 4 x mean) also stay with the streaming kernel
    return p.opt.csr_row_lines && !any_sliced(p) && !use_row_owner(p) && p.n > 0 && p.max_row <= 4096 &&
           p.nnz >= 16 * (int64_t)p.n && (int64_t)p.max_row * p.n <= 4 * p.nnz;
}
static int row_lines_resident_per_cu()
{
    static int nb = 0;
    if (!nb && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_csr_rl<false, true, false>, 256, 0) != hipSuccess || nb < 1))
        nb = 3;
    return nb;
}

// (BLOCK, TILE) instantiations of the offset-dict kernel.  TILE = entries staged in LDS per
// pass (9 bytes each); the launcher picks the smallest one that holds a whole row block of
// average density (+ alignment slack), so that a row block is one load phase + one gather
// phase and the LDS footprint stays small enough for 8 workgroups per CU.
#define SGM_DO_VARIANTS(X) X(256, 1024) X(256, 1536) X(256, 1920) X(256, 2048) X(256, 4096) X(512, 2048) X(512, 3072) X(512, 3840)
static int do_tile_for(const Part &p)
{
    const SpmvCfg &c = spmv_cfg();
    static const int t256[] = {1024, 1536, 1920, 2048, 4096}, t512[] = {2048, 3072, 3840};
    const int *tiles = c.block == 512 ? t512 : t256;
    // (the 4096-entry tile serves int32 columns only -- long rows: 33..64 entries 794 -> 770 us, 20..40 746 -> 719;
    // the 1-byte-code form keeps its 2048, measured with seven workgroups per CU)
    const int nt = c.block == 512 ? 3 : (use_offset_dict(p) ? 4 : 5);
    if (c.do_vpt) return tiles[std::min(std::max(c.do_vpt - 1, 0), nt - 1)];     // tuning override: 1..nt
    const double per_block = (double)p.nnz / (p.n > 0 ? p.n : 1) * c.block + 4;
    for (int i = 0; i < nt; ++i)
        if (per_block <= tiles[i]) return tiles[i];
    return tiles[nt - 1];
}

// Persistent grid: exactly the number of workgroups that are resident at once (LDS- or
// wave-limited), rounded down to a multiple of 8 for the XCD map -- a larger grid only adds
// a tail, a smaller one leaves CUs idle (measured: 7-point, 19.5 KiB LDS: 1536 beats 2048).
static int grid_for_rows(const Part &p, int64_t rows, int64_t limit, bool dots = true)
{
    const SpmvCfg &c = spmv_cfg();
    const int blk = any_sliced(p) ? kSlRows : use_row_lines(p) ? 256 : c.block;
    const int64_t nrb = (rows + blk - 1) / blk;
    int64_t g = ((nrb + 7) / 8) * 8;
    int64_t cap = c.maxgrid;
    // round-robin slices, not a persistent resident grid: 4096 workgroups; 8192 from 32768 slices on
    // (n >= 1.7e7: 464^3 1.54 -> 1.45 ms, 300^3 355 -> 345 us; below that the consumers' re-reduction of more partials costs more)
    // a product WITHOUT fused dots takes the 8192 grid at every size (C2: 98.5 -> 95.9 us with the block-cyclic map); with
    // them the 4096 one below 32768 slices (8192 partials per dot cost the CG update kernels more than the product gains)
    if (cap <= 0 && any_sliced(p)) cap = (nrb >= 32768 || !dots) ? kMaxGrid : kMaxGrid / 2;
    if (cap <= 0 && use_row_lines(p)) cap = (int64_t)row_lines_resident_per_cu() * g_rt.num_cu;
    if (cap <= 0) cap = (int64_t)resident_per_cu(use_row_owner(p), c.block, use_row_owner(p) ? do_tile_for(p) : c.vpt,
                                                 use_offset_dict(p) ? 1 : 4) * g_rt.num_cu;
    if (cap > limit) cap = limit;
    if (g > cap) g = cap / 8 * 8;
    if (g < 8) g = 8;
    return (int)g;
}

// Row ranges of one SpMV.  A part with halo columns is split so that the rows that touch
// only owned columns ("interior", one contiguous run of row blocks found at setup) can run
// while the halo exchange is still in flight; the head / tail ranges follow it.
struct RowRange { int32_t lo, hi; int grid, part_off; };
// whole: the halo is known to be in place (CG on a partition forms p's halo itself, option dist_halo_fused): no reason to cut
// the rows -- one launch like a part without halo columns.
static int spmv_ranges(const Part &p, RowRange out[3], bool dots = true, bool whole = false)
{
    int nr = 0, off = 0;
    auto add = [&](int32_t lo, int32_t hi) {       // grids sum to <= kMaxGrid partial slots
        if (hi <= lo) return;
        out[nr] = RowRange{lo, hi, grid_for_rows(p, hi - lo, nr == 0 ? kMaxGrid / 4 : kMaxGrid / 8), off};
        off += out[nr].grid;
        ++nr;
    };
    if (p.n_halo == 0 || p.int_hi <= p.int_lo || whole) {
        if (any_sliced(p)) {           // one range, all kMaxGrid partial slots are its own
            out[0] = RowRange{0, p.n, grid_for_rows(p, p.n > 0 ? p.n : 1, kMaxGrid, dots), 0};
            return 1;
        }
        add(0, p.n > 0 ? p.n : 1);
        if (nr) out[0].hi = p.n;
        return nr;
    }
    add(p.int_lo, p.int_hi);      // interior first: it is launched before the halo has arrived
    add(0, p.int_lo);
    add(p.int_hi, p.n);
    return nr;
}

int spmv_grid(const Part &p, bool whole)       // = number of partial sums one SpMV leaves per fused dot
{
    if (p.dot_grid_override) return p.dot_grid_override;
    if (p.ecol) return ell_grid(p);
    if (use_ell_colblock(p)) return ell_colblock_grid(p);
    RowRange r[3];
    const int nr = spmv_ranges(p, r, true, whole);
    return nr ? r[nr - 1].part_off + r[nr - 1].grid : 8;
}

template <int BLOCK, int VPT, bool NT, bool ADD>
static void launch_csr_cfg(const Part &p, int grid, const double *x, double *y, const double *w,
                           double *pwy, double *pyy, const int *flag, int gen, int remap)
{
    hipStream_t st = g_rt.stream;
#define L(DW, DY)                                                                                  \
    hipLaunchKernelGGL((k_csr_spmv<BLOCK, VPT, NT, ADD, DW, DY>), dim3(grid), dim3(BLOCK), 0, st,  \
                       p.n, p.rowptr, p.col, p.val, x, y, w, pwy, pyy, flag, gen, remap)
    if (w && pyy) L(true, true);
    else if (w) L(true, false);
    else if (pyy) L(false, true);
    else L(false, false);
#undef L
}

template <bool ADD>
static void launch_csr(const Part &p, int grid, const double *x, double *y, const double *w,
                       double *pwy, double *pyy, const int *flag, int gen)
{
    const SpmvCfg &c = spmv_cfg();
#define CFG(B, V)                                                                              \
    if (c.block == B && c.vpt == V) {                                                          \
        if (c.nt) launch_csr_cfg<B, V, true, ADD>(p, grid, x, y, w, pwy, pyy, flag, gen, c.remap | g_launch_flags);  \
        else launch_csr_cfg<B, V, false, ADD>(p, grid, x, y, w, pwy, pyy, flag, gen, c.remap | g_launch_flags);      \
        return;                                                                                \
    }
    CFG(256, 2) CFG(256, 4) CFG(256, 8) CFG(512, 2) CFG(512, 4) CFG(512, 8) CFG(1024, 2) CFG(1024, 4)
#undef CFG
    launch_csr_cfg<256, 2, true, ADD>(p, grid, x, y, w, pwy, pyy, flag, gen, c.remap | g_launch_flags);
}

template <bool ADD>
static void launch_csr_rl(const Part &p, int grid, const double *x, double *y, const double *w,
                          double *pwy, double *pyy, const int *flag, int gen)
{
    hipStream_t st = g_rt.stream;
    const int remap = (spmv_cfg().remap == 2 ? 2 : 1) | g_launch_flags;
#define L(DW, DY)                                                                                       \
    hipLaunchKernelGGL((k_csr_rl<ADD, DW, DY>), dim3(grid), dim3(256), 0, st, p.n, p.rowptr, p.col, p.val, x, y, w, \
                       pwy, pyy, flag, gen, remap)
    if (w && pyy) L(true, true);
    else if (w) L(true, false);
    else if (pyy) L(false, true);
    else L(false, false);
#undef L
}

template <bool ADD>
static void launch_csr_do(const Part &p, int grid, const double *x, double *y, const double *w,
                          double *pwy, double *pyy, const int *flag, int gen)
{
    const SpmvCfg &c = spmv_cfg();
    hipStream_t st = g_rt.stream;
    const int tile = do_tile_for(p);
    const bool dict = use_offset_dict(p);
#define L(B, T, DW, DY)                                                                                 \
    do {                                                                                                \
        if (dict)                                                                                       \
            hipLaunchKernelGGL((k_csr_do<B, T, 1, ADD, DW, DY>), dim3(grid), dim3(B), 0, st, p.n, p.rowptr, \
                               p.code, p.dict, p.val, x, y, w, pwy, pyy, flag, gen, c.remap | g_launch_flags); \
        else                                                                                            \
            hipLaunchKernelGGL((k_csr_do<B, T, 4, ADD, DW, DY>), dim3(grid), dim3(B), 0, st, p.n, p.rowptr, \
                               reinterpret_cast<const uint8_t *>(p.col), p.dict, p.val, x, y, w, pwy, pyy, flag, \
                               gen, c.remap | g_launch_flags);                                          \
    } while (0)
#define LV(B, T)                                \
    if (c.block == B && tile == T) {            \
        if (w && pyy) L(B, T, true, true);      \
        else if (w) L(B, T, true, false);       \
        else if (pyy) L(B, T, false, true);     \
        else L(B, T, false, false);             \
        return;                                 \
    }
    SGM_DO_VARIANTS(LV)
#undef LV
#undef L
}

#define SGM_SL_WIDTHS(X) X(3) X(5) X(7) X(8)
template <bool ADD>
static void launch_csr_sl(const Part &p, int grid, const double *x, double *y, const double *w,
                          double *pwy, double *pyy, const int *flag, int gen)
{
    const SpmvCfg &c = spmv_cfg();
    hipStream_t st = g_rt.stream;
    // XCD-block-cyclic slices (G = 32) below 32768 slices, plain round-robin on the 8192 grids above that
    // (measured: 300^3 363 vs 358 us) and wherever the grid is not a multiple of 8 G; SGM_SPMV_CFG's
    // remap field overrides (0 = round-robin, 3/4/5 = G 8/2/32)
    int mode = c.remap == 1 ? ((grid <= kMaxGrid / 2 || (int64_t)p.n < (int64_t)32768 * kSlRows) ? 5 : 0) : (c.remap >= 3 ? c.remap : 0);
    if (mode >= 3 && grid % (8 * (mode == 3 ? 8 : mode == 4 ? 2 : mode == 6 ? 64 : mode == 7 ? 128 : 32)) != 0) mode = 0;
#define L(WW, DW, DY)                                                                                   \
    hipLaunchKernelGGL((k_csr_sl<WW, ADD, DW, DY>), dim3(grid), dim3(256), 0, st, p.n, p.scode, p.dict, \
                       p.sval, x, y, w, pwy, pyy, flag, gen, mode | g_launch_flags, p.run_sched, p.run_iters)
#define LV(WW)                                \
    if (p.sw == WW) {                         \
        if (w && pyy) L(WW, true, true);      \
        else if (w) L(WW, true, false);       \
        else if (pyy) L(WW, false, true);     \
        else L(WW, false, false);             \
        return;                               \
    }
    SGM_SL_WIDTHS(LV)
#undef LV
#undef L
}

#define SGM_SL32_WIDTHS(X) X(3) X(5) X(7) X(8) X(12) X(16) X(20) X(24) X(28) X(32)
template <bool ADD>
static void launch_csr_sl32(const Part &p, int grid, const double *x, double *y, const double *w,
                            double *pwy, double *pyy, const int *flag, int gen)
{
    hipStream_t st = g_rt.stream;
    const SpmvCfg &c = spmv_cfg();
    int mode = c.remap == 1 ? ((grid <= kMaxGrid / 2 || (int64_t)p.n < (int64_t)32768 * kSlRows) ? 5 : 0) : (c.remap >= 3 ? c.remap : 0);
    if (mode >= 3 && grid % (8 * (mode == 3 ? 8 : mode == 4 ? 2 : mode == 6 ? 64 : mode == 7 ? 128 : 32)) != 0) mode = 0;
#define L(WW, DW, DY)                                                                                     \
    hipLaunchKernelGGL((k_csr_sl32<WW, ADD, DW, DY>), dim3(grid), dim3(256), 0, st, p.n, p.scol, p.sval, x, y, \
                       w, pwy, pyy, flag, gen, mode | g_launch_flags)
#define LV(WW)                                \
    if (p.sw == WW) {                         \
        if (w && pyy) L(WW, true, true);      \
        else if (w) L(WW, true, false);       \
        else if (pyy) L(WW, false, true);     \
        else L(WW, false, false);             \
        return;                               \
    }
    SGM_SL32_WIDTHS(LV)
#undef LV
#undef L
}

template <bool ADD>
static void launch_csr_sell(const Part &p, int grid, const double *x, double *y, const double *w,
                            double *pwy, double *pyy, const int *flag, int gen)
{
    hipStream_t st = g_rt.stream;
    const SpmvCfg &c = spmv_cfg();
    int mode = c.remap == 1 ? ((grid <= kMaxGrid / 2 || (int64_t)p.n < (int64_t)32768 * kSlRows) ? 5 : 0) : (c.remap >= 3 ? c.remap : 0);
    if (mode >= 3 && grid % (8 * (mode == 3 ? 8 : mode == 4 ? 2 : mode == 6 ? 64 : mode == 7 ? 128 : 32)) != 0) mode = 0;
#define L(DW, DY)                                                                                                    \
    hipLaunchKernelGGL((k_csr_sell<ADD, DW, DY>), dim3(grid), dim3(256), 0, st, p.n, (const int64_t *)p.sl_off,       \
                       (const uint16_t *)p.sl_perm, (const int32_t *)p.sl_col, (const double *)p.sl_val, x, y, w, pwy, pyy, flag, \
                       gen, mode | g_launch_flags)
#define LXG(DW, DY, GG)                                                                                              \
    do {                                                                                                             \
        static size_t attr = 0;                                                                                      \
        if (attr < lds) { (void)hipFuncSetAttribute((const void *)k_csr_sell<ADD, DW, DY, true, GG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = lds; } \
        hipLaunchKernelGGL((k_csr_sell<ADD, DW, DY, true, GG>), dim3(grid), dim3(256 * GG), lds, st, p.n, (const int64_t *)p.sl_off, \
                           (const uint16_t *)p.sl_perm, (const int32_t *)p.sl_col, (const double *)p.sl_val, x, y, w, pwy, pyy, flag, \
                           gen, mode | g_launch_flags, (const int32_t *)p.sl_win0, p.sl_span, (int32_t)p.xlen());     \
    } while (0)
#define LX(DW, DY) do { if (p.sl_gs == 2) LXG(DW, DY, 2); else LXG(DW, DY, 1); } while (0)
    if (p.sl_win0 && p.opt.csr_xwindow) {        // banded: the window of x of one (two) slice(s) through LDS
        const size_t lds = (size_t)p.sl_span * 8;
        if (w && pyy) LX(true, true);
        else if (w) LX(true, false);
        else if (pyy) LX(false, true);
        else LX(false, false);
        return;
    }
    if (w && pyy) L(true, true);
    else if (w) L(true, false);
    else if (pyy) L(false, true);
    else L(false, false);
#undef LX
#undef LXG
#undef L
}

template <bool ADD>
static void launch_csr_slb(const Part &p, int grid, const double *x, double *y, const double *w,
                           double *pwy, double *pyy, const int *flag, int gen)
{
    hipStream_t st = g_rt.stream;
    const SpmvCfg &c = spmv_cfg();
    int mode = c.remap == 1 ? ((grid <= kMaxGrid / 2 || (int64_t)p.n < (int64_t)32768 * kSlRows) ? 5 : 0) : (c.remap >= 3 ? c.remap : 0);
    if (mode >= 3 && grid % (8 * (mode == 3 ? 8 : mode == 4 ? 2 : mode == 6 ? 64 : mode == 7 ? 128 : 32)) != 0) mode = 0;
#define L(WW, DW, DY)                                                                                     \
    hipLaunchKernelGGL((k_csr_slb<WW, ADD, DW, DY>), dim3(grid), dim3(256), 0, st, p.n, p.sbcode, p.dict, p.sval, x, y, \
                       w, pwy, pyy, flag, gen, mode | g_launch_flags, p.run_sched, p.run_iters)
#define LV(WW)                                \
    if (p.sw == WW) {                         \
        if (w && pyy) L(WW, true, true);      \
        else if (w) L(WW, true, false);       \
        else if (pyy) L(WW, false, true);     \
        else L(WW, false, false);             \
        return;                               \
    }
    SGM_SLB_WIDTHS(LV)
#undef LV
#undef L
}

// slots in flight per lane, nontemporal matrix loads, grid cap
struct EllCfg { int u = 8, nt = 1, grid = 2048; };
static EllCfg &ell_cfg()
{
    static EllCfg c;
    static bool init = false;
    if (!init) {
        init = true;
        if (c.grid > kMaxGrid) c.grid = kMaxGrid;
    }
    return c;
}
static bool use_sliced_ell(const Part &p) { return p.ecol && p.scode && p.opt.csr_sliced && p.opt.ell_offset_dict; }
int ell_grid(const Part &p)
{
    if (use_ell_colblock(p)) return ell_colblock_grid(p);
    if (use_sliced_ell(p)) {           // k_csr_sl: 512-row slices round-robin over <= kMaxGrid workgroups
        const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
        return (int)std::max<int64_t>(1, std::min<int64_t>(nsl, nsl >= 32768 ? kMaxGrid : kMaxGrid / 2));
    }
    int64_t g = ((int64_t)p.n + kBlock - 1) / kBlock;
    int64_t cap = ell_cfg().grid;
    // k_ell_do<16> holds 16 values + 16 x entries per lane: 77-88 VGPRs, 5 waves/SIMD with the fused
    // dots -- a grid-stride launch sized for 8 resident workgroups per CU would run a second round
    if (p.ecode && p.opt.ell_offset_dict && p.emdp == 16) cap = std::min<int64_t>(cap, (int64_t)5 * g_rt.num_cu);
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, cap));
}

template <bool ADD>
static void launch_ell(const Part &p, int grid, const double *x, double *y, const double *w,
                       double *pwy, double *pyy, const int *flag, int gen)
{
    if (use_ell_colblock(p)) {         // random columns: products through LDS-resident x blocks, then ordered row sums
        (void)launch_ell_colblock(p, grid, x, y, ADD, (g_launch_flags & 256) != 0, w, pwy, pyy, flag, gen);
        return;
    }
    if (use_sliced_ell(p)) {           // structured ELLPACK in the sliced form: the CSR kernel as it is
        Part v;
        v.n = p.n; v.scode = p.scode; v.dict = p.dict; v.sval = p.sval; v.sw = p.sw;
        if (const SliceSched *ss = slice_sched(p, 0, p.n, grid)) { v.run_sched = ss->tab; v.run_iters = ss->iters; }
        launch_csr_sl<ADD>(v, grid, x, y, w, pwy, pyy, flag, gen);
        return;
    }
    hipStream_t st = g_rt.stream;
    const EllCfg &c = ell_cfg();
    if (p.ecode && p.opt.ell_offset_dict) {
#define LD(M, DW, DY)                                                                                   \
    hipLaunchKernelGGL((k_ell_do<M, ADD, DW, DY>), dim3(grid), dim3(kBlock), 0, st, p.n, p.max_d, p.ecode, \
                       p.dict, p.eval, x, y, w, pwy, pyy, flag, gen, g_launch_flags & 256)
#define LDV(M)                                   \
    {                                            \
        if (w && pyy) LD(M, true, true);         \
        else if (w) LD(M, true, false);          \
        else if (pyy) LD(M, false, true);        \
        else LD(M, false, false);                \
    }
        if (p.emdp == 4) LDV(4) else if (p.emdp == 8) LDV(8) else LDV(16)
#undef LDV
#undef LD
        return;
    }
#define L(UU, NTT, DW, DY)                                                                          \
    hipLaunchKernelGGL((k_ell_spmv<UU, NTT, ADD, DW, DY>), dim3(grid), dim3(kBlock), 0, st, p.n,     \
                       p.max_d, p.ecol, p.eval, x, y, w, pwy, pyy, flag, gen, g_launch_flags & 256)
#define LV(UU, NTT)                              \
    {                                            \
        if (w && pyy) L(UU, NTT, true, true);    \
        else if (w) L(UU, NTT, true, false);     \
        else if (pyy) L(UU, NTT, false, true);   \
        else L(UU, NTT, false, false);           \
    }
    if (c.u >= 16) { if (c.nt) LV(16, true) else LV(16, false) }
    else if (c.u <= 4) { if (c.nt) LV(4, true) else LV(4, false) }
    else { if (c.nt) LV(8, true) else LV(8, false) }
#undef LV
#undef L
}

// Workgroups of one kernel variant that fit on a CU at once (occupancy API, cached).
// `v` is the TILE for the offset-dict kernel and VPT for the int32 kernel.
// The fused-dot variants of one (BLOCK, TILE, CW) family have the same occupancy as the plain kernel
// (checked at build time with -Rpass-analysis=kernel-resource-usage: 8 waves/SIMD for tiles <= 1536,
// 7 for the 1-byte-code kernel with larger tiles), so one grid serves every epilogue.
static int resident_per_cu(bool dict, int block, int v, int cw)
{
    static std::vector<std::pair<int, int>> cache;
    const int key = (dict ? 1 << 30 : 0) | (cw == 1 ? 1 << 29 : 0) | (block << 16) | v;
    for (auto &kv : cache)
        if (kv.first == key) return kv.second;
    const void *fn = nullptr;
#define PICK_DO(B, T)                                                                                   \
    if (dict && block == B && v == T)                                                                   \
        fn = cw == 1 ? (const void *)k_csr_do<B, T, 1, false, true, false> : (const void *)k_csr_do<B, T, 4, false, true, false>;
#define PICK_ST(B, V) if (!dict && block == B && v == V) fn = (const void *)k_csr_spmv<B, V, true, false, false, false>;
    SGM_DO_VARIANTS(PICK_DO)
    PICK_ST(256, 2) PICK_ST(256, 4) PICK_ST(256, 8) PICK_ST(512, 2) PICK_ST(512, 4) PICK_ST(512, 8) PICK_ST(1024, 2) PICK_ST(1024, 4)
#undef PICK_DO
#undef PICK_ST
    int nb = 0;
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block, 0) != hipSuccess || nb < 1)
        nb = 2048 / block;
    cache.emplace_back(key, nb);
    return nb;
}

// ---- slice schedule ---------------------------------------------------------------------------------
// A 3-D grid's rows reference x a whole plane away (offset +-D, D >> one slice).  With slices handed out round-robin or
// block-cyclic, the slices D rows apart -- which read the same x lines -- run on different XCDs, so every x line enters
// three L2s (464^3: slice s and s + 420.5 land 4 XCDs apart).  The schedule cuts the period D into NB bands (NB a multiple
// of 8, bands of about `slice_sched_band` slices); band(s) = floor(NB * frac((512 s + 256) / D)), XCD x walks bands
// x, x + 8, ... one after the other, each in ascending slice order: a slice and its +-D neighbours sit one band width apart
// in the SAME XCD's sequence, inside or next to the window of slices that XCD has in flight.  Workgroup b (XCD b % 8,
// the hardware's round-robin) takes positions b / 8, b / 8 + grid / 8, ... of its XCD's sequence: tab[it * grid + b].
// Only the ORDER of whole slices changes: every row is still summed by one lane in stored order.
static void slice_sched_table(int64_t nsl, int64_t period_rows, int grid, int band_slices, std::vector<int32_t> &tab, int &iters)
{
    const double P = (double)period_rows / kSlRows;
    const int NB = 8 * std::max(1, (int)std::ceil(P / (8.0 * std::max(1, band_slices))));
    std::vector<int32_t> band((size_t)nsl);
    std::vector<int64_t> cnt((size_t)NB + 1, 0);
    for (int64_t sl = 0; sl < nsl; ++sl) {
        const double t = ((double)sl * kSlRows + kSlRows / 2) / (double)period_rows;
        int b = (int)((t - std::floor(t)) * NB);
        b = std::min(std::max(b, 0), NB - 1);
        band[(size_t)sl] = b;
        ++cnt[(size_t)b + 1];
    }
    // XCD x's sequence = bands x, x + 8, ... end to end; start[b] = position of band b's first slice inside it
    std::vector<int64_t> start((size_t)NB, 0), len(8, 0);
    for (int x = 0; x < 8; ++x)
        for (int b = x; b < NB; b += 8) { start[(size_t)b] = len[x]; len[x] += cnt[(size_t)b + 1]; }
    const int64_t L = grid / 8;
    const int64_t longest = *std::max_element(len.begin(), len.end());
    iters = (int)((longest + L - 1) / L);
    tab.assign((size_t)iters * grid, -1);
    for (int64_t sl = 0; sl < nsl; ++sl) {
        const int b = band[(size_t)sl], x = b & 7;
        const int64_t q = start[(size_t)b]++;
        tab[(size_t)((q / L) * grid + (q % L) * 8 + x)] = (int32_t)sl;
    }
}

extern "C" int sgm_slice_sched_host(int64_t n_slices, int64_t period_rows, int32_t grid, int32_t band_slices,
                                    int32_t *tab_out, int64_t capacity, int32_t *iters_out)
{
    if (n_slices < 1 || n_slices > INT32_MAX || period_rows < 1 || grid < 8 || grid % 8 || !iters_out)
        return fail(SGM_ERR_BAD_ARG, "sgm_slice_sched_host: n_slices %lld, period %lld, grid %d (a multiple of 8)",
                    (long long)n_slices, (long long)period_rows, grid);
    std::vector<int32_t> tab;
    int iters = 0;
    slice_sched_table(n_slices, period_rows, grid, band_slices, tab, iters);
    *iters_out = iters;
    if (tab_out) {
        if (capacity < (int64_t)tab.size()) return fail(SGM_ERR_BAD_ARG, "sgm_slice_sched_host: capacity %lld < %zu", (long long)capacity, tab.size());
        memcpy(tab_out, tab.data(), tab.size() * sizeof(int32_t));
    }
    return SGM_OK;
}

// the schedule of one row range of a part (built and uploaded on first use), or null: no far offset, option off,
// too few slices for it to matter
static const SliceSched *slice_sched(const Part &p, int32_t lo, int32_t hi, int grid)
{
    if (!p.opt.slice_sched || p.sched_period < 32 * kSlRows || grid < 8 || grid % 8) return nullptr;
    const int band = p.opt.slice_sched == 1 ? 64 : p.opt.slice_sched;
    const int64_t nsl = ((int64_t)hi - lo + kSlRows - 1) / kSlRows;
    if (nsl < 2 * (int64_t)grid || 2 * (int64_t)p.sched_period > (int64_t)hi - lo) return nullptr;
    // (a schedule built for another band width is stale: the option may change between products)
    for (int i = 0; i < p.nsched; ++i)
        if (p.sched[i].band != band) { free_slice_sched(const_cast<Part &>(p)); break; }
    for (int i = 0; i < p.nsched; ++i)
        if (p.sched[i].lo == lo && p.sched[i].hi == hi && p.sched[i].grid == grid) return p.sched[i].tab ? &p.sched[i] : nullptr;
    if (p.nsched >= 3) return nullptr;
    SliceSched &ss = p.sched[p.nsched++];
    ss.lo = lo; ss.hi = hi; ss.grid = grid; ss.tab = nullptr; ss.band = band;
    std::vector<int32_t> tab;
    int iters = 0;
    slice_sched_table(nsl, p.sched_period, grid, band, tab, iters);
    int32_t *d = nullptr;
    if (dalloc(&d, tab.size()) != SGM_OK) return nullptr;
    if (hipMemcpy(d, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { dfree(d); return nullptr; }
    ss.tab = d; ss.iters = iters;
    return &ss;
}
static void free_slice_sched(Part &p)
{
    for (int i = 0; i < p.nsched; ++i) { dfree(p.sched[i].tab); p.sched[i] = SliceSched(); }
    p.nsched = 0;
}

static void launch_range(sgm_mat A, const Part &p, const RowRange &r, const double *x, double *y, bool add,
                         const double *w, double *pwy, double *pyy, const int *flag_done, int gen)
{
    // a row range is the same kernel on shifted pointers: rowptr entries stay absolute offsets
    // into val/col/code; the offset-dict kernel forms columns as row + offset, so x shifts too
    Part v;
    v.opt = p.opt;
    v.n = r.hi - r.lo;
    v.nnz = (int64_t)((double)p.nnz * v.n / (p.n > 0 ? p.n : 1));   // same row density => same tile choice as the full part
    v.n_halo = p.n_halo; v.ncol_own = p.ncol_own;
    v.rowptr = p.rowptr + r.lo; v.col = p.col; v.val = p.val; v.code = p.code; v.dict = p.dict;
    v.max_row = p.max_row;
    const bool sliced = use_sliced(p), slicedb = !sliced && use_slicedb(p), sliced32 = !sliced && !slicedb && use_sliced32(p);   // range starts are multiples of the 512-row slices
    if (sliced || slicedb)
        if (const SliceSched *ss = slice_sched(p, r.lo, r.hi, r.grid)) { v.run_sched = ss->tab; v.run_iters = ss->iters; }
    if (sliced) { v.sval = p.sval + (int64_t)r.lo * p.sw; v.scode = p.scode + r.lo; v.sw = p.sw; }
    if (slicedb) { v.sval = p.sval + (int64_t)r.lo * p.sw; v.sbcode = p.sbcode + (int64_t)r.lo * ((p.sw + 7) / 8 * 8); v.sw = p.sw; }
    if (sliced32) { v.sval = p.sval + (int64_t)r.lo * p.sw; v.scol = p.scol + (int64_t)r.lo * p.sw; v.sw = p.sw; }
    const bool sell = !sliced && !slicedb && !sliced32 && use_sell(p);
    if (sell) {         // chunk offsets are absolute (into sl_val / sl_col); the chunk table and the positions shift with the range
        v.sl_val = p.sl_val; v.sl_col = p.sl_col;
        v.sl_off = p.sl_off + r.lo / kSellChunk; v.sl_perm = p.sl_perm + r.lo;
        v.sl_gs = p.sl_gs;          // (windows exist on parts without halo columns only: their one range starts at row 0)
        v.sl_win0 = p.sl_win0 ? p.sl_win0 + r.lo / (kSlRows * p.sl_gs) : nullptr; v.sl_span = p.sl_span;
    }
    const bool dict = use_offset_dict(p);
    const double *xs = dict ? x + r.lo : x;
    double *ys = y + r.lo;
    const double *ws = w ? w + r.lo : nullptr;
    double *pw = pwy ? pwy + r.part_off : nullptr, *py = pyy ? pyy + r.part_off : nullptr;
    if (sliced) {
        if (add) launch_csr_sl<true>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
        else launch_csr_sl<false>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
    } else if (slicedb) {
        if (add) launch_csr_slb<true>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
        else launch_csr_slb<false>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
    } else if (sliced32) {            // absolute columns: x is not shifted
        if (add) launch_csr_sl32<true>(v, r.grid, x, ys, ws, pw, py, flag_done, gen);
        else launch_csr_sl32<false>(v, r.grid, x, ys, ws, pw, py, flag_done, gen);
    } else if (sell) {
        if (add) launch_csr_sell<true>(v, r.grid, x, ys, ws, pw, py, flag_done, gen);
        else launch_csr_sell<false>(v, r.grid, x, ys, ws, pw, py, flag_done, gen);
    } else if (use_row_owner(p)) {
        if (add) launch_csr_do<true>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
        else launch_csr_do<false>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
    } else if (use_row_lines(p)) {
        if (add) launch_csr_rl<true>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
        else launch_csr_rl<false>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
    } else {
        if (add) launch_csr<true>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
        else launch_csr<false>(v, r.grid, xs, ys, ws, pw, py, flag_done, gen);
    }
}

static int composite_spmv(sgm_mat A, const double *x, double *y, bool add, const SpmvDots *dots,
                          const int *flag_done, int *grid_out, int gen);

int spmv_parts(sgm_mat A, const double *const *x, double *const *y, bool add,
               const SpmvDots *dots, const int *flag_done, int *grid_out, int gen, bool chain, bool halo_ready)
{
    if (A->fmt == SGM_FMT_COMPOSITE) return composite_spmv(A, x[0], y[0], add, dots, flag_done, grid_out, gen);
    const size_t P = A->parts.size();
    // a kernel other than the resident form's was asked for (options) on a lean part: its arrays come back and stay.  Resolved
    // here, before anything is launched (an allocation failure is SGM_ERR_ALLOC, not a kernel on null arrays; the solvers'
    // first, uncaptured iterations have been through here before a graph capture starts)
    if (A->fmt == SGM_FMT_CSR)
        for (const Part &p : A->parts)
            if (p.lean && !(lean_sell(p) ? use_sell(p) : use_sliced(p))) SGM_TRY(csr_need_arrays(p));
    g_launch_flags = chain ? 256 : 0;
    bool exchange = false;
    if (A->distributed() && !halo_ready)
        for (const Part &p : A->parts) exchange = exchange || !p.nbrs.empty();
    if (exchange) {
        // halo exchange on the communication stream, overlapped with the interior rows
        if (!g_rt.comm_stream) {
            SGM_HIP(hipStreamCreateWithFlags(&g_rt.comm_stream, hipStreamNonBlocking));
            SGM_HIP(hipEventCreateWithFlags(&g_rt.ev_x_ready, hipEventDisableTiming));
            SGM_HIP(hipEventCreateWithFlags(&g_rt.ev_halo_done, hipEventDisableTiming));
        }
        SGM_HIP(hipEventRecord(g_rt.ev_x_ready, g_rt.stream));
        SGM_HIP(hipStreamWaitEvent(g_rt.comm_stream, g_rt.ev_x_ready, 0));
        prof_begin(PH_HALO, g_rt.comm_stream);           // gather kernels + grouped send / recv, post -> done
        SGM_TRY(halo_exchange(A, const_cast<double *const *>(x), g_rt.comm_stream));
        prof_end(PH_HALO, g_rt.comm_stream);
        SGM_HIP(hipEventRecord(g_rt.ev_halo_done, g_rt.comm_stream));
    }
    hipEvent_t ev_int_end = nullptr;
    if (!exchange && prof_on()) prof_begin(PH_INTERIOR, g_rt.stream);      // no exchange: the whole product counts as interior rows
    for (int pass = 0; pass < 2; ++pass) {          // pass 0: ranges that need no halo; pass 1: the rest
        if (exchange && prof_on()) {
            if (pass == 0) prof_begin(PH_INTERIOR, g_rt.stream);
            else { prof_end(PH_INTERIOR, g_rt.stream); ev_int_end = prof_event(g_rt.stream); }
        }
        if (pass == 1 && exchange) SGM_HIP(hipStreamWaitEvent(g_rt.stream, g_rt.ev_halo_done, 0));
        if (pass == 1 && exchange && prof_on()) {
            // interior kernels: [begin, ev_int_end]; what the stream then waits for the halo: [ev_int_end, now]
            hipEvent_t ev_b0 = prof_event(g_rt.stream);
            prof_span(PH_HALO_WAIT, ev_int_end, ev_b0);
            prof_begin(PH_BOUNDARY, g_rt.stream);
        }
        for (size_t ip = 0; ip < P; ++ip) {
            const Part &p = A->parts[ip];
            const double *w = dots && dots->w ? dots->w[ip] : nullptr;
            double *pwy = dots && dots->part_wy ? dots->part_wy[ip] : nullptr;
            double *pyy = dots && dots->part_yy ? dots->part_yy[ip] : nullptr;
            if (A->fmt != SGM_FMT_CSR) {
                if (pass == 1) continue;
                const int grid = ell_grid(p);
                if (grid_out) *grid_out = grid;
                if (add) launch_ell<true>(p, grid, x[ip], y[ip], w, pwy, pyy, flag_done, gen);
                else launch_ell<false>(p, grid, x[ip], y[ip], w, pwy, pyy, flag_done, gen);
                continue;
            }
            if (use_ell_colblock(p)) {         // a CSR matrix with scattered columns (no halo: single-GPU parts only)
                if (pass == 1) continue;
                const int grid = ell_colblock_grid(p);
                if (grid_out) *grid_out = grid;
                SGM_TRY(launch_ell_colblock(p, grid, x[ip], y[ip], add, chain, w, pwy, pyy, flag_done, gen));
                continue;
            }
            RowRange r[3];
            // (halo_ready or not, the rows are cut the same way: the partial sums of a fused dot are laid out per range, and
            //  CG with dist_halo_fused must leave the bits of the exchanged-p path)
            const int nr = spmv_ranges(p, r, dots != nullptr);
            if (grid_out) *grid_out = spmv_grid(p);
            const bool split = nr > 1;              // r[0] is the interior range
            for (int k = 0; k < nr; ++k) {
                const bool needs_halo = !(split && k == 0) && p.n_halo > 0;
                if ((pass == 1) != needs_halo) continue;
                launch_range(A, p, r[k], x[ip], y[ip], add, w, pwy, pyy, flag_done, gen);
            }
        }
    }
    if (prof_on()) prof_end(exchange ? PH_BOUNDARY : PH_INTERIOR, g_rt.stream);
    g_launch_flags = 0;
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// partial sums of w.y and y.y (composite matrices cannot fuse them into one leaf kernel)
__global__ __launch_bounds__(kBlock) void k_dot_wy_yy(int64_t n, const double *__restrict__ w,
                                                      const double *__restrict__ y, double *part_wy,
                                                      double *part_yy, const int *flag_done, int gen)
{
    __shared__ double red[kBlock / 64];
    if (flag_done) { const int st = *flag_done; if (st && gen >= st) return; }
    double a = 0.0, b = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const double yi = y[i];
        if (w) a += w[i] * yi;
        b += yi * yi;
    }
    if (part_wy) { const double t = block_sum<kBlock>(a, red); if (threadIdx.x == 0) part_wy[blockIdx.x] = t; }
    if (part_yy) { const double t = block_sum<kBlock>(b, red); if (threadIdx.x == 0) part_yy[blockIdx.x] = t; }
}

// composite_matvec_add (sparse_matrix_composites.f90:1076-1099): for every row block, the
// column blocks in order, each `C%matvec_add(x(j1:j2), y(i1:i2))` -- one leaf launch per block
// on shifted pointers (the leaf kernels read x and write y with 8-byte accesses, so block
// boundaries need no alignment).
static int composite_spmv(sgm_mat A, const double *x, double *y, bool add, const SpmvDots *dots,
                          const int *flag_done, int *grid_out, int gen)
{
    const int nrb = (int)A->blk_row_ptr.size() - 1, ncb = (int)A->blk_col_ptr.size() - 1;
    // over distributed leaves (sgm_csr_create_dist_rect) the block offsets are LOCAL: x and y are the concatenation
    // of this rank's slices of the block vectors; a leaf reads [its slice of x_j | halo] out of its own staging vector
    const int64_t nloc = A->parts[0].n;
    if (!add) SGM_HIP(hipMemsetAsync(y, 0, (size_t)nloc * 8, g_rt.stream));     // y = 0 (matvec)
    for (int it = 0; it < nrb; ++it)
        for (int jt = 0; jt < ncb; ++jt) {
            sgm_mat C = A->blocks[(size_t)it * ncb + jt];
            if (!C) continue;
            const double *xs[1] = {x + A->blk_col_ptr[jt]};
            double *ys[1] = {y + A->blk_row_ptr[it]};
            if (C->comm) {
                Part &cp = C->parts[0];
                if (!cp.xext) SGM_TRY(dalloc(&cp.xext, (size_t)cp.xlen() + 2));
                SGM_HIP(hipMemcpyAsync(cp.xext, xs[0], (size_t)cp.ncol_own * 8, hipMemcpyDeviceToDevice, g_rt.stream));
                xs[0] = cp.xext;
            }
            SGM_TRY(spmv_parts(C, xs, ys, true, nullptr, flag_done, nullptr, gen, false));
        }
    if (dots && (dots->part_wy || dots->part_yy)) {
        const int grid = A->parts[0].dot_grid_override;
        hipLaunchKernelGGL(k_dot_wy_yy, dim3(grid), dim3(kBlock), 0, g_rt.stream, nloc,
                           dots->w ? dots->w[0] : nullptr, (const double *)y,
                           dots->part_wy ? dots->part_wy[0] : nullptr, dots->part_yy ? dots->part_yy[0] : nullptr,
                           flag_done, gen);
    }
    if (grid_out) *grid_out = A->parts[0].dot_grid_override;
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// Longest run of row blocks whose rows reference owned columns only (host index work at
// setup; ptr1/node1 are the part's 1-based local arrays).
void set_interior_range(Part &p, const int32_t *ptr1, const int32_t *node1)
{
    p.int_lo = 0;
    p.int_hi = 0;
    if (p.n_halo == 0 || p.n == 0) return;
    // ranges are cut at row-block boundaries of the kernel that will run them (512-row slices for the
    // sliced kernel; a multiple of the other kernels' 256-row blocks, so they can run the ranges too)
    const int B = p.sl_val ? kSellSigma : (p.scode || p.scol || p.sbcode) ? std::max(kSlRows, spmv_cfg().block) : spmv_cfg().block;      // (SELL: whole sort windows)
    const int32_t nb = (p.n + B - 1) / B;
    int32_t best_lo = 0, best_len = 0, run_lo = 0, run_len = 0;
    for (int32_t b = 0; b < nb; ++b) {
        const int32_t r0 = b * B, r1 = std::min(r0 + B, p.n);
        bool halo = false;
        for (int64_t k = ptr1[r0] - 1; k < ptr1[r1] - 1 && !halo; ++k) halo = node1[k] > p.ncol_own;
        if (halo) { run_len = 0; run_lo = b + 1; continue; }
        if (++run_len > best_len) { best_len = run_len; best_lo = run_lo; }
    }
    p.int_lo = best_lo * B;
    p.int_hi = std::min<int32_t>((best_lo + best_len) * B, p.n);
}

// ---- "csr_lean": the sliced form as the ONLY resident layout -----------------------------------------------------
// C2 kept 1.13 GB resident for a kernel that reads 0.44 GB of it: CSR-order values (400 MB), int32 columns (200), 1-byte
// codes (50) beside the sliced values + code words.  Slot u of a row in the sliced form IS the row's u-th stored entry
// (k_dict_encode / k_sl_pack), so the three arrays are a pure function of (rowptr, scode, dict, sval): they are released
// once the sliced form stands and rebuilt by k_sl_unpack for whoever reads them.
__global__ __launch_bounds__(256) void k_sl_unpack(int32_t n, int32_t W, const int32_t *__restrict__ rowptr, const uint32_t *__restrict__ scode,
                                                   const int32_t *__restrict__ dict, const double *__restrict__ sval,
                                                   int32_t *__restrict__ col, double *__restrict__ val, uint8_t *__restrict__ code)
{
    __shared__ int32_t dl[16];
    if (threadIdx.x < 16) dl[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t cw = scode[i];
        int32_t k = rowptr[i];
        const double *src = sval + ((int64_t)(i >> 9) * W) * kSlRows + (i & (kSlRows - 1));
        for (int u = 0; u < W; ++u) {
            const uint32_t c = (cw >> (4 * u)) & 15u;
            if (c == 15u) break;
            if (col) col[k] = i + dl[c];
            if (val) val[k] = src[(int64_t)u * kSlRows];
            if (code) code[k] = (uint8_t)c;
            ++k;
        }
    }
}
// the same out of the SELL-128-512 form: position q of chunk c holds row perm(c, q); slot u is its u-th stored entry
__global__ __launch_bounds__(256) void k_sell_unpack(int32_t n, const int32_t *__restrict__ rowptr, const int64_t *__restrict__ off,
                                                     const uint16_t *__restrict__ perm, const int32_t *__restrict__ scol,
                                                     const double *__restrict__ sval, int32_t *__restrict__ col, double *__restrict__ val)
{
    const int64_t nch = (((int64_t)n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nch; chunk += (int64_t)gridDim.x * 4) {
        const int64_t o0 = off[chunk];
        for (int h = 0; h < 2; ++h) {
            const int q = lane + 64 * h;
            const uint16_t pr = perm[chunk * kSellChunk + q];
            if (pr == 0xffffu) continue;
            const int64_t row = (chunk / (kSellSigma / kSellChunk)) * kSellSigma + pr;
            const int32_t k = rowptr[row], len = rowptr[row + 1] - k;
            for (int32_t u = 0; u < len; ++u) {
                if (col) col[k + u] = scol[o0 + (int64_t)u * kSellChunk + q];
                if (val) val[k + u] = sval[o0 + (int64_t)u * kSellChunk + q];
            }
        }
    }
}
static bool lean_sliced(const Part &p) { return p.scode && p.sval && p.dict && !p.ecol && !p.scol && !p.sbcode && p.sw <= 8; }
static bool lean_sell(const Part &p) { return p.sl_val && p.sl_col && !p.scode && !p.scol && !p.sbcode && !p.ecol; }
static bool lean_applies(const Part &p) { return p.opt.csr_lean && (lean_sliced(p) || lean_sell(p)); }
// after the sliced form has been built (or refreshed): keep only it
static void csr_go_lean(Part &p)
{
    if (!lean_applies(p)) return;
    dfree(p.val); dfree(p.col); dfree(p.code);
    p.val = nullptr; p.col = nullptr; p.code = nullptr;
    p.lean = true;
}
int csr_need_arrays(const Part &cp)
{
    Part &p = const_cast<Part &>(cp);
    if (!p.lean || (p.val && p.col && (p.code || lean_sell(p)))) return SGM_OK;
    hipStream_t st = g_rt.stream;
    const int64_t nnz = p.nnz;
    const bool mk_val = !p.val, mk_col = !p.col, mk_code = !p.code;
    if (mk_col) { SGM_TRY(dalloc(&p.col, (size_t)nnz + 4)); SGM_HIP(hipMemsetAsync(p.col + nnz, 0, 4 * sizeof(int32_t), st)); }
    if (mk_val) { SGM_TRY(dalloc(&p.val, (size_t)nnz + 2)); SGM_HIP(hipMemsetAsync(p.val + nnz, 0, 2 * sizeof(double), st)); }
    if (lean_sell(p)) {           // (no dictionary: no byte codes to bring back)
        const int64_t nch = (((int64_t)p.n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
        if (p.n > 0 && (mk_col || mk_val))
            hipLaunchKernelGGL(k_sell_unpack, dim3((unsigned)std::min<int64_t>((nch + 3) / 4, 65536)), dim3(256), 0, st, p.n,
                               (const int32_t *)p.rowptr, (const int64_t *)p.sl_off, (const uint16_t *)p.sl_perm, (const int32_t *)p.sl_col,
                               (const double *)p.sl_val, mk_col ? p.col : nullptr, mk_val ? p.val : nullptr);
        SGM_HIP(hipGetLastError());
        SGM_HIP(hipStreamSynchronize(st));     // (readers may use blocking copies, which do not order against this stream)
        return SGM_OK;
    }
    if (mk_code) { SGM_TRY(dalloc(&p.code, (size_t)nnz + 32)); SGM_HIP(hipMemsetAsync(p.code + nnz, 0, 32, st)); }
    if (p.n > 0)
        hipLaunchKernelGGL(k_sl_unpack, dim3((unsigned)std::min<int64_t>(((int64_t)p.n + 255) / 256, 4096)), dim3(256), 0, st, p.n, p.sw,
                           (const int32_t *)p.rowptr, (const uint32_t *)p.scode, (const int32_t *)p.dict, (const double *)p.sval,
                           mk_col ? p.col : nullptr, mk_val ? p.val : nullptr, mk_code ? p.code : nullptr);
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));         // (readers may use blocking copies, which do not order against this stream)
    return SGM_OK;
}
// a CSR-order value buffer to write new values into (they are then packed into the sliced form): allocated, not unpacked
static int lean_val_buffer(Part &p)
{
    if (!p.lean || p.val) return SGM_OK;
    SGM_TRY(dalloc(&p.val, (size_t)p.nnz + 2));
    SGM_HIP(hipMemsetAsync(p.val + p.nnz, 0, 2 * sizeof(double), g_rt.stream));
    return SGM_OK;
}
void csr_release_arrays(const Part &cp)
{
    Part &p = const_cast<Part &>(cp);
    if (!p.lean || !lean_applies(p)) return;         // (option switched off meanwhile: what was rebuilt stays)
    (void)hipStreamSynchronize(g_rt.stream);         // whoever asked for them has queued its reads on the stream
    dfree(p.val); dfree(p.col); dfree(p.code);
    p.val = nullptr; p.col = nullptr; p.code = nullptr;
}

// refresh the sliced copy of the values (no-op for parts without one)
int pack_sliced(Part &p)
{
    if (p.sl_val && p.n > 0) {          // SELL-128-512: the values of every slot again (columns stay)
        const int64_t nch = (((int64_t)p.n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
        hipLaunchKernelGGL(k_sell_fill, dim3((unsigned)std::min<int64_t>((nch + 3) / 4, 65536)), dim3(256), 0, g_rt.stream, p.n,
                           (const int32_t *)p.rowptr, (const int32_t *)nullptr, (const double *)p.val, (const int64_t *)p.sl_off,
                           (const uint16_t *)p.sl_perm, (int32_t *)nullptr, p.sl_val);
        SGM_HIP(hipGetLastError());
    }
    if ((!p.scode && !p.scol && !p.sbcode) || p.n == 0) return SGM_OK;
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
    if (p.ecol)
        hipLaunchKernelGGL(k_sl_pack_ell, dim3((unsigned)std::min<int64_t>(nsl, 65536)), dim3(256), 0, g_rt.stream, p.n, p.sw,
                           p.max_d, (const double *)p.eval, p.sval);
    else
        hipLaunchKernelGGL(k_sl_pack, dim3((unsigned)std::min<int64_t>(nsl, 65536)), dim3(256), 0, g_rt.stream, p.n, p.sw,
                           (const int32_t *)p.rowptr, (const double *)p.val, p.sval);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// Offset dictionary of a row block (host index work at setup): distinct (col - row) values in
// order of first appearance; gives up (p.code stays null) beyond 255 distinct offsets.
// ptr1/node1: optional 1-based host copies (otherwise the device arrays are read back).
__global__ void k_fill32(int64_t n, int32_t *a, int32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = v;
}

// ---- offset dictionary + sliced codes, built on the device ---------------------------------------
// pass 1: the set of distinct (col - row) offsets (open-addressing table of 1024 slots in global
// memory, atomicCAS insert; more than 255 live keys = overflow) and the longest row
constexpr int kDictSlots = 1024;
constexpr int32_t kDictEmpty = INT32_MIN;
__global__ __launch_bounds__(256) void k_dict_collect(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      int32_t *table, int *count, int *max_row)
{
    // Distinct offsets are collected per workgroup in an LDS hash table first and only its (few) entries go to the
    // global table at the end: every thread inserting its first row's offsets straight into the global table was
    // 2.6 M same-address atomics at n = 1e7 (14 ms of a 24 ms create).
    constexpr int kLocal = 512;
    __shared__ int32_t ltab[kLocal];
    __shared__ int lcount;
    for (int t = threadIdx.x; t < kLocal; t += 256) ltab[t] = kDictEmpty;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    int mr = 0;
    int32_t mine[8];                        // the offsets this lane met last (stencil rows repeat them)
#pragma unroll
    for (int t = 0; t < 8; ++t) mine[t] = kDictEmpty;
    int next = 0;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t s0 = rowptr[i], e = rowptr[i + 1];
        mr = max(mr, e - s0);
        for (int32_t k = s0; k < e; ++k) {
            const int32_t off = col[k] - i;
            bool known = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) known = known || mine[t] == off;
            if (known) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (t == next) mine[t] = off;
            next = (next + 1) & 7;
            if (*(volatile int *)&lcount > 255) break;                        // this workgroup alone overflows the dictionary
            uint32_t h = ((uint32_t)off * 2654435761u) >> 23;                // 9 bits
            for (int probe = 0; probe < kLocal; ++probe) {
                const int32_t prev = atomicCAS(&ltab[h], kDictEmpty, off);
                if (prev == kDictEmpty) { atomicAdd(&lcount, 1); break; }
                if (prev == off) break;
                h = (h + 1) & (kLocal - 1);
            }
        }
    }
    __syncthreads();
    if (lcount > 255) {
        if (threadIdx.x == 0) atomicAdd(count, 256);                          // overflow: more than 255 distinct offsets
    } else {
        for (int t = threadIdx.x; t < kLocal; t += 256) {
            const int32_t off = ltab[t];
            if (off == kDictEmpty || *(volatile int *)count > 255) continue;
            uint32_t h = ((uint32_t)off * 2654435761u) >> 22;                // 10 bits
            for (int probe = 0; probe < kDictSlots; ++probe) {
                const int32_t prev = atomicCAS(&table[h], kDictEmpty, off);
                if (prev == kDictEmpty) { atomicAdd(count, 1); break; }
                if (prev == off) break;
                h = (h + 1) & (kDictSlots - 1);
            }
        }
    }
    // one atomic per workgroup for the longest row
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mr = max(mr, __shfl_xor(mr, off, 64));
    __shared__ int wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mr;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_row, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}
// pass 2: 1-byte code of every entry (binary search in the sorted dictionary, held in LDS) and, when
// asked for, the row's word of 4-bit codes (15 = no entry)
__global__ __launch_bounds__(256) void k_dict_encode(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ dict, int ndict, uint8_t *__restrict__ code,
                                                     uint32_t *__restrict__ scode)
{
    __shared__ int32_t dl[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) dl[t] = t < ndict ? dict[t] : INT32_MAX;
    __syncthreads();
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t s0 = rowptr[i], e = rowptr[i + 1];
        uint32_t cw = 0xffffffffu;
        for (int32_t k = s0; k < e; ++k) {
            const int32_t off = col[k] - i;
            int lo = 0, hi = ndict - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (dl[mid] < off) lo = mid + 1; else hi = mid; }
            code[k] = (uint8_t)lo;
            if (scode && k - s0 < 8) cw = (cw & ~(15u << (4 * (k - s0)))) | ((uint32_t)lo << (4 * (k - s0)));
        }
        if (scode) scode[i] = cw;
    }
}

// The far offset most rows carry (a 3-D grid's plane stride, in rows), for the slice schedule: the largest |offset| that
// at least a quarter of the rows of a 512-row sample from the middle of the part use.  `codes` = the sample's dictionary
// codes (any order; 255 and codes >= ndict are ignored), `rows` = rows sampled.
static int32_t far_offset_of_sample(const std::vector<uint8_t> &codes, int64_t rows, const std::vector<int32_t> &dict, int ndict)
{
    std::vector<int64_t> freq(256, 0);
    for (uint8_t c : codes) ++freq[c];
    int64_t far = 0;
    for (int c = 0; c < ndict && c < 255; ++c)
        if (4 * freq[(size_t)c] >= rows) far = std::max<int64_t>(far, std::llabs((long long)dict[(size_t)c]));
    return (int32_t)std::min<int64_t>(far, INT32_MAX);
}
// SELL-128-512 of a part (see k_csr_sell): built for matrices without an offset dictionary that the uniform sliced form
// does not take, when sorting the rows of a slice keeps the padding below 30 % (a handful of very long rows among short
// ones -- an arrow matrix -- would blow their chunks up: those matrices stay with the CSR kernels)
static void free_sell(Part &p)
{
    dfree(p.sl_val); dfree(p.sl_col); dfree(p.sl_perm); dfree(p.sl_off); dfree(p.sl_win0);
    p.sl_val = nullptr; p.sl_col = nullptr; p.sl_perm = nullptr; p.sl_off = nullptr; p.sl_total = 0;
    p.sl_win0 = nullptr; p.sl_span = 0;
}
static int build_sell(Part &p)
{
    free_sell(p);
    if (!p.opt.csr_sliced || !p.opt.csr_sell || p.ecol || p.n < 1 || p.nnz < 4 * (int64_t)p.n || p.max_row < 1) return SGM_OK;
    // rows of up to 48 entries stay with the row-owner kernel UNLESS the slices' windows of x fit the LDS (decided below): its
    // tiles hold consecutive rows, whose x gathers share more L1 lines than a sorted chunk's (banded 20..40 entries per row:
    // 719-753 us against 793-819 here without the window -- and 553 with it; from 33..64 on SELL wins either way:
    // 620-650 against 757-794, 64..128: 649 against 792-816, 150..300: 750 against 922)
    const bool short_rows = p.max_row <= 48 && p.opt.csr_sell < 2;
    if (short_rows && (!p.opt.csr_xwindow || p.n_halo != 0 || p.max_row < 8)) return SGM_OK;
    hipStream_t st = g_rt.stream;
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows, nch = nsl * (kSlRows / kSellChunk);
    SGM_TRY(dalloc(&p.sl_perm, (size_t)nsl * kSlRows));
    SGM_TRY(dalloc(&p.sl_off, (size_t)nch + 1));
    SGM_HIP(hipMemsetAsync(p.sl_off + nch, 0, sizeof(int64_t), st));
    const int64_t nwin = ((int64_t)p.n + kSellSigma - 1) / kSellSigma;
    hipLaunchKernelGGL(k_sell_sort, dim3((unsigned)std::min<int64_t>(nwin, 65536)), dim3(256), 0, st, p.n, (const int32_t *)p.rowptr,
                       p.sl_perm, p.sl_off);
    void *tmp = nullptr;
    size_t tb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, p.sl_off, p.sl_off, (int)(nch + 1), st);
    if (hipMalloc(&tmp, std::max<size_t>(tb, 16)) != hipSuccess) { free_sell(p); return fail(SGM_ERR_ALLOC, "SELL build: scan workspace"); }
    (void)hipcub::DeviceScan::ExclusiveSum(tmp, tb, p.sl_off, p.sl_off, (int)(nch + 1), st);
    int64_t total = 0;
    const hipError_t e1 = hipMemcpyAsync(&total, p.sl_off + nch, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e1 != hipSuccess || e2 != hipSuccess) { free_sell(p); return fail(SGM_ERR_HIP, "SELL build: scan failed"); }
    if (total <= 0 || (double)total > 1.30 * (double)p.nnz) { free_sell(p); return SGM_OK; }
    p.sl_total = total;
    int rc = dalloc(&p.sl_val, (size_t)total + 2);
    if (rc == SGM_OK) rc = dalloc(&p.sl_col, (size_t)total + 2);
    if (rc != SGM_OK) { free_sell(p); return rc; }
    hipLaunchKernelGGL(k_sell_fill, dim3((unsigned)std::min<int64_t>((nch + 3) / 4, 65536)), dim3(256), 0, st, p.n,
                       (const int32_t *)p.rowptr, (const int32_t *)p.col, (const double *)p.val, (const int64_t *)p.sl_off,
                       (const uint16_t *)p.sl_perm, p.sl_col, p.sl_val);
    SGM_HIP(hipGetLastError());
    // the windows of x the slices gather from: where every one of them fits the LDS the kernel stages it there (XW).  Parts
    // with halo columns are left out (their windows span the halo region, and their row ranges are cut by slices).
    p.sl_gs = 1;
    if (p.n_halo == 0) {
        int32_t *mx = nullptr;
        SGM_TRY(dalloc(&p.sl_win0, (size_t)nsl));
        SGM_TRY(dalloc(&mx, 1));
        struct Tmp { int32_t *&a; ~Tmp() { dfree(a); } } tmpmx{mx};
        constexpr size_t kLdsCap = (size_t)152 * 1024;
        auto windows = [&](int gs, int32_t *span_out) -> int {
            SGM_HIP(hipMemsetAsync(mx, 0, 4, st));
            hipLaunchKernelGGL(k_sell_window, dim3((unsigned)std::min<int64_t>(nsl, 65536)), dim3(256), 0, st, nsl, gs, (const int64_t *)p.sl_off,
                               (const int32_t *)p.sl_col, p.sl_win0, mx);
            int32_t span = 0;
            SGM_HIP(hipMemcpyAsync(&span, mx, 4, hipMemcpyDeviceToHost, st));
            SGM_HIP(hipStreamSynchronize(st));
            *span_out = (span + 2) & ~1;                            // (even, and one spare entry for an odd tail)
            return SGM_OK;
        };
        int32_t span = 0;
        SGM_TRY(windows(1, &span));
        // a window beyond 72 KiB leaves room for ONE workgroup per CU: let it be a 512-thread one over two slices
        if ((size_t)span * 8 > (size_t)72 * 1024) {
            int32_t span2 = 0;
            SGM_TRY(windows(2, &span2));
            if ((size_t)span2 * 8 <= kLdsCap) { span = span2; p.sl_gs = 2; }
            else if ((size_t)span * 8 <= kLdsCap) SGM_TRY(windows(1, &span));       // (back to one slice per window)
        }
        // worth it when the window is re-used: a slice's rows must reference its columns several times over
        if (span < 2 || (size_t)span * 8 > kLdsCap || (double)span * (double)((nsl + p.sl_gs - 1) / p.sl_gs) > 0.5 * (double)total) {
            dfree(p.sl_win0); p.sl_win0 = nullptr; span = 0; p.sl_gs = 1;
        }
        p.sl_span = span;
    }
    if (short_rows && !p.sl_win0) { free_sell(p); return SGM_OK; }        // (short rows without a window: the row-owner kernel)
    SGM_HIP(hipStreamSynchronize(st));
    csr_go_lean(p);
    return SGM_OK;
}

static int detect_sched_period_csr(Part &p, const std::vector<int32_t> &dict)
{
    p.sched_period = 0;
    if (!p.code || p.n < 64 * kSlRows) return SGM_OK;
    const int32_t R = kSlRows, mid = (p.n / 2) / kSlRows * kSlRows;
    std::vector<int32_t> rp((size_t)R + 1);
    SGM_HIP(hipMemcpy(rp.data(), p.rowptr + mid, ((size_t)R + 1) * 4, hipMemcpyDeviceToHost));
    const int64_t cnt = (int64_t)rp[(size_t)R] - rp[0];
    if (cnt <= 0) return SGM_OK;
    std::vector<uint8_t> codes((size_t)cnt);
    SGM_HIP(hipMemcpy(codes.data(), p.code + rp[0], (size_t)cnt, hipMemcpyDeviceToHost));
    p.sched_period = far_offset_of_sample(codes, R, dict, p.ndict);
    return SGM_OK;
}

// Offset dictionary of a row block (index work at setup, on the device): the distinct (col - row)
// values in ascending order; gives up (p.code stays null) beyond 255 distinct offsets.
static int build_offset_dict(Part &p, const int32_t *, const int32_t *)
{
    const int64_t nnz = p.nnz;
    const int32_t n = p.n;
    if (nnz == 0 || n == 0) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *table = nullptr;
    int *cnt = nullptr;             // [0] distinct offsets, [1] longest row
    SGM_TRY(dalloc(&table, (size_t)kDictSlots));
    SGM_TRY(dalloc(&cnt, 2));
    hipLaunchKernelGGL(k_fill32, dim3(kDictSlots / 256), dim3(256), 0, st, (int64_t)kDictSlots, table, kDictEmpty);
    SGM_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(int), st));
    const int grid = (int)std::min<int64_t>(((int64_t)n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_dict_collect, dim3(grid), dim3(256), 0, st, n, (const int32_t *)p.rowptr, (const int32_t *)p.col, table,
                       cnt, cnt + 1);
    std::vector<int32_t> htab(kDictSlots);
    int hcnt[2] = {0, 0};
    SGM_HIP(hipMemcpyAsync(htab.data(), table, kDictSlots * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(hcnt, cnt, sizeof hcnt, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    dfree(table); dfree(cnt);
    p.max_row = hcnt[1];
    // without a dictionary (option off at creation, or more than 255 offsets): try the int32 sliced form
    auto sliced32 = [&]() -> int {
        // scattered columns (x far beyond the L2s' reach, no offset structure): the column-blocked two-phase form the ELLPACK
        // matrices of that kind get (sgm_ellcb.hip) -- every gather an LDS access; same products, same order of additions
        SGM_TRY(build_ell_colblock(p));
        if (p.cb_P) return SGM_OK;
        const int W = p.max_row <= 3 ? 3 : p.max_row <= 5 ? 5 : p.max_row <= 7 ? 7 : p.max_row <= 8 ? 8 : p.max_row <= 12 ? 12
                    : p.max_row <= 16 ? 16 : p.max_row <= 20 ? 20 : p.max_row <= 24 ? 24 : p.max_row <= 28 ? 28 : 32;
        if (!p.opt.csr_sliced || p.max_row < 1 || p.max_row > 32 || (double)W * n > 1.25 * (double)nnz) return build_sell(p);
        const size_t rows_padded = ((size_t)n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.scol, rows_padded * W));
        SGM_TRY(dalloc(&p.sval, rows_padded * W));
        p.sw = W;
        hipLaunchKernelGGL(k_sl_pack_cols, dim3((unsigned)std::min<size_t>(rows_padded / kSlRows, 65536)), dim3(256), 0, g_rt.stream,
                           n, W, (const int32_t *)p.rowptr, (const int32_t *)p.col, p.scol);
        SGM_HIP(hipGetLastError());
        return pack_sliced(p);
    };
    if (!p.opt.csr_offset_dict || hcnt[0] > 255) return sliced32();
    std::vector<int32_t> dict;
    for (int32_t v : htab) if (v != kDictEmpty) dict.push_back(v);
    std::sort(dict.begin(), dict.end());
    p.ndict = (int32_t)dict.size();
    p.dict_reach = 0;
    for (int32_t v : dict) p.dict_reach = std::max(p.dict_reach, v < 0 ? -v : v);
    dict.resize(256, 0);
    SGM_TRY(dalloc(&p.code, (size_t)nnz + 32));
    SGM_TRY(dalloc(&p.dict, (size_t)256));
    SGM_HIP(hipMemcpyAsync(p.dict, dict.data(), 256 * 4, hipMemcpyHostToDevice, st));
    SGM_HIP(hipMemsetAsync(p.code + nnz, 0, 32, st));
    // sliced form: short rows, few offsets, little padding
    const int W = p.max_row <= 3 ? 3 : p.max_row <= 5 ? 5 : p.max_row <= 7 ? 7 : 8;
    const bool sliced = p.opt.csr_sliced && p.ndict <= 15 && p.max_row >= 1 && p.max_row <= 8 && (double)W * n <= 1.25 * (double)nnz;
    size_t rows_padded = 0;
    if (sliced) {
        rows_padded = ((size_t)n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.scode, rows_padded));
        SGM_TRY(dalloc(&p.sval, rows_padded * W));
        hipLaunchKernelGGL(k_fill32, dim3(vec_grid(rows_padded)), dim3(256), 0, st, (int64_t)rows_padded,
                           reinterpret_cast<int32_t *>(p.scode), (int32_t)-1);
        p.sw = W;
    }
    hipLaunchKernelGGL(k_dict_encode, dim3(grid), dim3(256), 0, st, n, (const int32_t *)p.rowptr, (const int32_t *)p.col,
                       (const int32_t *)p.dict, p.ndict, p.code, sliced ? p.scode : nullptr);
    SGM_HIP(hipGetLastError());
    if (sliced) SGM_TRY(pack_sliced(p));
    // longer rows (9..32 entries), <= 255 offsets, little padding: slot-major slices with 1-byte codes (k_csr_slb)
    int Wb = 0;                                               // value slots: the smallest instantiated width that holds the longest row
#define PICK(WW) if (!Wb && p.max_row <= WW) Wb = WW;
    SGM_SLB_WIDTHS(PICK)
#undef PICK
    const int Wc = (Wb + 7) / 8 * 8;                          // code bytes per row in eights
    if (!sliced && p.opt.csr_sliced && Wb && p.ndict <= 255 && p.max_row > 8 && (double)Wb * n <= 1.35 * (double)nnz) {
        rows_padded = ((size_t)n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.sbcode, rows_padded * Wc));
        SGM_TRY(dalloc(&p.sval, rows_padded * Wb));
        p.sw = Wb;
        hipLaunchKernelGGL(k_slb_pack_codes, dim3((unsigned)std::min<size_t>(rows_padded / kSlRows, 65536)), dim3(256), 0, st, n, Wb,
                           (const int32_t *)p.rowptr, (const uint8_t *)p.code, p.sbcode);
        SGM_HIP(hipGetLastError());
        SGM_TRY(pack_sliced(p));
    }
    SGM_HIP(hipStreamSynchronize(st));       // `dict` (host staging of the upload) goes out of scope
    if (p.scode || p.sbcode) SGM_TRY(detect_sched_period_csr(p, dict));
    csr_go_lean(p);
    return SGM_OK;
}

// ELLPACK twin of build_offset_dict: codes for ALL max_d slots of every row (padding slots
// carry the last neighbour, so their offsets are already in the dictionary), row-major with
// the row padded to 4 / 8 / 16 bytes.  Skipped for max_d > 16 or > 255 distinct offsets.
// (device kernels of the ELLPACK twin: offsets of ALL max_d slots, slot-major columns)
__global__ __launch_bounds__(256) void k_ell_dict_collect(int32_t n, int32_t max_d, const int32_t *__restrict__ ecol, int32_t *table,
                                                          int *count)
{
    // per-workgroup LDS table first, its entries to the global table at the end (see k_dict_collect)
    constexpr int kLocal = 512;
    __shared__ int32_t ltab[kLocal];
    __shared__ int lcount;
    for (int t = threadIdx.x; t < kLocal; t += 256) ltab[t] = kDictEmpty;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    int32_t mine[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) mine[t] = kDictEmpty;
    int next = 0;
    bool over = false;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n && !over; i += gridDim.x * blockDim.x)
        for (int32_t k = 0; k < max_d; ++k) {
            const int32_t off = ecol[(int64_t)k * n + i] - i;
            bool known = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) known = known || mine[t] == off;
            if (known) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (t == next) mine[t] = off;
            next = (next + 1) & 7;
            if (*(volatile int *)&lcount > 255 || *(volatile int *)count > 255) { over = true; break; }
            uint32_t h = ((uint32_t)off * 2654435761u) >> 23;
            for (int probe = 0; probe < kLocal; ++probe) {
                const int32_t prev = atomicCAS(&ltab[h], kDictEmpty, off);
                if (prev == kDictEmpty) { atomicAdd(&lcount, 1); break; }
                if (prev == off) break;
                h = (h + 1) & (kLocal - 1);
            }
        }
    __syncthreads();
    if (lcount > 255) {
        if (threadIdx.x == 0) atomicAdd(count, 256);
        return;
    }
    for (int t = threadIdx.x; t < kLocal; t += 256) {
        const int32_t off = ltab[t];
        if (off == kDictEmpty || *(volatile int *)count > 255) continue;
        uint32_t h = ((uint32_t)off * 2654435761u) >> 22;
        for (int probe = 0; probe < kDictSlots; ++probe) {
            const int32_t prev = atomicCAS(&table[h], kDictEmpty, off);
            if (prev == kDictEmpty) { atomicAdd(count, 1); break; }
            if (prev == off) break;
            h = (h + 1) & (kDictSlots - 1);
        }
    }
}
__global__ __launch_bounds__(256) void k_ell_dict_encode(int32_t n, int32_t max_d, int32_t mdp, const int32_t *__restrict__ ecol,
                                                         const int32_t *__restrict__ dict, int ndict, uint8_t *__restrict__ ecode,
                                                         uint32_t *__restrict__ scode)
{
    __shared__ int32_t dl[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) dl[t] = t < ndict ? dict[t] : INT32_MAX;
    __syncthreads();
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t cw = 0xffffffffu;
        for (int32_t k = 0; k < max_d; ++k) {
            const int32_t off = ecol[(int64_t)k * n + i] - i;
            int lo = 0, hi = ndict - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (dl[mid] < off) lo = mid + 1; else hi = mid; }
            ecode[(int64_t)i * mdp + k] = (uint8_t)lo;
            if (scode && k < 8) cw = (cw & ~(15u << (4 * k))) | ((uint32_t)lo << (4 * k));
        }
        if (scode) scode[i] = cw;
    }
}

static int build_ell_offset_dict(Part &p)
{
    if (p.n == 0 || p.max_d == 0 || p.max_d > 16) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *table = nullptr;
    int *cnt = nullptr;
    SGM_TRY(dalloc(&table, (size_t)kDictSlots));
    SGM_TRY(dalloc(&cnt, 1));
    hipLaunchKernelGGL(k_fill32, dim3(kDictSlots / 256), dim3(256), 0, st, (int64_t)kDictSlots, table, kDictEmpty);
    SGM_HIP(hipMemsetAsync(cnt, 0, sizeof(int), st));
    const int grid = (int)std::min<int64_t>(((int64_t)p.n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_ell_dict_collect, dim3(grid), dim3(256), 0, st, p.n, p.max_d, (const int32_t *)p.ecol, table, cnt);
    std::vector<int32_t> htab(kDictSlots);
    int hcnt = 0;
    SGM_HIP(hipMemcpyAsync(htab.data(), table, kDictSlots * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(&hcnt, cnt, sizeof hcnt, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    dfree(table); dfree(cnt);
    if (hcnt > 255) return SGM_OK;              // too many offsets: the int32 slot-major kernel
    std::vector<int32_t> dict;
    for (int32_t v : htab) if (v != kDictEmpty) dict.push_back(v);
    std::sort(dict.begin(), dict.end());
    const int ndict = (int)dict.size();
    p.dict_reach = 0;
    for (int32_t v : dict) p.dict_reach = std::max(p.dict_reach, v < 0 ? -v : v);
    dict.resize(256, 0);
    const int mdp = p.max_d <= 4 ? 4 : p.max_d <= 8 ? 8 : 16;
    const size_t code_bytes = (size_t)p.n * mdp + 16;
    SGM_TRY(dalloc(&p.ecode, code_bytes));
    if (!p.dict) SGM_TRY(dalloc(&p.dict, (size_t)256));
    SGM_HIP(hipMemsetAsync(p.ecode, 0, code_bytes, st));
    SGM_HIP(hipMemcpyAsync(p.dict, dict.data(), 256 * 4, hipMemcpyHostToDevice, st));
    p.emdp = mdp;
    // sliced form (see k_csr_sl): every one of the max_d slots is an entry (padding slots keep their
    // 0.0 * x(last neighbour) term, like the reference), so the CSR kernel applies as it is
    const bool sliced = p.opt.csr_sliced && ndict <= 15 && p.max_d >= 1 && p.max_d <= 8;
    if (sliced) {
        const int W = p.max_d <= 3 ? 3 : p.max_d <= 5 ? 5 : p.max_d <= 7 ? 7 : 8;
        const size_t rows_padded = ((size_t)p.n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.scode, rows_padded));
        SGM_TRY(dalloc(&p.sval, rows_padded * W));
        hipLaunchKernelGGL(k_fill32, dim3(vec_grid(rows_padded)), dim3(256), 0, st, (int64_t)rows_padded,
                           reinterpret_cast<int32_t *>(p.scode), (int32_t)-1);
        p.sw = W;
    }
    hipLaunchKernelGGL(k_ell_dict_encode, dim3(grid), dim3(256), 0, st, p.n, p.max_d, mdp, (const int32_t *)p.ecol,
                       (const int32_t *)p.dict, ndict, p.ecode, sliced ? p.scode : nullptr);
    SGM_HIP(hipGetLastError());
    if (sliced) SGM_TRY(pack_sliced(p));
    SGM_HIP(hipStreamSynchronize(st));
    p.sched_period = 0;
    if (sliced && p.n >= 64 * kSlRows) {       // the slice schedule's period, from a 512-row sample (see far_offset_of_sample)
        const int32_t mid = (p.n / 2) / kSlRows * kSlRows;
        std::vector<uint8_t> rows((size_t)kSlRows * mdp), codes;
        SGM_HIP(hipMemcpy(rows.data(), p.ecode + (size_t)mid * mdp, rows.size(), hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < kSlRows; ++i)
            for (int32_t k = 0; k < p.max_d; ++k) codes.push_back(rows[(size_t)i * mdp + k]);
        p.sched_period = far_offset_of_sample(codes, kSlRows, dict, ndict);
    }
    return SGM_OK;
}

// Upload one CSR row block.  ptr1 is 1-based local (n+1), node1 is 1-based and already
// renumbered to [owned | halo]; `where` says where the three arrays live.  With `validate` the
// index arrays are checked on the device as they are converted (k_check_ptr1, k_dec1_check_cols):
// a malformed pointer array is SGM_ERR_BAD_ARG, a pointer array that does not end at nnz or a
// column outside 1..ncol_own+n_halo is SGM_ERR_DIMS, each naming the first offending row --
// never a memory fault inside a later product.
int build_csr_part(Part &p, int32_t n, int32_t ncol_own, int32_t n_halo, int64_t nnz,
                   const int32_t *ptr1, const int32_t *node1, const double *val, int where, bool validate)
{
    p.n = n;
    p.ncol_own = ncol_own;
    p.n_halo = n_halo;
    p.nnz = nnz;
    SGM_TRY(dalloc(&p.rowptr, (size_t)n + 1));
    SGM_TRY(dalloc(&p.col, (size_t)nnz + 4));          // k_csr_rl reads whole 16-byte pieces of col
    SGM_TRY(dalloc(&p.val, (size_t)nnz + 2));
    hipStream_t st = g_rt.stream;
    const hipMemcpyKind kind = where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    unsigned long long *bad = nullptr;                 // validation verdict: {first bad row, first bad entry}
    if (validate) {
        SGM_TRY(dalloc(&bad, 2));
        SGM_HIP(hipMemsetAsync(bad, 0xff, 2 * sizeof(unsigned long long), st));
    }
    SGM_HIP(hipMemsetAsync(p.col + nnz, 0, 4 * sizeof(int32_t), st));
    SGM_HIP(hipMemsetAsync(p.val + nnz, 0, 2 * sizeof(double), st));
    SGM_HIP(hipMemcpyAsync(p.rowptr, ptr1, ((size_t)n + 1) * sizeof(int32_t), kind, st));
    if (nnz) {
        SGM_HIP(hipMemcpyAsync(p.col, node1, (size_t)nnz * sizeof(int32_t), kind, st));
        SGM_HIP(hipMemcpyAsync(p.val, val, (size_t)nnz * sizeof(double), kind, st));
    }
    if (validate)
        hipLaunchKernelGGL(k_check_ptr1, dim3(vec_grid(n + 1)), dim3(kBlock), 0, st, (const int32_t *)p.rowptr, (int64_t)n, nnz, bad);
    hipLaunchKernelGGL(k_dec1, dim3(vec_grid(n + 1)), dim3(kBlock), 0, st, p.rowptr, (int64_t)n + 1);
    if (nnz) {
        if (validate)
            hipLaunchKernelGGL(k_dec1_check_cols, dim3(vec_grid(nnz)), dim3(kBlock), 0, st, p.col, nnz, (int64_t)ncol_own + n_halo, bad);
        else
            hipLaunchKernelGGL(k_dec1, dim3(vec_grid(nnz)), dim3(kBlock), 0, st, p.col, nnz);
    }
    SGM_HIP(hipGetLastError());
    unsigned long long hbad[2] = {~0ull, ~0ull};
    if (validate) SGM_HIP(hipMemcpyAsync(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));   // host staging buffers may go away after return
    dfree(bad);
    if (hbad[0] != ~0ull) {
        const int64_t i = (int64_t)hbad[0];
        int32_t v[2] = {0, 0};            // (0-based by now)
        SGM_HIP(hipMemcpy(v, p.rowptr + i, (i < n ? 2 : 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (i == n)
            return fail(SGM_ERR_DIMS, "csr create: ptr(%lld) - 1 = %lld entries, but nnz = %lld", (long long)n + 1, (long long)v[0],
                        (long long)nnz);
        if (i == 0 && v[0] != 0) return fail(SGM_ERR_BAD_ARG, "csr create: ptr(1) = %d, expected 1 (1-based row pointers)", v[0] + 1);
        return fail(SGM_ERR_BAD_ARG, "csr create: row pointers decrease at row %lld: ptr(%lld) = %d > ptr(%lld) = %d", (long long)i + 1,
                    (long long)i + 1, v[0] + 1, (long long)i + 2, v[1] + 1);
    }
    if (hbad[1] != ~0ull) {
        const int64_t k = (int64_t)hbad[1];
        int32_t c = 0;
        SGM_HIP(hipMemcpy(&c, p.col + k, sizeof c, hipMemcpyDeviceToHost));
        std::vector<int32_t> hp((size_t)n + 1);          // error path only: the row that holds entry k
        SGM_HIP(hipMemcpy(hp.data(), p.rowptr, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        const int64_t row = std::upper_bound(hp.begin(), hp.end(), (int32_t)k) - hp.begin();      // 1-based
        return fail(SGM_ERR_DIMS, "csr create: node(%lld) = %d in row %lld is outside 1..%lld", (long long)k + 1, c + 1, (long long)row,
                    (long long)ncol_own + n_halo);
    }
    return build_offset_dict(p, where == SGM_HOST ? ptr1 : nullptr, where == SGM_HOST ? node1 : nullptr);
}

// A plain-CSR copy of a single-part CSR matrix (device to device; no derived SpMV format): scratch for setup work that wants to
// permute a matrix without touching the caller's (the reordering preconditioner, sgm_pc.hip)
int clone_csr_plain(sgm_mat A, sgm_mat *out)
{
    *out = nullptr;
    if (!A || A->fmt != SGM_FMT_CSR || A->parts.size() != 1 || A->comm)
        return fail(SGM_ERR_UNSUPPORTED, "clone_csr_plain: single-GPU CSR matrices only");
    const Part &p = A->parts[0];
    SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{p};
    sgm_mat C = new sgm_mat_s;
    C->fmt = SGM_FMT_CSR; C->nrow = A->nrow; C->ncol = A->ncol; C->nnz = A->nnz;
    C->parts.resize(1);
    Part &q = C->parts[0];
    q.opt.csr_offset_dict = 0; q.opt.csr_sliced = 0; q.opt.csr_sell = 0; q.opt.csr_lean = 0; q.opt.slice_sched = 0;
    q.n = p.n; q.ncol_own = p.ncol_own; q.n_halo = 0; q.nnz = p.nnz; q.max_row = p.max_row;
    hipStream_t st = g_rt.stream;
    int rc = dalloc(&q.rowptr, (size_t)p.n + 1);
    if (rc == SGM_OK) rc = dalloc(&q.col, (size_t)p.nnz + 4);
    if (rc == SGM_OK) rc = dalloc(&q.val, (size_t)p.nnz + 2);
    if (rc != SGM_OK) { sgm_mat_destroy(C); return rc; }
    (void)hipMemcpyAsync(q.rowptr, p.rowptr, ((size_t)p.n + 1) * 4, hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(q.col + p.nnz, 0, 16, st);
    (void)hipMemsetAsync(q.val + p.nnz, 0, 16, st);
    if (p.nnz) {
        (void)hipMemcpyAsync(q.col, p.col, (size_t)p.nnz * 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(q.val, p.val, (size_t)p.nnz * 8, hipMemcpyDeviceToDevice, st);
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { sgm_mat_destroy(C); return fail(SGM_ERR_HIP, "clone_csr_plain: copy failed"); }
    *out = C;
    return SGM_OK;
}

// after a change of the index arrays (matrix permutation): drop and rebuild the derived formats
int rebuild_csr_formats(Part &p)
{
    SGM_TRY(csr_need_arrays(p));           // col / val are what the formats are rebuilt from
    p.lean = false;
    dfree(p.code); dfree(p.dict); dfree(p.sval); dfree(p.scode); dfree(p.scol); dfree(p.sbcode);
    p.code = nullptr; p.dict = nullptr; p.sval = nullptr; p.scode = nullptr; p.scol = nullptr; p.sbcode = nullptr;
    free_sell(p);
    free_ell_colblock(p);          // (a scattered matrix's column-blocked form: the new order may have an offset dictionary instead)
    p.ndict = 0; p.dict_reach = 0; p.sw = 0; p.max_row = 0; p.sched_period = 0;
    free_slice_sched(p);
    return build_offset_dict(p, nullptr, nullptr);
}
int rebuild_ell_formats(Part &p)
{
    dfree(p.ecode); dfree(p.dict); dfree(p.sval); dfree(p.scode);
    p.ecode = nullptr; p.dict = nullptr; p.dict_reach = 0; p.emdp = 0; p.sval = nullptr; p.scode = nullptr; p.sw = 0; p.sched_period = 0;
    free_slice_sched(p);
    SGM_TRY(build_ell_offset_dict(p));
    SGM_TRY(build_ell_colblock(p));
    return refresh_ell_colblock_values(p);
}
int sgm_invalidate_transpose(sgm_mat A)
{
    if (A->T) { sgm_mat_destroy(A->T); A->T = nullptr; }
    dfree(A->tperm);
    A->tperm = nullptr;
    A->t_stale = true;
    return SGM_OK;
}

void free_part(Part &p)
{
    dfree(p.rowptr); dfree(p.col); dfree(p.val); dfree(p.code); dfree(p.dict); dfree(p.sval); dfree(p.scode); dfree(p.scol); dfree(p.sbcode); dfree(p.ecol); dfree(p.eval); dfree(p.edeg); dfree(p.ecode); dfree(p.xext);
    for (auto &nb : p.nbrs) { dfree(nb.send_idx); dfree(nb.send_buf); }
    free_slice_sched(p);
    free_ell_colblock(p);
    free_sell(p);
    p = Part();
}

// Stage a caller vector on the device if it lives on the host (or is not 16-B aligned).
struct Staged {
    double *dev = nullptr;
    bool owned = false;
    ~Staged() { if (owned) dfree(dev); }
};
int stage_in(Staged &s, const double *v, int64_t n, int where, bool copy)
{
    if (where == SGM_DEVICE && (reinterpret_cast<uintptr_t>(v) & 15) == 0) {
        s.dev = const_cast<double *>(v);
        return SGM_OK;
    }
    SGM_TRY(dalloc(&s.dev, (size_t)n));
    s.owned = true;
    if (copy)
        SGM_HIP(hipMemcpyAsync(s.dev, v, (size_t)n * 8,
                               where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                               g_rt.stream));
    return SGM_OK;
}
int stage_out(const Staged &s, double *v, int64_t n, int where)
{
    if (!s.owned) return SGM_OK;
    SGM_HIP(hipMemcpyAsync(v, s.dev, (size_t)n * 8,
                           where == SGM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                           g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

__global__ void k_gather_perm(double *__restrict__ dst, const double *__restrict__ src,
                              const int32_t *__restrict__ perm, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[perm[i]];
}

// Transpose products (linear_operator_interface.f90:199-208 -> csc_matvec_add
// cs_matrices.f90:627-647 / ellpack_matvec_t_add ellpack_matrices.f90:670-693).  The reference
// scatters y(node(k)) += val(k)*x(j) for j = 1..n, k in stored order; a scatter needs atomics
// on a GPU and would lose the summation order.  Instead A^T is built once (device radix
// sort, stable in (j, k)), so y(i) is a ROW SUM over the same terms in the same order and the
// ordinary SpMV kernels apply (for matvec_t_add the sum is chained onto y(i), bit for bit
// like the scatter).  ELLPACK padding slots are kept (they add val=0 * x(j) like the reference).
// keys (= column of the entry) and source indices of all entries in (row j, slot k) order
__global__ void k_tr_keys_csr(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                              int32_t *__restrict__ key, int32_t *__restrict__ src, int32_t *__restrict__ rowid,
                              int32_t *__restrict__ count)
{
    const int32_t j = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    for (int32_t k = rowptr[j] + lane; k < rowptr[j + 1]; k += 64) {
        key[k] = col[k];
        src[k] = k;
        rowid[k] = j + 1;                        // 1-based row of A = column index in A^T
        atomicAdd(&count[col[k]], 1);
    }
}
__global__ void k_tr_keys_ell(int32_t n, int32_t max_d, const int32_t *__restrict__ ecol, int32_t *__restrict__ key,
                              int32_t *__restrict__ src, int32_t *__restrict__ rowid, int32_t *__restrict__ count)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // t = j*max_d + k
    if (t >= (int64_t)n * max_d) return;
    const int32_t j = (int32_t)(t / max_d), k = (int32_t)(t % max_d);
    const int64_t s = (int64_t)k * n + j;                                   // slot-major device layout
    const int32_t c = ecol[s];
    key[t] = c;
    src[t] = (int32_t)s;
    rowid[t] = j + 1;
    atomicAdd(&count[c], 1);
}
__global__ void k_tr_gather_rows(int64_t nnz, int64_t stride_t, int32_t max_d, int32_t n_src, const int32_t *__restrict__ src_sorted,
                                 const int32_t *__restrict__ rowid, int32_t *__restrict__ tnode)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        const int32_t s = src_sorted[i];
        // CSR: rowid is indexed by the entry; ELLPACK: by t = j*max_d + k with s = k*n + j
        tnode[i] = max_d ? rowid[(int64_t)(s % n_src) * max_d + s / n_src] : rowid[s];
    }
    (void)stride_t;
}
__global__ void k_inc1(int64_t n, int32_t *a)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] += 1;
}

static int ensure_transpose(sgm_mat A)
{
    if (A->distributed()) return fail(SGM_ERR_UNSUPPORTED, "matvec_t: not available on a row-partitioned matrix");
    Part &p = A->parts[0];
    const bool ell = A->fmt == SGM_FMT_ELL;
    const int64_t nnz = ell ? (int64_t)p.n * p.max_d : p.nnz;
    if (!A->T) {
        // A^T on the device: a STABLE radix sort of the entries by column (hipCUB) keeps them in
        // (row j, slot k) order inside every column, which is the order the reference's scatter adds
        // them in; the column histogram's prefix sum is A^T's row pointer.
        hipStream_t st = g_rt.stream;
        const int32_t nt = A->ncol;                        // rows of A^T
        const size_t m = (size_t)std::max<int64_t>(nnz, 1);
        int32_t *key = nullptr, *src = nullptr, *rowid = nullptr, *key2 = nullptr, *src2 = nullptr, *tptr = nullptr, *tnode = nullptr;
        double *zeros = nullptr;
        void *tmp = nullptr;
        size_t tb_sort = 0, tb_scan = 0;
        int rc = dalloc(&key, m);
        if (rc == SGM_OK) rc = dalloc(&src, m);
        if (rc == SGM_OK) rc = dalloc(&rowid, m);
        if (rc == SGM_OK) rc = dalloc(&key2, m);
        if (rc == SGM_OK) rc = dalloc(&src2, m);
        if (rc == SGM_OK) rc = dalloc(&tptr, (size_t)nt + 2);
        if (rc == SGM_OK) rc = dalloc(&tnode, m);
        if (rc == SGM_OK) rc = dalloc(&zeros, m);
        if (rc == SGM_OK) {
            int end_bit = 1;
            while (end_bit < 31 && (1ll << end_bit) <= (int64_t)nt) ++end_bit;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, key, key2, src, src2, (int)std::min<int64_t>(nnz, INT32_MAX), 0, end_bit, st);
            (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, tptr, tptr, nt + 1, st);
            if (hipMalloc(&tmp, std::max<size_t>(std::max(tb_sort, tb_scan), 16)) != hipSuccess) rc = fail(SGM_ERR_HIP, "matvec_t: sort workspace");
            if (rc == SGM_OK) {
                (void)hipMemsetAsync(tptr, 0, ((size_t)nt + 2) * 4, st);
                (void)hipMemsetAsync(zeros, 0, m * 8, st);
                if (nnz && !ell && csr_need_arrays(p) != SGM_OK) rc = SGM_ERR_ALLOC;
                if (nnz && rc == SGM_OK) {
                    if (!ell)
                        hipLaunchKernelGGL(k_tr_keys_csr, dim3((unsigned)(((int64_t)p.n * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                                           p.n, (const int32_t *)p.rowptr, (const int32_t *)p.col, key, src, rowid, tptr);
                    else
                        hipLaunchKernelGGL(k_tr_keys_ell, dim3((unsigned)((nnz + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, p.n, p.max_d,
                                           (const int32_t *)p.ecol, key, src, rowid, tptr);
                    (void)hipcub::DeviceRadixSort::SortPairs(tmp, tb_sort, key, key2, src, src2, (int)nnz, 0, end_bit, st);
                    hipLaunchKernelGGL(k_tr_gather_rows, dim3(vec_grid(nnz)), dim3(kBlock), 0, st, nnz, (int64_t)0, ell ? p.max_d : 0, p.n,
                                       (const int32_t *)src2, (const int32_t *)rowid, tnode);
                }
                (void)hipcub::DeviceScan::ExclusiveSum(tmp, tb_scan, tptr, tptr, nt + 1, st);
                hipLaunchKernelGGL(k_inc1, dim3(vec_grid(nt + 1)), dim3(kBlock), 0, st, (int64_t)nt + 1, tptr);     // 1-based, like the reference
                if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(SGM_ERR_HIP, "matvec_t: transpose build failed");
            }
        }
        sgm_mat T = nullptr;
        if (rc == SGM_OK) {
            T = new sgm_mat_s;
            T->fmt = SGM_FMT_CSR;
            T->nrow = A->ncol;
            T->ncol = A->nrow;
            T->nnz = nnz;
            T->parts.resize(1);
            T->parts[0].opt = p.opt;                       // A^T runs with A's options
            rc = build_csr_part(T->parts[0], T->nrow, T->ncol, 0, nnz, tptr, tnode, zeros, SGM_DEVICE, false);
        }
        if (tmp) (void)hipFree(tmp);
        dfree(key); dfree(src); dfree(rowid); dfree(key2); dfree(tptr); dfree(tnode); dfree(zeros);
        if (!ell) csr_release_arrays(p);
        if (rc != SGM_OK) { dfree(src2); if (T) sgm_mat_destroy(T); return rc; }
        A->tperm = src2;                                   // entry of A behind every entry of A^T
        A->T = T;
        A->t_stale = true;
    }
    if (A->t_stale && nnz) {
        Part &tp = A->T->parts[0];
        if (!ell) SGM_TRY(csr_need_arrays(p));             // A's values in CSR order (a lean part rebuilds them from its slices)
        SGM_TRY(lean_val_buffer(tp));
        hipLaunchKernelGGL(k_gather_perm, dim3(vec_grid(nnz)), dim3(kBlock), 0, g_rt.stream, tp.val,
                           (const double *)(ell ? p.eval : p.val), (const int32_t *)A->tperm, nnz);
        SGM_HIP(hipGetLastError());
        SGM_TRY(pack_sliced(tp));
        if (tp.cb_P) SGM_TRY(refresh_ell_colblock_values(tp));      // (a transpose with scattered columns has the column-blocked form)
        csr_release_arrays(tp);
        if (!ell) csr_release_arrays(p);
    }
    A->t_stale = false;
    return SGM_OK;
}

static int matvec_t_impl(sgm_mat A, const double *x, double *y, int where, bool add)
{
    SGM_TRY(require_init());
    if (!A || !x || !y) return fail(SGM_ERR_BAD_ARG, "matvec_t: null argument");
    if (A->fmt == SGM_FMT_COMPOSITE) {
        // composite_matvec_t_add (sparse_matrix_composites.f90:1104-1127): column blocks outer
        const int64_t nr = A->parts[0].n, nc = A->parts[0].ncol_own;        // (local lengths over distributed leaves)
        Staged sx, sy;
        SGM_TRY(stage_in(sx, x, nr, where, true));
        SGM_TRY(stage_in(sy, y, nc, where, add));
        if (!add) SGM_HIP(hipMemsetAsync(sy.dev, 0, (size_t)nc * 8, g_rt.stream));
        const int nrb = (int)A->blk_row_ptr.size() - 1, ncb = (int)A->blk_col_ptr.size() - 1;
        for (int jt = 0; jt < ncb; ++jt)
            for (int it = 0; it < nrb; ++it) {
                sgm_mat C = A->blocks[(size_t)it * ncb + jt];
                if (!C) continue;
                if (C->comm) {          // A^T of the leaf is a distributed matrix of its own (sgm_dist.hip)
                    SGM_TRY(matvec_t_dist(C, sx.dev + A->blk_row_ptr[it], sy.dev + A->blk_col_ptr[jt], SGM_DEVICE, true));
                    continue;
                }
                SGM_TRY(ensure_transpose(C));
                const double *xs[1] = {sx.dev + A->blk_row_ptr[it]};
                double *ys[1] = {sy.dev + A->blk_col_ptr[jt]};
                SGM_TRY(spmv_parts(C->T, xs, ys, true, nullptr, nullptr, nullptr, 0x7fffffff, true));
            }
        SGM_TRY(stage_out(sy, y, nc, where));
        return finish();
    }
    if (A->comm) return matvec_t_dist(A, x, y, where, add);
    SGM_TRY(ensure_transpose(A));
    Staged sx, sy;
    SGM_TRY(stage_in(sx, x, A->nrow, where, true));
    SGM_TRY(stage_in(sy, y, A->ncol, where, add));
    const double *xs[1] = {sx.dev};
    double *ys[1] = {sy.dev};
    SGM_TRY(spmv_parts(A->T, xs, ys, add, nullptr, nullptr, nullptr, 0x7fffffff, /*chain=*/add));
    SGM_TRY(stage_out(sy, y, A->ncol, where));
    return finish();
}

static int matvec_impl(sgm_mat A, const double *x, double *y, int where, bool add)
{
    SGM_TRY(require_init());
    if (!A || !x || !y) return fail(SGM_ERR_BAD_ARG, "matvec: null argument");
    const size_t P = A->parts.size();
    if (P == 1) {
        Part &p = A->parts[0];
        Staged sx, sy;
        SGM_TRY(stage_in(sx, x, p.xlen(), where, true));
        SGM_TRY(stage_in(sy, y, p.n, where, add));
        const double *xs[1] = {sx.dev};
        double *ys[1] = {sy.dev};
        SGM_TRY(spmv_parts(A, xs, ys, add, nullptr, nullptr, nullptr));
        SGM_TRY(stage_out(sy, y, p.n, where));
        return finish();
    }
    // in-process row partition: x and y are plain global-length vectors
    Staged sx, sy;
    SGM_TRY(stage_in(sx, x, A->ncol, where, true));
    SGM_TRY(stage_in(sy, y, A->nrow, where, add));
    std::vector<const double *> xs(P);
    std::vector<double *> ys(P);
    for (size_t ip = 0; ip < P; ++ip) {
        Part &p = A->parts[ip];
        SGM_HIP(hipMemcpyAsync(p.xext, sx.dev + p.row_begin, (size_t)p.ncol_own * 8,
                               hipMemcpyDeviceToDevice, g_rt.stream));
        xs[ip] = p.xext;
        ys[ip] = sy.dev + p.row_begin;
    }
    SGM_TRY(spmv_parts(A, xs.data(), ys.data(), add, nullptr, nullptr, nullptr));
    SGM_TRY(stage_out(sy, y, A->nrow, where));
    return finish();
}

// y = A x on device vectors laid out like sgm_mat_matvec's: one part (also one rank of a distributed matrix: x holds
// [owned | halo room]) or an in-process partition (plain global vectors); stream-ordered, no synchronisation
int matvec_plain(sgm_mat A, const double *x, double *y)
{
    const bool was_async = g_rt.async;
    g_rt.async = true;
    const int rc = matvec_impl(A, x, y, SGM_DEVICE, false);
    g_rt.async = was_async;
    return rc;
}

}  // namespace sgm

using namespace sgm;

extern "C" {

int sgm_csr_create(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t nnz, const int32_t *ptr,
                   const int32_t *node, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || nrow < 0 || ncol < 0 || nnz < 0 || !ptr || (nnz && (!node || !val)))
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_create: bad argument");
    if (nnz > INT32_MAX - 4) return fail(SGM_ERR_UNSUPPORTED, "sgm_csr_create: nnz exceeds int32 ptr");
    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_CSR;
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz = nnz;
    A->parts.resize(1);
    int rc = build_csr_part(A->parts[0], nrow, ncol, 0, nnz, ptr, node, val, where, true);
    if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
    *out = A;
    return SGM_OK;
}

int sgm_csr_set_values(sgm_mat A, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!A || A->fmt != SGM_FMT_CSR || !val) return fail(SGM_ERR_BAD_ARG, "sgm_csr_set_values: bad argument");
    A->t_stale = true;
    A->version += 1;
    int64_t off = 0;
    for (auto &p : A->parts) {
        SGM_TRY(lean_val_buffer(p));
        SGM_HIP(hipMemcpyAsync(p.val, val + off, (size_t)p.nnz * 8,
                               where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                               g_rt.stream));
        off += p.nnz;
        SGM_TRY(pack_sliced(p));
        if (p.cb_P) SGM_TRY(refresh_ell_colblock_values(p));
        csr_release_arrays(p);
    }
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

int sgm_ell_create(sgm_mat *out, int32_t nrow, int32_t ncol, int32_t max_d, const int32_t *node,
                   const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || nrow < 0 || ncol < 0 || max_d < 0 || (nrow && max_d && (!node || !val)))
        return fail(SGM_ERR_BAD_ARG, "sgm_ell_create: bad argument");
    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_ELL;
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz = (int64_t)nrow * max_d;
    A->parts.resize(1);
    Part &p = A->parts[0];
    p.n = nrow;
    p.ncol_own = ncol;
    p.max_d = max_d;
    const size_t total = (size_t)nrow * max_d;
    int rc = dalloc(&p.ecol, total);
    if (rc == SGM_OK) rc = dalloc(&p.eval, total);
    if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
    *out = A;
    if (total == 0) return SGM_OK;
    // (from here on *out owns A: an error return leaves a handle the caller may destroy -- except for rejected
    // index arrays, where nothing usable exists)
    int32_t *tn = nullptr;
    unsigned long long *bad = nullptr, hbad = ~0ull;
    SGM_TRY(dalloc(&bad, 1));
    SGM_HIP(hipMemsetAsync(bad, 0xff, sizeof(unsigned long long), g_rt.stream));
    const int32_t *src = node;
    if (where == SGM_HOST) {
        SGM_TRY(dalloc(&tn, total));
        SGM_HIP(hipMemcpyAsync(tn, node, total * 4, hipMemcpyHostToDevice, g_rt.stream));
        src = tn;
    }
    hipLaunchKernelGGL(k_ell_transpose, dim3(vec_grid(total)), dim3(kBlock), 0, g_rt.stream, src, (const double *)nullptr, p.ecol,
                       p.eval, nrow, max_d, ncol, bad);
    SGM_HIP(hipMemcpyAsync(&hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    dfree(tn);
    dfree(bad);
    if (hbad != ~0ull) {
        int32_t c = 0;
        const int64_t e = (int64_t)hbad;        // entry (slot k, row i) of the (max_d, n) array: e = i * max_d + k
        SGM_HIP(hipMemcpy(&c, node + e, sizeof c, where == SGM_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost));
        *out = nullptr;
        sgm_mat_destroy(A);
        return fail(SGM_ERR_DIMS, "ellpack create: node(%lld,%lld) = %d is outside 0..%d", (long long)(e % max_d) + 1,
                    (long long)(e / max_d) + 1, c, ncol);
    }
    SGM_TRY(build_ell_offset_dict(p));
    SGM_TRY(build_ell_colblock(p));
    return sgm_ell_set_values(A, val, where);
}

int sgm_ell_set_values(sgm_mat A, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!A || A->fmt != SGM_FMT_ELL || !val) return fail(SGM_ERR_BAD_ARG, "sgm_ell_set_values: bad argument");
    Part &p = A->parts[0];
    A->t_stale = true;
    A->version += 1;
    const size_t total = (size_t)p.n * p.max_d;
    if (!total) return SGM_OK;
    double *tv = nullptr;
    const double *src = val;
    if (where == SGM_HOST) {
        SGM_TRY(dalloc(&tv, total));
        SGM_HIP(hipMemcpyAsync(tv, val, total * 8, hipMemcpyHostToDevice, g_rt.stream));
        src = tv;
    }
    hipLaunchKernelGGL(k_ell_transpose, dim3(vec_grid(total)), dim3(kBlock), 0, g_rt.stream,
                       (const int32_t *)nullptr, src, p.ecol, p.eval, p.n, p.max_d);
    SGM_HIP(hipGetLastError());
    SGM_TRY(pack_sliced(p));
    SGM_TRY(refresh_ell_colblock_values(p));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    dfree(tv);
    return SGM_OK;
}

/* sgm_mat_set_option: this matrix's own copy of a kernel-selection option (sgm_set_option only changes what matrices
 * created LATER start with).  Options that choose among forms the handle already holds (csr_sliced, csr_offset_dict,
 * csr_row_owner, csr_row_lines, csr_sell, ell_offset_dict, ell_colblock 0 / nonzero, slice_sched) act from the next
 * product on; the ones a form is BUILT with (ell_colblock 0 <-> built, ell_colblock_cols / _rows, csr_lean)
 * rebuild / release that form here.  Every choice gives the same bits.  On a composite: applied to every block. */
int sgm_mat_set_option(sgm_mat A, const char *name, int value)
{
    SGM_TRY(require_init());
    if (!A || !name) return fail(SGM_ERR_BAD_ARG, "sgm_mat_set_option: null argument");
    int v = 0;
    SGM_TRY(normalise_option(name, value, &v));
    MatOptions probe;
    if (!mat_option_field(probe, name)) return fail(SGM_ERR_BAD_ARG, "sgm_mat_set_option: '%s' is not a matrix option", name);
    if (A->fmt == SGM_FMT_COMPOSITE) {
        for (sgm_mat_s *B : A->blocks)
            if (B) SGM_TRY(sgm_mat_set_option(B, name, value));
        return SGM_OK;
    }
    const bool cb_shape = !strcmp(name, "ell_colblock_cols") || !strcmp(name, "ell_colblock_rows");
    for (Part &p : A->parts) {
        int *f = mat_option_field(p.opt, name);
        const int old = *f;
        *f = v;
        if (old == v) continue;
        if ((p.ecol || (!p.lean && p.rowptr && p.col && p.val && p.n_halo == 0 && !p.sval && !p.sl_val)) &&
            (cb_shape || (!strcmp(name, "ell_colblock") && ((old != 0) != (v != 0) || v == 2 || old == 2)))) {
            SGM_TRY(build_ell_colblock(p));           // (frees the old form first; decides again whether the matrix wants one)
            SGM_TRY(refresh_ell_colblock_values(p));
            SGM_HIP(hipStreamSynchronize(g_rt.stream));
        }
        if (!strcmp(name, "csr_lean") && !p.ecol) {
            if (v == 0) { SGM_TRY(csr_need_arrays(p)); p.lean = false; }
            else csr_go_lean(p);
        }
        if (!strcmp(name, "slice_sched")) free_slice_sched(p);
    }
    if (A->T) SGM_TRY(sgm_mat_set_option(A->T, name, value));
    return SGM_OK;
}

int sgm_mat_matvec(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_impl(A, x, y, where, false);
}

int sgm_mat_matvec_add(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_impl(A, x, y, where, true);
}

int sgm_composite_create(sgm_mat *out, int32_t nrb, int32_t ncb, const int32_t *row_ptr, const int32_t *col_ptr,
                         const sgm_mat *blocks)
{
    SGM_TRY(require_init());
    if (!out || nrb < 1 || ncb < 1 || !row_ptr || !col_ptr || !blocks)
        return fail(SGM_ERR_BAD_ARG, "sgm_composite_create: bad argument");
    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_COMPOSITE;
    for (int i = 0; i <= nrb; ++i) A->blk_row_ptr.push_back(row_ptr[i] - 1);
    for (int j = 0; j <= ncb; ++j) A->blk_col_ptr.push_back(col_ptr[j] - 1);
    A->nrow = A->blk_row_ptr[nrb];
    A->ncol = A->blk_col_ptr[ncb];
    A->blocks.assign(blocks, blocks + (size_t)nrb * ncb);
    sgm_comm comm = nullptr;
    bool any_local = false;
    for (int it = 0; it < nrb; ++it)
        for (int jt = 0; jt < ncb; ++jt) {
            sgm_mat C = A->blocks[(size_t)it * ncb + jt];
            if (!C) continue;
            if (C->parts.size() != 1 || C->fmt == SGM_FMT_COMPOSITE || C->nrow != A->blk_row_ptr[it + 1] - A->blk_row_ptr[it] ||
                C->ncol != A->blk_col_ptr[jt + 1] - A->blk_col_ptr[jt]) {
                delete A;
                return fail(SGM_ERR_DIMS, "sgm_composite_create: block (%d,%d) does not fit its slot", it + 1, jt + 1);
            }
            if (C->comm) { if (comm && comm != C->comm) { delete A; return fail(SGM_ERR_BAD_ARG, "sgm_composite_create: leaves on different communicators"); } comm = C->comm; }
            else any_local = true;
            A->nnz += C->nnz;
        }
    int64_t nloc_r = A->nrow, nloc_c = A->ncol;
    if (comm) {
        // Leaves distributed over processes: block row i must be partitioned the same way in all its leaves, block
        // column j likewise, and (so that the operator maps a vector layout onto itself) block row i like block
        // column i.  The block offsets become the LOCAL ones: this rank's slices of the block vectors, concatenated.
        const int me = comm->rank;
        auto bad = [&](const char *why) { delete A; return fail(SGM_ERR_UNSUPPORTED, "sgm_composite_create over distributed leaves: %s", why); };
        if (any_local) return bad("every leaf must be distributed (sgm_csr_create_dist / _rect / sgm_ell_create_dist)");
        if (nrb != ncb) return bad("needs as many block rows as block columns");
        std::vector<const std::vector<int64_t> *> rpart(nrb, nullptr), cpart(ncb, nullptr);
        for (int it = 0; it < nrb; ++it)
            for (int jt = 0; jt < ncb; ++jt) {
                sgm_mat C = A->blocks[(size_t)it * ncb + jt];
                if (!C) continue;
                if (rpart[it] && *rpart[it] != C->row_starts) return bad("the leaves of a block row are partitioned differently");
                if (cpart[jt] && *cpart[jt] != C->col_starts) return bad("the leaves of a block column are partitioned differently");
                rpart[it] = &C->row_starts;
                cpart[jt] = &C->col_starts;
            }
        std::vector<int32_t> lr(1, 0), lc(1, 0);
        for (int it = 0; it < nrb; ++it) {
            if (!rpart[it] || !cpart[it]) return bad("a block row or column without any leaf has no partition");
            if (*rpart[it] != *cpart[it]) return bad("block row i must be partitioned like block column i");
            lr.push_back(lr.back() + (int32_t)((*rpart[it])[me + 1] - (*rpart[it])[me]));
            lc.push_back(lc.back() + (int32_t)((*cpart[it])[me + 1] - (*cpart[it])[me]));
        }
        A->blk_row_ptr = lr;
        A->blk_col_ptr = lc;
        A->comm = comm;
        nloc_r = lr.back();
        nloc_c = lc.back();
    }
    A->parts.resize(1);
    A->parts[0].n = (int32_t)nloc_r;
    A->parts[0].ncol_own = (int32_t)nloc_c;
    int64_t g = (nloc_r + 4 * kBlock - 1) / (4 * kBlock);
    A->parts[0].dot_grid_override = (int)std::max<int64_t>(1, std::min<int64_t>(g, 2048));
    *out = A;
    return SGM_OK;
}

int sgm_mat_matvec_t(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_t_impl(A, x, y, where, false);
}

int sgm_mat_matvec_t_add(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_t_impl(A, x, y, where, true);
}

int sgm_mat_get(sgm_mat A, const char *name, void *out, size_t bytes, size_t *needed)
{
    SGM_TRY(require_init());
    if (!A || !name) return fail(SGM_ERR_BAD_ARG, "sgm_mat_get: null argument");
    if (A->distributed() || A->fmt == SGM_FMT_COMPOSITE)
        return fail(SGM_ERR_UNSUPPORTED, "sgm_mat_get: leaf single-GPU matrices only");
    const Part &p = A->parts[0];
    const std::string nm(name);
    std::vector<int32_t> vi;
    std::vector<double> vd;
    const bool ell = A->fmt == SGM_FMT_ELL;
    if (!ell && (nm == "node" || nm == "val")) SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; bool on; ~Release() { if (on) csr_release_arrays(p); } } rel{p, !ell};
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    if (!ell && nm == "ptr") {
        vi.resize((size_t)p.n + 1);
        SGM_HIP(hipMemcpy(vi.data(), p.rowptr, vi.size() * 4, hipMemcpyDeviceToHost));
        for (auto &v : vi) v += 1;
    } else if (!ell && nm == "node") {
        vi.resize((size_t)p.nnz);
        if (p.nnz) SGM_HIP(hipMemcpy(vi.data(), p.col, vi.size() * 4, hipMemcpyDeviceToHost));
        for (auto &v : vi) v += 1;
    } else if (!ell && nm == "val") {
        vd.resize((size_t)p.nnz);
        if (p.nnz) SGM_HIP(hipMemcpy(vd.data(), p.val, vd.size() * 8, hipMemcpyDeviceToHost));
    } else if (ell && nm == "max_d") {
        vi.assign(1, p.max_d);
    } else if (ell && nm == "degrees" && p.edeg) {
        vi.resize((size_t)p.n);
        if (p.n) SGM_HIP(hipMemcpy(vi.data(), p.edeg, vi.size() * 4, hipMemcpyDeviceToHost));
    } else if (ell && (nm == "node" || nm == "val")) {
        const size_t total = (size_t)p.n * p.max_d;       // back to the reference's (max_d, n) order
        if (nm == "node") {
            std::vector<int32_t> t(total);
            if (total) SGM_HIP(hipMemcpy(t.data(), p.ecol, total * 4, hipMemcpyDeviceToHost));
            vi.resize(total);
            for (int32_t i = 0; i < p.n; ++i)
                for (int32_t k = 0; k < p.max_d; ++k) vi[(size_t)i * p.max_d + k] = t[(size_t)k * p.n + i] + 1;
        } else {
            std::vector<double> t(total);
            if (total) SGM_HIP(hipMemcpy(t.data(), p.eval, total * 8, hipMemcpyDeviceToHost));
            vd.resize(total);
            for (int32_t i = 0; i < p.n; ++i)
                for (int32_t k = 0; k < p.max_d; ++k) vd[(size_t)i * p.max_d + k] = t[(size_t)k * p.n + i];
        }
    } else {
        return fail(SGM_ERR_BAD_ARG, "sgm_mat_get: unknown array '%s' for this format", name);
    }
    const size_t sz = vi.size() * 4 + vd.size() * 8;
    if (needed) *needed = sz;
    if (out && sz) {
        if (bytes < sz) return fail(SGM_ERR_BAD_ARG, "sgm_mat_get: buffer too small (%zu < %zu)", bytes, sz);
        memcpy(out, vi.empty() ? (const void *)vd.data() : (const void *)vi.data(), sz);
    }
    return SGM_OK;
}

int sgm_mat_info(sgm_mat A, int32_t *nrow, int32_t *ncol, int64_t *nnz, int32_t *fmt, int64_t *x_len)
{
    if (!A) return fail(SGM_ERR_BAD_ARG, "sgm_mat_info: null matrix");
    if (nrow) *nrow = A->nrow;
    if (ncol) *ncol = A->ncol;
    if (nnz) *nnz = A->nnz;
    if (fmt) *fmt = A->fmt;
    if (x_len) *x_len = A->comm ? A->parts[0].xlen() : A->ncol;
    return SGM_OK;
}

int sgm_mat_kernel(sgm_mat A, char *buf, int len)
{
    if (!A || !buf || len < 1) return fail(SGM_ERR_BAD_ARG, "sgm_mat_kernel: bad argument");
    char name[64];
    if (A->fmt == SGM_FMT_COMPOSITE) snprintf(name, sizeof name, "composite");
    else {
        const Part &p = A->parts[0];
        if (A->fmt == SGM_FMT_ELL) {
            if (use_ell_colblock(p)) snprintf(name, sizeof name, "k_ellcb<cols=%d,R=%d>", p.cb_cols, p.cb_R);
            else if (use_sliced_ell(p)) snprintf(name, sizeof name, "k_csr_sl<W=%d>", p.sw);
            else if (p.ecode && p.opt.ell_offset_dict) snprintf(name, sizeof name, "k_ell_do<MDP=%d>", p.emdp);
            else snprintf(name, sizeof name, "k_ell_spmv");
        } else if (use_ell_colblock(p)) snprintf(name, sizeof name, "k_ellcb<cols=%d,R=%d,csr>", p.cb_cols, p.cb_R);
        else if (use_sliced(p)) snprintf(name, sizeof name, "k_csr_sl<W=%d>", p.sw);
        else if (use_slicedb(p)) snprintf(name, sizeof name, "k_csr_slb<W=%d>", p.sw);
        else if (use_sliced32(p)) snprintf(name, sizeof name, "k_csr_sl32<W=%d>", p.sw);
        else if (use_sell(p)) snprintf(name, sizeof name, p.sl_win0 && p.opt.csr_xwindow ? "k_csr_sell<pad=%.3f,xw=%dx%d>" : "k_csr_sell<pad=%.3f>",
                                       p.nnz ? (double)p.sl_total / (double)p.nnz : 1.0, p.sl_span, p.sl_gs);
        else if (use_offset_dict(p)) snprintf(name, sizeof name, "k_csr_do<256,%d,CW=1>", do_tile_for(p));
        else if (use_row_owner(p)) snprintf(name, sizeof name, "k_csr_do<256,%d,CW=4>", do_tile_for(p));
        else if (use_row_lines(p)) snprintf(name, sizeof name, "k_csr_rl");
        else snprintf(name, sizeof name, "k_csr_spmv");
    }
    snprintf(buf, (size_t)len, "%s", name);
    return SGM_OK;
}

// Bytes by construction: what lives in HBM for this handle, and what ONE
// y = A x moves with the kernel the current options select -- the stored format of that kernel
// (padded slices, codes, row pointers as it reads them), every x entry once, every y entry once.
static int64_t part_resident_bytes(const Part &p)
{
    int64_t b = 0;
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
    if (p.rowptr) b += 4 * ((int64_t)p.n + 1);
    if (p.col) b += 4 * (p.nnz + 4);
    if (p.val) b += 8 * (p.nnz + 2);
    if (p.code) b += p.nnz + 16;
    if (p.dict) b += 4 * 256;
    if (p.sval) b += 8 * nsl * kSlRows * p.sw;
    if (p.scode) b += 4 * nsl * kSlRows;
    if (p.scol) b += 4 * nsl * kSlRows * p.sw;
    if (p.sbcode) b += nsl * kSlRows * ((p.sw + 7) / 8 * 8);
    if (p.sl_val) b += 12 * p.sl_total + 2 * nsl * kSlRows + 8 * (nsl * (kSlRows / kSellChunk) + 1);
    if (p.ecol) b += 4 * (int64_t)p.n * p.max_d;
    if (p.eval) b += 8 * (int64_t)p.n * p.max_d;
    if (p.edeg) b += 4 * (int64_t)p.n;
    if (p.ecode) b += (int64_t)p.n * p.emdp;
    if (p.xext) b += 8 * p.xlen();
    b += ell_colblock_resident_bytes(p);
    for (const auto &nb : p.nbrs) b += (int64_t)nb.send_count * (nb.send_buf ? 12 : 4);
    return b;
}
static int64_t part_matvec_bytes(const sgm_mat_s *A, const Part &p)
{
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
    int64_t m;
    if (A->fmt == SGM_FMT_ELL) {
        if (use_ell_colblock(p)) return ell_colblock_matvec_bytes(p);
        if (use_sliced_ell(p)) m = nsl * kSlRows * (8 * (int64_t)p.sw + 4);
        else if (p.ecode && p.opt.ell_offset_dict) m = (int64_t)p.n * (8 * (int64_t)p.max_d + p.emdp);
        else m = 12 * (int64_t)p.n * p.max_d;
    } else if (use_ell_colblock(p)) return ell_colblock_matvec_bytes(p);
    else if (use_sliced(p)) m = nsl * kSlRows * (8 * (int64_t)p.sw + 4);
    else if (use_slicedb(p)) m = nsl * kSlRows * (8 * (int64_t)p.sw + (p.sw + 7) / 8 * 8);
    else if (use_sliced32(p)) m = nsl * kSlRows * 12 * (int64_t)p.sw;
    else if (use_sell(p)) {
        m = 12 * p.sl_total + 2 * nsl * kSlRows + 8 * nsl * (kSlRows / kSellChunk);      // slots (entries + padding), positions, chunk offsets
        if (p.sl_win0 && p.opt.csr_xwindow)              // every slice loads its window of x (instead of "every x entry once")
            return m + ((nsl + p.sl_gs - 1) / p.sl_gs) * (8 * (int64_t)p.sl_span + 4) + 8 * (int64_t)p.n;
    }
    else if (use_offset_dict(p)) m = 9 * p.nnz + 4 * ((int64_t)p.n + 1);
    else m = 12 * p.nnz + 4 * ((int64_t)p.n + 1);
    return m + 8 * p.xlen() + 8 * (int64_t)p.n;
}

int sgm_mat_footprint(sgm_mat A, int64_t *resident_bytes, int64_t *matvec_bytes)
{
    if (!A) return fail(SGM_ERR_BAD_ARG, "sgm_mat_footprint: null matrix");
    int64_t res = 0, mv = 0;
    if (A->fmt == SGM_FMT_COMPOSITE) {
        for (sgm_mat C : A->blocks) {
            if (!C) continue;
            int64_t r = 0, m = 0;
            SGM_TRY(sgm_mat_footprint(C, &r, &m));
            mv += m + 8 * (int64_t)C->nrow;       // a block leaf adds onto y: one more read of its rows
        }
        mv += 8 * (int64_t)A->nrow;               // y = 0
    } else {
        for (const Part &p : A->parts) { res += part_resident_bytes(p); mv += part_matvec_bytes(A, p); }
        if (A->T) { int64_t r = 0; SGM_TRY(sgm_mat_footprint(A->T, &r, nullptr)); res += r + 4 * A->nnz; }
    }
    if (resident_bytes) *resident_bytes = res;
    if (matvec_bytes) *matvec_bytes = mv;
    return SGM_OK;
}

int sgm_mat_destroy(sgm_mat A)
{
    if (!A) return SGM_OK;
    for (auto &p : A->parts) free_part(p);
    if (A->T) sgm_mat_destroy(A->T);
    dfree(A->tperm);
    delete A;
    return SGM_OK;
}

}  // extern "C"
