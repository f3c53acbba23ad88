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
#include "sgm_spmv_select.hpp"
#include "sgm_plan_host.hpp"

namespace sgm {

// ---------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------



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
            if (kend - k >= 24) {
                // a LONG row (a tile may hold nothing else): groups of eight without predicates, the next group's LDS reads in
                // flight while this one is added -- the lane then runs at the chain's own pace, one dependent add per entry
                // (4.2 ns; tools/probes/long_row_probe.py: 33 ns per entry with the predicated loop below alone, which is what a
                // product with ONE row of 1e5 entries waited for)
                const double *pp = prod + (k - ts);
                double a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = pp[u];
                int32_t left = kend - k;
                while (left >= 24) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) b[u] = pp[8 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) z = z + a[u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = pp[16 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) z = z + b[u];
                    pp += 16; left -= 16;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) z = z + a[u];
                left -= 8;
                k = kend - left;
            }
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
// W = value slots per row (compile time: a run-time W costs 10 %): the instantiated widths below, the smallest one that
// holds the longest row; code bytes come in chunks of 8 per row.
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

static int g_launch_flags = 0;        // 256: chained accumulation (see k_csr_* `chain`)

int spmv_grid(const Part &p, bool whole)       // = number of partial sums one SpMV leaves per fused dot
{
    if (p.dot_grid_override) return p.dot_grid_override;
    if (p.ecol) return ell_grid(p);
    if (use_ell_colblock(p)) return ell_colblock_grid(p);
    RowRange r[3];
    const int nr = spmv_ranges(p, r, true, whole);
    return nr ? r[nr - 1].part_off + r[nr - 1].grid : 8;
}

int row_lines_resident_per_cu()
{
    static int nb = 0;
    if (!nb && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_csr_rl<false, true, false>, 256, 0) != hipSuccess || nb < 1))
        nb = 3;
    return nb;
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
int resident_per_cu(bool dict, int block, int v, int cw)
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
// slice_sched_table (what the schedule is and why: sgm_plan_host.hpp) is host-only index work and lives with the other planners.
static_assert(kPlanSliceRows == kSlRows, "sgm_plan_host.hpp keeps its own copy of the slice height");
extern "C" int sgm_slice_sched_host(int64_t n_slices, int64_t period_rows, int32_t grid, int32_t band_slices,
                                    int32_t *tab_out, int64_t capacity, int32_t *iters_out)
{
    return host_slice_sched_host(n_slices, period_rows, grid, band_slices, tab_out, capacity, iters_out);
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
void free_slice_sched(Part &p)
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

}  // namespace sgm
