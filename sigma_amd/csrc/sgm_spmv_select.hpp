// Which product kernel a part runs and with what grid: the predicates over a Part's layouts and options, the row ranges of a
// launch, the launch configuration.  Shared by sgm_spmv.hip (kernels + dispatch), sgm_layouts.hip (the layouts built at create)
// and sgm_mat.hip (the matrix handles of the C ABI: sgm_mat_kernel / sgm_mat_footprint name and price what would run).
#pragma once
#include "sgm_internal.hpp"

#include <type_traits>

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

namespace sgm {

// vector types of the kernels' 8- / 16-byte accesses; the instantiated slot widths of the sliced kernels
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
#define SGM_SLB_WIDTHS(X) X(9) X(12) X(15) X(16) X(19) X(20) X(24) X(25) X(27) X(28) X(32)
#define SGM_SL_WIDTHS(X) X(3) X(5) X(7) X(8)
#define SGM_SL32_WIDTHS(X) X(3) X(5) X(7) X(8) X(12) X(16) X(20) X(24) X(28) X(32)
// geometry of the sliced layouts (kernels: sgm_spmv.hip; builders: sgm_layouts.hip)
constexpr int kSlRows = 512;       // rows per slice = 2 x workgroup size
constexpr int kSellChunk = 128;
constexpr int kSellSigma = 512;       // rows sorted together (a multiple of the 512-row slices).  Measured on the banded test matrices: windows of
                                      // 2048 rows cut the padding from 10-12 % to 2-4 % and were 35-40 % SLOWER -- a chunk's 128 rows then come from a
                                      // 2048-row neighbourhood and their x gathers no longer fit the CU's L1 (33..64 entries per row: 656 -> 920 us)
// the sliced / SELL form as a part's ONLY resident layout (option csr_lean: sgm_layouts.hip)
static bool lean_sliced(const Part &p) { return p.scode && p.sval && p.dict && !p.ecol && !p.scol && !p.sbcode && p.sw <= 8; }
static bool lean_sell(const Part &p) { return p.sl_val && p.sl_col && !p.scode && !p.scol && !p.sbcode && !p.ecol; }
static bool lean_applies(const Part &p) { return p.opt.csr_lean && (lean_sliced(p) || lean_sell(p)); }
// after the sliced form has been built (or refreshed): keep only it

// ---------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------

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

int resident_per_cu(bool dict, int block, int v, int cw = 4);         // (sgm_spmv.hip)
static const SliceSched *slice_sched(const Part &p, int32_t lo, int32_t hi, int grid);
void free_slice_sched(Part &p);                                         // (sgm_spmv.hip)
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
static bool any_sliced(const Part &p) { return use_sliced(p) || use_sliced32(p) || use_slicedb(p) || use_sell(p); }
static bool use_sliced_ell(const Part &p) { return p.ecol && p.scode && p.opt.csr_sliced && p.opt.ell_offset_dict; }
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
int row_lines_resident_per_cu();          // (sgm_spmv.hip: asks the runtime about k_csr_rl)

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


// sgm_layouts.hip
void free_part(Part &p);
void csr_go_lean(Part &p);
int lean_val_buffer(Part &p);
int pack_sliced(Part &p);
int build_ell_offset_dict(Part &p);
// sgm_layouts.hip: kernels other units launch
__global__ void k_ell_transpose(const int32_t *__restrict__ node, const double *__restrict__ val, int32_t *__restrict__ ecol,
                                double *__restrict__ eval, int32_t n, int32_t max_d, int32_t ncol = 0, unsigned long long *bad = nullptr);
}  // namespace sgm
