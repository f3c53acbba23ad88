// Preconditioners for gfx950: Jacobi (jacobi_solvers.f90:37-81) and ILDU(0)
// (ldu_solvers.f90:95-176, :208-265, :275-440).
//
// Jacobi: idiag(i) = 1/A(i,i) is extracted on the device by a row scan (the reference
// calls A%get_value(i,i) per row, cs_matrices.f90:709-724); apply is one elementwise pass.
//
// ILDU(0): pattern pass and factorisation run on the DEVICE (k_ildu_count / _split / _init /
// _factor_level): the reference's algorithm is a sequential IKJ sweep built on
// get/set/add_value row scans; row i only reads rows k < i of its L pattern, so the rows of
// one dependency level of L run side by side, each lane executing its row's statements in
// the reference's order -- L-I, D, U-I are bit-identical.  Everything the applies read is
// filled from the device factors by kernels; the host keeps the level computation, the
// grid / slab detection and lazy copies for sgm_pc_get.
// The APPLY is the hot part: x=b ; (I+L)^-1 ; x/D ; (I+U)^-1, each triangular solve a
// row recurrence (ldu_solvers.f90:227-236).  Rows are grouped into dependency LEVELS at
// setup; rows of one level are independent, each lane does its row's
// z = z - val(k)*x(node(k)) left to right, so the result is bit-identical to the
// sequential sweep.  The solve runs in "position space": vectors are permuted into level
// order (xp[pos]), every row is a 64-byte record {count, first 4 (dependency position, value)}
// so that ONE independent load brings a row and can be issued a level ahead, and
// dependencies are positions.  Wide levels get one launch each; runs of narrow levels
// (<= 4096 rows) are walked by ONE 1024-thread workgroup that keeps the last 8192 results in
// an LDS ring: a level then costs LDS reads + a barrier instead of four dependent global
// round trips (3.9 us -> see DESIGN.md).  Grid-like factors take the strip / slab pipelines
// (k_trsv_strip, sgm_trsv3.hip), factors of a few levels (colour orderings) the row-space
// sweeps (k_trsv_rows): no position space at all.
#include <hipcub/hipcub.hpp>
#include "sgm_internal.hpp"

#include <chrono>
#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace sgm {
struct Staged {
    double *dev = nullptr;
    bool owned = false;
    ~Staged() { if (owned) dfree(dev); }
};
int stage_in(Staged &s, const double *v, int64_t n, int where, bool copy);
int stage_out(const Staged &s, double *v, int64_t n, int where);
}  // namespace sgm
using namespace sgm;

namespace sgm { int rebuild_csr_formats(Part &p); }          // sgm_spmv.hip

namespace {

constexpr int kTrsvBlock = 1024;
constexpr int kNarrow = 4096;        // levels with <= this many rows are walked by one workgroup (4 rows per lane)
constexpr int kRing = 8192;          // LDS ring of recent results (64 KiB): covers two narrow levels
constexpr int kInline = 4;           // dependencies stored inside the row record
constexpr int kRowLevels = 32;       // factors of at most this many levels are swept in row space, one launch per level

struct TrsvRec {                     // one row of a triangular factor, in level order (64 bytes)
    int32_t cnt, k0;                 // entries of the row; offset of its entries in pq / pv
    int32_t q[kInline];              // position (in level order) of the first dependencies
    double v[kInline];               // their values
    int32_t pad[2];
};
static_assert(sizeof(TrsvRec) == 64, "TrsvRec is one 64-byte record");

struct TriFactor {                   // strictly triangular factor on the device, level order
    int32_t *order = nullptr;        // device: pos -> row
    TrsvRec *recs = nullptr;         // device: n records
    int32_t *pq = nullptr;           // device: dependency positions of ALL entries, rows in level order
    double *pv = nullptr;            // device: their values
    int32_t *level_ptr_dev = nullptr;
    int32_t *wq = nullptr;           // device: dependency POSITIONS, kInline slots, slot-major (-1 = none): wide levels
    uint32_t *dq32 = nullptr;        // device: low halves of dq, contiguous (runs with <= 2 dependencies per row)
    uint64_t *dq = nullptr;          // device: ring-walker copy, 4 x 16-bit position deltas per row (0 = none)
    double *dv = nullptr;            // device: ring-walker copy, kInline value slots, slot-major (slot*n + pos)
    std::vector<uint64_t> h_dq;
    size_t nstride = 0;              // entries per value slot of dv (n + padding)
    std::vector<int32_t> level_ptr;  // host: offsets into the level order per level
    std::vector<int32_t> h_order, h_pos;             // host: pos -> row, row -> pos
    int32_t *src = nullptr;                          // device: level-order entry -> entry of the factor's val array
    std::vector<TrsvRec> h_recs;
    std::vector<int32_t> h_pq;
    // cls: -1 = one wave (levels of <= 64 rows), 0..2 = 256/512/1024 threads (one row per lane), 3/4 = 2/4 rows per lane; ring: k_trsv_walk_ring
    // applies; c: most dependencies of a row in the run
    struct Launch { int32_t l0, l1; bool narrow; int cls; bool ring; int c; };
    std::vector<Launch> schedule;
    // row-space copy for factors of a few levels (colour orderings: one per colour): dependency ROWS and values, rc slots,
    // slot-major over the level order -- the sweeps then run on the vectors themselves, one launch per level (k_trsv_rows)
    struct RowLevel { int32_t b, e, c, row0; };      // positions [b, e); most entries of a row; row0 >= 0: rows row0, row0+1, ...
    std::vector<RowLevel> row_levels;
    bool rows_on = false;
    bool have_levels = false, have_walkers = false;   // index work done: levels (+ row-space copy) / the walkers' structures
    int rc = 0;
    int32_t *rq = nullptr;
    double *rv = nullptr;
    // the dependency rows once more as 4-bit codes (rc <= 8 and at most 15 distinct offsets "dependency row - own row" in the
    // whole factor -- any stencil matrix in any of the reference's orderings): rcode[p] = eight codes of position p (15 =
    // no entry), rdict = the offsets.  4 bytes per row where rq holds 4 * rc: the fused PCG sweeps read these.
    uint32_t *rcode = nullptr;
    int32_t *rdict = nullptr;
    int nrdict = 0;
};

// A strictly triangular factor whose rows depend only on the previous row (r-1) and on the row one grid
// line back (r-w): ILDU(0) factors of 5-point / banded matrices in natural order.  The grid is cut into
// STRIPS of 64 columns; a strip's rows are re-laid in a skewed order: lane l of the strip's chain wave handles
// column i0+l and, at step t, grid line t-l, so that the (i-1, j) neighbour is lane l-1's result of the previous
// step (one DPP shift), the (i, j-1) neighbour the lane's own, and every access of a step is one coalesced
// line of the skewed layout (position = strip base + step * 64 + lane).  See k_trsv_strip.
struct StripRec { double cS, cW, rhs; uint64_t code; };    // 32 bytes per (step, lane): coefficients of the r-w / r-1
                                                           // dependency, right-hand side, bit0 has r-w, bit1 has r-1,
                                                           // bit2 r-1 comes FIRST in the row's stored order
struct GridTri {
    bool on = false;
    int32_t w = 0, nj = 0, NI = 0, S = 0;                       // grid width / lines, strips, steps per strip
    int order = 2;                                              // 0 / 1: every two-term row has its r-w / r-1 term first; 2: mixed
    int64_t NP = 0;                                             // positions (incl. padding) = NI * S * 64
    StripRec *rec = nullptr;                                    // device
    int32_t *row = nullptr;                                     // device: position -> row (-1 = padding)
    double *edge = nullptr;                                     // device: NI x (S + 72): lane 63's result of every step (kEdgeEmpty = not yet), 2 clocks
    int32_t *progress = nullptr;                                // device: NI + 1: steps whose edge values are published; [NI] = abort
    int32_t *pos = nullptr;                                     // device: row -> position (index work only; freed after it)
    int32_t *srcS = nullptr, *srcW = nullptr;                   // device: position -> entry of the factor's val array (-1 = none)
    uint8_t *code = nullptr;                                    // device: presence / order bits per position
};

struct PartPC {
    double *idiag = nullptr;
    int32_t n = 0;                   // rows of the part (on a matrix distributed over ranks: this rank's, not the global count)
};

}  // namespace

// ILDU(0) of one diagonal block (the whole matrix on one GPU; with a row partition, the owned
// rows x owned columns of each part: block-Jacobi ILDU)
struct IlduState {
    PcOptions opt = g_opt.pc;        // the owning preconditioner's options (kept equal to sgm_pc_s::opt)
    int32_t n = 0;
    TriFactor L, U;
    double *D = nullptr;
    double *xpL = nullptr, *xpU = nullptr, *Dp = nullptr;   // level-order work vectors, D in U's level order
    int32_t *mapLU = nullptr;                                // U position -> L position of the same row
    std::vector<int32_t> hLptr, hLnode, hUptr, hUnode;      // 1-based, as the reference holds them
    // the factors live on the device (0-based pattern copies, values in the pattern's order; D = the array above): the
    // factorisation runs there, level by level of L's dependency graph (L.order / L.level_ptr), and every structure the
    // applies read is filled from these by kernels.  Host copies of the VALUES only on request (sgm_pc_get, self-check).
    int32_t *dLptr = nullptr, *dLnode = nullptr, *dUptr = nullptr, *dUnode = nullptr;
    double *dLval = nullptr, *dUval = nullptr;
    std::vector<double> hLval, hUval, hD;
    bool host_vals = false;
    int32_t maxL = 0, maxU = 0;                              // longest row of each factor
    int32_t nnzL = 0, nnzU = 0;
    // grid-like factors (found on the device, grid_detect_device): the factorisation walks the anti-diagonals of the grid;
    // L's true dependency levels are then only built if something asks for them
    int32_t *forder = nullptr;
    std::vector<int32_t> flevel_ptr;
    int32_t dev_wl = 0, dev_wu = 0;                          // grid widths found on the device (0: not grid-like / not looked)
    bool dev_slab = false;                                   // a 3-D grid's factors, found on the device
    // strip-pipeline path (both factors grid-like, see GridTri): results in position space and the L -> U hand-over
    GridTri gL, gU;
    double *gxL = nullptr, *gxU = nullptr, *gDp = nullptr;
    int32_t *gmapLU = nullptr;
    bool grid_ok = false;                                   // the strip path reproduced the level-scheduled apply at setup
    // slab-pipeline path (3-D grid factors, sgm_trsv3.hip); slab_ok: it reproduced the level-scheduled apply at setup
    Slab3 *slab = nullptr;
    bool slab_ok = false;
    // the level-scheduled structures are built on first need when a pipelined path serves the pattern
    bool levels_ready = false, levels_pattern = false;
    bool walk_ready = false, walk_pattern = false;          // the same for the level walkers' structures (ensure_walkers)
    // row-space sweeps (apply_rows): L's level 0 is the entry-less run of rows 0 .. rows_n0-1 (0: it is not); L's last level
    // and U's level 0 are the same entry-less run of rows
    int32_t rows_n0 = 0;
    bool rows_fin = false;
};

struct sgm_pc_s {
    int kind = 0;
    int32_t n = 0;
    std::vector<PartPC> parts;       // jacobi
    std::vector<IlduState> ild;      // ildu: one block per part
    std::vector<double> hidiag;
    int32_t *abort_sticky = nullptr; // device: set by a pipelined triangular sweep that gave up; cleared by the host only
    int retired = 0;                 // pipelines switched off after an abort (diagnostics: sgm_pc_get "pipeline_retired")
    PcOptions opt = g_opt.pc;        // this preconditioner's options: the defaults at its creation, then sgm_pc_set_option
    // option "ildu_reorder": the factors are those of P A P^T -- on a row partition of P_k A_kk P_k^T for every part k, each
    // part ordering its own diagonal block (no communication; halo columns keep their numbers).  perm = p (1-based, local:
    // row i of the part is row p(i) of the permuted part), rp / zp = right-hand side and result in the permuted order
    // hmap (device, n_halo entries; null: the halo keeps its order) = the part's halo slots re-ordered by the permuted rows they
    // attach to; send_order[k] (device) = where entry j this part sends over its k-th link goes in the RECEIVER's re-ordered halo
    struct Reorder {
        int32_t *perm = nullptr; double *rp = nullptr, *zp = nullptr; int32_t n = 0, colors = 0;
        int32_t *hmap = nullptr; std::vector<int32_t> hmap_host; std::vector<int32_t *> send_order;
    };
    std::vector<Reorder> ro;            // one per part; empty = natural order
    uint64_t ro_serial = 0, ro_pattern = 0;     // the matrix (serial number, pattern version) the orderings were found for
    double reorder_ms[3] = {0, 0, 0};  // last setup: ordering, permuted copy, (factorisation is in the regular phases)
    // the permuted matrix itself, kept (with A's kernel forms) for the Krylov solvers: they run the whole solve in the
    // permuted order -- b and x permuted once each way -- instead of permuting r and z in every apply (in_permuted: vectors
    // handed to pc_apply_parts are in that order already)
    sgm_mat Ap = nullptr;
    uint64_t Ap_serial = 0, Ap_version = 0;            // ... of the matrix it is the permutation of
    bool in_permuted = false;
};

namespace {

// ------------------------------------------------------------------------------ kernels
// rows [row0, row0 + count) of a leaf; the diagonal of global row i sits at local column
// (local row) + dcol of this leaf (dcol = 0 for a plain matrix; row-block minus column-block
// offset for a block of a composite, composite_mat_get_value sparse_matrix_composites.f90:465-485)
__global__ void k_jacobi_setup_csr(int32_t count, int32_t row0, int32_t dcol, const int32_t *__restrict__ rowptr,
                                   const int32_t *__restrict__ col, const double *__restrict__ val,
                                   double *__restrict__ idiag)
{
    int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int32_t i = row0 + t, want = i + dcol;
    double z = 0.0;                                   // get_value: 0 when the entry is absent
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (col[k] == want) z = val[k];
    idiag[t] = 1.0 / z;
}
__global__ void k_jacobi_setup_ell(int32_t count, int32_t row0, int32_t dcol, int32_t n, int32_t max_d,
                                   const int32_t *__restrict__ ecol, const double *__restrict__ eval,
                                   double *__restrict__ idiag)
{
    int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const int32_t i = row0 + t, want = i + dcol;
    // ellpack get_value scans the first degrees(i) slots (ellpack_matrices.f90:232-235);
    // padding repeats the last real neighbour with val 0, real neighbours are unique, so
    // the FIRST hit is the real slot.
    double z = 0.0;
    for (int32_t k = 0; k < max_d; ++k)
        if (ecol[(int64_t)k * n + i] == want) { z = eval[(int64_t)k * n + i]; break; }
    idiag[t] = 1.0 / z;
}
__global__ void k_fill_inf(int32_t count, double *idiag)
{
    int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) idiag[t] = 1.0 / 0.0;
}
// (vectors are 16-byte aligned: 16-byte accesses for the pairs, the odd tail element alone)
__global__ void k_scale_by(int64_t n, const double *__restrict__ d, const double *__restrict__ r,
                           double *__restrict__ z, const int *flag)
{
    if (flag && *flag) return;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n2 = n >> 1;
    const double2 *d2 = reinterpret_cast<const double2 *>(d), *r2 = reinterpret_cast<const double2 *>(r);
    double2 *z2 = reinterpret_cast<double2 *>(z);
    for (int64_t i = gtid; i < n2; i += stride) {            // x = idiag * b
        const double2 a = d2[i], b = r2[i];
        z2[i] = make_double2(a.x * b.x, a.y * b.y);
    }
    if ((n & 1) && gtid == 0) z[n - 1] = d[n - 1] * r[n - 1];
}
__global__ void k_div_by(int64_t n, const double *__restrict__ d, double *__restrict__ x, const int *flag)
{
    if (flag && *flag) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) x[i] = x[i] / d[i];       // x = x / D
}
__global__ void k_copy(int64_t n, const double *__restrict__ s, double *__restrict__ d, const int *flag)
{
    if (flag && *flag) return;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) d[i] = s[i];
}

__global__ void k_perm_gather(int64_t n, double *__restrict__ xp, const double *__restrict__ src,
                              const int32_t *__restrict__ order, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) xp[p] = src[order[p]];            // x = b, in level order
}
__global__ void k_lu_transition(int64_t n, double *__restrict__ xpU, const double *__restrict__ xpL,
                                const int32_t *__restrict__ mapLU, const double *__restrict__ Dp, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) xpU[p] = xpL[mapLU[p]] / Dp[p];   // x = x / D, re-ordered for the U sweep
}
__global__ void k_perm_scatter(int64_t n, double *__restrict__ dst, const double *__restrict__ xp,
                               const int32_t *__restrict__ order, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < n; p += stride) dst[order[p]] = xp[p];
}

// one wide level: one lane per row, dependencies read from global memory
__global__ void k_trsv_wide(const TrsvRec *__restrict__ recs, const int32_t *__restrict__ pq,
                            const double *__restrict__ pv, int32_t begin, int32_t end, double *xp, const int *flag)
{
    if (flag && *flag) return;
    const int32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    const TrsvRec r = recs[p];
    double z = xp[p];
#pragma unroll
    for (int j = 0; j < kInline; ++j)
        if (j < r.cnt) z = z - r.v[j] * xp[r.q[j]];
    for (int32_t k = r.k0 + kInline; k < r.k0 + r.cnt; ++k) z = z - pv[k] * xp[pq[k]];
    xp[p] = z;
}

// the same on the structure-of-arrays copy (all rows of the level have <= kInline dependencies):
// positions and values slot-major, every load coalesced (the 64-byte records cost one cache line per
// lane and load instruction)
template <int C>
__global__ void k_trsv_wide_soa(const int32_t *__restrict__ wq, const double *__restrict__ dv, uint32_t nstride,
                                int32_t begin, int32_t end, double *xp, const int *flag)
{
    if (flag && *flag) return;
    const int32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    double z = xp[p];
    int32_t q[C];
    double v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { q[c] = wq[(size_t)c * nstride + p]; v[c] = dv[(size_t)c * nstride + p]; }
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (q[c] >= 0) z = z - v[c] * xp[q[c]];
    xp[p] = z;
}

// Row-space tables rq / rv, slice-major: the rc slots of 512 consecutive positions lie side by side, so a sweep's tile reads ONE
// contiguous run (rc * 6 KiB for both tables) instead of 2 * rc streams a whole vector apart.  (Measured in round 4 against
// the slot-major layout it replaced: the same time -- the sweeps are not bound by the number of open streams.)  Size <=
// (n + 511) * rc entries.
__host__ __device__ inline size_t rs_at(int c, uint32_t p, int rc) { return ((size_t)(p >> 9) * rc + c) * 512 + (p & 511u); }

// One level in ROW space (factors of a few levels -- what the reference's greedy colouring makes of a matrix: one level per
// colour): out[i] = src[i] (/ D[i]) - sum over the row's entries, stored order, of val * out[row of the entry];
// i = the level's rows, through `order` or -- a level that is a run of consecutive rows -- counted from row0.  No gather
// into level order before the sweeps, no re-ordering between them, no scatter after: the L sweep reads r and writes the
// work vector, the U sweep divides by D as it picks its right-hand side up and writes z.  C = slots read (the most
// entries of a row of the level; -1: `rc` of them in a loop).  Same operations in the same order as k_trsv_wide_soa
// after k_perm_gather / k_lu_transition, so the same bits.
// MODE 0: a level of the L sweep, y_i = r_i - sum val * y(node); MODE 1: the same for a level whose rows have no U entries
// at all (U's level 0 -- with a colour ordering: the last colour), finished on the spot: z_i = y_i / D_i, y_i is never
// stored; MODE 2: a level of the U sweep, z_i = y_i / D_i - sum val * z(node).  Rows below n0 are L's level 0 when that is
// the run of rows 0 .. n0-1 (the first colour): their y IS r, so nobody copies it -- whoever wants y(q), q < n0, reads r(q).
template <int C, int MODE>
__global__ void k_trsv_rows(const int32_t *__restrict__ rq, const double *__restrict__ rv, uint32_t nstride, int rc,
                            const int32_t *__restrict__ order, int32_t row0, int32_t begin, int32_t end, const double *r,
                            double *y, const double *__restrict__ D, double *z, int32_t n0, const int *flag)
{
    if (flag && *flag) return;
    // tile -> workgroup: the dispatcher deals workgroups b, b + 8, ... to one XCD; they take CONSECUTIVE tiles of 256 positions
    // (XCD k: the k-th eighth of the level), so that a row and its neighbours a grid line away -- other tiles, the same
    // vector entries -- meet in one L2 instead of being fetched once per XCD (the grid is 8 * ceil(tiles / 8) workgroups)
    const int32_t tiles_per_xcd = gridDim.x >> 3;
    const int32_t tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    const int32_t p = begin + tile * (int32_t)blockDim.x + threadIdx.x;
    if (p >= end) return;
    const int32_t i = row0 >= 0 ? row0 + (p - begin) : order[p];
    double t;
    if (MODE == 2) t = (i < n0 ? r[i] : y[i]) / D[i];
    else t = r[i];
    auto dep = [&](int32_t q) -> double { return MODE == 2 ? z[q] : (q < n0 ? r[q] : y[q]); };
    if (C >= 0) {
        int32_t q[C > 0 ? C : 1];
        double v[C > 0 ? C : 1];
#pragma unroll
        for (int c = 0; c < C; ++c) {                       // (read once per sweep: past the caches the gathers live in)
            q[c] = __builtin_nontemporal_load(rq + rs_at(c, p, rc));
            v[c] = __builtin_nontemporal_load(rv + rs_at(c, p, rc));
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (q[c] >= 0) t = t - v[c] * dep(q[c]);
    } else {
        for (int c = 0; c < rc; ++c) {
            const int32_t q = rq[rs_at(c, p, rc)];
            if (q < 0) break;                              // (a row's entries fill its first slots)
            t = t - rv[rs_at(c, p, rc)] * dep(q);
        }
    }
    if (MODE == 0) y[i] = t;
    else if (MODE == 1) z[i] = t / D[i];
    else z[i] = t;
}

// PCG's  r = r - alpha q ; z = M^-1 r ; partial r.z  inside the two launches of a TWO-level factorisation (what the greedy
// colouring makes of a 5- / 7-point matrix: L = the rows of colour 2 reading colour 1, U = the rows of colour 1 reading
// colour 2) -- the r update (k_elem<FCgR<2>>: read r, q, write r) and the dot (k_elem<FDot2>: read r, z) cost 43 + 24 us of a
// 403 us iteration at n = 1e7 as launches of their own.  alpha = res2 / dpr from the partial sums like FCgR's prepare.
//   (before them the caller has updated the entry-less rows 0 .. n0-1 -- a streaming launch over a third of the bytes: with
//    r_j - alpha q_j formed on the fly in MODE 1 its gathers doubled and the launch ran at 4.3 TB/s, 120 us)
//   MODE 1 (rows n0 .. n-1): r_i -= alpha q_i ; z_i = (r_i - sum val * r_j) / D_i      [j < n0]
//   MODE 2 (rows 0 .. n0-1): z_i = r_i / D_i - sum val * z_j                            [j >= n0]
// Same statements and operand order per row as FCgR<2> + k_trsv_rows<C, 1 / 2>: r and z bit-identical; the dot is summed per
// block of this grid instead of k_elem's (tree order either way).
template <int C, int MODE>
__global__ __launch_bounds__(kBlock) void k_trsv_rows_cg(const int32_t *__restrict__ rq, const double *__restrict__ rv, uint32_t nstride, int rc,
                                                         const int32_t *__restrict__ order, int32_t row0, int32_t begin, int32_t end, double *r,
                                                         const double *__restrict__ q, ScalarRef res2, ScalarRef dpr, const double *__restrict__ D,
                                                         double *z, double *part, const int *flag, int gen)
{
    __shared__ double red[2 * (kBlock / 64)];
    const int st = flag ? *flag : 0;
    const ScalarRef rs[2] = {res2, dpr};
    double sc[2];
    load_scalars<kBlock, 2>(rs, sc, red);
    if (st && gen >= st) return;
    const double alpha = sc[0] / sc[1];
    double s = 0.0;
    // (tiles of 256 positions; XCD k = workgroups k, k + 8, ... walks the k-th eighth of them in order: k_trsv_rows' map)
    const int32_t tiles = (end - begin + kBlock - 1) / kBlock, tiles_per_xcd = (tiles + 7) >> 3, wg_per_xcd = gridDim.x >> 3;
    const int32_t t_end = min(tiles, ((int32_t)(blockIdx.x & 7) + 1) * tiles_per_xcd);
    for (int32_t tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3); tile < t_end; tile += wg_per_xcd) {
        const int32_t p = begin + tile * kBlock + threadIdx.x;
        if (p >= end) continue;
        const int32_t i = row0 >= 0 ? row0 + (p - begin) : order[p];
        const double ri = MODE == 1 ? r[i] - alpha * q[i] : r[i];
        if (MODE == 1) r[i] = ri;
        double t = MODE == 2 ? ri / D[i] : ri;
        auto dep = [&](int32_t j) -> double { return MODE == 2 ? z[j] : r[j]; };
        if (C >= 0) {
            int32_t qq[C > 0 ? C : 1];
            double v[C > 0 ? C : 1];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                qq[c] = __builtin_nontemporal_load(rq + rs_at(c, p, rc));
                v[c] = __builtin_nontemporal_load(rv + rs_at(c, p, rc));
            }
#pragma unroll
            for (int c = 0; c < C; ++c)
                if (qq[c] >= 0) t = t - v[c] * dep(qq[c]);
        } else {
            for (int c = 0; c < rc; ++c) {
                const int32_t j = rq[rs_at(c, p, rc)];
                if (j < 0) break;
                t = t - rv[rs_at(c, p, rc)] * dep(j);
            }
        }
        const double zi = MODE == 1 ? t / D[i] : t;
        z[i] = zi;
        s += ri * zi;
    }
    const double tot = block_sum<kBlock>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// The same with TWO consecutive rows per lane and 16-byte accesses (8-byte for the row numbers) on every stream -- the level
// is a run of consecutive rows (row0 >= 0) with an even first position and first row, slot stride even; an odd last row
// goes alone.  Half the memory instructions per byte: the one-row form streams at 4.3-4.8 TB/s where the vector kernels
// reach 5.6-6.6.  Per row the same statements in the same order: same bits.
typedef double f64x2p __attribute__((ext_vector_type(2)));
typedef int32_t i32x2p __attribute__((ext_vector_type(2)));
// CODED: the dependency rows come from the 4-bit codes (rcode, rdict) instead of rq: 4 bytes per row instead of 4 * C.
typedef uint32_t u32x2p __attribute__((ext_vector_type(2)));
template <int C, int MODE, bool CODED>
__global__ __launch_bounds__(kBlock) void k_trsv_rows_cg2(const int32_t *__restrict__ rq, const double *__restrict__ rv, int rc,
                                                          const uint32_t *__restrict__ rcode, const int32_t *__restrict__ rdict,
                                                          int32_t row0, int32_t begin, int32_t end, double *r, const double *__restrict__ q,
                                                          ScalarRef res2, ScalarRef dpr, const double *__restrict__ D, double *z, double *part,
                                                          const int *flag, int gen)
{
    __shared__ double red[2 * (kBlock / 64)];
    __shared__ int32_t dl[16];
    if (CODED) {
        if (threadIdx.x < 16) dl[threadIdx.x] = rdict[threadIdx.x];
        __syncthreads();
    }
    const int st = flag ? *flag : 0;
    const ScalarRef rs[2] = {res2, dpr};
    double sc[2];
    load_scalars<kBlock, 2>(rs, sc, red);
    if (st && gen >= st) return;
    const double alpha = sc[0] / sc[1];
    double s = 0.0;
    constexpr int TILE = 2 * kBlock;
    auto dep = [&](int32_t j) -> double { return MODE == 2 ? z[j] : r[j]; };
    // a level that starts at an odd position / row (both odd: the caller checks): its first row alone, the pairs from the next
    const int32_t peel = begin & 1;
    if (peel && blockIdx.x == 0 && threadIdx.x == 0 && begin < end) {
        const int32_t p = begin, i = row0;
        const double ri = MODE == 1 ? r[i] - alpha * q[i] : r[i];
        if (MODE == 1) r[i] = ri;
        double t = MODE == 2 ? ri / D[i] : ri;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int32_t j = rq[rs_at(c, p, rc)];
            if (j >= 0) t = t - rv[rs_at(c, p, rc)] * dep(j);
        }
        const double zi = MODE == 1 ? t / D[i] : t;
        z[i] = zi;
        s += ri * zi;
    }
    begin += peel;
    row0 += peel;
    const int32_t tiles = (end - begin + TILE - 1) / TILE, tiles_per_xcd = (tiles + 7) >> 3, wg_per_xcd = gridDim.x >> 3;
    const int32_t t_end = min(tiles, ((int32_t)(blockIdx.x & 7) + 1) * tiles_per_xcd);
    for (int32_t tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3); tile < t_end; tile += wg_per_xcd) {
        const int32_t p = begin + tile * TILE + 2 * (int32_t)threadIdx.x;
        if (p >= end) continue;
        const int32_t i = row0 + (p - begin);
        if (p + 1 < end) {
            const f64x2p rr = *reinterpret_cast<const f64x2p *>(r + i), dd = *reinterpret_cast<const f64x2p *>(D + i);
            f64x2p qo;
            qo.x = 0.0; qo.y = 0.0;
            if (MODE == 1) qo = *reinterpret_cast<const f64x2p *>(q + i);
            i32x2p jj[C > 0 ? C : 1];
            f64x2p vv[C > 0 ? C : 1];
            u32x2p cw;
            if (CODED) cw = __builtin_nontemporal_load(reinterpret_cast<const u32x2p *>(rcode + p));
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (CODED) {
                    const uint32_t na = (cw.x >> (4 * c)) & 15u, nb = (cw.y >> (4 * c)) & 15u;
                    jj[c].x = na == 15u ? -1 : i + dl[na];
                    jj[c].y = nb == 15u ? -1 : i + 1 + dl[nb];
                } else
                    jj[c] = __builtin_nontemporal_load(reinterpret_cast<const i32x2p *>(rq + rs_at(c, p, rc)));
                vv[c] = __builtin_nontemporal_load(reinterpret_cast<const f64x2p *>(rv + rs_at(c, p, rc)));
            }
            const double ra = MODE == 1 ? rr.x - alpha * qo.x : rr.x, rb = MODE == 1 ? rr.y - alpha * qo.y : rr.y;
            if (MODE == 1) {
                f64x2p rn; rn.x = ra; rn.y = rb;
                *reinterpret_cast<f64x2p *>(r + i) = rn;
            }
            double ta = MODE == 2 ? ra / dd.x : ra, tb = MODE == 2 ? rb / dd.y : rb;
            // (the two rows' entries side by side: every slot's operands are requested before either row uses them)
            double da[C > 0 ? C : 1], db[C > 0 ? C : 1];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                da[c] = jj[c].x >= 0 ? dep(jj[c].x) : 0.0;
                db[c] = jj[c].y >= 0 ? dep(jj[c].y) : 0.0;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (jj[c].x >= 0) ta = ta - vv[c].x * da[c];
                if (jj[c].y >= 0) tb = tb - vv[c].y * db[c];
            }
            f64x2p zn;
            zn.x = MODE == 1 ? ta / dd.x : ta;
            zn.y = MODE == 1 ? tb / dd.y : tb;
            *reinterpret_cast<f64x2p *>(z + i) = zn;
            s += ra * zn.x;
            s += rb * zn.y;
        } else {
            const double ri = MODE == 1 ? r[i] - alpha * q[i] : r[i];
            if (MODE == 1) r[i] = ri;
            double t = MODE == 2 ? ri / D[i] : ri;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int32_t j = rq[rs_at(c, p, rc)];
                if (j >= 0) t = t - rv[rs_at(c, p, rc)] * dep(j);
            }
            const double zi = MODE == 1 ? t / D[i] : t;
            z[i] = zi;
            s += ri * zi;
        }
    }
    const double tot = block_sum<kBlock>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// k_trsv_rows for a level that is a run of consecutive rows: two rows per lane, 16-byte accesses, the dependency rows from
// the 4-bit codes where the factor has them (CODED) -- the forms k_trsv_rows_cg2 measured (83 / 69 us against 98 / 79 with
// one row per lane and 4-byte row numbers, n = 1e7).  Per row the statements of k_trsv_rows in their order: same bits.
template <int C, int MODE, bool CODED>
__global__ __launch_bounds__(kBlock) void k_trsv_rows2(const int32_t *__restrict__ rq, const double *__restrict__ rv, int rc,
                                                       const uint32_t *__restrict__ rcode, const int32_t *__restrict__ rdict, int32_t row0,
                                                       int32_t begin, int32_t end, const double *r, double *y, const double *__restrict__ D,
                                                       double *z, int32_t n0, const int *flag)
{
    __shared__ int32_t dl[16];
    if (flag && *flag) return;
    if (CODED) {
        if (threadIdx.x < 16) dl[threadIdx.x] = rdict[threadIdx.x];
        __syncthreads();
    }
    auto dep = [&](int32_t j) -> double { return MODE == 2 ? z[j] : (j < n0 ? r[j] : y[j]); };
    auto rhs = [&](int32_t k) -> double { return MODE == 2 ? (k < n0 ? r[k] : y[k]) : r[k]; };
    // a level that starts at an odd position / row (both odd, its rows on one side of n0: the caller checks): the first row alone
    const int32_t peel = begin & 1;
    if (peel && blockIdx.x == 0 && threadIdx.x == 0 && begin < end) {
        const int32_t p = begin, i = row0;
        double t = MODE == 2 ? rhs(i) / D[i] : rhs(i);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int32_t j = rq[rs_at(c, p, rc)];
            if (j >= 0) t = t - rv[rs_at(c, p, rc)] * dep(j);
        }
        if (MODE == 0) y[i] = t;
        else if (MODE == 1) z[i] = t / D[i];
        else z[i] = t;
    }
    begin += peel;
    row0 += peel;
    const int32_t tiles_per_xcd = gridDim.x >> 3;
    const int32_t tile = (blockIdx.x & 7) * tiles_per_xcd + (blockIdx.x >> 3);
    const int32_t p = begin + tile * 2 * kBlock + 2 * (int32_t)threadIdx.x;
    if (p >= end) return;
    const int32_t i = row0 + (p - begin);
    if (p + 1 < end) {
        // (i even: rows i, i + 1 lie on one side of the even n0 or straddle nothing -- n0 odd is the caller's scalar case)
        const double *src = MODE == 2 ? (i < n0 ? r : y) : r;
        const f64x2p rr = *reinterpret_cast<const f64x2p *>(src + i);
        f64x2p dd;
        dd.x = 1.0; dd.y = 1.0;
        if (MODE != 0) dd = *reinterpret_cast<const f64x2p *>(D + i);
        i32x2p jj[C > 0 ? C : 1];
        f64x2p vv[C > 0 ? C : 1];
        u32x2p cw;
        if (CODED) cw = __builtin_nontemporal_load(reinterpret_cast<const u32x2p *>(rcode + p));
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (CODED) {
                const uint32_t na = (cw.x >> (4 * c)) & 15u, nb = (cw.y >> (4 * c)) & 15u;
                jj[c].x = na == 15u ? -1 : i + dl[na];
                jj[c].y = nb == 15u ? -1 : i + 1 + dl[nb];
            } else
                jj[c] = __builtin_nontemporal_load(reinterpret_cast<const i32x2p *>(rq + rs_at(c, p, rc)));
            vv[c] = __builtin_nontemporal_load(reinterpret_cast<const f64x2p *>(rv + rs_at(c, p, rc)));
        }
        double ta = MODE == 2 ? rr.x / dd.x : rr.x, tb = MODE == 2 ? rr.y / dd.y : rr.y;
        double da[C > 0 ? C : 1], db[C > 0 ? C : 1];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            da[c] = jj[c].x >= 0 ? dep(jj[c].x) : 0.0;
            db[c] = jj[c].y >= 0 ? dep(jj[c].y) : 0.0;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            if (jj[c].x >= 0) ta = ta - vv[c].x * da[c];
            if (jj[c].y >= 0) tb = tb - vv[c].y * db[c];
        }
        f64x2p out;
        out.x = MODE == 1 ? ta / dd.x : ta;
        out.y = MODE == 1 ? tb / dd.y : tb;
        *reinterpret_cast<f64x2p *>((MODE == 0 ? y : z) + i) = out;
    } else {
        double t = MODE == 2 ? rhs(i) / D[i] : rhs(i);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int32_t j = rq[rs_at(c, p, rc)];
            if (j >= 0) t = t - rv[rs_at(c, p, rc)] * dep(j);
        }
        if (MODE == 0) y[i] = t;
        else if (MODE == 1) z[i] = t / D[i];
        else z[i] = t;
    }
}

// a run of narrow levels [l0, l1) walked by ONE workgroup.  The row records and right-hand
// sides are independent of the solve, so they are requested D levels ahead (registers; one
// HBM round trip is ~2 us, one level's arithmetic a fraction of that); results of the current
// run live in an LDS ring indexed by position, so the dependencies of the next level are LDS
// reads; anything older than the ring (or produced before this run) is read from xp, which
// the workgroup fence + barrier before a ring wrap keeps valid.  RPT = rows per lane and
// level (levels of up to RPT*1024 rows); RPT*D records are in flight per lane.
template <int RPT, int D>
__global__ __launch_bounds__(kTrsvBlock) void k_trsv_walk(const TrsvRec *__restrict__ recs,
                                                          const int32_t *__restrict__ pq,
                                                          const double *__restrict__ pv,
                                                          const int32_t *__restrict__ level_ptr, int32_t l0,
                                                          int32_t l1, int32_t n, double *xp, const int *flag)
{
    // Every prefetch load and every result store is issued by ALL lanes on EVERY level (lanes
    // without a row use a clamped record and the scratch slots xp[n + lane]): the compiler can
    // then count the younger requests exactly and waits for a prefetched record with
    // s_waitcnt vmcnt(k > 0); a conditional load or store in the loop would turn every wait
    // into vmcnt(0), i.e. one HBM round trip per level.
    __shared__ double ring[kRing];
    if (flag && *flag) return;
    const int tid = threadIdx.x;
    TrsvRec pre[D][RPT];
    double z0pre[D][RPT];
    int32_t lb[D], le[D];
    // a level's bounds are a scalar load: requested at the top of a step (overlaps the LDS
    // reads), consumed by the row requests at its end
    auto bounds = [&](int32_t l, int32_t &b, int32_t &e) {
        const int32_t lc = min(l, l1 - 1);           // past the run: an empty level (requests are clamped)
        b = level_ptr[lc];
        e = level_ptr[lc + 1];
        if (l >= l1) b = e;
    };
    auto fetch = [&](int slot, int32_t b, int32_t e) {
        lb[slot] = b;
        le[slot] = e;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int32_t p = b + tid + r * kTrsvBlock;
            const bool ok = p < e;
            pre[slot][r] = recs[ok ? p : n - 1];
            z0pre[slot][r] = xp[ok ? p : n + tid + r * kTrsvBlock];   // right-hand side: only this row ever writes it
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) {
        int32_t b, e;
        bounds(l0 + j, b, e);
        fetch(j, b, e);
    }
    int32_t fpos = level_ptr[l0];   // every position < fpos is visible in xp (written before a workgroup fence)
    for (int32_t l = l0; l < l1; l += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {                // levels past l1 are empty: same instruction stream
            const int32_t b = lb[j], e = le[j];
            int32_t nb, ne;
            bounds(l + j + D, nb, ne);
            if (e - fpos > kRing) {
                // the ring is about to lose positions that were never fenced: make all stores of
                // this run visible in xp first (rare: once per ~2 widest levels at most)
                __threadfence_block();
                __syncthreads();
                fpos = b;
            }
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                const int32_t p = b + tid + r * kTrsvBlock;
                const bool ok = p < e;
                const int32_t cnt = ok ? pre[j][r].cnt : 0;
                double z = z0pre[j][r];
                bool fast = cnt <= kInline;
#pragma unroll
                for (int i = 0; i < kInline; ++i) fast = fast & ((i >= cnt) | (pre[j][r].q[i] >= fpos));
                if (fast) {                          // every dependency is in the LDS ring: no memory wait
#pragma unroll
                    for (int i = 0; i < kInline; ++i) {
                        const double t = z - pre[j][r].v[i] * ring[pre[j][r].q[i] & (kRing - 1)];
                        z = i < cnt ? t : z;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kInline; ++i)      // (static indices: the records stay in registers)
                        if (i < cnt) {
                            const int32_t q = pre[j][r].q[i];
                            const double xv = q >= fpos ? ring[q & (kRing - 1)] : xp[q];
                            z = z - pre[j][r].v[i] * xv;
                        }
                    for (int32_t k = pre[j][r].k0 + kInline; k < pre[j][r].k0 + cnt; ++k) {
                        const int32_t q = pq[k];
                        const double xv = q >= fpos ? ring[q & (kRing - 1)] : xp[q];
                        z = z - pv[k] * xv;
                    }
                }
                if (ok) ring[p & (kRing - 1)] = z;
                xp[ok ? p : n + tid + r * kTrsvBlock] = z;     // drains in the background; readers use the ring
            }
            // slot j is free again: request level l+j+D into the same registers (issued after the
            // last use, so the compiler needs no second register set and no copies at the back-edge)
            fetch(j, nb, ne);
            // level barrier on the LDS ring only: the global stores above stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
}

// The same walk for runs whose rows all have <= kInline dependencies, every one of them within
// kRing positions below the end of the row's own level (checked on the host at setup; true for
// grid-like factors).  The ring is pre-loaded with the kRing results that precede the run, so
// EVERY dependency is an LDS read: no fence, no branch, and no memory request inside the loop
// other than the D-level-ahead prefetch and the result store -- the loop is one straight
// instruction stream, which lets the compiler wait for a prefetched row with an exact
// s_waitcnt vmcnt(k).  One workgroup on one CU is bound by the CU's memory pipeline (measured:
// 64-byte records, one per lane = 64 cache lines per load instruction, ~0.9 us per 1000-row
// level), so these runs read a structure-of-arrays copy instead: per row ONE 8-byte word of
// four 16-bit ring slots (position & (kRing-1); kRing = "no entry", a slot that holds 0.0 and
// is paired with the value 0.0) and C values, slot-major -- every load is a coalesced 8 bytes
// per lane, C + 3 memory instructions and ~3 ALU instructions per dependency.
// A = adjacent rows per lane (1 or 2): with A = 2 a lane owns rows 2t and 2t+1 of the level and
// every load moves 16 bytes per lane -- half the memory instructions for the same bytes.
template <class T, int A> struct alignas(sizeof(T)) RowPack { T v[A]; };
template <int TB, int RPT, int D, int C, int A>
__global__ __launch_bounds__(TB) void k_trsv_walk_ring(const uint64_t *__restrict__ dq,
                                                       const uint32_t *__restrict__ dq32,
                                                       const double *__restrict__ dv, uint32_t nstride,
                                                       const int32_t *__restrict__ level_ptr, int32_t l0,
                                                       int32_t l1, int32_t n, double *xp, const int *flag)
{
    // ring[kRing] is a constant 0.0 (the slot absent dependencies point at, with value 0.0:
    // z - 0.0*0.0 == z for every z, so they need no branch); ring[kRing+1+..]: parking
    __shared__ double ring[kRing + 1 + 2 * kTrsvBlock];
    if (flag && *flag) return;
    const uint32_t tid = threadIdx.x;
    {
        const int32_t base = level_ptr[l0];
        for (int32_t q = base - 1 - (int32_t)tid; q >= 0 && q >= base - kRing; q -= TB) ring[q & (kRing - 1)] = xp[q];
        if (tid == 0) ring[kRing] = 0.0;
    }
    // byte offsets fit 32 bits (checked on the host): scalar base + 32-bit lane offset addressing
    // (C <= 2 reads the 32-bit copy of the slot words: of a 64-bit word only the low half would
    //  be used, the register allocator would re-use the idle half, and a write to a register
    //  with a load in flight has to wait for that load)
    using WQ = typename std::conditional<(C <= 2), uint32_t, uint64_t>::type;
    const char *dqb = C <= 2 ? reinterpret_cast<const char *>(dq32) : reinterpret_cast<const char *>(dq);
    const char *dvb[C];
#pragma unroll
    for (int i = 0; i < C; ++i) dvb[i] = reinterpret_cast<const char *>(dv + (size_t)i * nstride);
    char *xpb = reinterpret_cast<char *>(xp);
    RowPack<WQ, A> wq[D][RPT];
    RowPack<double, A> wv[D][RPT][C], z0pre[D][RPT];
    int32_t lb[D], le[D];
    auto bounds = [&](int32_t l, int32_t &b, int32_t &e) {
        const int32_t lc = min(l, l1 - 1);           // past the run: an empty level
        b = level_ptr[lc];
        e = level_ptr[lc + 1];
        if (l >= l1) b = e;
    };
    auto fetch = [&](int slot, int32_t b, int32_t e) {
        lb[slot] = b;
        le[slot] = e;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            // lanes past the level's end read the rows that follow (the arrays are padded by
            // kNarrow entries); what they compute lands in the parking slots
            const uint32_t off = ((uint32_t)b + A * (tid + r * TB)) * 8u;
            wq[slot][r] = *reinterpret_cast<const RowPack<WQ, A> *>(dqb + (C <= 2 ? off / 2 : off));
#pragma unroll
            for (int i = 0; i < C; ++i) wv[slot][r][i] = *reinterpret_cast<const RowPack<double, A> *>(dvb[i] + off);
            z0pre[slot][r] = *reinterpret_cast<const RowPack<double, A> *>(xpb + off);
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) {
        int32_t b, e;
        bounds(l0 + j, b, e);
        fetch(j, b, e);
    }
    __syncthreads();
    const char *ringb = reinterpret_cast<const char *>(ring);
    for (int32_t l = l0; l < l1; l += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {                // levels past l1 are empty: same instruction stream
            const int32_t b = lb[j], e = le[j];
            int32_t nb, ne;
            bounds(l + j + D, nb, ne);
#pragma unroll
            for (int r = 0; r < RPT; ++r)
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const uint32_t lane_row = A * (tid + r * TB) + a;      // < A * RPT * TB <= 2 * kTrsvBlock ... kNarrow
                    const uint32_t p = (uint32_t)b + lane_row;
                    const bool ok = p < (uint32_t)e;
                    double z = z0pre[j][r].v[a];
#pragma unroll
                    for (int i = 0; i < C; ++i) {
                        const uint32_t slot = (uint32_t)(wq[j][r].v[a] >> (16 * i)) & 0xffffu;     // ring slot of the dependency
                        z = z - wv[j][r][i].v[a] * *reinterpret_cast<const double *>(ringb + slot * 8u);
                    }
                    // rows past the level's end: results go to parking slots nobody reads
                    const uint32_t park = A * tid + a;
                    *reinterpret_cast<double *>(const_cast<char *>(ringb) + (ok ? (p & (kRing - 1)) : kRing + 1 + park) * 8u) = z;
                    *reinterpret_cast<double *>(xpb + (ok ? p : (uint32_t)n + lane_row) * 8u) = z;
                }
            // slot j is free again: request level l+j+D into the same registers (issued after the
            // last use, so the compiler needs no second register set and no copies at the back-edge)
            fetch(j, nb, ne);
            // level barrier on the LDS ring only: the global stores above stay in flight
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // keep the next level's address arithmetic below this point: hoisted above, it would
            // pull the wait for that level's (still in flight) row up here
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---- strip-pipelined triangular solve (GridTri) ------------------------------------------------
// ONE launch per triangular solve, one workgroup of two waves per 64-column strip, all strips running at once:
//   chain wave   walks its strip top to bottom.  A step = shift the previous results one lane up (DPP wave_shr,
//                no LDS on the chain), two products, two subtractions in the row's STORED order (an absent
//                dependency contributes an exact 0.0 whatever its operand holds).  Its only vector-memory traffic is
//                the 32-byte records DEPTH steps ahead (static register slots: exact vmcnt waits) and the result
//                store; lane 0's left neighbours come out of an LDS ring, lane 63's results go into another.
//   helper waves talk to the neighbours: one forwards this strip's edge values to memory, one loads the left strip's
//                into the LDS ring -- sc1 (agent-scope relaxed) accesses, valid across XCDs.  The edge values are their
//                own flags (kEdgeEmpty until written): a hand-off costs one memory round trip.
// Strip ib only ever waits for strip ib-1 -- a workgroup with a smaller index, dispatched no later -- so the launch
// cannot deadlock; every wait loop is bounded all the same and raises the abort word instead of hanging.
constexpr int kStripDepth = 32;          // records in flight per lane (16: 0.91 / 1.81 ms per PCG iteration at 1000^2 / 2000^2, 32: 0.85 / 1.66)
constexpr int kStripChunk = 8;           // steps between LDS hand-offs
constexpr int kStripRing = 512;          // edge values the LDS rings hold (steps)
constexpr int kStripSpinLimit = 1 << 22;
// A wait that gives up marks the sweep (abort_word: cleared by the next sweep's gather, read by the setup self-check) AND the
// preconditioner's sticky word, which only the host clears: the solvers and sgm_pc_apply read it whenever they synchronise
// anyway, redo the work with the level walkers and retire the pipeline for this handle -- a result the library itself
// spoiled never reaches the caller (sgm::pc_abort_word / pc_retire_pipelines).
__device__ inline void raise_abort(int32_t *abort_word, int32_t *sticky)
{
    __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sticky) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kEdgePad = 72;             // an edge row: S + 64 values (a strip reads its left neighbour's step t + 63), 2 clocks
// "not yet written": a SIGNALLING NaN no subtraction can produce (arithmetic quiets NaNs), so the edge values are their own flags
constexpr unsigned long long kEdgeEmpty = 0x7FF4A5A5A5A5A5A5ull;
typedef double f64x2s __attribute__((ext_vector_type(2)));
__device__ inline double dpp_shift_up(double v, double lane0)
{
    // lane l receives lane l-1's v (wave_shr:1 crosses the 16-lane DPP rows on gfx9); lane 0 receives lane0
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_update_dpp(__double2loint(lane0), lo, 0x138, 0xf, 0xf, false);
    const int shi = __builtin_amdgcn_update_dpp(__double2hiint(lane0), hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(shi, slo);
}
// ORDER: 0 = every row subtracts its r-w term first, 1 = every row its r-1 term first, 2 = per-row flag (bit 2)
template <int DEPTH, int CH, int ORDER, int LA = DEPTH>
__global__ __launch_bounds__(192) void k_trsv_strip(int32_t NI, int32_t S, const StripRec *__restrict__ rec, double *__restrict__ xp,
                                                    double *edge, int32_t *progress, const int *flag, int one_xcd, int spin_limit,
                                                    int32_t *sticky)
{
    __shared__ double in_ring[kStripRing], out_ring[kStripRing], out_scratch[64 + CH];
    __shared__ int in_avail, out_count, out_sent, lds_abort; // steps of left-edge values available / produced by the chain / forwarded
    if (flag && *flag) return;
    // one_xcd: the grid is 8 x NI and only every eighth workgroup works, so that all strips sit on ONE XCD (round-robin
    // dispatch) and the neighbour hand-offs are served by one L2; placement only, any mapping is correct
    if (one_xcd && (blockIdx.x & 7)) return;
    const int lane = threadIdx.x & 63;
    const bool chain = threadIdx.x < 64;
    const int32_t ib = one_xcd ? blockIdx.x >> 3 : blockIdx.x;
    int32_t *abort_word = progress + NI;
    if (threadIdx.x == 0) { in_avail = ib == 0 ? S + kStripRing : 0; out_count = 0; out_sent = 0; lds_abort = 0; }
    for (int q = threadIdx.x; q < kStripRing; q += 192) { in_ring[q] = 0.0; out_ring[q] = 0.0; }
    __syncthreads();
    if (chain) {
        const int64_t base = (int64_t)ib * S * 64;
        const f64x2s *R = reinterpret_cast<const f64x2s *>(rec + base + lane);     // 2 x 16 bytes per record
        double *X = xp + base + lane;
        f64x2s ra[DEPTH], rb[DEPTH];
        auto fetch = [&](int slot, int32_t t) {
            const int32_t tc = min(t, S - 1);
            ra[slot] = R[(int64_t)tc * 128];          // (plain loads: the records are re-read by every apply)
            rb[slot] = R[(int64_t)tc * 128 + 1];
        };
#pragma unroll
        // look-ahead LA steps of the DEPTH register slots: two loads and a store per step, and vmcnt counts to 63
        for (int j = 0; j < LA; ++j) fetch(j, j);
        const long long clk0 = wall_clock64();
        double prev = 0.0;
        double eE[CH];                       // lane 0's left neighbours of the current chunk (read out of the ring at its start)
        double *ow = &out_scratch[lane];     // where this lane's results of the current chunk go in LDS
        for (int32_t t0 = 0; t0 < S; t0 += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int32_t t = t0 + j;
                if (j % CH == 0) {
                    // lane 0's left neighbours of this chunk must be in the ring
                    int spins = 0;
                    // ... and the helper must have forwarded what the out ring is about to overwrite
                    while (__hip_atomic_load(&in_avail, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t + CH ||
                           t + CH - __hip_atomic_load(&out_sent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > kStripRing - CH) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > spin_limit || __hip_atomic_load(&lds_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            if (lane == 0) raise_abort(abort_word, sticky);
                            return;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < CH; ++u) eE[u] = in_ring[(t + u) & (kStripRing - 1)];
                    ow = lane == 63 ? &out_ring[t & (kStripRing - 1)] : &out_scratch[lane];
                }
                const double left = dpp_shift_up(prev, eE[j % CH]);
                double z = rb[j].x;
                if (ORDER == 2) {                       // per-row order: flag word (bit0 has r-w, bit1 has r-1, bit2 r-1 first)
                    const uint32_t cc = (uint32_t)__double_as_longlong(rb[j].y);
                    const double pS = (cc & 1u) ? ra[j].x * prev : 0.0;
                    const double pW = (cc & 2u) ? ra[j].y * left : 0.0;
                    const bool wfirst = (cc & 4u) != 0;
                    z = z - (wfirst ? pW : pS);
                    z = z - (wfirst ? pS : pW);
                } else {                                // uniform order: the code word holds two 32-bit AND masks (all ones = present)
                    const uint64_t mk = (uint64_t)__double_as_longlong(rb[j].y);
                    const uint32_t mS = (uint32_t)mk, mW = (uint32_t)(mk >> 32);
                    const double rS = ra[j].x * prev, rW = ra[j].y * left;
                    const double pS = __hiloint2double(__double2hiint(rS) & (int)mS, __double2loint(rS) & (int)mS);
                    const double pW = __hiloint2double(__double2hiint(rW) & (int)mW, __double2loint(rW) & (int)mW);
                    z = z - (ORDER == 1 ? pW : pS);
                    z = z - (ORDER == 1 ? pS : pW);
                }
                __builtin_nontemporal_store(z, X + (int64_t)t * 64);
                ow[j % CH] = z;                         // lane 63: the out ring; the other lanes: scratch
                prev = z;
                fetch((j + LA) % DEPTH, t + LA);
                if (j % CH == CH - 1 && lane == 0)
                    __hip_atomic_store(&out_count, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        // diagnostics (sgm_pc_get "strip_clocks"): start / end of this strip's chain in the two unused tail slots of its edge row
        if (lane == 0) {
            long long *tail = reinterpret_cast<long long *>(edge + (int64_t)ib * (S + kEdgePad) + S + 64);
            tail[0] = clk0;
            tail[1] = wall_clock64();
        }
        return;
    }
    // ---- helper waves: wave 1 forwards this strip's edge values, wave 2 fetches the left strip's
    const bool forwarder = threadIdx.x < 128;
    double *my_edge = edge + (int64_t)ib * (S + kEdgePad);
    const double *left_edge = edge + (int64_t)(ib > 0 ? ib - 1 : 0) * (S + kEdgePad);
    int spins = 0;
    if (forwarder) {
        // edge values are their own flags (kEdgeEmpty until written, reset before every sweep): no progress word, no wait
        // for the stores to be acknowledged
        int32_t sent = 0;            // steps of this strip's edge values stored
        while (sent < S) {
            const int32_t made = min(__hip_atomic_load(&out_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP), S);
            if (made > sent) {
                for (int32_t q = sent + lane; q < made; q += 64)
                    __hip_atomic_store(my_edge + q, out_ring[q & (kStripRing - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) {
                    __hip_atomic_store(progress + ib, made, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);         // (diagnostics only)
                    __hip_atomic_store(&out_sent, made, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                sent = made;
                spins = 0;
                continue;
            }
            __builtin_amdgcn_s_sleep(4);            // (the helpers share the CU's LDS and memory pipeline with the chain wave: poll gently)
            if (++spins > spin_limit || __hip_atomic_load(&lds_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                if (lane == 0) raise_abort(abort_word, sticky);
                return;
            }
        }
        // the right strip reads 63 entries past the last step (padding rows there: any value that is not kEdgeEmpty)
        __hip_atomic_store(my_edge + S + lane, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (ib == 0) return;
    int32_t got = 0;                 // steps of left-edge values copied into the ring; lane 0 at step t needs the left strip's step t + 63
    while (got < S) {
        // never more than a ring ahead of what the chain has consumed (it has produced out_count steps)
        const int32_t done = __hip_atomic_load(&out_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int32_t room = done + kStripRing - 2 * CH;
        if (got - done > 128) { __builtin_amdgcn_s_sleep(32); continue; }       // comfortably ahead of the chain: stay out of its way
        const int32_t cnt = min(64, min(S, room) - got);
        if (cnt > 0) {
            // ONE memory round trip per look: load the next entries and keep the leading ones that have been written
            double v = 0.0;
            if (lane < cnt) v = __hip_atomic_load(left_edge + got + 63 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = lane >= cnt || (unsigned long long)__double_as_longlong(v) != kEdgeEmpty;
            const unsigned long long miss = ~__ballot(ok);
            const int32_t nvalid = miss ? min(cnt, (int32_t)__builtin_ctzll(miss)) : cnt;
            if (nvalid > 0) {
                if (lane < nvalid) in_ring[(got + lane) & (kStripRing - 1)] = v;
                got += nvalid;
                if (lane == 0) __hip_atomic_store(&in_avail, got >= S ? S + kStripRing : got, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                spins = 0;
                continue;
            }
        }
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ||
            __hip_atomic_load(&lds_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            if (lane == 0) {
                raise_abort(abort_word, sticky);
                __hip_atomic_store(&lds_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return;
        }
    }
}
// position-space gather / hand-over / scatter of the strip path (padding positions hold 0)
// (gather and transition also clear the progress words of the sweep that follows)
__global__ void k_grid_gather(int64_t np, StripRec *__restrict__ rec, const double *__restrict__ src,
                              const int32_t *__restrict__ row, int32_t *__restrict__ progress, int32_t nprog,
                              unsigned long long *__restrict__ edge, int64_t nedge, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = p; q < nprog; q += stride) progress[q] = 0;
    for (int64_t q = p; q < nedge; q += stride) edge[q] = kEdgeEmpty;
    for (; p < np; p += stride) { const int32_t r = row[p]; rec[p].rhs = r >= 0 ? src[r] : 0.0; }
}
__global__ void k_grid_transition(int64_t np, StripRec *__restrict__ recU, const double *__restrict__ xpL,
                                  const int32_t *__restrict__ mapLU, const double *__restrict__ Dp, int32_t *__restrict__ progress,
                                  int32_t nprog, unsigned long long *__restrict__ edge, int64_t nedge, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = p; q < nprog; q += stride) progress[q] = 0;
    for (int64_t q = p; q < nedge; q += stride) edge[q] = kEdgeEmpty;
    for (; p < np; p += stride) { const int32_t q = mapLU[p]; recU[p].rhs = q >= 0 ? xpL[q] / Dp[p] : 0.0; }   // x = x / D
}
__global__ void k_grid_scatter(int64_t np, double *__restrict__ dst, const double *__restrict__ xp,
                               const int32_t *__restrict__ row, const int *flag)
{
    if (flag && *flag) return;
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < np; p += stride) { const int32_t r = row[p]; if (r >= 0) dst[r] = xp[p]; }
}

// ---- ILDU(0) on the device ----------------------------------------------------------------------------------------
// get_value / set_value / add_value of the reference's csr_matrix on one row of a factor (cs_matrices.f90: a scan of the
// row; the LAST matching entry answers a get, EVERY matching entry takes a set / add)
__host__ __device__ inline double row_get(const int32_t *node, const double *val, int32_t b, int32_t e, int32_t j)
{
    double z = 0.0;
    for (int32_t k = b; k < e; ++k)
        if (node[k] == j) z = val[k];
    return z;
}
__host__ __device__ inline void row_set(const int32_t *node, double *val, int32_t b, int32_t e, int32_t j, double z)
{
    for (int32_t k = b; k < e; ++k)
        if (node[k] == j) val[k] = z;
}
__host__ __device__ inline void row_add(const int32_t *node, double *val, int32_t b, int32_t e, int32_t j, double z)
{
    for (int32_t k = b; k < e; ++k)
        if (node[k] == j) val[k] = val[k] + z;
}

// incomplete_ldu_sparsity_pattern, level 0 (ldu_solvers.f90:397-440): entries of A in stored order, i > j -> L,
// j > i -> U.  Two passes over the rows of the part's diagonal block (columns >= ncol_own are halo slots: dropped):
// counts (an exclusive scan between the launches makes the row pointers), fill.
__global__ void k_ildu_count(int32_t n, int32_t ncol_own, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                             int32_t *__restrict__ lcnt, int32_t *__restrict__ ucnt, int32_t *longest)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    int32_t l = 0, u = 0;
    if (i < n)
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int32_t j = col[k];
            if (j >= ncol_own) continue;
            l += j < i;
            u += j > i;
        }
    lcnt[i] = l;                 // (slot n: 0 -- the scan's total lands there)
    ucnt[i] = u;
    if (l) atomicMax(longest, l);
    if (u) atomicMax(longest + 1, u);
}
__global__ void k_ildu_split(int32_t n, int32_t ncol_own, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                             const int32_t *__restrict__ Lptr, int32_t *__restrict__ Lnode,
                             const int32_t *__restrict__ Uptr, int32_t *__restrict__ Unode)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t l = Lptr[i], u = Uptr[i];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j >= ncol_own) continue;
        if (j < i) Lnode[l++] = j;
        else if (j > i) Unode[u++] = j;
    }
}

// Dependency levels of a strictly triangular pattern on the device, for factors of a FEW levels (colour orderings):
// level(i) = 1 + max level(node) over the row's entries, relaxed in place until nothing moves (<= levels sweeps; values
// only grow and never pass the true level).  flags[0]: something moved; flags[1]: a level reached `cap` -- too many
// levels for this path, the host computes them.
__global__ void k_level_relax(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ node, int32_t *level,
                              int32_t cap, int32_t *flags)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t lv = 0;
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) lv = max(lv, level[node[k]] + 1);
    if (lv != level[i]) {
        level[i] = lv;
        flags[0] = 1;
        if (lv >= cap) flags[1] = 1;
    }
}
// (a handful of levels: the counts are gathered per workgroup in LDS first -- millions of atomics on two addresses crawl)
__global__ void k_level_hist(int32_t n, const int32_t *__restrict__ level, int32_t *__restrict__ count, int32_t *__restrict__ rows)
{
    __shared__ int32_t h[kRowLevels + 2];
    for (int q = threadIdx.x; q < kRowLevels + 2; q += blockDim.x) h[q] = 0;
    __syncthreads();
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        atomicAdd(&h[level[i]], 1);
        rows[i] = i;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < kRowLevels + 2; q += blockDim.x)
        if (h[q]) atomicAdd(count + q, h[q]);
}
// per level (positions [begin[l], begin[l+1]) of the level order): most entries of a row, whether its rows are consecutive,
// its first row
__global__ void k_level_info(int32_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ level,
                             const int32_t *__restrict__ ptr, const int32_t *__restrict__ begin, int32_t *cmax, int32_t *notrun,
                             int32_t *first)
{
    __shared__ int32_t m[kRowLevels + 2], nr[kRowLevels + 2];
    for (int q = threadIdx.x; q < kRowLevels + 2; q += blockDim.x) { m[q] = 0; nr[q] = 0; }
    __syncthreads();
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        const int32_t i = order[p], l = level[i];
        atomicMax(&m[l], ptr[i + 1] - ptr[i]);
        if (p == begin[l]) first[l] = i;
        else if (order[p - 1] + 1 != i) nr[l] = 1;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < kRowLevels + 2; q += blockDim.x) {
        if (m[q]) atomicMax(cmax + q, m[q]);
        if (nr[q]) notrun[q] = 1;
    }
}

// grid_width on the device (the host version below reads a host copy of the pattern; this one keeps it where it is).
// Pass 1: info[0] / info[1] = smallest / largest dependency distance > 1, info[2] = some row breaks the shape (a
// dependency on the wrong side, more than two entries, the same column twice).  Pass 2, with the width w those agree on:
// info[3] = an r-1 / r+1 dependency across a grid line, or a distance that is neither 1 nor w.
__global__ void k_grid_detect1(int32_t n, int lower, const int32_t *__restrict__ ptr, const int32_t *__restrict__ node, int32_t *info)
{
    __shared__ int32_t lo, hi, bad;
    if (threadIdx.x == 0) { lo = INT32_MAX; hi = 0; bad = 0; }
    __syncthreads();
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const int32_t b = ptr[r], cnt = ptr[r + 1] - b;
        if (cnt > 2 || (cnt == 2 && node[b] == node[b + 1])) bad = 1;
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
__global__ void k_grid_detect2(int32_t n, int lower, int32_t w, const int32_t *__restrict__ ptr, const int32_t *__restrict__ node, int32_t *info)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    for (int32_t k = ptr[r]; k < ptr[r + 1]; ++k) {
        const int32_t dlt = lower ? r - node[k] : node[k] - r;
        if (dlt == 1) { if (lower ? r % w == 0 : (r + 1) % w == 0) info[3] = 1; }
        else if (dlt != w) info[3] = 1;
    }
}
// rows of a w-wide grid keyed by their anti-diagonal i + j: a valid levelling of a factor whose rows depend on r-1 and
// r-w only (each of them one anti-diagonal back) -- the order its rows are factorised in
// (h > 0: a w x h x nk grid, rows depend on r-1, r-w, r-w*h: keyed by i + j + k)
__global__ void k_grid_keys(int32_t n, int32_t w, int32_t h, int32_t *__restrict__ key, int32_t *__restrict__ rows, int32_t *__restrict__ count)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int32_t k = h > 0 ? r % w + (r / w) % h + r / (w * h) : r % w + r / w;
    key[r] = k;
    rows[r] = r;
    atomicAdd(count + k, 1);
}

// sparse_static_pattern_ldu_factorization, first loop (ldu_solvers.f90:300-318): A's entries into L, D, U through
// set_value, row by row in stored order
__global__ void k_ildu_init(int32_t n, int32_t ncol_own, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                            const double *__restrict__ val, const int32_t *__restrict__ Lptr, const int32_t *__restrict__ Lnode,
                            double *Lval, const int32_t *__restrict__ Uptr, const int32_t *__restrict__ Unode, double *Uval,
                            double *D)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t lb = Lptr[i], le = Lptr[i + 1], ub = Uptr[i], ue = Uptr[i + 1];
    for (int32_t k = lb; k < le; ++k) Lval[k] = 0.0;
    for (int32_t k = ub; k < ue; ++k) Uval[k] = 0.0;
    double d = 0.0;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j >= ncol_own) continue;
        const double v = val[k];
        if (i > j) row_set(Lnode, Lval, lb, le, j, v);
        else if (j > i) row_set(Unode, Uval, ub, ue, j, v);
        else d = v;
    }
    D[i] = d;
}

// its main loop (ldu_solvers.f90:334-382), the statements of one row in the reference's order; the rows of one
// dependency level of L side by side (row i reads rows k < i of its L pattern only -- final since an earlier level --
// and writes its own).  One lane per row.
__host__ __device__ inline void ildu_factor_row(int32_t i, const int32_t *Lptr, const int32_t *Lnode, double *Lval, const int32_t *Uptr,
                                                const int32_t *Unode, double *Uval, double *D);
__global__ void k_ildu_factor_level(const int32_t *__restrict__ order, int32_t begin, int32_t end,
                                    const int32_t *__restrict__ Lptr, const int32_t *__restrict__ Lnode, double *Lval,
                                    const int32_t *__restrict__ Uptr, const int32_t *__restrict__ Unode, double *Uval, double *D)
{
    const int32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    ildu_factor_row(order[p], Lptr, Lnode, Lval, Uptr, Unode, Uval, D);
}
// (also run row after row on the HOST for factors that are chains: see pc_setup_ordered -- the same statements compiled with the
// same -ffp-contract=off, the same bits)
__host__ __device__ inline void ildu_factor_row(int32_t i, const int32_t *Lptr, const int32_t *Lnode, double *Lval, const int32_t *Uptr,
                                                const int32_t *Unode, double *Uval, double *D)
{
    const int32_t lb = Lptr[i], le = Lptr[i + 1], ub = Uptr[i], ue = Uptr[i + 1];
    double Di = D[i];
    for (int32_t a = lb; a < le; ++a) {
        const int32_t k = Lnode[a];
        const int32_t kb = Uptr[k], ke = Uptr[k + 1];
        double Lik = row_get(Lnode, Lval, lb, le, k);
        const double Uki = row_get(Unode, Uval, kb, ke, i);
        const double Dk = D[k];
        row_set(Lnode, Lval, lb, le, k, Lik / Dk);
        Lik = Lik / Dk;
        for (int32_t c = lb; c < le; ++c) {
            const int32_t j = Lnode[c];
            if (j > k) {
                const double Ukj = row_get(Unode, Uval, kb, ke, j);
                row_add(Lnode, Lval, lb, le, j, -Lik * Dk * Ukj);
            }
        }
        Di = Di - Lik * Dk * Uki;
        for (int32_t c = ub; c < ue; ++c) {
            const int32_t j = Unode[c];
            const double Ukj = row_get(Unode, Uval, kb, ke, j);
            row_add(Unode, Uval, ub, ue, j, -Lik * Dk * Ukj);
        }
    }
    for (int32_t c = ub; c < ue; ++c) {
        const int32_t k = Unode[c];
        const double Uik = row_get(Unode, Uval, ub, ue, k);
        row_set(Unode, Uval, ub, ue, k, Uik / Di);
    }
    D[i] = Di;
}

// index work of the strips' skewed layout, one lane per row: position of the row, its entries' places in the factor's val
// array (r-w term / r-1 term) and the presence / order bits; flags[0] / [1]: some two-term row has its r-w / r-1 term first.
// The upper factor is the lower one of the reversed numbering: i' = w-1-i, j' = nj-1-j.
__global__ void k_grid_build(int32_t n, int32_t w, int32_t nj, int32_t S, int lower, const int32_t *__restrict__ ptr,
                             const int32_t *__restrict__ node, int32_t *__restrict__ row, int32_t *__restrict__ srcS,
                             int32_t *__restrict__ srcW, uint8_t *__restrict__ code, int32_t *__restrict__ pos, int32_t *flags)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int32_t i = r % w, j = r / w;
    if (!lower) { i = w - 1 - i; j = nj - 1 - j; }
    const int32_t ib = i / 64, l = i % 64;
    const int64_t p = (int64_t)ib * S * 64 + (int64_t)(j + l) * 64 + l;
    pos[r] = (int32_t)p;
    row[p] = r;
    uint8_t c = 0;
    int seen = 0;
    for (int32_t k = ptr[r]; k < ptr[r + 1]; ++k, ++seen) {
        const int32_t dlt = lower ? r - node[k] : node[k] - r;
        if (dlt == 1) { c |= 2; srcW[p] = k; if (seen == 0) c |= 4; }
        else { c |= 1; srcS[p] = k; }
    }
    code[p] = c;
    if ((c & 3) == 3) flags[(c & 4) ? 1 : 0] = 1;                  // (single-term rows fit either order)
}
__global__ void k_grid_map(int32_t n, const int32_t *__restrict__ posU, const int32_t *__restrict__ posL, int32_t *__restrict__ map)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) map[posU[r]] = posL[r];
}

__global__ void k_check_vector(int64_t n, double *__restrict__ r)       // the self-check's right-hand side
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) r[i] = 1.0 + 0.25 * (double)(i % 7) - 0.125 * (double)(i % 3);
}

// One sweep of ldu_solve checked row by row (setup self-check of the pipelined sweeps): row i of the result must be what
// the reference's recurrence (ldu_solvers.f90:227-236, :254-263) makes of the right-hand side and of the RESULT's own
// earlier rows -- t = rhs_i (/ D_i); t = t - val * x(node) over the row's entries in stored order -- bit for bit.  If
// that holds for every row the result IS the sequential sweep's (induction along the dependencies), and every row can
// be checked independently.
__global__ void k_sweep_check(int32_t n, const int32_t *__restrict__ ptr, const int32_t *__restrict__ node, const double *__restrict__ val,
                              const double *__restrict__ rhs, const double *__restrict__ D, const double *__restrict__ x, int32_t *bad)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = rhs[i];
    if (D) t = t / D[i];
    for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) t = t - val[k] * x[node[k]];
    if (__double_as_longlong(t) != __double_as_longlong(x[i])) atomicAdd(bad, 1);
}

// The same for factors whose rows are short (every row of L at most ML entries, of U at most MU: 5-, 7-, 9-point
// matrices): the row's own entries and the rows k it reads are fetched into registers up front -- five dependent memory
// round trips (order, row pointers, own entries, pointers / D of the rows k, their entries) instead of the eleven or so
// the scans above make one after the other; a launch of a narrow level is nothing but that chain.  Then the same
// statements in the same order on the registers, and one store of the row.
template <int M>
__device__ inline double reg_get(const int32_t (&nd)[M], const double (&vl)[M], int cnt, int32_t j)
{
    double z = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m)
        if (m < cnt && nd[m] == j) z = vl[m];
    return z;
}
template <int M>
__device__ inline void reg_set(const int32_t (&nd)[M], double (&vl)[M], int cnt, int32_t j, double z)
{
#pragma unroll
    for (int m = 0; m < M; ++m)
        if (m < cnt && nd[m] == j) vl[m] = z;
}
template <int M>
__device__ inline void reg_add(const int32_t (&nd)[M], double (&vl)[M], int cnt, int32_t j, double z)
{
#pragma unroll
    for (int m = 0; m < M; ++m)
        if (m < cnt && nd[m] == j) vl[m] = vl[m] + z;
}
template <int ML, int MU>
__global__ void k_ildu_factor_level_short(const int32_t *__restrict__ order, int32_t begin, int32_t end,
                                          const int32_t *__restrict__ Lptr, const int32_t *__restrict__ Lnode, double *Lval,
                                          const int32_t *__restrict__ Uptr, const int32_t *__restrict__ Unode, double *Uval, double *D)
{
    const int32_t p = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= end) return;
    const int32_t i = order[p];
    const int32_t lb = Lptr[i], ub = Uptr[i];
    const int dl = Lptr[i + 1] - lb, du = Uptr[i + 1] - ub;
    int32_t ln[ML], un[MU];
    double lv[ML], uv[MU];
#pragma unroll
    for (int m = 0; m < ML; ++m) { ln[m] = m < dl ? Lnode[lb + m] : -1; lv[m] = m < dl ? Lval[lb + m] : 0.0; }
#pragma unroll
    for (int m = 0; m < MU; ++m) { un[m] = m < du ? Unode[ub + m] : -1; uv[m] = m < du ? Uval[ub + m] : 0.0; }
    double Di = D[i];
    int32_t kb[ML];
    int kc[ML];
    double dk[ML];
#pragma unroll
    for (int a = 0; a < ML; ++a) {
        const int32_t k = a < dl ? ln[a] : 0;
        kb[a] = a < dl ? Uptr[k] : 0;
        kc[a] = a < dl ? Uptr[k + 1] - kb[a] : 0;
        dk[a] = a < dl ? D[k] : 1.0;
    }
    int32_t kn[ML][MU];
    double kv[ML][MU];
#pragma unroll
    for (int a = 0; a < ML; ++a)
#pragma unroll
        for (int m = 0; m < MU; ++m) {
            kn[a][m] = m < kc[a] ? Unode[kb[a] + m] : -1;
            kv[a][m] = m < kc[a] ? Uval[kb[a] + m] : 0.0;
        }
#pragma unroll
    for (int a = 0; a < ML; ++a) {
        if (a < dl) {
            const int32_t k = ln[a];
            double Lik = reg_get<ML>(ln, lv, dl, k);
            const double Uki = reg_get<MU>(kn[a], kv[a], kc[a], i);
            const double Dk = dk[a];
            reg_set<ML>(ln, lv, dl, k, Lik / Dk);
            Lik = Lik / Dk;
#pragma unroll
            for (int c = 0; c < ML; ++c) {
                if (c < dl && ln[c] > k) {
                    const double Ukj = reg_get<MU>(kn[a], kv[a], kc[a], ln[c]);
                    reg_add<ML>(ln, lv, dl, ln[c], -Lik * Dk * Ukj);
                }
            }
            Di = Di - Lik * Dk * Uki;
#pragma unroll
            for (int c = 0; c < MU; ++c) {
                if (c < du) {
                    const double Ukj = reg_get<MU>(kn[a], kv[a], kc[a], un[c]);
                    reg_add<MU>(un, uv, du, un[c], -Lik * Dk * Ukj);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MU; ++c) {
        if (c < du) {
            const double Uik = reg_get<MU>(un, uv, du, un[c]);
            reg_set<MU>(un, uv, du, un[c], Uik / Di);
        }
    }
#pragma unroll
    for (int m = 0; m < ML; ++m)
        if (m < dl) Lval[lb + m] = lv[m];
#pragma unroll
    for (int m = 0; m < MU; ++m)
        if (m < du) Uval[ub + m] = uv[m];
    D[i] = Di;
}

// values into the structures the applies read
__global__ void k_grid_records(int64_t np, const int32_t *__restrict__ srcS, const int32_t *__restrict__ srcW,
                               const uint8_t *__restrict__ code, int order, const double *__restrict__ val, StripRec *__restrict__ rec)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < np; p += stride) {
        StripRec r;
        r.cS = srcS[p] >= 0 ? val[srcS[p]] : 0.0;
        r.cW = srcW[p] >= 0 ? val[srcW[p]] : 0.0;
        r.rhs = 0.0;
        const uint8_t c = code[p];
        if (order == 2) r.code = c;                                // flag word
        else r.code = ((c & 1) ? 0xffffffffull : 0ull) | ((c & 2) ? 0xffffffff00000000ull : 0ull);   // AND masks
        rec[p] = r;
    }
}
__global__ void k_pos_diag(int64_t np, const int32_t *__restrict__ row, const double *__restrict__ D, double *__restrict__ Dp)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < np; p += stride) Dp[p] = row[p] >= 0 ? D[row[p]] : 1.0;
}
__global__ void k_tri_entries(int64_t nnz, const int32_t *__restrict__ src, const double *__restrict__ val, double *__restrict__ pv)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) pv[k] = val[src[k]];
}
// row-space copy of a factor (k_trsv_rows): slot j of position p = entry j of row order[p] -- its column, its value
__global__ void k_rows_index(int32_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ ptr, const int32_t *__restrict__ node,
                             uint32_t nstride, int rc, int32_t *__restrict__ rq)
{
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t i = order[p], b = ptr[i], cnt = ptr[i + 1] - b;
    for (int j = 0; j < rc; ++j) rq[rs_at(j, p, rc)] = j < cnt ? node[b + j] : -1;
}
__global__ void k_rows_values(int32_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ ptr, const double *__restrict__ val,
                              uint32_t nstride, int rc, double *__restrict__ rv)
{
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t i = order[p], b = ptr[i], cnt = ptr[i + 1] - b;
    for (int j = 0; j < rc; ++j) rv[rs_at(j, p, rc)] = j < cnt ? val[b + j] : 0.0;
}

// distinct offsets (dependency row - own row) of the row-space copy into a 64-slot table (INT32_MIN = free); *overflow: more
__global__ void k_rows_offsets(int32_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ rq, int rc, int32_t *table, int *overflow)
{
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t i = order[p];
    for (int c = 0; c < rc; ++c) {
        const int32_t j = rq[rs_at(c, p, rc)];
        if (j < 0) continue;
        const int32_t d = j - i;
        uint32_t h = ((uint32_t)d * 2654435761u) >> 26;
        int probe = 0;
        for (; probe < 64; ++probe, h = (h + 1) & 63u) {
            int32_t cur = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (almost always: already there)
            if (cur == d) break;
            if (cur == INT32_MIN) {
                cur = atomicCAS(table + h, INT32_MIN, d);
                if (cur == INT32_MIN || cur == d) break;
            }
        }
        if (probe == 64) *overflow = 1;
    }
}
__global__ void k_rows_encode(int32_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ rq, int rc, const int32_t *__restrict__ dict,
                              int ndict, uint32_t *__restrict__ rcode)
{
    __shared__ int32_t dl[16];
    if (threadIdx.x < 16) dl[threadIdx.x] = (int)threadIdx.x < ndict ? dict[threadIdx.x] : INT32_MIN;
    __syncthreads();
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t i = order[p];
    uint32_t code = 0;
    for (int c = 0; c < 8; ++c) {
        uint32_t nib = 15u;
        const int32_t j = c < rc ? rq[rs_at(c, p, rc)] : -1;
        if (j >= 0)
            for (int k = 0; k < ndict; ++k)
                if (dl[k] == j - i) { nib = (uint32_t)k; break; }
        code |= nib << (4 * c);
    }
    rcode[p] = code;
}
// after k_rows_index: the codes, where the factor allows them (see TriFactor::rcode)
int rows_encode(TriFactor &T, int32_t n)
{
    dfree(T.rcode); dfree(T.rdict);
    T.rcode = nullptr; T.rdict = nullptr; T.nrdict = 0;
    if (!T.rows_on || T.rc > 8 || n < 1) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *table = nullptr;
    struct Guard { int32_t *&t; ~Guard() { dfree(t); } } guard{table};
    SGM_TRY(dalloc(&table, 64 + 1));
    std::vector<int32_t> h(65, INT32_MIN);
    h[64] = 0;
    SGM_HIP(hipMemcpyAsync(table, h.data(), 65 * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rows_offsets, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)T.order, (const int32_t *)T.rq, T.rc,
                       table, reinterpret_cast<int *>(table + 64));
    SGM_HIP(hipMemcpyAsync(h.data(), table, 65 * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    std::vector<int32_t> dict;
    for (int k = 0; k < 64; ++k)
        if (h[k] != INT32_MIN) dict.push_back(h[k]);
    if (h[64] != 0 || dict.size() > 15) return SGM_OK;
    std::sort(dict.begin(), dict.end());
    dict.resize(16, 0);
    T.nrdict = 0;
    for (int k = 0; k < 64; ++k) T.nrdict += h[k] != INT32_MIN;
    SGM_TRY(dalloc(&T.rdict, 16));
    SGM_TRY(dalloc(&T.rcode, (size_t)n + 2));
    SGM_HIP(hipMemcpyAsync(T.rdict, dict.data(), 16 * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_rows_encode, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)T.order, (const int32_t *)T.rq, T.rc,
                       (const int32_t *)T.rdict, T.nrdict, T.rcode);
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));                   // (dict is a local)
    return SGM_OK;
}

// the inline values of the row records and their slot-major copy (dv: kInline slots)
__global__ void k_tri_slots(int32_t n, TrsvRec *recs, const double *__restrict__ pv, uint32_t nstride, double *__restrict__ dv)
{
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int32_t cnt = recs[p].cnt, k0 = recs[p].k0;
#pragma unroll
    for (int j = 0; j < kInline; ++j) {
        const double v = j < cnt ? pv[k0 + j] : 0.0;
        recs[p].v[j] = v;
        dv[(size_t)j * nstride + p] = v;
    }
}

// dependency levels of a strictly triangular pattern (1-based): level_ptr / order (position -> row, rows of a level in
// ascending order) / pos (row -> position)
void tri_levels(int32_t n, const std::vector<int32_t> &ptr1, const std::vector<int32_t> &node1, bool lower,
                std::vector<int32_t> &level_ptr, std::vector<int32_t> &order, std::vector<int32_t> *pos)
{
    std::vector<int32_t> level(std::max(n, 1), 0);
    int32_t nlev = 0;
    auto visit = [&](int32_t i) {
        int32_t lv = 0;
        for (int32_t k = ptr1[i] - 1; k < ptr1[i + 1] - 1; ++k) lv = std::max(lv, level[node1[k] - 1] + 1);
        level[i] = lv;
        nlev = std::max(nlev, lv + 1);
    };
    if (lower) for (int32_t i = 0; i < n; ++i) visit(i);
    else for (int32_t i = n - 1; i >= 0; --i) visit(i);
    level_ptr.assign(nlev + 1, 0);
    for (int32_t i = 0; i < n; ++i) level_ptr[level[i] + 1]++;
    for (int32_t l = 0; l < nlev; ++l) level_ptr[l + 1] += level_ptr[l];
    order.assign(std::max(n, 1), 0);
    if (pos) pos->assign(std::max(n, 1), 0);
    std::vector<int32_t> cursor(level_ptr.begin(), level_ptr.end() - 1);
    for (int32_t i = 0; i < n; ++i) {
        const int32_t p = cursor[level[i]]++;
        order[p] = i;
        if (pos) (*pos)[i] = p;
    }
}

void free_tri(TriFactor &T);
void free_grid(GridTri &G);
void free_ildu(IlduState &S)
{
    free_tri(S.L);
    free_tri(S.U);
    dfree(S.D); dfree(S.xpL); dfree(S.xpU); dfree(S.Dp); dfree(S.mapLU);
    dfree(S.dLptr); dfree(S.dLnode); dfree(S.dUptr); dfree(S.dUnode); dfree(S.dLval); dfree(S.dUval); dfree(S.forder);
    free_grid(S.gL); free_grid(S.gU);
    dfree(S.gxL); dfree(S.gxU); dfree(S.gDp); dfree(S.gmapLU);
    slab3_free(S.slab);
    const PcOptions keep = S.opt;          // (the owning preconditioner's options outlive a rebuild of its factors)
    S = IlduState();
    S.opt = keep;
}

void free_tri(TriFactor &T)
{
    dfree(T.order); dfree(T.recs); dfree(T.pq); dfree(T.pv); dfree(T.level_ptr_dev); dfree(T.dq); dfree(T.dq32); dfree(T.dv); dfree(T.wq);
    dfree(T.rq); dfree(T.rv); dfree(T.src); dfree(T.rcode); dfree(T.rdict);
    T = TriFactor();
}

// The same index work entirely on the device, for a factor of at most kRowLevels levels (what a colour ordering leaves):
// levels by relaxation, the level order by a stable radix sort of the row numbers on their levels, per-level facts by
// one more pass.  *served = false (and T untouched) when the factor has more levels than that or rows too long for the
// row-space copy: the host pass (tri_levels_dev) then does it from the pattern's host copy.  No host copy of the pattern
// is needed here; T.h_order / T.h_pos stay empty until the level walkers want them (tri_host_order).
int tri_levels_device(TriFactor &T, int32_t n, const int32_t *dptr, const int32_t *dnode, bool *served)
{
    *served = false;
    if (T.have_levels) { *served = T.rows_on; return SGM_OK; }
    if (n < 1 || (size_t)n + kNarrow >= (size_t)500000000) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *level = nullptr, *small = nullptr, *rows = nullptr, *order = nullptr, *keys = nullptr;
    void *tmp = nullptr;
    struct Tmp { int32_t *&a, *&b, *&c, *&d, *&e; void *&t; ~Tmp() { dfree(a); dfree(b); dfree(c); dfree(d); dfree(e); if (t) (void)hipFree(t); } }
        guard{level, small, rows, order, keys, tmp};
    SGM_TRY(dalloc(&level, (size_t)n));
    SGM_TRY(dalloc(&small, (size_t)2 + 5 * (kRowLevels + 2)));      // flags[2] | count | begin | cmax | notrun | first
    SGM_HIP(hipMemsetAsync(level, 0, (size_t)n * 4, st));
    int32_t *flags = small;                              // [2]
    const int grid = (n + kBlock - 1) / kBlock;
    bool done = false;
    for (int it = 0; it <= kRowLevels + 1 && !done; ++it) {
        int32_t hf[2] = {0, 0};
        SGM_HIP(hipMemsetAsync(flags, 0, 8, st));
        hipLaunchKernelGGL(k_level_relax, dim3(grid), dim3(kBlock), 0, st, n, dptr, dnode, level, (int32_t)kRowLevels, flags);
        SGM_HIP(hipMemcpyAsync(hf, flags, 8, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipStreamSynchronize(st));
        if (hf[1]) return SGM_OK;                        // too many levels for this path
        done = !hf[0];
    }
    if (!done) return SGM_OK;
    // histogram -> level_ptr; stable sort of 0 .. n-1 on the levels -> level order (rows of a level ascending)
    int32_t *count = small + 2, *begin = count + kRowLevels + 2, *cmax = begin + kRowLevels + 2, *notrun = cmax + kRowLevels + 2,
            *first = notrun + kRowLevels + 2;
    SGM_HIP(hipMemsetAsync(count, 0, (size_t)5 * (kRowLevels + 2) * 4, st));
    SGM_TRY(dalloc(&rows, (size_t)n));
    SGM_TRY(dalloc(&order, (size_t)n));
    SGM_TRY(dalloc(&keys, (size_t)n));
    hipLaunchKernelGGL(k_level_hist, dim3(grid), dim3(kBlock), 0, st, n, (const int32_t *)level, count, rows);
    int32_t hcount[kRowLevels + 2];
    SGM_HIP(hipMemcpyAsync(hcount, count, sizeof hcount, hipMemcpyDeviceToHost, st));
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const int32_t *)level, keys, (const int32_t *)rows, order, n, 0, 6, st);
    SGM_HIP(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    SGM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const int32_t *)level, keys, (const int32_t *)rows, order, n, 0, 6, st));
    SGM_HIP(hipStreamSynchronize(st));
    int32_t nlev = 0;
    for (int l = 0; l < kRowLevels + 2; ++l) if (hcount[l]) nlev = l + 1;
    if (nlev < 1 || nlev > kRowLevels) return SGM_OK;
    std::vector<int32_t> lp((size_t)nlev + 1, 0);
    for (int l = 0; l < nlev; ++l) lp[l + 1] = lp[l] + hcount[l];
    SGM_HIP(hipMemcpyAsync(begin, lp.data(), (size_t)(nlev + 1) * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_level_info, dim3(grid), dim3(kBlock), 0, st, n, (const int32_t *)order, (const int32_t *)level, dptr,
                       (const int32_t *)begin, cmax, notrun, first);
    int32_t hinfo[3 * (kRowLevels + 2)];
    SGM_HIP(hipMemcpyAsync(hinfo, cmax, sizeof hinfo, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    const int32_t *hc = hinfo, *hn = hinfo + kRowLevels + 2, *hfst = hinfo + 2 * (kRowLevels + 2);
    int cm = 0;
    for (int l = 0; l < nlev; ++l) cm = std::max(cm, hc[l]);
    if (cm > 64) return SGM_OK;
    // commit
    free_tri(T);
    T.level_ptr = lp;
    T.nstride = (size_t)n + kNarrow;
    T.order = order; order = nullptr;                    // (the sorted row numbers ARE the level order)
    SGM_TRY(dalloc(&T.level_ptr_dev, T.level_ptr.size()));
    SGM_HIP(hipMemcpy(T.level_ptr_dev, T.level_ptr.data(), T.level_ptr.size() * 4, hipMemcpyHostToDevice));
    for (int l = 0; l < nlev; ++l) T.row_levels.push_back({lp[l], lp[l + 1], hc[l], hn[l] ? -1 : hfst[l]});
    T.rows_on = true;
    T.rc = cm <= 4 ? std::max(cm, 1) : cm <= 6 ? 6 : cm <= 8 ? 8 : cm;      // (the unrolled kernels read 6 / 8 slots)
    SGM_TRY(dalloc(&T.rq, T.nstride * (size_t)T.rc));
    SGM_TRY(dalloc(&T.rv, T.nstride * (size_t)T.rc));
    SGM_HIP(hipMemsetAsync(T.rq, 0xff, T.nstride * (size_t)T.rc * 4, st));
    SGM_HIP(hipMemsetAsync(T.rv, 0, T.nstride * (size_t)T.rc * 8, st));
    hipLaunchKernelGGL(k_rows_index, dim3(grid), dim3(kBlock), 0, st, n, (const int32_t *)T.order, dptr, dnode, (uint32_t)T.nstride, T.rc, T.rq);
    SGM_HIP(hipGetLastError());
    SGM_TRY(rows_encode(T, n));
    T.have_levels = true;
    *served = true;
    return SGM_OK;
}
// host copies of the level order for what still reads them (the level walkers' index work)
int tri_host_order(TriFactor &T, int32_t n)
{
    if (!T.h_order.empty() || n < 1) return SGM_OK;
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    T.h_order.resize((size_t)n);
    T.h_pos.resize((size_t)n);
    SGM_TRY(copy_big(T.h_order.data(), T.order, (size_t)n * 4, hipMemcpyDeviceToHost));
    for (int32_t p2 = 0; p2 < n; ++p2) T.h_pos[T.h_order[p2]] = p2;
    return SGM_OK;
}

// Dependency levels of a strictly triangular factor and, for one of at most kRowLevels levels, its row-space copy.
// lower: rows depend on smaller rows (forward sweep 1..n); upper: on larger rows (backward sweep n..1).  ptr1 / node1:
// the pattern on the host (1-based); dptr / dnode / dval: the factor on the device (0-based, values in pattern order;
// dval null: index work only).
// (ptr1 / node1 may be EMPTY when the device pass is known to have served this factor: tri_levels_device below)
int tri_levels_dev(TriFactor &T, int32_t n, const std::vector<int32_t> &ptr1, const std::vector<int32_t> &node1,
                   const int32_t *dptr, const int32_t *dnode, const double *dval, bool lower)
{
    hipStream_t st = g_rt.stream;
    if (!T.have_levels) {
        free_tri(T);
        tri_levels(n, ptr1, node1, lower, T.level_ptr, T.h_order, &T.h_pos);
        const int32_t nlev = (int32_t)T.level_ptr.size() - 1;
        T.nstride = (size_t)n + kNarrow;          // (padded by kNarrow rows: lanes of the walkers past a level's end read valid memory)
        SGM_TRY(dalloc(&T.order, (size_t)std::max(n, 1)));
        SGM_TRY(dalloc(&T.level_ptr_dev, T.level_ptr.size()));
        if (n) SGM_TRY(copy_big(T.order, T.h_order.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        SGM_HIP(hipMemcpy(T.level_ptr_dev, T.level_ptr.data(), T.level_ptr.size() * 4, hipMemcpyHostToDevice));
        // a few levels (whatever their widths): the row-space copy (dependency rows, slot-major over the level order)
        T.rows_on = false;
        T.rc = 0;
        T.row_levels.clear();
        if (nlev >= 1 && nlev <= kRowLevels && (size_t)n + kNarrow < (size_t)500000000) {
            int cm = 0;
            for (int32_t l = 0; l < nlev; ++l) {
                const int32_t b = T.level_ptr[l], e = T.level_ptr[l + 1];
                int c = 0;
                bool run = true;
                for (int32_t p2 = b; p2 < e; ++p2) {
                    const int32_t i = T.h_order[p2];
                    c = std::max(c, ptr1[i + 1] - ptr1[i]);
                    if (p2 > b) run = run && i == T.h_order[p2 - 1] + 1;
                }
                T.row_levels.push_back({b, e, c, run ? T.h_order[b] : -1});
                cm = std::max(cm, c);
            }
            if (cm <= 64) {
                T.rows_on = true;
                T.rc = cm <= 4 ? std::max(cm, 1) : cm <= 6 ? 6 : cm <= 8 ? 8 : cm;      // (the unrolled kernels read 6 / 8 slots)
                SGM_TRY(dalloc(&T.rq, T.nstride * (size_t)T.rc));
                SGM_TRY(dalloc(&T.rv, T.nstride * (size_t)T.rc));
                SGM_HIP(hipMemsetAsync(T.rq, 0xff, T.nstride * (size_t)T.rc * 4, st));
                SGM_HIP(hipMemsetAsync(T.rv, 0, T.nstride * (size_t)T.rc * 8, st));
                if (n) hipLaunchKernelGGL(k_rows_index, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)T.order, dptr,
                                          dnode, (uint32_t)T.nstride, T.rc, T.rq);
                SGM_TRY(rows_encode(T, n));
            }
        }
        T.have_levels = true;
    }
    if (T.rows_on && n && dval)
        hipLaunchKernelGGL(k_rows_values, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)T.order, dptr, dval,
                           (uint32_t)T.nstride, T.rc, T.rv);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// The level walkers' structures of a factor (tri_levels_dev has run): records in level order, schedule, ring copies --
// index work when the pattern is new, values (from the device factor) every time.
int tri_walkers(TriFactor &T, int32_t n, const std::vector<int32_t> &ptr1, const std::vector<int32_t> &node1, const double *val)
{
    const size_t nnz = node1.size();
    if (!T.have_walkers) {
        SGM_TRY(tri_host_order(T, n));
        dfree(T.recs); dfree(T.pq); dfree(T.pv); dfree(T.dq); dfree(T.dq32); dfree(T.dv); dfree(T.wq); dfree(T.src);
        T.recs = nullptr; T.pq = nullptr; T.pv = nullptr; T.dq = nullptr; T.dq32 = nullptr; T.dv = nullptr; T.wq = nullptr; T.src = nullptr;
        T.schedule.clear();
        std::vector<int32_t> h_src(std::max<size_t>(nnz, 1), 0);      // level-order entry -> factor entry
        const int32_t nlev = (int32_t)T.level_ptr.size() - 1;
        // rows in level order: dependency POSITIONS in stored order
        T.h_recs.assign(std::max(n, 1), TrsvRec());
        T.h_pq.assign(std::max<size_t>(nnz, 1), 0);
        int32_t kk = 0;
        for (int32_t p = 0; p < n; ++p) {
            const int32_t i = T.h_order[p];
            TrsvRec &r = T.h_recs[p];
            r.cnt = ptr1[i + 1] - ptr1[i];
            r.k0 = kk;
            for (int32_t k = ptr1[i] - 1; k < ptr1[i + 1] - 1; ++k, ++kk) {
                T.h_pq[kk] = T.h_pos[node1[k] - 1];
                h_src[kk] = k;
                if (kk - r.k0 < kInline) r.q[kk - r.k0] = T.h_pq[kk];
            }
        }
        // schedule: wide levels alone, runs of narrow levels together
        constexpr int narrow = kNarrow;
        std::vector<int8_t> lev_cls(nlev, 0);
        {
            std::vector<int8_t> raw(nlev, 0);
            for (int32_t l = 0; l < nlev; ++l) {
                const int32_t w = T.level_ptr[l + 1] - T.level_ptr[l];
                raw[l] = w <= 64 ? -1 : w <= 256 ? 0 : w <= 512 ? 1 : w <= kTrsvBlock ? 2 : w <= 2 * kTrsvBlock ? 3 : w <= narrow ? 4 : 5;      // (-1: one wave)
            }
            for (int32_t l = 0; l < nlev; ++l) {          // window maximum over narrow neighbours
                int8_t m = raw[l];
                if (m < 5) {
                    for (int32_t k = l - 1; k >= std::max(0, l - 8) && raw[k] < 5; --k) m = std::max(m, raw[k]);
                    for (int32_t k = l + 1; k <= std::min(nlev - 1, l + 8) && raw[k] < 5; ++k) m = std::max(m, raw[k]);
                }
                lev_cls[l] = m;
            }
        }
        for (int32_t l = 0; l < nlev;) {
            const int32_t sz = T.level_ptr[l + 1] - T.level_ptr[l];
            if (sz > narrow) {
                int cm = 0;
                for (int32_t p = T.level_ptr[l]; p < T.level_ptr[l + 1]; ++p) cm = std::max(cm, T.h_recs[p].cnt);
                T.schedule.push_back({l, l + 1, false, 0, false, cm});      // c: most dependencies of a row of the level
                ++l;
                continue;
            }
            // runs are cut by width class: 256 / 512 / 1024 threads with one row per lane, then 2 and
            // 4 rows per lane (classes 0..4, smoothed so that a run is at least ~16 levels long)
            const int c = lev_cls[l];
            int32_t e = l;
            while (e < nlev && T.level_ptr[e + 1] - T.level_ptr[e] <= narrow && lev_cls[e] == c) ++e;
            // all dependencies inline and within the ring's reach?  (see k_trsv_walk_ring)
            bool ring_ok = true;
            int cmax = 0;
            for (int32_t lev = l; lev < e && ring_ok; ++lev)
                for (int32_t p = T.level_ptr[lev]; p < T.level_ptr[lev + 1] && ring_ok; ++p) {
                    const TrsvRec &r = T.h_recs[p];
                    ring_ok = r.cnt <= kInline;
                    cmax = std::max(cmax, r.cnt);
                    for (int32_t k = r.k0; k < r.k0 + r.cnt && ring_ok; ++k)
                        ring_ok = T.h_pq[k] >= T.level_ptr[lev + 1] - kRing && T.h_pq[k] < p;
                }
            T.schedule.push_back({l, e, true, c, ring_ok, cmax});
            l = e;
        }
        // ring-walker copy of the structure: 16-bit ring slots (only read in ring runs), padded
        // by kNarrow rows so that lanes past a level's end read valid memory
        T.h_dq.assign(T.nstride, 0);
        for (int32_t p = 0; p < n; ++p) {
            const TrsvRec &r = T.h_recs[p];
            uint64_t w = 0;
            for (int j = 0; j < kInline; ++j)
                w |= (uint64_t)(j < r.cnt ? (r.q[j] & (kRing - 1)) : kRing) << (16 * j);
            T.h_dq[p] = w;
        }
        SGM_TRY(dalloc(&T.dq, T.nstride));
        SGM_TRY(dalloc(&T.dv, T.nstride * kInline));
        SGM_HIP(hipMemcpy(T.dq, T.h_dq.data(), T.h_dq.size() * 8, hipMemcpyHostToDevice));
        {
            std::vector<uint32_t> lo(T.nstride);
            for (size_t p = 0; p < T.nstride; ++p) lo[p] = (uint32_t)T.h_dq[p];
            SGM_TRY(dalloc(&T.dq32, T.nstride));
            SGM_HIP(hipMemcpy(T.dq32, lo.data(), lo.size() * 4, hipMemcpyHostToDevice));
        }
        {
            std::vector<int32_t> wq(T.nstride * kInline, -1);
            for (int32_t p = 0; p < n; ++p)
                for (int j = 0; j < kInline && j < T.h_recs[p].cnt; ++j) wq[(size_t)j * T.nstride + p] = T.h_recs[p].q[j];
            SGM_TRY(dalloc(&T.wq, wq.size()));
            SGM_HIP(hipMemcpy(T.wq, wq.data(), wq.size() * 4, hipMemcpyHostToDevice));
        }
        SGM_TRY(dalloc(&T.recs, (size_t)std::max(n, 1)));
        SGM_TRY(dalloc(&T.pq, nnz));
        SGM_TRY(dalloc(&T.pv, nnz));
        SGM_TRY(dalloc(&T.src, nnz));
        if (nnz) SGM_TRY(copy_big(T.pq, T.h_pq.data(), nnz * 4, hipMemcpyHostToDevice));
        if (nnz) SGM_TRY(copy_big(T.src, h_src.data(), nnz * 4, hipMemcpyHostToDevice));
        if (n) SGM_TRY(copy_big(T.recs, T.h_recs.data(), (size_t)n * sizeof(TrsvRec), hipMemcpyHostToDevice));     // (values: k_tri_slots)
        SGM_HIP(hipMemsetAsync(T.dv, 0, T.nstride * kInline * 8, g_rt.stream));          // (the padding slots stay zero)
        T.have_walkers = true;
    }
    // values (every setup), on the device: level-order copy, the inline part of the records, the slot-major copy
    hipStream_t st = g_rt.stream;
    if (nnz) hipLaunchKernelGGL(k_tri_entries, dim3(vec_grid((int64_t)nnz)), dim3(kBlock), 0, st, (int64_t)nnz, (const int32_t *)T.src, val, T.pv);
    if (n) hipLaunchKernelGGL(k_tri_slots, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, T.recs, (const double *)T.pv,
                              (uint32_t)T.nstride, T.dv);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// ---- strip path: host side ---------------------------------------------------------------------
void free_grid(GridTri &G)
{
    dfree(G.rec); dfree(G.row); dfree(G.edge); dfree(G.progress); dfree(G.srcS); dfree(G.srcW); dfree(G.code); dfree(G.pos);
    G = GridTri();
}

// Is the factor grid-like?  lower: deps of row r within {r-1, r-w}, the r-1 one never across a grid
// line (r % w != 0); upper: {r+1, r+w}, (r+1) % w != 0.  Returns w (0 = no).
int32_t grid_width(int32_t n, const std::vector<int32_t> &ptr1, const std::vector<int32_t> &node1, bool lower)
{
    int32_t w = 0;
    for (int32_t r = 0; r < n; ++r)
        for (int32_t k = ptr1[r] - 1; k < ptr1[r + 1] - 1; ++k) {
            const int32_t dlt = lower ? r - (node1[k] - 1) : (node1[k] - 1) - r;
            if (dlt <= 0) return 0;
            if (dlt == 1) continue;
            if (!w) w = dlt;
            if (dlt != w) return 0;
        }
    if (w < 2) return 0;
    for (int32_t r = 0; r < n; ++r) {
        if (ptr1[r + 1] - ptr1[r] > 2) return 0;
        for (int32_t k = ptr1[r] - 1; k < ptr1[r + 1] - 1; ++k) {
            const int32_t c = node1[k] - 1;
            if (lower && c == r - 1 && r % w == 0) return 0;
            if (!lower && c == r + 1 && (r + 1) % w == 0) return 0;
        }
        if (ptr1[r + 1] - ptr1[r] == 2 && node1[ptr1[r] - 1] == node1[ptr1[r]]) return 0;
    }
    return w;
}

// index work of the skewed layout (once per pattern), on the device from the factor's pattern there (0-based)
int build_grid(GridTri &G, int32_t n, int32_t w, const int32_t *dptr, const int32_t *dnode, bool lower)
{
    free_grid(G);
    G.w = w;
    G.nj = (n + w - 1) / w;
    G.NI = (w + 63) / 64;
    G.S = (G.nj + 63 + 31) / 32 * 32;                         // a multiple of every look-ahead depth
    G.NP = (int64_t)G.NI * G.S * 64;
    if (G.NP >= INT32_MAX) return SGM_OK;                     // (positions are int32)
    hipStream_t st = g_rt.stream;
    int32_t *flags = nullptr;
    SGM_TRY(dalloc(&G.rec, (size_t)G.NP));
    SGM_TRY(dalloc(&G.row, (size_t)G.NP));
    SGM_TRY(dalloc(&G.edge, (size_t)G.NI * (G.S + kEdgePad)));
    SGM_TRY(dalloc(&G.progress, (size_t)G.NI + 1));
    SGM_TRY(dalloc(&G.srcS, (size_t)G.NP));
    SGM_TRY(dalloc(&G.srcW, (size_t)G.NP));
    SGM_TRY(dalloc(&G.code, (size_t)G.NP));
    SGM_TRY(dalloc(&G.pos, (size_t)std::max(n, 1)));
    SGM_TRY(dalloc(&flags, 2));
    SGM_HIP(hipMemsetAsync(G.row, 0xff, (size_t)G.NP * 4, st));       // -1 = padding / no such term
    SGM_HIP(hipMemsetAsync(G.srcS, 0xff, (size_t)G.NP * 4, st));
    SGM_HIP(hipMemsetAsync(G.srcW, 0xff, (size_t)G.NP * 4, st));
    SGM_HIP(hipMemsetAsync(G.code, 0, (size_t)G.NP, st));
    SGM_HIP(hipMemsetAsync(flags, 0, 8, st));
    SGM_HIP(hipMemsetAsync(G.edge, 0, (size_t)G.NI * (G.S + kEdgePad) * 8, st));
    if (n) hipLaunchKernelGGL(k_grid_build, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, w, G.nj, G.S, lower ? 1 : 0, dptr, dnode,
                              G.row, G.srcS, G.srcW, G.code, G.pos, flags);
    int32_t hf[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(hf, flags, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    dfree(flags);
    SGM_HIP(e);
    G.order = hf[0] && hf[1] ? 2 : hf[1] ? 1 : 0;
    G.on = true;
    return SGM_OK;
}

// records in the skewed layout (every setup), from the factor's values on the device
int refresh_grid_values(GridTri &G, const double *val)
{
    if (!G.on) return SGM_OK;
    hipLaunchKernelGGL(k_grid_records, dim3(vec_grid(G.NP)), dim3(kBlock), 0, g_rt.stream, G.NP, (const int32_t *)G.srcS,
                       (const int32_t *)G.srcW, (const uint8_t *)G.code, G.order, val, G.rec);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

void trsv_grid(const GridTri &G, double *xp, const int *flag, int spin_limit, int32_t *sticky)
{
    hipStream_t st = g_rt.stream;
    constexpr int depth = kStripDepth;
    constexpr int one_xcd = 0;            // (all strips on one XCD measured 0.83 vs 0.87 ms at 1000^2, 2.06 vs 1.85 at 2000^2: within noise, off)
    // 96 KiB of (unused) dynamic LDS per workgroup: at most ONE strip per CU, so that no two chain waves share a SIMD
    constexpr size_t lds_pad = (size_t)96 * 1024;
#define STRIP_K(DD, OO, LL)                                                                                                      \
    do {                                                                                                                         \
        static bool attr = false;                                                                                                \
        if (!attr) { (void)hipFuncSetAttribute((const void *)k_trsv_strip<DD, kStripChunk, OO, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pad); attr = true; } \
        hipLaunchKernelGGL((k_trsv_strip<DD, kStripChunk, OO, LL>), dim3(one_xcd ? G.NI * 8 : G.NI), dim3(192), lds_pad, st, G.NI, G.S, \
                           (const StripRec *)G.rec, xp, G.edge, G.progress, flag, one_xcd, spin_limit, sticky);                  \
    } while (0)
    // look-ahead (kStripDepth): 32 register slots with 20 steps in flight (3 memory operations per step, vmcnt counts to 63),
    // 32 with all 32 in flight (the compiler then drains the queue once per trip of the unrolled loop), or 16
    if (depth == 20) { if (G.order == 0) STRIP_K(32, 0, 20); else if (G.order == 1) STRIP_K(32, 1, 20); else STRIP_K(32, 2, 20); }
    else if (depth >= 32) { if (G.order == 0) STRIP_K(32, 0, 32); else if (G.order == 1) STRIP_K(32, 1, 32); else STRIP_K(32, 2, 32); }
    else { if (G.order == 0) STRIP_K(16, 0, 16); else if (G.order == 1) STRIP_K(16, 1, 16); else STRIP_K(16, 2, 16); }
#undef STRIP_K
}

// z = (I+U)^-1 D^-1 (I+L)^-1 r through the strip path
void apply_grid(const IlduState *S, const double *r, double *z, const int *flag, int spin_limit, int32_t *sticky)
{
    hipStream_t st = g_rt.stream;
    const int gl = vec_grid(S->gL.NP), gu = vec_grid(S->gU.NP);
    hipLaunchKernelGGL(k_grid_gather, dim3(gl), dim3(kBlock), 0, st, S->gL.NP, S->gL.rec, r, (const int32_t *)S->gL.row, S->gL.progress,
                       S->gL.NI + 1, reinterpret_cast<unsigned long long *>(S->gL.edge), (int64_t)S->gL.NI * (S->gL.S + kEdgePad), flag);
    trsv_grid(S->gL, S->gxL, flag, spin_limit, sticky);                                       // (I+L) x = b
    hipLaunchKernelGGL(k_grid_transition, dim3(gu), dim3(kBlock), 0, st, S->gU.NP, S->gU.rec, (const double *)S->gxL,
                       (const int32_t *)S->gmapLU, (const double *)S->gDp, S->gU.progress, S->gU.NI + 1,
                       reinterpret_cast<unsigned long long *>(S->gU.edge), (int64_t)S->gU.NI * (S->gU.S + kEdgePad), flag);       // x = x / D
    trsv_grid(S->gU, S->gxU, flag, spin_limit, sticky);                                       // (I+U) x = x
    hipLaunchKernelGGL(k_grid_scatter, dim3(gu), dim3(kBlock), 0, st, S->gU.NP, z, (const double *)S->gxU,
                       (const int32_t *)S->gU.row, flag);
}

// Dependency levels of both factors, their row-space copies when they have few levels (index work when the pattern is
// new, values always) and the work vector of the row-space sweeps.  At setup when no pipelined path serves the pattern,
// otherwise on first need.
int ensure_host_pattern(IlduState *S);
int ensure_levels(IlduState *S)
{
    if (S->levels_ready) return SGM_OK;
    const int32_t n = S->n;
    const bool fresh = !S->levels_pattern;
    // factors of a few levels: all index work on the device; otherwise from the pattern's host copy
    bool ls = false, us = false;
    SGM_TRY(tri_levels_device(S->L, n, S->dLptr, S->dLnode, &ls));
    SGM_TRY(tri_levels_device(S->U, n, S->dUptr, S->dUnode, &us));
    if (!S->L.have_levels || !S->U.have_levels) SGM_TRY(ensure_host_pattern(S));
    SGM_TRY(tri_levels_dev(S->L, n, S->hLptr, S->hLnode, S->dLptr, S->dLnode, S->dLval, true));
    SGM_TRY(tri_levels_dev(S->U, n, S->hUptr, S->hUnode, S->dUptr, S->dUnode, S->dUval, false));
    if (fresh) {
        dfree(S->xpL);
        S->xpL = nullptr;
        SGM_TRY(dalloc(&S->xpL, (size_t)n + kNarrow));     // + scratch slots of the level walker
    }
    S->rows_n0 = 0;
    S->rows_fin = false;
    if (S->L.rows_on && S->U.rows_on && !S->L.row_levels.empty() && !S->U.row_levels.empty()) {
        const auto &l0 = S->L.row_levels.front(), &ll = S->L.row_levels.back(), &u0 = S->U.row_levels.front();
        if (l0.c == 0 && l0.row0 == 0) S->rows_n0 = l0.e - l0.b;
        S->rows_fin = S->L.row_levels.size() >= 2 && u0.c == 0 && u0.row0 >= 0 && u0.row0 == ll.row0 && u0.e - u0.b == ll.e - ll.b;
    }
    S->levels_pattern = true;
    S->levels_ready = true;
    return SGM_OK;
}

// The level walkers' structures (records, schedules, ring copies, the L -> U hand-over in position space): built when
// neither a pipelined path nor the row-space sweeps serve the pattern, otherwise on first need (an option switched
// off, a retired pipeline).
int ensure_walkers(IlduState *S)
{
    SGM_TRY(ensure_levels(S));
    if (S->walk_ready) return SGM_OK;
    SGM_TRY(ensure_host_pattern(S));
    const int32_t n = S->n;
    const bool fresh = !S->walk_pattern;
    SGM_TRY(tri_walkers(S->L, n, S->hLptr, S->hLnode, S->dLval));
    SGM_TRY(tri_walkers(S->U, n, S->hUptr, S->hUnode, S->dUval));
    if (fresh) {
        dfree(S->xpU); dfree(S->Dp); dfree(S->mapLU);
        S->xpU = S->Dp = nullptr; S->mapLU = nullptr;
        SGM_TRY(dalloc(&S->xpU, (size_t)n + kNarrow));
        SGM_TRY(dalloc(&S->Dp, (size_t)std::max(n, 1)));
        SGM_TRY(dalloc(&S->mapLU, (size_t)std::max(n, 1)));
        std::vector<int32_t> map((size_t)std::max(n, 1));
        for (int32_t p = 0; p < n; ++p) map[p] = S->L.h_pos[S->U.h_order[p]];
        if (n) SGM_TRY(copy_big(S->mapLU, map.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    if (n) hipLaunchKernelGGL(k_pos_diag, dim3(vec_grid(n)), dim3(kBlock), 0, g_rt.stream, (int64_t)n, (const int32_t *)S->U.order,
                              (const double *)S->D, S->Dp);                  // D in U's level order
    S->walk_pattern = true;
    S->walk_ready = true;
    return SGM_OK;
}

// the factor values on the host (sgm_pc_get only)
int ensure_host_values(IlduState *S)
{
    if (S->host_vals) return SGM_OK;
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    S->hLval.resize((size_t)S->nnzL);
    S->hUval.resize((size_t)S->nnzU);
    S->hD.resize((size_t)S->n);
    if (!S->hLval.empty()) SGM_TRY(copy_big(S->hLval.data(), S->dLval, S->hLval.size() * 8, hipMemcpyDeviceToHost));
    if (!S->hUval.empty()) SGM_TRY(copy_big(S->hUval.data(), S->dUval, S->hUval.size() * 8, hipMemcpyDeviceToHost));
    if (S->n) SGM_TRY(copy_big(S->hD.data(), S->D, (size_t)S->n * 8, hipMemcpyDeviceToHost));
    S->host_vals = true;
    return SGM_OK;
}

void trsv(const TriFactor &T, double *xp, const int *flag);
// z = (I+U)^-1 D^-1 (I+L)^-1 r through the level-scheduled walkers
void apply_levels(const IlduState *S, const double *r, double *z, const int *flag)
{
    hipStream_t st = g_rt.stream;
    const int64_t n = S->n;
    const int g = vec_grid(n);
    hipLaunchKernelGGL(k_perm_gather, dim3(g), dim3(kBlock), 0, st, n, S->xpL, r, (const int32_t *)S->L.order, flag);
    trsv(S->L, S->xpL, flag);                                             // (I+L) x = b
    hipLaunchKernelGGL(k_lu_transition, dim3(g), dim3(kBlock), 0, st, n, S->xpU, (const double *)S->xpL,
                       (const int32_t *)S->mapLU, (const double *)S->Dp, flag);                     // x = x / D
    trsv(S->U, S->xpU, flag);                                             // (I+U) x = x
    hipLaunchKernelGGL(k_perm_scatter, dim3(g), dim3(kBlock), 0, st, n, z, (const double *)S->xpU,
                       (const int32_t *)S->U.order, flag);
}

// the same through the row-space levels (both factors a few wide levels): one launch per level, nothing else
bool rows_serve(const IlduState *S) { return S->opt.ildu_rows && S->levels_ready && S->L.rows_on && S->U.rows_on; }
void launch_rows(const TriFactor &T, const TriFactor::RowLevel &L, int mode, const double *r, double *y, const double *D, double *z,
                 int32_t n0, const int *flag)
{
    hipStream_t st = g_rt.stream;
    const int32_t b = L.b, e = L.e;
    // two rows per lane: row and position of the level of one parity (an odd start is peeled), and no pair astride n0 -- n0 even, or
    // the level's rows all on one side of it with the pairs aligned to the level's own start
    const bool pairs_ok = L.row0 >= 0 && ((L.row0 ^ b) & 1) == 0 &&
                          (((n0 & 1) == 0 && (L.row0 & 1) == 0) || L.row0 >= n0 || L.row0 + (e - b) <= n0);
    if (pairs_ok && L.c >= 1 && L.c <= 4) {
        const bool coded = T.rcode != nullptr;
        const dim3 g2(8 * (((e - b + 2 * kBlock - 1) / (2 * kBlock) + 7) / 8));
#define R2_M(CC, MM, CD)                                                                                                        \
    hipLaunchKernelGGL((k_trsv_rows2<CC, MM, CD>), g2, dim3(kBlock), 0, st, (const int32_t *)T.rq, (const double *)T.rv, T.rc,   \
                       (const uint32_t *)T.rcode, (const int32_t *)T.rdict, L.row0, b, e, r, y, D, z, n0, flag)
#define R2_C(CC, MM) do { if (coded) R2_M(CC, MM, true); else R2_M(CC, MM, false); } while (0)
#define R2(CC) do { if (mode == 0) R2_C(CC, 0); else if (mode == 1) R2_C(CC, 1); else R2_C(CC, 2); } while (0)
        switch (L.c) {
        case 1: R2(1); break;
        case 2: R2(2); break;
        case 3: R2(3); break;
        default: R2(4); break;
        }
#undef R2
#undef R2_C
#undef R2_M
        return;
    }
    const dim3 g(8 * (((e - b + kBlock - 1) / kBlock + 7) / 8));           // (a multiple of 8: see the tile map in the kernel)
#define ROWS_M(CC, MM)                                                                                                         \
    hipLaunchKernelGGL((k_trsv_rows<CC, MM>), g, dim3(kBlock), 0, st, (const int32_t *)T.rq, (const double *)T.rv, (uint32_t)T.nstride, \
                       T.rc, (const int32_t *)T.order, L.row0, b, e, r, y, D, z, n0, flag)
#define ROWS(CC)                                                                                                               \
    do {                                                                                                                      \
        if (mode == 0) ROWS_M(CC, 0); else if (mode == 1) ROWS_M(CC, 1); else ROWS_M(CC, 2);                                    \
    } while (0)
    switch (L.c) {
    case 0: ROWS(0); break;
    case 1: ROWS(1); break;
    case 2: ROWS(2); break;
    case 3: ROWS(3); break;
    case 4: ROWS(4); break;
    case 5: case 6: ROWS(6); break;
    case 7: case 8: ROWS(8); break;
    default: ROWS(-1); break;
    }
#undef ROWS
#undef ROWS_M
}
// (fused: ildu_rows = 1 -- L's first level, when it is rows 0 .. n0-1 without entries, is not copied; the L level that is
// also U's level 0 is finished in the L sweep.  ildu_rows = 2 launches every level of both sweeps.)
void apply_rows(const IlduState *S, const double *r, double *z, const int *flag)
{
    const auto &Ls = S->L.row_levels, &Us = S->U.row_levels;
    const bool fused = S->opt.ildu_rows == 1;
    const int32_t n0 = fused ? S->rows_n0 : 0;
    const bool fin = fused && S->rows_fin;
    for (size_t k = n0 > 0 ? 1 : 0; k < Ls.size(); ++k)                    // (I+L) y = r
        launch_rows(S->L, Ls[k], fin && k + 1 == Ls.size() ? 1 : 0, r, S->xpL, S->D, z, n0, flag);
    for (size_t k = fin ? 1 : 0; k < Us.size(); ++k)                       // (I+U) z = y / D
        launch_rows(S->U, Us[k], 2, r, S->xpL, S->D, z, n0, flag);
}

// both factors exactly two row-space levels, the outer ones entry-less (a two-colour ordering): k_trsv_rows_cg's case
bool rows_two_level(const IlduState *S)
{
    return rows_serve(S) && S->opt.ildu_rows == 1 && S->rows_n0 > 0 && S->rows_fin && S->L.row_levels.size() == 2 && S->U.row_levels.size() == 2;
}
constexpr int kRowsCgGrid = 2048;            // blocks per launch (grid-stride): 2 x 2048 partial sums <= kMaxGrid
void launch_rows_cg(const TriFactor &T, const TriFactor::RowLevel &L, int mode, double *r, const double *q, ScalarRef res2, ScalarRef dpr,
                    const double *D, double *z, double *part, int grid, const int *flag, int gen)
{
    hipStream_t st = g_rt.stream;
    if (L.row0 >= 0 && ((L.row0 ^ L.b) & 1) == 0 && L.c >= 1 && L.c <= 4) {           // (both even, or both odd: the kernel peels the first row)
        const bool coded = T.rcode != nullptr;
#define ROWS2_M(CC, MM, CD)                                                                                                    \
    hipLaunchKernelGGL((k_trsv_rows_cg2<CC, MM, CD>), dim3(grid), dim3(kBlock), 0, st, (const int32_t *)T.rq, (const double *)T.rv, T.rc, \
                       (const uint32_t *)T.rcode, (const int32_t *)T.rdict, L.row0, L.b, L.e, r, q, res2, dpr, D, z, part, flag, gen)
#define ROWS2(CC) do { if (mode == 1) { if (coded) ROWS2_M(CC, 1, true); else ROWS2_M(CC, 1, false); }                          \
                       else { if (coded) ROWS2_M(CC, 2, true); else ROWS2_M(CC, 2, false); } } while (0)
        switch (L.c) {
        case 1: ROWS2(1); break;
        case 2: ROWS2(2); break;
        case 3: ROWS2(3); break;
        default: ROWS2(4); break;
        }
#undef ROWS2
#undef ROWS2_M
        return;
    }
#define ROWS_M(CC, MM)                                                                                                         \
    hipLaunchKernelGGL((k_trsv_rows_cg<CC, MM>), dim3(grid), dim3(kBlock), 0, st, (const int32_t *)T.rq, (const double *)T.rv, (uint32_t)T.nstride, \
                       T.rc, (const int32_t *)T.order, L.row0, L.b, L.e, r, q, res2, dpr, D, z, part, flag, gen)
#define ROWS(CC) do { if (mode == 1) ROWS_M(CC, 1); else ROWS_M(CC, 2); } while (0)
    switch (L.c) {
    case 0: ROWS(0); break;
    case 1: ROWS(1); break;
    case 2: ROWS(2); break;
    case 3: ROWS(3); break;
    case 4: ROWS(4); break;
    case 5: case 6: ROWS(6); break;
    case 7: case 8: ROWS(8); break;
    default: ROWS(-1); break;
    }
#undef ROWS
#undef ROWS_M
}

// triangular solve in position space: xp holds the right-hand side on entry, the solution on exit
void trsv(const TriFactor &T, double *xp, const int *flag)
{
    const int32_t n = (int32_t)T.h_order.size();
    hipStream_t st = g_rt.stream;
    for (const auto &L : T.schedule) {
        if (L.narrow) {
#define WALK(R, DD)                                                                                          \
    hipLaunchKernelGGL((k_trsv_walk<R, DD>), dim3(1), dim3(kTrsvBlock), 0, st, (const TrsvRec *)T.recs,       \
                       (const int32_t *)T.pq, (const double *)T.pv, (const int32_t *)T.level_ptr_dev, L.l0, L.l1, \
                       n, xp, flag)
#define RING(TT, R, DD, CC, AA)                                                                              \
    hipLaunchKernelGGL((k_trsv_walk_ring<TT, R, DD, CC, AA>), dim3(1), dim3(TT), 0, st, (const uint64_t *)T.dq, \
                       (const uint32_t *)T.dq32, (const double *)T.dv, (uint32_t)T.nstride, (const int32_t *)T.level_ptr_dev, L.l0, L.l1, n, xp, \
                       flag)
#define RINGC(TT, R, DD, AA)                                                  \
    do {                                                                      \
        if (L.c <= 2) RING(TT, R, DD, 2, AA); else if (L.c == 3) RING(TT, R, DD, 3, AA); else RING(TT, R, DD, 4, AA); \
    } while (0)
            if (L.ring && T.nstride < (size_t)500000000) {       // (32-bit byte offsets)
                // class = widest level of the run: <= 256, 512, 1024, 2048, 4096 rows (two rows per lane pair from 512 on)
                // (-1: levels of at most 64 rows -- chains: ONE wave, whose level barrier costs nothing)
                if (L.cls < 0) RINGC(64, 1, 4, 1);
                else if (L.cls == 0) RINGC(256, 1, 4, 1);
                else if (L.cls == 1) RINGC(256, 1, 4, 2);
                else if (L.cls == 2) RINGC(512, 1, 4, 2);
                else if (L.cls == 3) RINGC(1024, 1, 2, 2);
                else RINGC(1024, 2, 1, 2);
            } else if (L.cls <= 2) WALK(1, 2);
            else if (L.cls == 3) WALK(2, 1);
            else WALK(4, 1);
#undef RINGC
#undef RING
#undef WALK
        } else {
            const int32_t b = T.level_ptr[L.l0], e = T.level_ptr[L.l1];
            const dim3 g((e - b + kBlock - 1) / kBlock);
#define WSOA(CC) hipLaunchKernelGGL((k_trsv_wide_soa<CC>), g, dim3(kBlock), 0, st, (const int32_t *)T.wq, \
                                    (const double *)T.dv, (uint32_t)T.nstride, b, e, xp, flag)
            if (L.c <= 2 && T.nstride < (size_t)500000000) WSOA(2);
            else if (L.c == 3 && T.nstride < (size_t)500000000) WSOA(3);
            else if (L.c == 4 && T.nstride < (size_t)500000000) WSOA(4);
            else
                hipLaunchKernelGGL(k_trsv_wide, g, dim3(kBlock), 0, st, (const TrsvRec *)T.recs, (const int32_t *)T.pq,
                                   (const double *)T.pv, b, e, xp, flag);
#undef WSOA
        }
    }
}

// grid_width without a host copy of the pattern: 0 = not grid-like
int grid_width_device(int32_t n, const int32_t *dptr, const int32_t *dnode, bool lower, int32_t *w_out)
{
    *w_out = 0;
    if (n < 1) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *info = nullptr;
    SGM_TRY(dalloc(&info, 4));
    struct Tmp { int32_t *&a; ~Tmp() { dfree(a); } } guard{info};
    const int32_t init[4] = {INT32_MAX, 0, 0, 0};
    int32_t h[4];
    SGM_HIP(hipMemcpyAsync(info, init, 16, hipMemcpyHostToDevice, st));
    const int grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_grid_detect1, dim3(grid), dim3(kBlock), 0, st, n, lower ? 1 : 0, dptr, dnode, info);
    SGM_HIP(hipMemcpyAsync(h, info, 16, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    if (h[2] || h[0] == INT32_MAX || h[0] != h[1] || h[0] < 2) return SGM_OK;
    hipLaunchKernelGGL(k_grid_detect2, dim3(grid), dim3(kBlock), 0, st, n, lower ? 1 : 0, h[0], dptr, dnode, info);
    SGM_HIP(hipMemcpyAsync(h, info, 16, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    if (!h[3]) *w_out = h[0];
    return SGM_OK;
}
// the factorisation order of a grid-like factor pair: rows sorted (stably) on their anti-diagonal
int grid_factor_order(IlduState *S, int32_t n, int32_t w, int32_t h)
{
    hipStream_t st = g_rt.stream;
    const int32_t nj = (n + w - 1) / w;
    const int32_t nkeys = h > 0 ? w + h + (int32_t)((n + (int64_t)w * h - 1) / ((int64_t)w * h)) : w + nj;   // keys 0 .. w-1 + nj-1
    int bits = 1;
    while ((1 << bits) < nkeys) ++bits;
    int32_t *key = nullptr, *key2 = nullptr, *rows = nullptr, *count = nullptr;
    void *tmp = nullptr;
    struct Tmp { int32_t *&a, *&b, *&c, *&d; void *&t; ~Tmp() { dfree(a); dfree(b); dfree(c); dfree(d); if (t) (void)hipFree(t); } }
        guard{key, key2, rows, count, tmp};
    SGM_TRY(dalloc(&key, (size_t)n));
    SGM_TRY(dalloc(&key2, (size_t)n));
    SGM_TRY(dalloc(&rows, (size_t)n));
    SGM_TRY(dalloc(&count, (size_t)nkeys));
    dfree(S->forder);
    S->forder = nullptr;
    SGM_TRY(dalloc(&S->forder, (size_t)n));
    SGM_HIP(hipMemsetAsync(count, 0, (size_t)nkeys * 4, st));
    hipLaunchKernelGGL(k_grid_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, w, h, key, rows, count);
    size_t tb = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, (const int32_t *)key, key2, (const int32_t *)rows, S->forder, n, 0, bits, st);
    SGM_HIP(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    SGM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, (const int32_t *)key, key2, (const int32_t *)rows, S->forder, n, 0, bits, st));
    std::vector<int32_t> hc((size_t)nkeys);
    SGM_HIP(hipMemcpyAsync(hc.data(), count, (size_t)nkeys * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    S->flevel_ptr.assign(1, 0);
    for (int32_t k = 0; k < nkeys; ++k)
        if (hc[k]) S->flevel_ptr.push_back(S->flevel_ptr.back() + hc[k]);
    return SGM_OK;
}

// The factors' patterns on the device (0-based) from the part's CSR-order arrays ...
// The real entries of an ELLPACK part -- the first degrees(i) slots of every row, in slot order: what the reference's cursor
// hands out (ellpack_graphs.f90:310-369) -- as 0-based CSR arrays.  Padding slots (the last neighbour repeated, value 0) and
// empty rows' node = 0 never appear.  ELL = false: the rows are fixed-length CSR rows (an ELLPACK matrix over ranks,
// sgm_ell_create_dist, whose padding slots are stored entries for the product's sake) and the same first degrees(i) are taken.
template <bool ELL>
__global__ void k_real_entries(int32_t n, const int32_t *__restrict__ src_ptr, const int32_t *__restrict__ scol, const double *__restrict__ sval,
                               const int32_t *__restrict__ rowptr, int32_t *__restrict__ col, double *__restrict__ val)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t b = rowptr[i], d = rowptr[i + 1] - b;
    const int64_t s0 = ELL ? i : src_ptr[i];
    for (int32_t k = 0; k < d; ++k) {
        const int64_t s = ELL ? (int64_t)k * n + s0 : s0 + k;
        col[b + k] = scol[s];
        val[b + k] = sval[s];
    }
}
int real_entries_as_csr(const Part &p, bool ell, Part &v)
{
    hipStream_t st = g_rt.stream;
    const int32_t n = p.n;
    if (!p.edeg && n && (!ell || p.max_d)) return fail(SGM_ERR_UNSUPPORTED, "ILDU(0) on an ELLPACK matrix needs its degrees (this handle has none)");
    v.n = n;
    v.ncol_own = p.ncol_own;
    v.n_halo = p.n_halo;
    v.lean = false;
    SGM_TRY(dalloc(&v.rowptr, (size_t)n + 1));
    SGM_HIP(hipMemsetAsync(v.rowptr, 0, ((size_t)n + 1) * 4, st));
    if (n && p.edeg) SGM_HIP(hipMemcpyAsync(v.rowptr, p.edeg, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    size_t tb = 0;
    void *tmp = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, v.rowptr, v.rowptr, n + 1, st);
    SGM_HIP(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    struct Tmp { void *t; ~Tmp() { (void)hipFree(t); } } guard{tmp};
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, v.rowptr, v.rowptr, n + 1, st));
    int32_t total = 0;
    SGM_HIP(hipMemcpyAsync(&total, v.rowptr + n, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    v.nnz = total;
    SGM_TRY(dalloc(&v.col, (size_t)total + 4));
    SGM_TRY(dalloc(&v.val, (size_t)total + 2));
    SGM_HIP(hipMemsetAsync(v.col + total, 0, 4 * sizeof(int32_t), st));
    SGM_HIP(hipMemsetAsync(v.val + total, 0, 2 * sizeof(double), st));
    const dim3 grid((n + kBlock - 1) / kBlock);
    if (n && ell)
        hipLaunchKernelGGL(k_real_entries<true>, grid, dim3(kBlock), 0, st, n, (const int32_t *)nullptr, (const int32_t *)p.ecol,
                           (const double *)p.eval, (const int32_t *)v.rowptr, v.col, v.val);
    else if (n)
        hipLaunchKernelGGL(k_real_entries<false>, grid, dim3(kBlock), 0, st, n, (const int32_t *)p.rowptr, (const int32_t *)p.col,
                           (const double *)p.val, (const int32_t *)v.rowptr, v.col, v.val);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

int ildu_pattern(IlduState *S, const Part &P, int32_t own)
{
    const int32_t n = P.n;
    hipStream_t st = g_rt.stream;
    int32_t *longest = nullptr;
    SGM_TRY(dalloc(&S->dLptr, (size_t)n + 1));
    SGM_TRY(dalloc(&S->dUptr, (size_t)n + 1));
    SGM_TRY(dalloc(&longest, 2));
    struct Tmp { int32_t *&a; void *t = nullptr; ~Tmp() { dfree(a); if (t) (void)hipFree(t); } } guard{longest};
    SGM_HIP(hipMemsetAsync(longest, 0, 8, st));
    const int grid = (n + 1 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_ildu_count, dim3(grid), dim3(kBlock), 0, st, n, own, (const int32_t *)P.rowptr, (const int32_t *)P.col,
                       S->dLptr, S->dUptr, longest);
    size_t tb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, S->dLptr, S->dLptr, n + 1, st);
    SGM_HIP(hipMalloc(&guard.t, std::max<size_t>(tb, 16)));
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(guard.t, tb, S->dLptr, S->dLptr, n + 1, st));
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(guard.t, tb, S->dUptr, S->dUptr, n + 1, st));
    int32_t tot[2] = {0, 0}, lg[2] = {0, 0};
    SGM_HIP(hipMemcpyAsync(&tot[0], S->dLptr + n, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(&tot[1], S->dUptr + n, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(lg, longest, 8, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    S->nnzL = tot[0]; S->nnzU = tot[1];
    S->maxL = lg[0]; S->maxU = lg[1];
    SGM_TRY(dalloc(&S->dLnode, (size_t)std::max(tot[0], 1)));
    SGM_TRY(dalloc(&S->dUnode, (size_t)std::max(tot[1], 1)));
    if (n) hipLaunchKernelGGL(k_ildu_split, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, own, (const int32_t *)P.rowptr,
                              (const int32_t *)P.col, (const int32_t *)S->dLptr, S->dLnode, (const int32_t *)S->dUptr, S->dUnode);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}
// ... and their 1-based host copies, when something asks: sgm_pc_get, the host's level pass (factors of many levels), the
// grid / slab detection, the level walkers' index work
int ensure_host_pattern(IlduState *S)
{
    if (!S->hLptr.empty() || !S->dLptr) return SGM_OK;
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    auto down = [](std::vector<int32_t> &h, const int32_t *d, size_t cnt) -> int {
        h.resize(cnt);
        if (cnt) SGM_TRY(copy_big(h.data(), d, cnt * 4, hipMemcpyDeviceToHost));
        for (auto &v : h) v += 1;
        return SGM_OK;
    };
    SGM_TRY(down(S->hLptr, S->dLptr, (size_t)S->n + 1));
    SGM_TRY(down(S->hUptr, S->dUptr, (size_t)S->n + 1));
    SGM_TRY(down(S->hLnode, S->dLnode, (size_t)S->nnzL));
    SGM_TRY(down(S->hUnode, S->dUnode, (size_t)S->nnzU));
    return SGM_OK;
}

}  // namespace

namespace sgm {

int pc_kind(sgm_pc pc) { return pc ? pc->kind : 0; }
// an apply that is a handful of launches (Jacobi; ILDU through the strip / slab pipeline or over a few wide levels) lets the
// solvers queue a whole batch of iterations between two looks at the stop flag; thousands of level launches per apply do not
bool pc_apply_is_short(sgm_pc pc)
{
    if (!pc || pc->kind == SGM_PC_JACOBI) return true;
    // (a colour-ordered matrix has two or three levels per factor: its level-scheduled apply is seven to nine launches)
    for (const auto &S : pc->ild) {
        if (pc->opt.ildu_strips && (S.grid_ok || S.slab_ok)) continue;
        if (!S.levels_ready) return false;
        if (pc->opt.ildu_rows && S.L.rows_on && S.U.rows_on) continue;         // at most 2 * kRowLevels launches
        if (!S.walk_ready || 3 + S.L.schedule.size() + S.U.schedule.size() > 35) return false;
    }
    return true;
}
const double *pc_idiag(sgm_pc pc, size_t part) { return pc->parts[part].idiag; }

// ILDU(0) of the colour-ordered matrix (option ildu_reorder): the permuted matrix the factors belong to (null: none / natural
// order).  A solver that finds one runs in the permuted order: x and b through pc_permute_vec once each way, the products on
// this matrix, and pc_in_permuted(pc, true) around the solve so that the applies skip their own two permutations.
// Only for the very matrix it was made from, unchanged since (the reference lets any matrix be solved with any preconditioner:
// for another one the applies permute r and z themselves).
sgm_mat pc_permuted_matrix(sgm_pc pc, sgm_mat A)
{
    if (!pc || pc->kind != SGM_PC_ILDU0 || pc->ro.empty() || !pc->Ap || !A) return nullptr;
    return pc->Ap_serial == A->serial && pc->Ap_version == A->version ? pc->Ap : nullptr;
}
void pc_in_permuted(sgm_pc pc, bool on) { if (pc) pc->in_permuted = on; }

// The sticky abort word of a preconditioner whose apply runs through a pipelined triangular solve right now (null otherwise:
// nothing to watch).  Whoever synchronises after such applies copies it back; nonzero = some sweep gave up and its result --
// and everything computed from it -- is not to be used.
int32_t *pc_abort_word(sgm_pc pc)
{
    if (!pc || pc->kind != SGM_PC_ILDU0 || !pc->opt.ildu_strips || !pc->abort_sticky) return nullptr;
    for (const auto &S : pc->ild)
        if (S.grid_ok || S.slab_ok) return pc->abort_sticky;
    return nullptr;
}
// After an abort: the pipelines of this handle are retired (every later apply takes the level walkers, built here if they
// were never needed) and the word is cleared.  Loud on stderr: it should not happen on a GPU this process owns.
int pc_retire_pipelines(sgm_pc pc)
{
    fprintf(stderr, "[sigma_hip] ILDU pipelined triangular solve gave up waiting (preempted / shared GPU?): result discarded, "
                    "redone with the level-scheduled solves; the pipeline is retired for this preconditioner\n");
    for (auto &S : pc->ild) {
        S.grid_ok = false;
        S.slab_ok = false;
        SGM_TRY(ensure_levels(&S));
        if (!rows_serve(&S)) SGM_TRY(ensure_walkers(&S));
    }
    pc->retired += 1;
    SGM_HIP(hipMemsetAsync(pc->abort_sticky, 0, sizeof(int32_t), g_rt.stream));
    return SGM_OK;
}

// dst[p(i) - 1] = src[i]  /  dst[i] = src[p(i) - 1]
__global__ void k_perm_to(int32_t n, const int32_t *__restrict__ p1, const double *__restrict__ src, double *__restrict__ dst,
                          const int *__restrict__ flag)
{
    if (flag && *flag) return;
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[p1[i] - 1] = src[i];
}
__global__ void k_perm_from(int32_t n, const int32_t *__restrict__ p1, const double *__restrict__ src, double *__restrict__ dst,
                            const int *__restrict__ flag)
{
    if (flag && *flag) return;
    int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t stride = gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[p1[i] - 1];
}

static int pc_apply_parts_ordered(sgm_pc pc, sgm_mat A, const double *const *r, double *const *z, const int *const *flags);

// part ip's slice: dst = P src (to_permuted) or dst = P^T src
void pc_permute_vec(sgm_pc pc, size_t ip, const double *src, double *dst, bool to_permuted)
{
    const int32_t n = pc->ro[ip].n;
    if (!n) return;
    if (to_permuted) hipLaunchKernelGGL(k_perm_to, dim3(vec_grid(n)), dim3(kBlock), 0, g_rt.stream, n, (const int32_t *)pc->ro[ip].perm, src, dst, (const int *)nullptr);
    else hipLaunchKernelGGL(k_perm_from, dim3(vec_grid(n)), dim3(kBlock), 0, g_rt.stream, n, (const int32_t *)pc->ro[ip].perm, src, dst, (const int *)nullptr);
}

// CG's "r -= alpha q; z = M^-1 r; partial sums of r.z" as the two launches of a two-level row-space factorisation
// (k_trsv_rows_cg).  false: this preconditioner is not of that kind here -- the caller launches the three steps itself.
// *count = partial sums left in `part`.
// pc_cg_fused_rows: 0 = not that kind, else n0 -- the caller first updates r(0 .. n0-1) -= alpha q (the rows without L
// entries: nobody else writes them), then calls pc_cg_fused, which updates the others as its first sweep reaches them
int32_t pc_cg_fused_rows(sgm_pc pc, size_t ip)
{
    if (!pc || pc->kind != SGM_PC_ILDU0 || ip >= pc->ild.size() || (!pc->ro.empty() && !pc->in_permuted)) return 0;
    const IlduState *S = &pc->ild[ip];
    if ((S->opt.ildu_strips && (S->grid_ok || S->slab_ok)) || !S->levels_ready || !rows_two_level(S)) return 0;
    return S->rows_n0;
}
bool pc_cg_fused(sgm_pc pc, size_t ip, ScalarRef res2, ScalarRef dpr, const double *q, double *r, double *z, double *part, int *count, const int *flag, int gen)
{
    if (!pc_cg_fused_rows(pc, ip)) return false;
    const IlduState *S = &pc->ild[ip];
    const auto &L1 = S->L.row_levels[1], &U1 = S->U.row_levels[1];
    auto grid_for = [](int32_t rows) { return 8 * std::max(1, std::min(kRowsCgGrid / 8, ((rows + kBlock - 1) / kBlock + 7) / 8)); };
    const int g1 = grid_for(L1.e - L1.b), g2 = grid_for(U1.e - U1.b);
    launch_rows_cg(S->L, L1, 1, r, q, res2, dpr, S->D, z, part, g1, flag, gen);
    launch_rows_cg(S->U, U1, 2, r, q, res2, dpr, S->D, z, part + g1, g2, flag, gen);
    *count = g1 + g2;
    return true;
}

int pc_apply_parts(sgm_pc pc, sgm_mat A, const double *const *r, double *const *z, const int *const *flags)
{
    if (pc->kind == SGM_PC_ILDU0 && !pc->ro.empty() && !pc->in_permuted) {
        // z = P^T M^-1 P r, part by part: into the colour order, the sweeps there, back
        hipStream_t st = g_rt.stream;
        const size_t P = pc->ro.size();
        std::vector<const double *> rr(P);
        std::vector<double *> zz(P);
        for (size_t ip = 0; ip < P; ++ip) {
            const auto &R = pc->ro[ip];
            if (R.n) hipLaunchKernelGGL(k_perm_to, dim3(vec_grid(R.n)), dim3(kBlock), 0, st, R.n, (const int32_t *)R.perm, r[ip], R.rp, flags ? flags[ip] : nullptr);
            rr[ip] = R.rp; zz[ip] = R.zp;
        }
        SGM_TRY(pc_apply_parts_ordered(pc, A, rr.data(), zz.data(), flags));
        for (size_t ip = 0; ip < P; ++ip) {
            const auto &R = pc->ro[ip];
            if (R.n) hipLaunchKernelGGL(k_perm_from, dim3(vec_grid(R.n)), dim3(kBlock), 0, st, R.n, (const int32_t *)R.perm, (const double *)R.zp, z[ip], flags ? flags[ip] : nullptr);
        }
        SGM_HIP(hipGetLastError());
        return SGM_OK;
    }
    return pc_apply_parts_ordered(pc, A, r, z, flags);
}

static int pc_apply_parts_ordered(sgm_pc pc, sgm_mat A, const double *const *r, double *const *z, const int *const *flags)
{
    hipStream_t st = g_rt.stream;
    if (pc->kind == SGM_PC_JACOBI) {
        for (size_t ip = 0; ip < A->parts.size(); ++ip) {
            const int64_t n = A->parts[ip].n;
            hipLaunchKernelGGL(k_scale_by, dim3(vec_grid(n)), dim3(kBlock), 0, st, n, pc->parts[ip].idiag, r[ip], z[ip],
                               flags ? flags[ip] : nullptr);
        }
    } else {
        for (size_t ip = 0; ip < pc->ild.size(); ++ip) {      // block-Jacobi over the parts: no exchange
            const IlduState *S = &pc->ild[ip];
            const int *flag = flags ? flags[ip] : nullptr;
            const int spin = pc->opt.pipeline_spin_limit > 0 ? pc->opt.pipeline_spin_limit : kStripSpinLimit;
            if (S->grid_ok && S->opt.ildu_strips) {                 // grid-like factors: one strip-pipelined launch per sweep
                apply_grid(S, r[ip], z[ip], flag, spin, pc->abort_sticky);
                continue;
            }
            if (S->slab_ok && S->opt.ildu_strips) {                 // 3-D grid factors: one slab-pipelined launch per sweep
                slab3_apply(S->slab, r[ip], z[ip], flag, spin, pc->abort_sticky);
                continue;
            }
            SGM_TRY(ensure_levels(&pc->ild[ip]));              // (built on first need when a pipelined path served the pattern so far)
            if (rows_serve(S)) { apply_rows(S, r[ip], z[ip], flag); continue; }
            SGM_TRY(ensure_walkers(&pc->ild[ip]));
            apply_levels(S, r[ip], z[ip], flag);
        }
    }
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

}  // namespace sgm

extern "C" {

static int pc_setup_ordered(sgm_pc pc, sgm_mat A);
int sgm_pc_info(sgm_pc pc, int32_t part, int32_t *out4, double *est_us, char *path_name, int len);

// SGM_TRACE: one line per setup naming the sweeps that will serve the applies (sgm_pc_info) -- a chain-bound `ldu()` says so
static void trace_setup(sgm_pc pc)
{
    if (!trace_on()) return;
    const size_t P = pc->kind == SGM_PC_JACOBI ? 1 : pc->ild.size();
    for (size_t ip = 0; ip < P && ip < 2; ++ip) {
        int32_t o[4]; double us = 0.0; char nm[160];
        if (sgm_pc_info(pc, (int32_t)ip, o, &us, nm, (int)sizeof nm) != SGM_OK) return;
        fprintf(stderr, "[sigma_hip] %s setup%s: %s, about %.0f us per apply%s%s\n", pc->kind == SGM_PC_JACOBI ? "jacobi" : "ildu",
                P > 1 ? (ip == 0 ? " (part 0)" : " (part 1)") : "", nm, us, o[3] ? ", colour-ordered" : "",
                o[2] >= 2 && !o[3] ? " -- a dependency chain: ldu(reorder=\"colour\") / option ildu_reorder makes it two bandwidth-bound sweeps" : "");
    }
}

static void free_reorder(sgm_pc pc)
{
    for (auto &R : pc->ro) { dfree(R.perm); dfree(R.rp); dfree(R.zp); dfree(R.hmap); for (int32_t *q : R.send_order) dfree(q); }
    pc->ro.clear();
}

int sgm_pc_setup(sgm_pc pc, sgm_mat A)
{
    SGM_TRY(require_init());
    if (!pc || !A) return fail(SGM_ERR_BAD_ARG, "sgm_pc_setup: null argument");
    const bool reorder = pc->kind == SGM_PC_ILDU0 && pc->opt.ildu_reorder && A->fmt == SGM_FMT_CSR && A->nrow == A->ncol && A->nrow > 0;
    if (!reorder) {
        if (!pc->ro.empty()) { free_reorder(pc); for (auto &S : pc->ild) free_ildu(S); pc->ild.clear(); }
        if (pc->Ap) { sgm_mat_destroy(pc->Ap); pc->Ap = nullptr; }
        const int rc0 = pc_setup_ordered(pc, A);
        if (rc0 == SGM_OK) trace_setup(pc);
        return rc0;
    }
    // ILDU(0) of the colour-ordered matrix: the ordering once per pattern (ldu_solvers.f90:117-125 builds the pattern once),
    // a permuted copy of A per setup (the values may have changed), the regular device-side setup on that copy.  On a row
    // partition every part orders its own DIAGONAL block (greedy_color_ordering of A_kk's graph, permutations.f90:83-205; no
    // communication) and the copy's part k is P_k A_k [P_k^T (+) I]: rows and owned columns renumbered, halo columns and
    // the neighbours' request lists' meaning kept (the lists are mapped through P_k) -- block-Jacobi ILDU(0) of the ordered
    // blocks, SURVEY 8e.
    auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { (void)hipStreamSynchronize(g_rt.stream); return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const size_t P = A->parts.size();
    bool same = pc->ro.size() == P && pc->ro_serial == A->serial && pc->ro_pattern == A->pattern_version;
    for (size_t ip = 0; same && ip < P; ++ip) same = pc->ro[ip].n == A->parts[ip].n && pc->ro[ip].perm && pc->ro[ip].rp && pc->ro[ip].zp;
    if (!same) {
        free_reorder(pc);
        for (auto &S : pc->ild) free_ildu(S);
        pc->ild.clear();
        pc->ro.resize(P);
        struct Undo { sgm_pc pc; bool armed = true; ~Undo() { if (armed) free_reorder(pc); } } undo{pc};      // (a failure below leaves no half-made ordering behind)
        for (size_t ip = 0; ip < P; ++ip) {
            sgm_mat B = nullptr;
            SGM_TRY(diag_block_plain(A->parts[ip], &B));
            std::vector<int32_t> ptrs;
            const int rc = color_order_device(B, &pc->ro[ip].perm, ptrs);
            sgm_mat_destroy(B);
            if (rc != SGM_OK) return rc;
            pc->ro[ip].n = A->parts[ip].n;
            pc->ro[ip].colors = (int32_t)ptrs.size() - 1;
            SGM_TRY(dalloc(&pc->ro[ip].rp, (size_t)pc->ro[ip].n + 2));
            SGM_TRY(dalloc(&pc->ro[ip].zp, (size_t)pc->ro[ip].n + 2));
        }
        // Halo slots in the order of the permuted rows they attach to (index work, once per pattern): in the colour order a
        // grid part's halo columns would otherwise sit at a different offset from every row -- the 15-entry offset dictionary
        // overflows and the product falls from k_csr_sl (8.5 B per slot) to k_csr_sl32 (12 B).  Every receiver orders the
        // slots of each neighbour's segment and the senders permute their lists to match (ranks: one exchange of int32 lists).
        for (size_t ip = 0; ip < P; ++ip) {
            std::vector<std::pair<int32_t, int32_t>> seg;
            if (A->comm) {
                for (const HaloNbr &nb : A->parts[ip].nbrs)
                    if (nb.recv_count) seg.emplace_back(nb.recv_offset, nb.recv_count);
            } else {
                for (size_t is = 0; is < P; ++is)
                    for (const HaloNbr &nb : A->parts[is].nbrs)
                        if ((size_t)nb.peer == ip && nb.send_count) seg.emplace_back(nb.recv_offset, nb.send_count);
            }
            auto &R = pc->ro[ip];
            SGM_TRY(halo_attach_order(A->parts[ip], R.perm, seg, R.hmap_host));
            if (!R.hmap_host.empty()) {
                SGM_TRY(dalloc(&R.hmap, R.hmap_host.size()));
                SGM_HIP(hipMemcpyAsync(R.hmap, R.hmap_host.data(), R.hmap_host.size() * 4, hipMemcpyHostToDevice, g_rt.stream));
                SGM_HIP(hipStreamSynchronize(g_rt.stream));
            }
        }
        if (A->comm) SGM_TRY(exchange_halo_orders(A, pc->ro[0].hmap_host, pc->ro[0].send_order));
        else
            for (size_t is = 0; is < P; ++is) {
                auto &S = pc->ro[is];
                S.send_order.assign(A->parts[is].nbrs.size(), nullptr);
                for (size_t k = 0; k < A->parts[is].nbrs.size(); ++k) {
                    const HaloNbr &nb = A->parts[is].nbrs[k];
                    if (!nb.send_count) continue;
                    const std::vector<int32_t> &hm = pc->ro[(size_t)nb.peer].hmap_host;
                    std::vector<int32_t> rel((size_t)nb.send_count);
                    for (int32_t t = 0; t < nb.send_count; ++t) rel[(size_t)t] = hm[(size_t)nb.recv_offset + t] - nb.recv_offset;
                    SGM_TRY(dalloc(&S.send_order[k], (size_t)nb.send_count));
                    SGM_HIP(hipMemcpyAsync(S.send_order[k], rel.data(), (size_t)nb.send_count * 4, hipMemcpyHostToDevice, g_rt.stream));
                    SGM_HIP(hipStreamSynchronize(g_rt.stream));
                }
            }
        undo.armed = false;
        pc->ro_serial = A->serial;
        pc->ro_pattern = A->pattern_version;
        pc->reorder_ms[0] = ms_since(t0);
    }
    t0 = std::chrono::steady_clock::now();
    if (pc->Ap) { sgm_mat_destroy(pc->Ap); pc->Ap = nullptr; }
    sgm_mat Ap = new sgm_mat_s;
    Ap->fmt = SGM_FMT_CSR; Ap->nrow = A->nrow; Ap->ncol = A->ncol; Ap->nnz = A->nnz;
    Ap->comm = A->comm; Ap->row_starts = A->row_starts; Ap->col_starts = A->col_starts; Ap->halo_cols = A->halo_cols;
    Ap->parts.resize(P);
    int rc = SGM_OK;
    for (size_t ip = 0; rc == SGM_OK && ip < P; ++ip)
        rc = permuted_part(A->parts[ip], pc->ro[ip].perm, Ap->parts[ip], pc->ro[ip].hmap, &pc->ro[ip].send_order);
    if (A->comm && !Ap->halo_cols.empty() && pc->ro[0].hmap_host.size() == Ap->halo_cols.size())      // (global column of every halo slot, in the new order)
        for (size_t h = 0; h < pc->ro[0].hmap_host.size(); ++h) Ap->halo_cols[(size_t)pc->ro[0].hmap_host[h]] = A->halo_cols[h];
    pc->reorder_ms[1] = ms_since(t0);
    t0 = std::chrono::steady_clock::now();
    if (rc == SGM_OK) rc = pc_setup_ordered(pc, Ap);
    pc->reorder_ms[2] = ms_since(t0);
    if (rc == SGM_OK) { pc->Ap = Ap; pc->Ap_serial = A->serial; pc->Ap_version = A->version; }
    else sgm_mat_destroy(Ap);
    if (rc == SGM_OK) trace_setup(pc);
    return rc;
}

static int pc_setup_ordered(sgm_pc pc, sgm_mat A)
{
    if (A->nrow != A->ncol)      // jacobi_solvers.f90:46-50, ldu_solvers.f90:104-108
        return fail(SGM_ERR_DIMS, "Cannot make a %s solver for a non-square matrix",
                    pc->kind == SGM_PC_JACOBI ? "Jacobi" : "LDU");
    hipStream_t st = g_rt.stream;
    auto jacobi_rows = [&](const Part &p, int32_t fmt, int32_t row0, int32_t count, int32_t dcol, double *out) {
        const int grid = (count + kBlock - 1) / kBlock;
        if (!grid) return;
        if (fmt == SGM_FMT_CSR) {
            if (csr_need_arrays(p) != SGM_OK) return;
            hipLaunchKernelGGL(k_jacobi_setup_csr, dim3(grid), dim3(kBlock), 0, st, count, row0, dcol, p.rowptr, p.col, p.val, out);
            csr_release_arrays(p);
        } else
            hipLaunchKernelGGL(k_jacobi_setup_ell, dim3(grid), dim3(kBlock), 0, st, count, row0, dcol, p.n, p.max_d, p.ecol,
                               p.eval, out);
    };
    if (A->fmt == SGM_FMT_COMPOSITE) {
        // jacobi_setup only needs A%get_value(i,i) (jacobi_solvers.f90:59-61), which a composite answers
        // from the block that owns (i,i) (sparse_matrix_composites.f90:465-485).  ILDU on a composite has no
        // reference behaviour to match: its pattern pass walks the composite's get_edges cursor, whose block
        // advance skips block (2,1) and runs past the last column block (sparse_matrix_composites.f90:724-727).
        if (pc->kind != SGM_PC_JACOBI)
            return fail(SGM_ERR_UNSUPPORTED, "ILDU(0) needs a leaf CSR matrix, not a composite (the reference's own "
                                             "pattern pass is broken on composites)");
        if (pc->parts.size() != 1) {
            for (auto &pp : pc->parts) dfree(pp.idiag);
            pc->parts.assign(1, PartPC());
        }
        pc->n = A->nrow;
        pc->parts[0].n = A->nrow;
        if (!pc->parts[0].idiag) SGM_TRY(dalloc(&pc->parts[0].idiag, (size_t)A->nrow + 2));
        const int nrb = (int)A->blk_row_ptr.size() - 1, ncb = (int)A->blk_col_ptr.size() - 1;
        for (int it = 0; it < nrb; ++it)
            for (int jt = 0; jt < ncb; ++jt) {
                const int32_t lo = std::max(A->blk_row_ptr[it], A->blk_col_ptr[jt]);
                const int32_t hi = std::min(A->blk_row_ptr[it + 1], A->blk_col_ptr[jt + 1]);
                if (hi <= lo) continue;               // this block holds no diagonal entry
                sgm_mat C = A->blocks[(size_t)it * ncb + jt];
                double *out = pc->parts[0].idiag + lo;
                if (!C) {
                    hipLaunchKernelGGL(k_fill_inf, dim3((hi - lo + kBlock - 1) / kBlock), dim3(kBlock), 0, st, hi - lo, out);
                    continue;
                }
                jacobi_rows(C->parts[0], C->fmt, lo - A->blk_row_ptr[it], hi - lo, A->blk_row_ptr[it] - A->blk_col_ptr[jt], out);
            }
        SGM_HIP(hipGetLastError());
        return finish();
    }
    if (pc->kind == SGM_PC_JACOBI) {
        if (pc->parts.size() != A->parts.size()) {
            for (auto &pp : pc->parts) dfree(pp.idiag);
            pc->parts.assign(A->parts.size(), PartPC());
        }
        pc->n = A->nrow;
        for (size_t ip = 0; ip < A->parts.size(); ++ip) {
            const Part &p = A->parts[ip];
            if (!pc->parts[ip].idiag) SGM_TRY(dalloc(&pc->parts[ip].idiag, (size_t)p.n + 2));
            pc->parts[ip].n = p.n;
            jacobi_rows(p, A->fmt, 0, p.n, 0, pc->parts[ip].idiag);
        }
        SGM_HIP(hipGetLastError());
        return finish();
    }
    // ILDU(0); on a row partition: of every part's diagonal block (block-Jacobi ILDU -- exact
    // parity with the reference holds for one part, more parts change the iteration counts).
    // An ELLPACK operand: sparse_ldu_setup takes any sparse_matrix_interface (ldu_solvers.f90:95-130); the pattern pass reads A
    // through its get_edges cursor (:397-440) and the fill through get_entries (:306-321), and the ELLPACK cursor yields row
    // after row the first degrees(i) slots of node(:, i) / val(:, i) (ellpack_graphs.f90:310-369) -- never the padding.  That
    // edge stream is a CSR matrix's: the rows' real entries in slot order are laid out as one (real_entries_as_csr) for the
    // length of this setup, and everything below runs on it statement for statement.
    if (A->fmt != SGM_FMT_CSR && A->fmt != SGM_FMT_ELL)
        return fail(SGM_ERR_UNSUPPORTED, "ILDU(0) needs a CSR or ELLPACK matrix");
    if (!pc->abort_sticky) SGM_TRY(dalloc(&pc->abort_sticky, 1));
    SGM_HIP(hipMemsetAsync(pc->abort_sticky, 0, sizeof(int32_t), g_rt.stream));
    if (pc->ild.size() != A->parts.size()) {
        for (auto &S : pc->ild) free_ildu(S);
        pc->ild.assign(A->parts.size(), IlduState());
        for (auto &S0 : pc->ild) S0.opt = pc->opt;
    }
    pc->n = A->nrow;
    for (size_t ip = 0; ip < A->parts.size(); ++ip) {
        IlduState *S = &pc->ild[ip];
        Part ellview;
        struct EllView { Part &v; ~EllView() { dfree(v.rowptr); dfree(v.col); dfree(v.val); v.rowptr = nullptr; v.col = nullptr; v.val = nullptr; } } ellguard{ellview};
        const bool trim = A->fmt == SGM_FMT_ELL || A->parts[ip].edeg;     // (edeg on a CSR part: ELLPACK rows over ranks, sgm_ell_create_dist)
        if (trim) {
            const Part &src = A->parts[ip];
            if (A->fmt == SGM_FMT_CSR) SGM_TRY(csr_need_arrays(src));
            const int rc = real_entries_as_csr(src, A->fmt == SGM_FMT_ELL, ellview);
            if (A->fmt == SGM_FMT_CSR) csr_release_arrays(src);
            SGM_TRY(rc);
        }
        const Part &P = trim ? ellview : A->parts[ip];
        static const bool timing = getenv("SGM_PC_TIMING") != nullptr;       // phase times of the setup on stderr (tuning aid)
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto t_prev = now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            (void)hipStreamSynchronize(g_rt.stream);
            const auto t = now();
            fprintf(stderr, "[sigma_hip] ildu setup: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
            t_prev = t;
        };
        const int32_t n = P.n;
        const bool fresh = S->n != n || !S->dLptr;             // ldu_solvers.f90:117-125: pattern once
        SGM_TRY(csr_need_arrays(P));          // (a part that kept only its sliced form rebuilds col / val for the setup)
        struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{P};
        const int32_t own = P.n_halo == 0 ? INT32_MAX : P.ncol_own;
        bool few = false;                     // L has at most kRowLevels levels (found on the device): no pipeline applies, no host pattern needed
        if (fresh) {
            free_ildu(*S);
            S->n = n;
            SGM_TRY(ildu_pattern(S, P, own));
            lap("pattern (device)");
            // L's dependency levels: the order the rows are factorised in (and what its sweeps use later) -- on the device
            // when they are few, else from the pattern's host copy
            SGM_TRY(tri_levels_device(S->L, n, S->dLptr, S->dLnode, &few));
            if (!S->L.have_levels && S->opt.ildu_strips) {
                // many levels: a grid-like pair (what the strip pipeline serves)?  Then the anti-diagonals are the order
                SGM_TRY(grid_width_device(n, S->dLptr, S->dLnode, true, &S->dev_wl));
                if (S->dev_wl >= 64) SGM_TRY(grid_width_device(n, S->dUptr, S->dUnode, false, &S->dev_wu));
                if (S->dev_wl >= 64 && S->dev_wl == S->dev_wu && (n + S->dev_wl - 1) / S->dev_wl >= 64)
                    SGM_TRY(grid_factor_order(S, n, S->dev_wl, 0));
                else {
                    S->dev_wl = S->dev_wu = 0;
                    // ... or a 3-D grid's (what the slab pipeline serves: the same bounds as slab3_build)?
                    int32_t wl3, hl3, wu3 = 0, hu3 = 0;
                    SGM_TRY(slab_dims_device(n, S->dLptr, S->dLnode, true, &wl3, &hl3));
                    if (wl3) SGM_TRY(slab_dims_device(n, S->dUptr, S->dUnode, false, &wu3, &hu3));
                    const int64_t wh3 = (int64_t)wl3 * hl3;
                    if (wl3 && wl3 == wu3 && hl3 == hu3 && wl3 >= 32 && wl3 <= 256 && hl3 >= 8 && (n + wh3 - 1) / wh3 >= 8) {
                        SGM_TRY(grid_factor_order(S, n, wl3, hl3));
                        S->dev_slab = true;
                    }
                }
                lap("grid detection, anti-diagonal order");
            }
            if (!S->L.have_levels && !S->forder) {
                SGM_TRY(ensure_host_pattern(S));
                lap("host copy of the pattern");
                SGM_TRY(tri_levels_dev(S->L, n, S->hLptr, S->hLnode, S->dLptr, S->dLnode, nullptr, true));
            }
            SGM_TRY(dalloc(&S->dLval, (size_t)std::max(S->nnzL, 1)));
            SGM_TRY(dalloc(&S->dUval, (size_t)std::max(S->nnzU, 1)));
            SGM_TRY(dalloc(&S->D, (size_t)std::max(n, 1)));
            lap("levels of L");
        }
        S->n = n;
        S->host_vals = false;
        bool host_factor = false;
        if (n) {
            // sparse_static_pattern_ldu_factorization (ldu_solvers.f90:275-387) on the device: the fill, then one launch per
            // dependency level of L
            hipLaunchKernelGGL(k_ildu_init, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, own, (const int32_t *)P.rowptr,
                               (const int32_t *)P.col, (const double *)P.val, (const int32_t *)S->dLptr, (const int32_t *)S->dLnode,
                               S->dLval, (const int32_t *)S->dUptr, (const int32_t *)S->dUnode, S->dUval, S->D);
            // (rows in the order of L's dependency levels, or -- grid-like factors -- of the grid's anti-diagonals)
            const std::vector<int32_t> &flp = S->forder ? S->flevel_ptr : S->L.level_ptr;
            const int32_t *ford = S->forder ? S->forder : S->L.order;
            // A factor that is (nearly) a chain -- thousands of levels of a few rows each: bands with their first off-diagonal,
            // 1-D problems -- would be one launch per row (n = 4e5: 1.9 s of launches; tools/probes/chain_setup.py).  Its rows are
            // factored on the HOST instead, one after the other in natural order (row i reads rows k < i only: the reference's own
            // loop order), by the very statements of k_ildu_factor_level, and the values go back: two copies and ~0.1 us per row.
            const size_t nlev = flp.size() - 1;
            // (short rows only -- the host loop is O(len^3) per row and single-threaded: a chain of WIDE rows stays on the device)
            host_factor = nlev > 4096 && (int64_t)nlev * 8 > (int64_t)n && S->maxL + S->maxU <= 16;
            if (host_factor) {
                std::vector<int32_t> hLp((size_t)n + 1), hUp((size_t)n + 1), hLn((size_t)std::max(S->nnzL, 1)), hUn((size_t)std::max(S->nnzU, 1));
                std::vector<double> hLv((size_t)std::max(S->nnzL, 1)), hUv((size_t)std::max(S->nnzU, 1)), hD((size_t)n);
                SGM_HIP(hipMemcpyAsync(hLp.data(), S->dLptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
                SGM_HIP(hipMemcpyAsync(hUp.data(), S->dUptr, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, st));
                if (S->nnzL) SGM_HIP(hipMemcpyAsync(hLn.data(), S->dLnode, (size_t)S->nnzL * 4, hipMemcpyDeviceToHost, st));
                if (S->nnzU) SGM_HIP(hipMemcpyAsync(hUn.data(), S->dUnode, (size_t)S->nnzU * 4, hipMemcpyDeviceToHost, st));
                if (S->nnzL) SGM_HIP(hipMemcpyAsync(hLv.data(), S->dLval, (size_t)S->nnzL * 8, hipMemcpyDeviceToHost, st));
                if (S->nnzU) SGM_HIP(hipMemcpyAsync(hUv.data(), S->dUval, (size_t)S->nnzU * 8, hipMemcpyDeviceToHost, st));
                SGM_HIP(hipMemcpyAsync(hD.data(), S->D, (size_t)n * 8, hipMemcpyDeviceToHost, st));
                SGM_HIP(hipStreamSynchronize(st));
                for (int32_t i = 0; i < n; ++i) ildu_factor_row(i, hLp.data(), hLn.data(), hLv.data(), hUp.data(), hUn.data(), hUv.data(), hD.data());
                if (S->nnzL) SGM_HIP(hipMemcpyAsync(S->dLval, hLv.data(), (size_t)S->nnzL * 8, hipMemcpyHostToDevice, st));
                if (S->nnzU) SGM_HIP(hipMemcpyAsync(S->dUval, hUv.data(), (size_t)S->nnzU * 8, hipMemcpyHostToDevice, st));
                SGM_HIP(hipMemcpyAsync(S->D, hD.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
                SGM_HIP(hipStreamSynchronize(st));          // (the host vectors go out of scope)
            }
            for (size_t l = 0; !host_factor && l + 1 < flp.size(); ++l) {
                const int32_t b = flp[l], e = flp[l + 1];
                if (S->maxL <= 4 && S->maxU <= 4) {
                    hipLaunchKernelGGL((k_ildu_factor_level_short<4, 4>), dim3((e - b + 63) / 64), dim3(64), 0, st,
                                       ford, b, e, (const int32_t *)S->dLptr, (const int32_t *)S->dLnode, S->dLval,
                                       (const int32_t *)S->dUptr, (const int32_t *)S->dUnode, S->dUval, S->D);
                    continue;
                }
                hipLaunchKernelGGL(k_ildu_factor_level, dim3((e - b + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                                   ford, b, e, (const int32_t *)S->dLptr, (const int32_t *)S->dLnode, S->dLval,
                                   (const int32_t *)S->dUptr, (const int32_t *)S->dUnode, S->dUval, S->D);
            }
            SGM_HIP(hipGetLastError());
        }
        lap(host_factor ? "factorisation (host: a chain)" : "factorisation (device)");
        // (the level-scheduled structures: ensure_levels, below or on first need)
        S->levels_ready = false;
        S->walk_ready = false;
        if (fresh) S->levels_pattern = S->walk_pattern = false;
        if (fresh) {        // grid-like factors get the strip layout
            free_grid(S->gL); free_grid(S->gU);
            dfree(S->gxL); dfree(S->gxU); dfree(S->gDp); dfree(S->gmapLU);
            S->gxL = S->gxU = S->gDp = nullptr; S->gmapLU = nullptr;
            S->grid_ok = false;
            int32_t wl = S->dev_wl, wu = S->dev_wu;          // (found on the device already when the pair is grid-like)
            if (!few && !wl && !S->dev_slab) {
                SGM_TRY(ensure_host_pattern(S));
                wl = grid_width(n, S->hLptr, S->hLnode, true);
                wu = grid_width(n, S->hUptr, S->hUnode, false);
            }
            if (S->opt.ildu_strips && wl >= 64 && wl == wu && (n + wl - 1) / wl >= 64) {
                SGM_TRY(build_grid(S->gL, n, wl, S->dLptr, S->dLnode, true));
                SGM_TRY(build_grid(S->gU, n, wl, S->dUptr, S->dUnode, false));
                if (S->gL.on && S->gU.on) {
                    SGM_TRY(dalloc(&S->gxL, (size_t)S->gL.NP));
                    SGM_TRY(dalloc(&S->gxU, (size_t)S->gU.NP));
                    SGM_TRY(dalloc(&S->gDp, (size_t)S->gU.NP));
                    SGM_TRY(dalloc(&S->gmapLU, (size_t)S->gU.NP));
                    SGM_HIP(hipMemsetAsync(S->gmapLU, 0xff, (size_t)S->gU.NP * 4, st));
                    hipLaunchKernelGGL(k_grid_map, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)S->gU.pos,
                                       (const int32_t *)S->gL.pos, S->gmapLU);
                    SGM_HIP(hipStreamSynchronize(st));
                    dfree(S->gL.pos); dfree(S->gU.pos);
                    S->gL.pos = S->gU.pos = nullptr;
                } else { free_grid(S->gL); free_grid(S->gU); }
            }
            slab3_free(S->slab);
            S->slab = nullptr;
            S->slab_ok = false;
            if (S->opt.ildu_strips && !few && !(S->gL.on && S->gU.on) && !S->dev_slab) SGM_TRY(ensure_host_pattern(S));
            if (S->opt.ildu_strips && !few && !(S->gL.on && S->gU.on))
                SGM_TRY(slab3_build(&S->slab, n, S->hLptr, S->hLnode, S->hUptr, S->hUnode, S->dLptr, S->dLnode, S->dUptr, S->dUnode));
        }
        lap("strip / slab index work");
        if (S->slab) SGM_TRY(slab3_refresh(S->slab, S->dLval, S->dUval, S->D));
        const bool have_grid = S->gL.on && S->gU.on;
        if (have_grid) {
            SGM_TRY(refresh_grid_values(S->gL, S->dLval));
            SGM_TRY(refresh_grid_values(S->gU, S->dUval));
            hipLaunchKernelGGL(k_pos_diag, dim3(vec_grid(S->gU.NP)), dim3(kBlock), 0, st, S->gU.NP, (const int32_t *)S->gU.row,
                               (const double *)S->D, S->gDp);
        }
        if (!have_grid && !S->slab) {         // no pipelined path for this pattern: the row-space sweeps or the level walkers serve it
            SGM_TRY(ensure_levels(S));
            lap("levels, row-space copy");
            if (!rows_serve(S)) {
                SGM_TRY(ensure_walkers(S));
                lap("level walkers' structures");
            }
        }
        lap("strip / slab records");
        if ((have_grid || S->slab) && fresh) {
            // the pipelines hand data between workgroups inside one launch: before one is trusted with this pattern it
            // must reproduce the row-by-row sweeps of ldu_solve (ldu_solvers.f90:160-176, :208-265) bit for bit on a test
            // vector, and raise no abort.  Checked on the device, every row against the recurrence (k_sweep_check).
            double *dr = nullptr, *dz = nullptr, *dy = nullptr;
            int32_t *dbad = nullptr;
            struct Tmp { double *&a, *&b, *&c; int32_t *&d; ~Tmp() { dfree(a); dfree(b); dfree(c); dfree(d); } } tmp{dr, dz, dy, dbad};
            SGM_TRY(dalloc(&dr, (size_t)n));
            SGM_TRY(dalloc(&dz, (size_t)n));
            SGM_TRY(dalloc(&dy, (size_t)n));
            SGM_TRY(dalloc(&dbad, 1));
            hipStream_t st2 = g_rt.stream;
            hipLaunchKernelGGL(k_check_vector, dim3(vec_grid(n)), dim3(kBlock), 0, st2, (int64_t)n, dr);
            (void)hipMemsetAsync(dz, 0, (size_t)n * 8, st2);
            (void)hipMemsetAsync(dy, 0, (size_t)n * 8, st2);
            (void)hipMemsetAsync(dbad, 0, 4, st2);
            if (have_grid) {
                apply_grid(S, dr, dz, nullptr, kStripSpinLimit, nullptr);
                hipLaunchKernelGGL(k_grid_scatter, dim3(vec_grid(S->gL.NP)), dim3(kBlock), 0, st2, S->gL.NP, dy, (const double *)S->gxL,
                                   (const int32_t *)S->gL.row, (const int *)nullptr);
            } else {
                slab3_apply(S->slab, dr, dz, nullptr, kStripSpinLimit, nullptr);
                slab3_lower_result(S->slab, dy);
            }
            const int cg = (n + kBlock - 1) / kBlock;
            hipLaunchKernelGGL(k_sweep_check, dim3(cg), dim3(kBlock), 0, st2, n, (const int32_t *)S->dLptr, (const int32_t *)S->dLnode,
                               (const double *)S->dLval, (const double *)dr, (const double *)nullptr, (const double *)dy, dbad);
            hipLaunchKernelGGL(k_sweep_check, dim3(cg), dim3(kBlock), 0, st2, n, (const int32_t *)S->dUptr, (const int32_t *)S->dUnode,
                               (const double *)S->dUval, (const double *)dy, (const double *)S->D, (const double *)dz, dbad);
            int32_t abL = 0, abU = 0, bad = 0;
            (void)hipMemcpyAsync(&bad, dbad, 4, hipMemcpyDeviceToHost, st2);
            if (have_grid) {
                (void)hipMemcpyAsync(&abL, S->gL.progress + S->gL.NI, 4, hipMemcpyDeviceToHost, st2);
                (void)hipMemcpyAsync(&abU, S->gU.progress + S->gU.NI, 4, hipMemcpyDeviceToHost, st2);
            }
            const hipError_t e = hipStreamSynchronize(st2);
            if (!have_grid && e == hipSuccess) (void)slab3_aborted(S->slab, &abL, &abU);
            const bool same = e == hipSuccess && !abL && !abU && bad == 0;
            if (have_grid) S->grid_ok = same; else S->slab_ok = same;
            if (!same)
                fprintf(stderr, "[sigma_hip] ILDU %s pipeline disabled for this matrix (self-check: abort %d/%d, %d rows differ)\n",
                        have_grid ? "strip" : "slab", abL, abU, bad);
            lap("self-check");
            if (!same) {
                SGM_TRY(ensure_levels(S));
                if (!rows_serve(S)) SGM_TRY(ensure_walkers(S));
                lap("levels, walkers' structures");
            }
        }
    }
    return SGM_OK;
}

/* sgm_pc_create: the factory alone -- jacobi() / ldu() (jacobi_solvers.f90:23-31, ldu_solvers.f90:73-86) return an object
 * that has seen no matrix yet; options can be set on it before the first sgm_pc_setup builds its sweeps. */
int sgm_pc_create(sgm_pc *out, int32_t kind)
{
    if (!out || (kind != SGM_PC_JACOBI && kind != SGM_PC_ILDU0)) return fail(SGM_ERR_BAD_ARG, "sgm_pc_create: kind is SGM_PC_JACOBI or SGM_PC_ILDU0");
    sgm_pc pc = new sgm_pc_s;
    pc->kind = kind;
    *out = pc;
    return SGM_OK;
}

int sgm_jacobi_create(sgm_pc *out, sgm_mat A)
{
    if (!out) return fail(SGM_ERR_BAD_ARG, "sgm_jacobi_create: null out pointer");
    sgm_pc pc = new sgm_pc_s;
    pc->kind = SGM_PC_JACOBI;
    int rc = sgm_pc_setup(pc, A);
    if (rc != SGM_OK) { sgm_pc_destroy(pc); return rc; }
    *out = pc;
    return SGM_OK;
}

int sgm_ildu0_create(sgm_pc *out, sgm_mat A)
{
    if (!out) return fail(SGM_ERR_BAD_ARG, "sgm_ildu0_create: null out pointer");
    sgm_pc pc = new sgm_pc_s;
    pc->kind = SGM_PC_ILDU0;
    int rc = sgm_pc_setup(pc, A);
    if (rc != SGM_OK) { sgm_pc_destroy(pc); return rc; }
    *out = pc;
    return SGM_OK;
}

/* sgm_pc_set_option: this preconditioner's own copy of "ildu_strips", "ildu_rows", "pipeline_spin_limit" (sgm_set_option
 * only changes what preconditioners created LATER start with).  Which sweeps exist is decided at setup: switching a path
 * off acts from the next apply on, switching one on that was off at setup takes effect at the next sgm_pc_setup. */
int sgm_pc_set_option(sgm_pc pc, const char *name, int value)
{
    if (!pc || !name) return fail(SGM_ERR_BAD_ARG, "sgm_pc_set_option: null argument");
    int v = 0;
    SGM_TRY(normalise_option(name, value, &v));
    int *f = pc_option_field(pc->opt, name);
    if (!f) return fail(SGM_ERR_BAD_ARG, "sgm_pc_set_option: '%s' is not a preconditioner option", name);
    *f = v;
    for (auto &S : pc->ild) S.opt = pc->opt;
    return SGM_OK;
}

int sgm_pc_apply(sgm_pc pc, const double *r, double *z, int where)
{
    SGM_TRY(require_init());
    if (!pc || !r || !z) return fail(SGM_ERR_BAD_ARG, "sgm_pc_apply: null argument");
    if (pc->kind == SGM_PC_JACOBI && pc->parts.size() != 1)
        return fail(SGM_ERR_UNSUPPORTED, "sgm_pc_apply: stand-alone apply needs a single-part matrix");
    // the vectors hold THIS process's rows: all of them on one GPU or an in-process partition, this rank's block when the
    // matrix is distributed over ranks (pc->n is the global count there)
    int64_t nloc = 0;
    if (pc->kind == SGM_PC_ILDU0) for (const auto &S : pc->ild) nloc += S.n;
    else nloc = pc->parts.empty() ? 0 : pc->parts[0].n;
    Staged sr, sz;
    SGM_TRY(stage_in(sr, r, nloc, where, true));
    SGM_TRY(stage_in(sz, z, nloc, where, false));
    // a throw-away matrix view with the right part count for pc_apply_parts (block-Jacobi ILDU on an in-process partition:
    // the caller's vectors are global, part k's slice starts where the rows of the parts before it end)
    const size_t NP = pc->kind == SGM_PC_ILDU0 ? std::max<size_t>(pc->ild.size(), 1) : 1;
    sgm_mat_s view;
    view.parts.resize(NP);
    if (NP == 1) view.parts[0].n = (int32_t)nloc;
    else
        for (size_t ip = 0; ip < NP; ++ip) view.parts[ip].n = pc->ild[ip].n;
    // in-place apply (r == z on the device) through a pipelined sweep: a sweep that gives up has scattered its "not yet
    // written" patterns over z = r by the time anyone notices, so the redo below needs a right-hand side of its own
    Staged rkeep;
    if (sr.dev == sz.dev && pc_abort_word(pc)) {
        SGM_TRY(dalloc(&rkeep.dev, (size_t)nloc));
        rkeep.owned = true;
        SGM_HIP(hipMemcpyAsync(rkeep.dev, sr.dev, (size_t)nloc * sizeof(double), hipMemcpyDeviceToDevice, g_rt.stream));
    }
    std::vector<const double *> rsv(NP);
    std::vector<double *> zsv(NP);
    { int64_t off = 0; for (size_t ip = 0; ip < NP; ++ip) { rsv[ip] = (rkeep.dev ? rkeep.dev : sr.dev) + off; zsv[ip] = sz.dev + off; off += view.parts[ip].n; } }
    const double *const *rs = rsv.data();
    double *const *zs = zsv.data();
    SGM_TRY(pc_apply_parts(pc, &view, rs, zs, nullptr));
    if (int32_t *ab = pc_abort_word(pc)) {
        // a pipelined sweep may give up (bounded waits): look before the result leaves -- one 4-byte copy and a
        // synchronisation against an apply of a millisecond, also in async mode -- and redo it with the level walkers
        int32_t aborted = 0;
        SGM_HIP(hipMemcpyAsync(&aborted, ab, sizeof(int32_t), hipMemcpyDeviceToHost, g_rt.stream));
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
        if (aborted) {
            SGM_TRY(pc_retire_pipelines(pc));
            SGM_TRY(pc_apply_parts(pc, &view, rs, zs, nullptr));
        }
    }
    SGM_TRY(stage_out(sz, z, nloc, where));
    return finish();
}

int sgm_pc_get(sgm_pc pc, const char *name, void *out, size_t bytes, size_t *needed)
{
    if (!pc || !name) return fail(SGM_ERR_BAD_ARG, "sgm_pc_get: null argument");
    const void *src = nullptr;
    size_t sz = 0;
    std::string nm(name);
    static const char kEmpty = 0;
    if (pc->kind == SGM_PC_JACOBI && nm == "idiag") {
        if (pc->parts.size() != 1) return fail(SGM_ERR_UNSUPPORTED, "sgm_pc_get(idiag): single-part only");
        pc->hidiag.resize((size_t)pc->n);
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
        if (pc->n) SGM_HIP(hipMemcpy(pc->hidiag.data(), pc->parts[0].idiag, (size_t)pc->n * 8, hipMemcpyDeviceToHost));
        src = pc->hidiag.data(); sz = pc->hidiag.size() * 8;
    } else if (pc->kind == SGM_PC_ILDU0) {
        if (pc->ild.size() != 1) return fail(SGM_ERR_UNSUPPORTED, "sgm_pc_get: single-part ILDU only");
        IlduState *S = &pc->ild[0];
        if (nm == "Lval" || nm == "Uval" || nm == "D") SGM_TRY(ensure_host_values(S));
        if (nm == "Lptr" || nm == "Lnode" || nm == "Uptr" || nm == "Unode") SGM_TRY(ensure_host_pattern(S));
        if (nm == "Lptr") { src = S->hLptr.data(); sz = S->hLptr.size() * 4; }
        else if (nm == "Lnode") { src = S->hLnode.data(); sz = S->hLnode.size() * 4; }
        else if (nm == "Lval") { src = S->hLval.data(); sz = S->hLval.size() * 8; }
        else if (nm == "Uptr") { src = S->hUptr.data(); sz = S->hUptr.size() * 4; }
        else if (nm == "Unode") { src = S->hUnode.data(); sz = S->hUnode.size() * 4; }
        else if (nm == "Uval") { src = S->hUval.data(); sz = S->hUval.size() * 8; }
        else if (nm == "D") { src = S->hD.data(); sz = S->hD.size() * 8; }
        else if (nm == "strips") {          // strip pipeline in use: {strips per sweep, steps per strip, order variant of L, of U}; zeros = off
            static int32_t sv[4];
            const bool on = S->grid_ok && S->opt.ildu_strips;
            sv[0] = on ? S->gL.NI : 0; sv[1] = on ? S->gL.S : 0; sv[2] = on ? S->gL.order : 0; sv[3] = on ? S->gU.order : 0;
            src = sv; sz = sizeof sv;
        }
        else if (nm == "strip_clocks" && S->grid_ok) {     // per strip of the L sweep: chain start, end (100 MHz ticks)
            static std::vector<long long> ck;
            ck.assign((size_t)2 * S->gL.NI, 0);
            SGM_HIP(hipStreamSynchronize(g_rt.stream));
            for (int32_t i = 0; i < S->gL.NI; ++i)
                SGM_HIP(hipMemcpy(&ck[2 * i], S->gL.edge + (int64_t)i * (S->gL.S + kEdgePad) + S->gL.S + 64, 16, hipMemcpyDeviceToHost));
            src = ck.data(); sz = ck.size() * 8;
        }
        else if (nm == "slabs") {           // slab pipeline in use: {strips, line groups, lines per group, steps, order of L, of U}; zeros = off
            static int32_t sv[6];
            memset(sv, 0, sizeof sv);
            if (S->slab_ok && S->opt.ildu_strips) slab3_info(S->slab, sv);
            src = sv; sz = sizeof sv;
        }
        else if (nm == "slab_clocks" && S->slab_ok) {
            static std::vector<long long> ck;
            SGM_HIP(hipStreamSynchronize(g_rt.stream));
            SGM_TRY(slab3_clocks(S->slab, ck));
            src = ck.data(); sz = ck.size() * 8;
        }
        else if (nm == "perm") {                 // option ildu_reorder: p (1-based; row i of A = row p(i) of the factorised matrix); empty = natural order
            static std::vector<int32_t> hp;
            hp.assign((size_t)(!pc->ro.empty() ? pc->n : 0), 0);
            if (!pc->ro.empty() && pc->n) { SGM_HIP(hipStreamSynchronize(g_rt.stream)); SGM_HIP(hipMemcpy(hp.data(), pc->ro[0].perm, hp.size() * 4, hipMemcpyDeviceToHost)); }
            src = hp.data(); sz = hp.size() * 4;
            if (!sz) src = &kEmpty;
        }
        else if (nm == "reorder_ms") {           // last setup with ildu_reorder: {ordering, permuted copy, setup on the copy, colours}
            static double rm[4];
            rm[0] = pc->reorder_ms[0]; rm[1] = pc->reorder_ms[1]; rm[2] = pc->reorder_ms[2]; rm[3] = !pc->ro.empty() ? pc->ro[0].colors : 0;
            src = rm; sz = sizeof rm;
        }
        else if (nm == "pipeline_retired") {     // how often a pipelined sweep gave up and the pipelines were retired (0 = never)
            static int32_t rv[1];
            rv[0] = pc->retired;
            src = rv; sz = sizeof rv;
        }
        else if (nm == "row_levels") {           // row-space level path in use: {1, launches of the L sweep, of the U sweep}; zeros = off
            static int32_t rl[3];
            const bool on = !(S->opt.ildu_strips && (S->grid_ok || S->slab_ok)) && rows_serve(S);
            const bool fu = S->opt.ildu_rows == 1;
            rl[0] = on;
            rl[1] = on ? (int32_t)S->L.row_levels.size() - (fu && S->rows_n0 > 0 ? 1 : 0) : 0;
            rl[2] = on ? (int32_t)S->U.row_levels.size() - (fu && S->rows_fin ? 1 : 0) : 0;
            src = rl; sz = sizeof rl;
        }
        else if (nm == "levels") {
            SGM_TRY(ensure_levels(&pc->ild[0]));
            static int32_t lv[2];
            lv[0] = (int32_t)S->L.level_ptr.size() - 1;
            lv[1] = (int32_t)S->U.level_ptr.size() - 1;
            src = lv; sz = sizeof lv;
        }
    }
    const bool known = nm == "strips" || nm == "strip_clocks" || nm == "slabs" || nm == "slab_clocks" || nm == "pipeline_retired" || nm == "idiag" || nm == "Lptr" || nm == "Lnode" || nm == "Lval" || nm == "Uptr" ||
                       nm == "Unode" || nm == "Uval" || nm == "D" || nm == "levels" || nm == "row_levels" || nm == "perm" || nm == "reorder_ms";
    if (!known || (!src && sz)) return fail(SGM_ERR_BAD_ARG, "sgm_pc_get: unknown array '%s'", name);
    if (!src) src = &kEmpty;
    if (needed) *needed = sz;
    if (out && sz) {
        if (bytes < sz) return fail(SGM_ERR_BAD_ARG, "sgm_pc_get: buffer too small (%zu < %zu)", bytes, sz);
        memcpy(out, src, sz);
    }
    return SGM_OK;
}

/* sgm_pc_info: which sweeps serve part `part` of this preconditioner and what an apply costs -- so that a caller who builds
 * `ldu()` the way the reference's tests do (solver_test_incomplete_cholesky.f90:137-141) can SEE that the factors of a
 * naturally ordered grid are a dependency chain before paying for it.
 *   out[0], out[1]  dependency levels of L and of U (the row recurrences of ldu_solvers.f90:227-236, :254-263 can start a level
 *                   only when the one before it is done); 1 for a diagonal preconditioner
 *   out[2]          path: 0 diagonal scaling (Jacobi), 1 row-space sweeps (one launch per level, bandwidth-bound), 2 strip
 *                   pipeline (2-D grid factors), 3 slab pipeline (3-D grid factors), 4 level walkers
 *   out[3]          colours of the ordering the factors belong to (option ildu_reorder); 0 = the matrix's own order
 *   est_us          estimated microseconds per apply on this GPU (from the path's measured constants)
 *   path_name       the same in words */
int sgm_pc_info(sgm_pc pc, int32_t part, int32_t *out4, double *est_us, char *path_name, int len)
{
    if (!pc) return fail(SGM_ERR_BAD_ARG, "sgm_pc_info: null preconditioner");
    int32_t o[4] = {1, 1, 0, 0};
    double us = 0.0;
    char nm[160] = "";
    if (pc->kind == SGM_PC_JACOBI) {
        if (part < 0 || (size_t)part >= std::max<size_t>(pc->parts.size(), 1)) return fail(SGM_ERR_BAD_ARG, "sgm_pc_info: part %d", part);
        us = 24.0 * pc->n / 5.5e6;
        snprintf(nm, sizeof nm, "diagonal scaling, 1 level");
    } else {
        if (part < 0 || (size_t)part >= pc->ild.size()) return fail(SGM_ERR_BAD_ARG, "sgm_pc_info: part %d of %zu (set the preconditioner up first)", part, pc->ild.size());
        IlduState *S = &pc->ild[(size_t)part];
        o[3] = (size_t)part < pc->ro.size() ? pc->ro[(size_t)part].colors : 0;
        if (S->grid_ok && S->opt.ildu_strips) {
            // a w x nj grid in natural order: rows (i, j) with i + j equal form a level
            o[0] = o[1] = S->gL.w + S->gL.nj - 1;
            o[2] = 2;
            us = 2.0 * (0.075 * S->gL.S + 9.9 * std::max(0, S->gL.NI - 1)) + 3.0 * 16.0 * S->gL.NP / 5.5e6;
            snprintf(nm, sizeof nm, "strip pipeline, %d levels", o[0]);
        } else if (S->slab_ok && S->opt.ildu_strips) {
            int32_t w = 0, h = 0, sv[6] = {0, 0, 0, 0, 0, 0};
            slab3_dims(S->slab, &w, &h);
            slab3_info(S->slab, sv);
            const int32_t nk = (int32_t)((S->n + (int64_t)w * h - 1) / ((int64_t)w * h));
            o[0] = o[1] = w + h + nk - 2;
            o[2] = 3;
            us = 2.0 * (0.135 * sv[3] + 6.5 * std::max(0, sv[1] - 1)) + 3.0 * 16.0 * S->n / 5.5e6;
            snprintf(nm, sizeof nm, "slab pipeline, %d levels", o[0]);
        } else {
            SGM_TRY(ensure_levels(S));
            o[0] = (int32_t)S->L.level_ptr.size() - 1;
            o[1] = (int32_t)S->U.level_ptr.size() - 1;
            if (rows_serve(S)) {
                o[2] = 1;
                // one launch per level and sweep (+ the two permutations of a stand-alone apply in a colour order), about 4.5 us each
                // when the levels are small: a 2-level apply of 1e5 rows is 32 us of launches around 2 us of traffic
                us = (12.0 * ((double)S->nnzL + S->nnzU) + 56.0 * S->n) / 5.5e6 + 4.5 * ((double)o[0] + o[1] + (o[3] ? 2 : 0));
                snprintf(nm, sizeof nm, "row space, %d levels", std::max(o[0], o[1]));
            } else {
                o[2] = 4;
                // (0.3 us per level and sweep, 0.16 where the levels are a wave wide at most -- chains --, measured on factors of
                //  1e3 ... 4e5 levels: tools/pc_survey.py)
                const double per_level = (int64_t)std::max(o[0], 1) * 64 >= (int64_t)S->n ? 0.16 : 0.30;
                us = per_level * ((double)o[0] + o[1]) + (12.0 * ((double)S->nnzL + S->nnzU) + 56.0 * S->n) / 5.5e6;
                snprintf(nm, sizeof nm, "level walkers, %d levels", std::max(o[0], o[1]));
            }
        }
    }
    // a colour-ordered part: the kernel the PRODUCT on the permuted copy runs with (the solvers iterate on that copy)
    if (pc->kind == SGM_PC_ILDU0 && pc->Ap && (size_t)part < pc->Ap->parts.size() && (size_t)part < pc->ro.size() && pc->ro[(size_t)part].colors) {
        char kn[64];
        part_kernel_name(pc->Ap->parts[(size_t)part], pc->Ap->fmt, kn, sizeof kn);
        const size_t used = strlen(nm);
        snprintf(nm + used, sizeof nm - used, "; product of the ordered part: %s", kn);
    }
    if (out4) memcpy(out4, o, sizeof o);
    if (est_us) *est_us = us;
    if (path_name && len > 0) snprintf(path_name, (size_t)len, "%s", nm);
    return SGM_OK;
}

int sgm_pc_destroy(sgm_pc pc)
{
    if (!pc) return SGM_OK;
    for (auto &pp : pc->parts) dfree(pp.idiag);
    for (auto &S : pc->ild) free_ildu(S);
    dfree(pc->abort_sticky);
    free_reorder(pc);
    if (pc->Ap) sgm_mat_destroy(pc->Ap);
    delete pc;
    return SGM_OK;
}

}  // extern "C"
