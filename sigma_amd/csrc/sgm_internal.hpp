// Internal declarations of libsigma_hip.so (gfx950 / CDNA4 only).
// Public surface: include/sigma_hip.h.  Nothing here includes or links oracle/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <utility>
#include <vector>

#include "../../include/sigma_hip.h"

namespace sgm {

// ------------------------------------------------------------------ errors
extern std::string g_err;
int fail(int code, const char *fmt, ...);

#define SGM_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess)                                                     \
            return sgm::fail(SGM_ERR_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__,   \
                             #call, hipGetErrorString(e__));                       \
    } while (0)
#define SGM_TRY(call)                        \
    do {                                     \
        int rc__ = (call);                   \
        if (rc__ != SGM_OK) return rc__;     \
    } while (0)

// ------------------------------------------------------------------ runtime
// Options are PER HANDLE: a matrix / solver / preconditioner copies the process-wide defaults (sgm_set_option) when it
// is created and keeps its own copy from then on (sgm_mat_set_option / sgm_solver_set_option / sgm_pc_set_option); two
// handles of one process may run different kernels, and changing a default never touches an existing handle.
constexpr int kEllcbMaxCols = 16384;   // x entries of one column block = 128 KiB of LDS (20480 = all 160 KiB measured slower: profiles/r04/c4_cols_rows_sweep.jsonl)
struct MatOptions {
    int csr_offset_dict = 1;       // use the 1-byte column code kernel when a matrix allows it
    int ell_offset_dict = 1;       // ELLPACK twin of csr_offset_dict (max_d <= 16)
    int csr_row_owner = 1;         // int32 columns, rows <= 64 entries: gather by the row's owner lane
    int csr_row_lines = 1;         // int32 columns, longer rows (up to 4096 entries): one 128-byte line of val per row and pass (k_csr_rl)
    int csr_sliced = 1;            // rows <= 8 entries, <= 15 offsets: slot-major slices + 4-bit codes (k_csr_sl); its siblings k_csr_slb / k_csr_sl32
    int csr_sell = 1;              // general matrices whose rows are too long / uneven for the uniform sliced form: SELL-128-512 (2 = whenever the padding allows)
    int csr_xwindow = 1;           // SELL form of a banded matrix: every slice's window of x staged in LDS (k_csr_sell<.., XW>); 0 = gathers from L2
    int csr_lean = 1;              // a matrix served by the sliced / SELL form keeps ONLY that form (+ row pointers) resident
    int ell_colblock = 1;          // ELLPACK with random columns: column-blocked two-phase product (0 never, 1 automatic, 2 always)
    int ell_colblock_cols = 16384; // its column block: x entries staged in LDS per workgroup (even, <= kEllcbMaxCols)
    int ell_colblock_rows = 0;     // rows per tile of its sum phase: 0 automatic, 256 or 512
    int coloring_pass = 0;         // greedy colouring / colour ordering of this matrix's graph: 0 = the fastest pass that applies (union-find
                                   // parity on the device, level sweep on the device, sequential host pass), 1 = from the level sweep on,
                                   // 2 = the host pass -- the same colours whichever (tests compare the passes)
    int slice_sched = 0;           // sliced kernels on matrices with a far stencil offset (3-D grids): tile-ordered slice schedule per XCD;
                                   // 0 off, 1 = bands of 64 slices, n > 1 = bands of n slices.  Off: the counters drop (fabric reads 9.9 ->
                                   // 8.6 / 7.1 GB at 464^3) but the time is within run-to-run noise of the computed maps (round 4: 1514 -> 1456 us
                                   // in one sweep, 1512 -> 1527 in the next; CG 367 -> 354 it/s): profiles/r04/c5_slice_sched_sweep.txt
};
struct SolverOptions {
    int cg_small = 1;              // CG (plain or Jacobi) on a small CSR matrix: the whole solve in one workgroup (k_cg_small);
                                   // 0 off, 1 on (50000 iterations per launch), n > 1 on with n iterations per launch
    int bicgstab_small = 1;        // BiCGStab (plain or Jacobi) on a CSR matrix of <= 4096 rows: the whole solve in one workgroup
    int krylov_graph = 1;          // CG / BiCGStab launch loops (one GPU, plain or Jacobi): replay a captured group of 16 iterations (hipGraph);
                                   // 0 off, 1 = once the solve has run 64 iterations, n > 1 = after n (rounded up to a multiple of 16)
    int dot_order = 0;             // dot products of CG / BiCGStab: 0 = tree (per-workgroup partial sums), 1 = the reference's order
    int gmres_cgs2 = 1;            // GMRES: 1 = low-synchronisation CGS-2 (k_gsl: basis read twice per step), 0 = modified Gram-Schmidt
    int coop_spin_limit = 0;       // cooperative CG / BiCGStab: polls before a hand-off gives up (0 = built-in 2^19; tests set 1 to force the fall-back)
    int cg_coop_variant = 0;       // cooperative CG / BiCGStab: low 4 bits pin the rows per thread (1, 2, 4, 8; 0 = by size), +16 = never the one-XCD variant
    int reorder_solve = 2;         // a preconditioner with ildu_reorder: 2 = the solve runs in the permuted order and CG folds its r update and
                                   // r.z into the two sweeps, 1 = permuted order, separate steps, 0 = r and z permuted around every apply
    int dist_halo_fused = 1;       // CG on a row partition: the boundary rows of r (z) travel beside the all-reduce of r.r (r.z) and p's
                                   // halo is formed locally -- no exchange in front of the product; 1 = one RCCL group with the all-reduce,
                                   // 2 = its own send / recv group just before it, 0 = off (p's halo exchanged by every product)
};
struct PcOptions {
    int ildu_strips = 1;           // ILDU(0) factors of grid-like matrices (deps r-1, r-w[, r-wh]): strip- / slab-pipelined triangular solves
    int ildu_rows = 1;             // ILDU(0) factors of a few levels (colour orderings): row-space sweeps, one launch per level (2 = every level launched)
    int pipeline_spin_limit = 0;   // strip / slab triangular solves: polls before a wait gives up (0 = built-in limit; tests set 1 to force an abort)
    int ildu_reorder = 0;          // ILDU(0) of the colour-ordered matrix P A P^T (the reference's greedy_color_ordering), applied as
                                   // z = P^T M^-1 P r: a factorisation of a few wide levels whatever order A is in; 0 = A as it is
};
struct Options { MatOptions mat; SolverOptions solver; PcOptions pc; };      // the defaults handles are created with
// name -> field of one of the three groups (nullptr: not in that group); shared by sgm_set_option and the per-handle setters
int *mat_option_field(MatOptions &o, const char *name);
int *solver_option_field(SolverOptions &o, const char *name);
int *pc_option_field(PcOptions &o, const char *name);
int normalise_option(const char *name, int value, int *out);     // range checks / rounding shared by all setters
extern Options g_opt;

struct Runtime {
    bool ready = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;      // halo exchange, overlapped with interior rows
    hipEvent_t ev_x_ready = nullptr, ev_halo_done = nullptr;
    bool async = false;
    int num_cu = 256;
};
extern Runtime g_rt;
int require_init();
bool trace_on();   // SGM_TRACE set: which path ran, one line per solve / setup on stderr
int finish();   // hipStreamSynchronize unless async
// pageable host <-> device, chunked through pinned buffers (synchronous)
int copy_big(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);

// Heartbeat (sgm_heartbeat): where the host thread of this process is inside the library, readable from ANOTHER thread while
// that thread is blocked in a synchronisation or a collective -- what makes a hang on a multi-GPU run say where it hangs.
// Plain counters written by the one thread that drives the library; a reader only ever looks.
enum { HB_IDLE = 0, HB_CREATE_DIST = 1, HB_HALO_POST = 2, HB_ALLREDUCE_POST = 3, HB_SOLVER_ENQUEUE = 4, HB_SOLVER_WAIT = 5,
       HB_SEQ_CHAIN = 6, HB_TRANSPOSE_DIST = 7, HB_PHASES = 8 };
struct Heartbeat {
    volatile int32_t phase = HB_IDLE;
    volatile int64_t beats = 0;            // bumped at every phase change and every solver batch
    volatile int64_t iteration = 0;        // iterations of the running solve the host has queued so far
    volatile int64_t halo_posts = 0;       // halo exchanges posted (send/recv groups) since sgm_init
    volatile int64_t allreduce_posts = 0;  // all-reduces posted
    volatile int64_t solves = 0;           // solver calls entered
};
extern Heartbeat g_hb;
inline void hb_phase(int ph) { g_hb.phase = ph; g_hb.beats = g_hb.beats + 1; }

template <class T>
int dalloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess)
        return fail(SGM_ERR_ALLOC, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
#ifdef SGM_POISON_ALLOC
    // debugging build (`make -C sigma_amd/csrc POISON=1`, tools/poison_visit.sh): every allocation starts as 0xFF bytes -- NaN
    // doubles, -1 indices -- so that a read of memory nothing has written yet shows in the first result instead of depending on
    // what the allocator handed back (a fresh GPU box hands back zeros; a busy one does not)
    // (hipMemset on the null stream is not ordered against the library's non-blocking stream: wait for it, or the poison lands
    //  on top of what is written next)
    (void)hipMemset(*p, 0xFF, count * sizeof(T));
    (void)hipDeviceSynchronize();
#endif
    return SGM_OK;
}
inline void dfree(void *p) { if (p) (void)hipFree(p); }

// ------------------------------------------------------------------ launch geometry
constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kMaxGrid = 8192;       // partial-sum slots per dot (>= the largest grid)

inline int vec_grid(int64_t n)
{
    int64_t g = (n + 4 * kBlock - 1) / (4 * kBlock);     // >= 4 elements per thread
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;            // 256 CUs x 8 resident 256-thread blocks
    return (int)g;
}

// A scalar that lives on the device as `count` partial sums (count == 1: already
// reduced, e.g. after an all-reduce).  Every consumer block re-reduces the partials in
// the same fixed order, so all blocks see the same bits and no host round trip or
// atomic is needed between a dot product and its consumers.
struct ScalarRef {
    const double *ptr;
    int count;
};

// ------------------------------------------------------------------ matrices
struct HaloNbr {
    int peer = -1;                 // rank (RCCL) or part index (local)
    int32_t send_count = 0;        // my owned entries the peer needs
    int32_t *send_idx = nullptr;   // device: local owned indices, in the peer's halo order
    double *send_buf = nullptr;    // device staging (RCCL only)
    int32_t recv_offset = 0;       // offset inside my halo region
    int32_t recv_count = 0;
};

// Slice schedule of the sliced kernels (sgm_spmv.hip, slice_sched): the order in which the workgroups of every XCD
// take the 512-row slices of one row range; tab[it * grid + workgroup] = slice or -1
struct SliceSched { int32_t *tab = nullptr; int32_t lo = 0, hi = 0, grid = 0, iters = 0, band = 0; };

struct Part {
    MatOptions opt = g_opt.mat;    // this matrix's options: the defaults at the moment the part is made; every part of a matrix holds the same
    int32_t n = 0;                 // owned rows
    int32_t ncol_own = 0;          // owned columns (== n for the square partitions used)
    int32_t n_halo = 0;
    int32_t int_lo = 0, int_hi = 0; // interior rows [int_lo, int_hi): only owned columns (when n_halo > 0)
    int64_t nnz = 0;
    int64_t row_begin = 0;         // first owned global row
    // CSR (device, 0-based; col indexes [owned | halo]); val/col padded by 2 entries
    int32_t *rowptr = nullptr;
    int32_t *col = nullptr;
    double *val = nullptr;
    // offset-dictionary form of col (built when the matrix has <= 255 distinct col-row
    // offsets, e.g. any stencil / banded matrix): col(k) = row + dict[code(k)], 1 byte/nnz
    uint8_t *code = nullptr;
    int32_t *dict = nullptr;       // 256 entries
    int32_t ndict = 0;
    int32_t dict_reach = 0;        // largest |offset| of the dictionary (set where the dictionary is built: the cooperative
                                   // kernels size their halo window from it)
    int32_t max_row = 0;           // longest row (entries); picks the row-owner kernel for short rows
    // sliced form (rows <= 8 entries, <= 15 distinct offsets, little padding): slices of 256
    // rows, values slot-major inside a slice (entry (slot u, row r) at ((r/256)*sw + u)*256 + r%256),
    // the row's column offsets as 8 four-bit dictionary codes in one word (15 = no entry)
    double *sval = nullptr;
    uint32_t *scode = nullptr;
    int32_t *scol = nullptr;       // sliced form WITHOUT a dictionary: the int32 column of every slot (-1 = no entry), same layout as sval
    uint8_t *sbcode = nullptr;     // sliced form for rows of 9..32 entries with a dictionary (k_csr_slb): 1-byte codes (255 = no entry),
                                   // per slice and chunk of 8 slots the 8 bytes of every row: ((slice * sw/8 + chunk) * 512 + row) * 8 + slot % 8
    bool lean = false;             // val / col / code were released after the sliced form was built (option "csr_lean"): csr_need_arrays
                                   // brings them back (from the sliced form) for whoever reads them, csr_release_arrays drops them again
    // SELL-128-512 (general matrices, any columns, rows of ANY length that differ inside a neighbourhood): the rows of every
    // window of 512 rows sorted by length, chunks of 128 sorted rows stored slot-major with the chunk's own width (k_csr_sell)
    double *sl_val = nullptr;      // values, chunk c at sl_off[c], slot u of position q at + u * 128 + q
    int32_t *sl_col = nullptr;     // columns beside them (-1 = no entry)
    uint16_t *sl_perm = nullptr;   // position -> row inside its sort window (0xffff = no row), n rounded up to whole slices
    int64_t *sl_off = nullptr;     // chunks + 1 offsets (entries)
    int64_t sl_total = 0;          // stored slots (entries + padding)
    int32_t *sl_win0 = nullptr;    // banded matrices: first column (even) of the window of x every 512-row slice gathers from ...
    int32_t sl_gs = 1;             // slices behind one window (1, or 2 for wide windows: a 512-thread workgroup)
    int32_t sl_span = 0;           // ... and the longest window (entries, even): k_csr_sell stages it in LDS (null / 0: not banded enough)
    int32_t sw = 0;                // slots per row in sval (3, 5, 7 or 8)
    int32_t sched_period = 0;      // rows: the far offset most rows carry (a 3-D grid's plane), 0 = none / near
    mutable SliceSched sched[3];   // built on first use, one per row range launched (whole part, or interior / head / tail)
    mutable int nsched = 0;
    const int32_t *run_sched = nullptr;   // (views of one launch) the schedule to walk, or null = computed maps
    int32_t run_iters = 0;
    // ELLPACK (device, slot-major: entry (slot k, row i) at k*n + i)
    int32_t max_d = 0;
    int32_t *ecol = nullptr;
    double *eval = nullptr;
    int32_t *edeg = nullptr;       // degrees(n), only when built by sgm_ell_from_edges
    uint8_t *ecode = nullptr;      // ELLPACK offset-dictionary codes, row-major n x emdp (emdp = max_d rounded up)
    int32_t emdp = 0;
    // column-blocked two-phase form of an ELLPACK matrix with random columns (sgm_ellcb.hip)
    int32_t *cb_perm = nullptr;    // sorted position -> entry i*max_d + k
    double *cb_sval = nullptr;     // values in (column block, row, slot) order
    uint16_t *cb_lcol = nullptr;   // column inside its block
    int32_t *cb_bstart = nullptr;  // nb+1: where every column block starts in the sorted order
    uint16_t *cb_lpos = nullptr;   // slot-major: position of entry (k, i) in its tile's LDS image
    void *cb_fdesc = nullptr;      // ntiles x nb run descriptors {sorted position, length | LDS base << 16}
    double *cb_P = nullptr;        // products val * x in sorted order (written by phase 1)
    int32_t cb_cols = 0, cb_nb = 0, cb_R = 0, cb_ntiles = 0, cb_chunks = 0;
    int32_t cb_maxd = 0;           // slots per row of that form: max_d (ELLPACK) or the longest row (a CSR matrix with scattered columns)
    int64_t cb_count = 0;          // sorted positions that hold entries (= n * max_d for an ELLPACK matrix, nnz for a CSR one)
    double col_spread = -1.0;      // mean |column - row| over a sample of rows, when it was looked at (-1: not)
    std::vector<HaloNbr> nbrs;
    double *xext = nullptr;        // owned+halo staging for plain-vector matvec (multi-part only)
    int dot_grid_override = 0;     // composite matrices: grid of the separate dot kernel
    int64_t xlen() const { return (int64_t)ncol_own + n_halo; }
};

}  // namespace sgm

struct sgm_comm_s {
    int rank = 0, nranks = 1;
    void *nccl = nullptr;          // ncclComm_t
    void *nccl_halo = nullptr;     // optional second communicator: the halo send / recv pairs (sgm_comm_attach_halo_comm)
    // does this transport take ONE group holding neighbour send / recv pairs AND an all-reduce (what CG posts per iteration with
    // option dist_halo_fused = 1)?  Probed once, collectively, by sgm_comm_init and agreed over all ranks; 0 = the solvers
    // post the pairs and the all-reduce separately (what dist_halo_fused = 2 does), whatever the option says
    int group_ok = 1;
};

// every matrix handle has a serial number of its own, and a version that every change of its entries or their order bumps:
// what a cached derivative (a preconditioner's permuted copy) checks before it stands in for the matrix
inline uint64_t next_mat_serial() { static std::atomic<uint64_t> c{0}; return ++c; }
struct sgm_mat_s {
    uint64_t serial = next_mat_serial(), version = 0;
    uint64_t pattern_version = 0;  // bumped by what changes the POSITIONS of the entries (the permutations); version counts those and value changes
    int32_t fmt = 0;
    int32_t nrow = 0, ncol = 0;    // global
    int64_t nnz = 0;               // global (local sum when distributed)
    std::vector<sgm::Part> parts;  // 1 unless created with sgm_csr_create_partitioned
    sgm_comm comm = nullptr;       // RCCL communicator when distributed over processes
    std::vector<int64_t> row_starts;   // distributed: the row partition (nranks+1, 0-based)
    std::vector<int64_t> col_starts;   // distributed: the partition of x (== row_starts unless created by sgm_csr_create_dist_rect)
    std::vector<int32_t> halo_cols;    // distributed: global 1-based column of every halo slot
    bool distributed() const { return comm != nullptr || parts.size() > 1; }
    // explicit transpose for matvec_t (built on first use; rows sorted by (source row, slot) so
    // that every y(i) receives its terms in the reference's order)
    // composite ("matrix of matrices", sparse_matrix_composites.f90:41-162): blocks are not owned
    std::vector<int32_t> blk_row_ptr, blk_col_ptr;      // 0-based offsets, nrb+1 / ncb+1
    std::vector<sgm_mat_s *> blocks;                    // nrb x ncb, row-major, may hold nullptr
    sgm_mat_s *T = nullptr;
    int32_t *tperm = nullptr;      // device: position in this matrix's val/eval of each entry of T
    bool t_stale = true;
};

namespace sgm {

// y = [y +] A^T x on a matrix distributed over processes (sgm_dist.hip): x = owned rows, y = owned columns
int matvec_t_dist(sgm_mat A, const double *x, double *y, int where, bool add);
// exchange the halo part of an extended vector set (one pointer per local part)
int halo_exchange(sgm_mat A, double *const *xext, hipStream_t st);
// sum `count` scalar slots across parts / ranks (in place, every part gets the total)
int allreduce_slots(sgm_mat A, double *const *slot_ptrs, int count);
// both at once on the launch stream (over RCCL: one group); uext = one extended vector per local part
int halo_exchange_allreduce(sgm_mat A, double *const *uext, double *const *slot_ptrs, int count, bool grouped);

// Phase timers of the row-partitioned path (sgm_dist_profile): HIP events around the phases of every product and dot, so
// that a multi-GPU run says where an iteration's time goes.  Off: no event is recorded.
enum { PH_HALO = 0, PH_INTERIOR = 1, PH_HALO_WAIT = 2, PH_BOUNDARY = 3, PH_DOT_REDUCE = 4, PH_ALLREDUCE = 5, PH_COUNT = 6 };
struct PhaseMark { hipEvent_t a = nullptr, b = nullptr; };
bool prof_on();
void prof_begin(int phase, hipStream_t st);              // records the phase's start event on st
void prof_end(int phase, hipStream_t st);                // ... and its end event
hipEvent_t prof_event(hipStream_t st);                   // a pooled event recorded on st now (null when off)
void prof_span(int phase, hipEvent_t a, hipEvent_t b);   // a phase between two already recorded events

// dot_order = 1 across ranks: one running sum (a device double) travels rank 0 -> 1 -> ... -> R-1, every rank continuing it over its
// own rows in between; seq_chain_recv takes delivery from rank-1 (no-op on rank 0), seq_chain_share passes it on to rank+1
// and leaves the total (what the last rank holds) in `run` on every rank.
int seq_chain_recv(sgm_mat A, double *run);
int seq_chain_share(sgm_mat A, double *run);

// y = [y +] A x on every part; optional fused dots: partial sums of w[i]*y[i] (dot_w) and
// y[i]*y[i] (dot_yy) per block into partials arrays (kMaxGrid doubles each).
struct SpmvDots {
    const double *const *w = nullptr;   // per part; nullptr = no w.y dot
    double *const *part_wy = nullptr;   // per part partial arrays
    double *const *part_yy = nullptr;   // per part partial arrays or nullptr
};
int spmv_parts(sgm_mat A, const double *const *x, double *const *y, bool add,
               const SpmvDots *dots, const int *flag_done, int *grid_out, int gen = 0x7fffffff,
               bool chain = false, bool halo_ready = false);      // halo_ready: x's halo slots are current -- no exchange

// upload one CSR row block (1-based arrays as the Fortran holds them) and build its device formats; `validate` checks the
// index arrays on the device as they are converted (SGM_ERR_BAD_ARG / SGM_ERR_DIMS naming the first offending row)
int build_csr_part(Part &p, int32_t n, int32_t ncol_own, int32_t n_halo, int64_t nnz,
                   const int32_t *ptr1, const int32_t *node1, const double *val, int where, bool validate = true);
int clone_csr_plain(sgm_mat A, sgm_mat *out);                                   // plain-CSR device copy (setup scratch)
int color_order_device(sgm_mat A, int32_t **dp, std::vector<int32_t> &ptrs);    // greedy_color_ordering, p left on the device (sgm_order.hip)
// (sgm_order.hip) a row block's DIAGONAL block (its rows x its owned columns) as a plain single-GPU CSR matrix: what a part
// of a row partition colours / orders locally
int diag_block_plain(const Part &p, sgm_mat *out);
// (sgm_order.hip) q = the row block p with row i moved to row p1(i), its owned columns renumbered by p1 (1-based, device) and
// its halo columns as they are; stored order inside the rows kept; p's kernel forms (options) and halo links (the sender's
// row numbers mapped through p1).  No interior range: a product on q runs after its halo has arrived.
void part_kernel_name(const Part &p, int fmt, char *name, size_t len);       // (sgm_mat.hip) the SpMV kernel a part runs with
int permuted_part(const Part &p, const int32_t *p1_dev, Part &q, const int32_t *hmap_dev = nullptr,
                  const std::vector<int32_t *> *send_order = nullptr);
// (sgm_order.hip) hmap: old halo slot -> new, the slots of every neighbour's segment (offset, count) ordered by the permuted index
// of the row they attach to; send_order[k] (device, one per entry of p.nbrs): where entry j of that link goes in the receiver's halo
int halo_attach_order(const Part &p, const int32_t *p1_dev, const std::vector<std::pair<int32_t, int32_t>> &segments,
                      std::vector<int32_t> &hmap_host);
// (sgm_dist.hip) ranks: every receiver tells each sender the order it wants that link's entries in (one grouped send / recv of
// int32 lists); `mine` = this rank's hmap (host); out[k] = device list for nbrs[k] (null where nothing is sent)
int exchange_halo_orders(sgm_mat A, const std::vector<int32_t> &mine, std::vector<int32_t *> &out);
int spmv_grid(const Part &p, bool whole = false);      // whole: the product launched with halo_ready (one range)
int matvec_plain(sgm_mat A, const double *x, double *y);     // device vectors, sgm_mat_matvec's layout, stream-ordered
// "csr_lean": the CSR-order arrays of a part that kept only its sliced form, on demand (no-op otherwise)
int csr_need_arrays(const Part &p);
void csr_release_arrays(const Part &p);

// sgm_trsv3.hip: slab-pipelined triangular solves for ILDU(0) factors of 3-D grids (deps r-1, r-w, r-w*h)
struct Slab3;
int slab3_build(Slab3 **out, int32_t n, const std::vector<int32_t> &Lptr, const std::vector<int32_t> &Lnode,
                const std::vector<int32_t> &Uptr, const std::vector<int32_t> &Unode, const int32_t *dLptr, const int32_t *dLnode,
                const int32_t *dUptr, const int32_t *dUnode);      // *out null: not applicable
int slab3_refresh(Slab3 *S, const double *Lval, const double *Uval, const double *D);     // (device pointers)
void slab3_apply(const Slab3 *S, const double *r, double *z, const int *flag, int spin_limit, int32_t *sticky);
void slab3_lower_result(const Slab3 *S, double *dst);
int slab_dims_device(int32_t n, const int32_t *dptr, const int32_t *dnode, bool lower, int32_t *w, int32_t *h);
void slab3_dims(const Slab3 *S, int32_t *w, int32_t *h);
int slab3_aborted(const Slab3 *S, int32_t *abL, int32_t *abU);
void slab3_info(const Slab3 *S, int32_t out[6]);
int slab3_clocks(const Slab3 *S, std::vector<long long> &out);
void slab3_free(Slab3 *S);

}  // namespace sgm

// ====================================================================== device helpers
#if defined(__HIPCC__)
namespace sgm {

// Sum over the 64 lanes of a wave, every lane the same bits: the butterfly
//     for (off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
// without its six trips through the LDS crossbar (ds_bpermute: ~130 cycles each for a double, a dependent chain -- 0.75 us
// per block sum of a 1024-thread workgroup, four of them in an iteration of the single-launch CG kernels).  Offsets 32 and 16
// are gfx950's v_permlane32_swap / v_permlane16_swap of the register with a copy of itself: one result holds the lower
// half's (the even rows') values in every lane, the other the upper half's (the odd rows'), and their sum is own + partner or
// partner + own -- the same bits.  After them every row of 16 lanes holds the same 16 values, and after each further step the
// period halves: lane j's partner j ^ 8, j ^ 4, j ^ 2, j ^ 1 holds what the lane 8, 4, 2, 1 places round the row holds, which
// is a DPP row rotation on the operand read.  Bit-identical to the butterfly (tools/probes/wave_sum_probe.cpp compares them).
template <int CTRL>
__device__ inline double dpp_lane(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ inline double wave_sum(double v)
{
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
    }
    v += dpp_lane<0x128>(v);        // row_ror:8
    v += dpp_lane<0x124>(v);        // row_ror:4
    v += dpp_lane<0x122>(v);        // row_ror:2
    v += dpp_lane<0x121>(v);        // row_ror:1
    return v;
}

// Deterministic block-wide sum; every thread returns the same bits.
template <int BLOCK>
__device__ inline double block_sum(double v, double *red /* BLOCK/64 doubles of LDS */)
{
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) s += red[w];
    return s;
}

// two block sums sharing their barriers (each summed exactly as block_sum does: same bits); red: 2 * BLOCK / 64 doubles
template <int BLOCK>
__device__ inline void block_sum2(double a, double b, double *red, double &sa, double &sb)
{
    a = wave_sum(a);
    b = wave_sum(b);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[wave] = a; red[BLOCK / 64 + wave] = b; }
    __syncthreads();
    double s = red[0], t = red[BLOCK / 64];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) { s += red[w]; t += red[BLOCK / 64 + w]; }
    sa = s;
    sb = t;
}

template <int BLOCK>
__device__ inline double load_scalar(ScalarRef r, double *red)
{
    if (r.count == 1) return r.ptr[0];
    double v = 0.0;
    for (int i = threadIdx.x; i < r.count; i += BLOCK) v += r.ptr[i];
    return block_sum<BLOCK>(v, red);
}

// K device scalars at once: the first round of every scalar's partial sums is requested before any of them is used, and
// the K block sums share their two barriers -- ONE memory round trip and one reduction where K calls of load_scalar take K
// of each (the prologue of the Krylov update kernels: 2-6 scalars; at n = 1e5 those round trips were a third of a CG
// iteration).  Every scalar is summed in exactly load_scalar's order: same bits.  red: K * BLOCK / 64 doubles of LDS.
template <int BLOCK, int K>
__device__ inline void load_scalars(const ScalarRef (&r)[K], double (&out)[K], double *red)
{
    double v[K];
    bool any = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        // (count == 1: already reduced -- every thread reads the value itself, nothing is added to it)
        const int i = r[k].count == 1 ? 0 : (int)threadIdx.x;
        v[k] = i < r[k].count ? r[k].ptr[i] : 0.0;
        any = any || r[k].count > 1;
    }
    if (!any) {
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = v[k];
        return;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (r[k].count > 1) {
            v[k] = 0.0 + v[k];
            for (int i = threadIdx.x + BLOCK; i < r[k].count; i += BLOCK) v[k] += r[k].ptr[i];
            v[k] = wave_sum(v[k]);
        }
    }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[k * (BLOCK / 64) + wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (r[k].count > 1) {
            double s = red[k * (BLOCK / 64)];
#pragma unroll
            for (int w = 1; w < BLOCK / 64; ++w) s += red[k * (BLOCK / 64) + w];
            out[k] = s;
        } else {
            out[k] = v[k];
        }
    }
}

}  // namespace sgm
#endif
