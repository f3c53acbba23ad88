/*
 * sigma_hip.h -- C ABI of the MI355X-native SpMV + Krylov path for SiGMA.
 *
 * This is the drop-in boundary.  The reference (danshapero/sigma) has
 * no usable FFI for this path: src/wrapper.f90 is dead code that wraps graph edits only
 * (src/wrapper.f90:100-303, excluded at src/CMakeLists.txt:39-40).  What the Fortran
 * host binds with ISO_C_BINDING is therefore the set of type-bound procedures that make
 * up the path; every entry point below names the one it replaces.  Conventions follow
 * src/wrapper.f90: opaque handles (type(c_ptr)), `integer(c_int), value` scalars, and
 * index arrays are handed over EXACTLY as the Fortran holds them -- 1-based int32
 * (src/graph/formats/cs_graphs.f90:16, src/graph/formats/ellpack_graphs.f90:14) --
 * conversion to the device layout is this library's job.
 *
 * Errors: the reference prints and calls exit(1) (src/solver/cg_solvers.f90:61-65).
 * Here every function returns an int status (0 = ok) and sgm_last_error() holds the
 * message; sigma_amd/fortran/sigma_hip.f90 turns nonzero into print + exit(1).
 *
 * `where` arguments say where the caller's arrays live: SGM_HOST (copied across PCIe
 * inside the call) or SGM_DEVICE (HBM pointers, e.g. from sgm_malloc or a torch tensor).
 * All calls are synchronous on return unless sgm_set_async(1) was called.
 * Not thread-safe (the reference is single-threaded); one process drives one GPU.
 */
#ifndef SIGMA_HIP_H
#define SIGMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgm_mat_s *sgm_mat;       /* class(sparse_matrix_interface) leaf: csr_matrix / ellpack_matrix */
typedef struct sgm_pc_s *sgm_pc;         /* class(linear_solver) used as preconditioner: jacobi / ldu      */
typedef struct sgm_solver_s *sgm_solver; /* class(linear_solver): cg / bicgstab / gmres                    */
typedef struct sgm_comm_s *sgm_comm;     /* row-partition communicator (RCCL over xGMI); no reference analogue */

enum {
    SGM_OK = 0,
    SGM_ERR_BAD_ARG = 1,
    SGM_ERR_DIMS = 2,          /* non-square matrix handed to a solver: cg_solvers.f90:61-65 */
    SGM_ERR_HIP = 3,
    SGM_ERR_RCCL = 4,
    SGM_ERR_NOT_CONVERGED = 5, /* only when a max_iter extension is set and hit */
    SGM_ERR_NO_DEVICE = 6,
    SGM_ERR_ALLOC = 7,
    SGM_ERR_UNSUPPORTED = 8
};
enum { SGM_HOST = 0, SGM_DEVICE = 1 };
enum { SGM_FMT_CSR = 1, SGM_FMT_ELL = 2, SGM_FMT_COMPOSITE = 3 };
enum { SGM_SOLVER_CG = 1, SGM_SOLVER_BICGSTAB = 2, SGM_SOLVER_GMRES = 3 };
enum { SGM_PC_JACOBI = 1, SGM_PC_ILDU0 = 2 };

/* ---- runtime -------------------------------------------------------------------- */
int sgm_init(int device);                 /* hipSetDevice + stream; fails loudly without a GPU */
int sgm_finalize(void);
const char *sgm_last_error(void);
int sgm_set_stream(void *hip_stream);     /* adopt the caller's hipStream_t (NULL = library stream) */
int sgm_set_async(int on);                /* 1: calls return without hipStreamSynchronize */
int sgm_synchronize(void);
/* ---- options ------------------------------------------------------------------------- *
 * Options choose among device formats / kernels / loop structures that all give the SAME bits (the one exception,
 * "dot_order", is spelled out below).  They are PER HANDLE: a matrix, solver or preconditioner copies the process-wide
 * defaults when it is created and keeps its own copy from then on --
 *     sgm_set_option(name, v)            the default that handles created LATER start with (touches no existing handle)
 *     sgm_mat_set_option(A, name, v)     this matrix      (composite: every block; A^T follows A)
 *     sgm_solver_set_option(s, name, v)  this solver      (read at the next solve)
 *     sgm_pc_set_option(pc, name, v)     this preconditioner
 * -- so two handles of one process may run different kernels, and no caller can change another caller's solver.
 * Unknown names, or a name of the wrong group, are SGM_ERR_BAD_ARG.
 *
 * Matrix options
 *   "csr_offset_dict" (1)  CSR matrices whose column-minus-row offsets take <= 255 distinct values (every stencil / banded
 *                          matrix) hold their columns as 1-byte dictionary codes: 9 instead of 12 bytes per entry; 0 = int32 kernels
 *   "ell_offset_dict" (1)  the same for ELLPACK matrices with max_d <= 16
 *   "csr_sliced" (1)       rows of <= 8 entries from <= 15 offsets (1-D / 2-D / 3-D stencils): values in slices of 512 rows,
 *                          slot-major, one 32-bit word of 4-bit codes per row, a lane owns two adjacent rows, no row pointers
 *                          read (8 W + 4 bytes per row of width W; k_csr_sl).  Its siblings for rows of 9..32 entries with a
 *                          dictionary (k_csr_slb) and for <= 32 similar-length entries at arbitrary columns (k_csr_sl32)
 *   "csr_row_owner" (1)    int32-column matrices with rows <= 64 entries: row-owner gather kernel; 0 = kernels for longer rows
 *   "csr_row_lines" (1)    longer rows of similar length (mean >= 16, none beyond 4096 or 4 x the mean): line-staged row-owner
 *                          kernel; 0 = the balanced streaming-gather kernel (any row length)
 *   "csr_sell" (1)         general matrices whose rows are too long / uneven for the uniform sliced form: SELL-128-512 (rows of
 *                          every 512-row window sorted by length, chunks of 128 rows slot-major with the chunk's own width),
 *                          taken from a longest row of 49 entries on; 2 = whenever the padding allows; 0 = the CSR kernels
 *   "csr_xwindow" (1)      the SELL form of a BANDED matrix (every 512-row slice gathers from a window of x that fits 144 KiB of
 *                          LDS, and re-uses it): the slice's window is staged in LDS with coalesced loads and every gather is an
 *                          LDS read, instead of one 128-byte line moved L2 -> L1 per 8-byte gather; 0 = gathers from L2
 *   "csr_lean" (1)         a matrix served by the sliced / SELL form keeps ONLY that form (+ row pointers) in HBM (C2: 0.48
 *                          instead of 1.13 GB); its CSR-order arrays are rebuilt on the device for whoever reads them
 *   "ell_colblock" (1)     ELLPACK matrices -- and CSR matrices with rows of similar length -- whose columns have no locality
 *                          (x >= 7 MiB, >= 8 slots per row, no offset dictionary): column-blocked two-phase product (products
 *                          through LDS-resident x blocks, then ordered row sums); 0 never, 2 always
 *   "ell_colblock_cols" (16384 = 128 KiB of LDS, the maximum)  x entries per block
 *   "ell_colblock_rows" (0 = automatic; 256 or 512)  rows per tile of the second phase
 *   "coloring_pass" (0)    sgm_graph_greedy_coloring / _greedy_color_order on this matrix's graph: 0 = the fastest pass that
 *                          applies (parities by union-find on the device, level sweep on the device, the reference's sequential
 *                          pass on the host), 1 = from the level sweep on, 2 = the host pass; the same colours whichever
 *   "slice_sched" (0)      sliced matrices most of whose rows carry a far offset (the plane stride of a 3-D grid): the slices
 *                          are handed to the XCDs tile by tile -- the plane is cut into bands (1 = of 64 slices, n > 1 = of n
 *                          slices), XCD x sweeps bands x, x + 8, ... plane after plane, so that the three planes a band reads
 *                          share ONE XCD's L2.  464^3: fabric reads 9.9 -> 8.6 GB per product with bands of 64, 7.1 GB
 *                          with bands of 8 -- and the time stays within run-to-run noise (the re-fetched planes were
 *                          Infinity-Cache hits; profiles/r04/c5_slice_sched_sweep.txt), hence off; only the ORDER of whole slices changes
 * Solver options
 *   "cg_small" (1)         CG (plain / Jacobi) with the WHOLE SOLVE in one launch: on a single-GPU matrix of <= 10240 rows
 *                          (stencil; 4096 otherwise) and <= 49k stored slots as ONE workgroup -- p in LDS, x and r in
 *                          registers --; on a larger stencil matrix, up to 256 x 4096 rows, as up to 256 co-resident workgroups
 *                          (one per CU) that own 1024..4096 rows each and meet twice per iteration through sc1 stores / polls
 *                          (k_cg_coop: n = 1e6 17.7 instead of 36 us per iteration, n = 1e5 10.7 instead of 13.0); every wait
 *                          is bounded, a hand-off that gives up hands the solve to the launch loop.  A launch runs at most
 *                          50000 iterations (n > 1: at most n) and the solve continues in the next one from parked r, p,
 *                          res2, bit-identical to the uncut solve; 0 = the launch loop (three kernels per iteration)
 *   "bicgstab_small" (1)   the same for BiCGStab (one workgroup <= 4096 rows; the cooperative kernel k_bicg_coop beyond, for
 *                          the systems the cooperative CG kernel takes)
 *   "krylov_graph" (1)     the CG / BiCGStab launch loops on one GPU (plain / Jacobi) go on as replays of ONE captured group
 *                          of 16 iterations (a hipGraph) once a solve has run 64 iterations (n > 1: n, rounded up to a
 *                          multiple of 16); 0 = launch every kernel
 *   "gmres_cgs2" (1)       how GMRES orthogonalises: 1 = classical Gram-Schmidt applied twice in its low-synchronisation form --
 *                          the stored basis vector is the ONCE-projected one and the second projection lives in the Cholesky
 *                          factor R of the stored columns' Gram matrix (V = S R^-1 never formed; H = R Gs R^-1): the basis is
 *                          read twice per step and two reductions (all-reduces) are taken, 2 k + 3 vector passes at basis
 *                          size k; 0 = modified Gram-Schmidt (k + 2 dependent passes and reductions, 4 k + 8 vector passes),
 *                          the checker.  (Round 4's blocked CGS-2 with the second projection applied -- three passes, three
 *                          reductions -- was 1265 it/s on C3 against 1644 and is gone.)
 *   "dot_order" (0)        how CG / BiCGStab add up their dot products.  0 = tree order (per-workgroup partial sums,
 *                          re-reduced in a fixed order): a legal order for the Fortran intrinsic, deterministic, the fast one.
 *                          1 = the order the reference build uses (amdflang -O2 turns dot_product into ONE accumulator fed
 *                          first element to last, cg_solvers.f90:131,135,140): every iterate, iteration count and residual is
 *                          then BIT-IDENTICAL to the reference's, also on row partitions and across ranks; about 4 ns per
 *                          element -- a VALIDATION mode for n up to ~1e5.  GMRES (no reference counterpart) keeps the tree order
 *   "coop_spin_limit" (0 = built-in, 2^19 polls)  how often a hand-off of the cooperative CG / BiCGStab kernels polls before
 *                          it gives up (the launch loop then takes the solve from the caller's x); tests set 1 to force that path
 *   "cg_coop_variant" (0)  which cooperative kernel serves a system, all the same statements: low four bits = rows per thread
 *                          pinned (1, 2, 4 or 8; 0 = by size), + 16 = never the variant that keeps a small system on one XCD
 *   "reorder_solve" (2)    with a preconditioner that factorised the colour-ordered matrix ("ildu_reorder"): 2 = the whole solve
 *                          runs in that order (x, b permuted once each way, products on P A P^T) and CG folds its r update and
 *                          r.z into the two sweeps; 1 = that order, separate steps; 0 = r and z permuted around every apply
 *   "dist_halo_fused" (1)  CG on a row partition (ranks or in-process parts): the boundary rows of r (z with a preconditioner)
 *                          travel in the same step as the all-reduce of r.r (r.z) -- over RCCL ONE ncclGroup holding the
 *                          send / recv pairs and the all-reduce -- and every part forms its halo copy of p itself by the owner's
 *                          statement p = r + beta p (cg_solvers.f90:142): the product starts without an exchange and without a
 *                          wait for one.  Same operands, same statement: the same bits as exchanging p.  2 = the send / recv
 *                          group on its own just before the all-reduce; 0 = p's halo exchanged in front of every product
 *                          (on the communication stream, overlapped with the interior rows)
 * Preconditioner options
 *   "ildu_strips" (1)      ILDU(0) factors of grid-like matrices use the strip- / slab-pipelined triangular solves; 0 = the
 *                          level-scheduled walkers
 *   "ildu_rows" (1)        ILDU(0) factors of <= 32 levels (what a colour ordering leaves) are swept in row space, one launch
 *                          per level on the vectors themselves; 2 = every level launched (1 skips two); 0 = the walkers
 *   "pipeline_spin_limit" (0 = built-in, 2^22 polls)  how often a wait inside the pipelined sweeps polls before it gives up
 *                          (a sweep that gives up is redone with the walkers and the pipeline retired for that handle:
 *                          sgm_pc_get "pipeline_retired"); tests set 1 to force that path
 *   "ildu_reorder" (0)     1 = the preconditioner is ILDU(0) of the COLOUR-ORDERED matrix P A P^T (P = the reference's
 *                          greedy_color_ordering of A's graph, permutations.f90:162-205; on the device for bipartite graphs such
 *                          as the 5- / 7-point grids); sgm_pc_apply is z = P^T M^-1 P r.  Its factors have one dependency level per
 *                          colour whatever order A is in, so an apply is a few bandwidth-bound launches instead of a
 *                          dependency chain of nx + ny levels.  The caller keeps A, b, x as they are; the preconditioner keeps
 *                          P A P^T, and sgm_solver_solve -- handed the matrix the preconditioner was set up with, unchanged
 *                          since -- runs the whole solve in that order (x, b permuted once each way).  A different
 *                          (equally valid) preconditioner than ILDU(0) in A's own order: iteration counts are those of the
 *                          permuted system -- off by default because the reference's `ldu()` factors A in the given order
 * Process-wide (sgm_set_option only)
 *   "dist_force_collectives" (0)  1 = a matrix distributed over ONE rank still issues its all-reduces (the fixed cost of the
 *                          RCCL code path, measurable on a single-GPU box: bench.py's `dist_overhead_1rank`)          */
int sgm_set_option(const char *name, int value);
int sgm_mat_set_option(sgm_mat A, const char *name, int value);
int sgm_solver_set_option(sgm_solver s, const char *name, int value);
int sgm_pc_set_option(sgm_pc pc, const char *name, int value);
/* The schedule itself, host-only (no HIP call; what the library uploads for a row range of n_slices
 * 512-row slices): tab_out[it * grid + workgroup] = slice or -1, iters_out = entries per workgroup.
 * tab_out may be NULL to ask for iters_out only; capacity in entries (>= iters * grid).          */
int sgm_slice_sched_host(int64_t n_slices, int64_t period_rows, int32_t grid, int32_t band_slices,
                         int32_t *tab_out, int64_t capacity, int32_t *iters_out);
/* sgm_heartbeat: where the thread that drives the library is right now -- callable from ANOTHER host thread while that one
 * is blocked in a synchronisation or a collective (bench.py's watchdog: a multi-GPU run that hangs says where).
 * out6 = {phase code, beats (bumped at every phase change and solver batch), iterations the running solve has queued,
 * halo exchanges posted, all-reduces posted, solver calls entered}; phase_name (optional) receives the phase in words.
 * Two ranks stuck with different post counts name the rank that fell behind.  No HIP call; never blocks.              */
int sgm_heartbeat(int64_t *out6, char *phase_name, int len);
int sgm_malloc(void **p, size_t bytes);   /* HBM buffer for hosts without a device allocator */
int sgm_free(void *p);
int sgm_memcpy(void *dst, const void *src, size_t bytes, int kind); /* 0 h2d, 1 d2h, 2 d2d */

/* ---- matrices -------------------------------------------------------------------- *
 * sgm_csr_create    <- csr_matrix: g%ptr(n+1), g%node(nnz), val(nnz)
 *                      src/matrix/formats/cs_matrices.f90:32-38,112-151; cs_graphs.f90:16
 * sgm_ell_create    <- ellpack_matrix: g%node(max_d,n), val(max_d,n) column-major,
 *                      padding = last neighbour / 0.0
 *                      src/matrix/formats/ellpack_matrices.f90:28-33; ellpack_graphs.f90:14,164
 * sgm_*_set_values  <- re-upload after host-side set_value/add_value
 *                      (test/solver_test_jacobi.f90:240-274 edits A, then re-runs setup)
 * sgm_mat_matvec    <- A%matvec(x,y): y = 0 ; matvec_add
 *                      src/linear_operator/linear_operator_interface.f90:185-194
 * sgm_mat_matvec_add<- csr_matvec_add cs_matrices.f90:600-622 /
 *                      ellpack_matvec_add ellpack_matrices.f90:640-665
 * sgm_mat_destroy   <- A%destroy()
 * Results are bit-identical to the reference loops: per row, products are rounded
 * individually and added left to right in stored order (no FMA, no reassociation).   */
int sgm_csr_create(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t nnz,
                   const int32_t *ptr_1based, const int32_t *node_1based,
                   const double *val, int where);
int sgm_csr_set_values(sgm_mat A, const double *val, int where);
int sgm_ell_create(sgm_mat *out, int32_t nrow, int32_t ncol, int32_t max_d,
                   const int32_t *node_1based_colmajor, const double *val_colmajor,
                   int where);
int sgm_ell_set_values(sgm_mat A, const double *val_colmajor, int where);
int sgm_mat_matvec(sgm_mat A, const double *x, double *y, int where);
int sgm_mat_matvec_add(sgm_mat A, const double *x, double *y, int where);
/* sgm_mat_matvec_t     <- A%matvec_t(x,y): y = 0 ; matvec_t_add
 *                         src/linear_operator/linear_operator_interface.f90:199-208
 * sgm_mat_matvec_t_add <- csc_matvec_add (the csr transpose kernel) cs_matrices.f90:627-647 /
 *                         ellpack_matvec_t_add ellpack_matrices.f90:670-693
 * x has nrow entries, y has ncol.  The reference scatters y(node(k)) += val(k)*x(j); here an
 * explicit transpose is built on first use so that every y(i) is summed in that same order
 * (bit-identical, no atomics).  On a matrix distributed over processes (sgm_csr_create_dist) the call is
 * collective: A^T is built once as another distributed matrix (every entry travels to the rank owning its
 * column), x = this rank's owned rows, y = its owned columns.  Not available on in-process partitions.   */
int sgm_mat_matvec_t(sgm_mat A, const double *x, double *y, int where);
int sgm_mat_matvec_t_add(sgm_mat A, const double *x, double *y, int where);
/* sgm_csr_from_edges / sgm_ell_from_edges <- the assembly sequence of the reference's tests
 *   g%add_edge(i,j)... ; convert_graph_type(g, "compressed sparse" | "ellpack") ; A%set_graph(g) ;
 *   A%set_value(i,j,z)...        (test/solver_test_jacobi.f90:73-128)
 * i.e. ll_graphs.f90:355-370 (repeated edges ignored) + cs_graphs.f90:109-197 /
 * ellpack_graphs.f90:105-170 + cs_matrices.f90:840-863 (the last value written wins).
 * The edge list (1-based, INSERTION order) is turned into the same ptr/node/val (or
 * node(max_d,n)/val/degrees) arrays on the device -- bit-identical index work.
 * sgm_mat_get reads a leaf matrix back in the reference's layout: "ptr","node","val" (CSR),
 * "max_d","degrees","node","val" (ELLPACK, (max_d,n) Fortran order).                     */
int sgm_csr_from_edges(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t ne, const int32_t *ei_1based,
                       const int32_t *ej_1based, const double *ev, int where);
int sgm_ell_from_edges(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t ne, const int32_t *ei_1based,
                       const int32_t *ej_1based, const double *ev, int where);
int sgm_mat_get(sgm_mat A, const char *name, void *out_host, size_t bytes, size_t *needed);
/* sgm_composite_create <- type(sparse_matrix), the block "matrix of matrices"
 *                         src/matrix/sparse_matrix_composites.f90:41-162; matvec_add = loop over
 *                         the blocks `C%matvec_add(x(j1:j2), y(i1:i2))`, :1076-1099 (row blocks
 *                         outer, column blocks inner), matvec_t_add :1104-1127 (column blocks
 *                         outer).  row_ptr/col_ptr are the 1-based block offsets (nrb+1 / ncb+1
 *                         entries); blocks[it*ncb + jt] are leaf handles (NULL = empty block) that
 *                         stay owned by the caller.  The handle works with matvec(_t)(_add) and
 *                         with the unpreconditioned solvers.                               */
int sgm_composite_create(sgm_mat *out, int32_t nrb, int32_t ncb, const int32_t *row_ptr_1based,
                         const int32_t *col_ptr_1based, const sgm_mat *blocks);
int sgm_mat_info(sgm_mat A, int32_t *nrow, int32_t *ncol, int64_t *nnz, int32_t *fmt,
                 int64_t *x_len /* entries matvec reads from x: ncol, or owned+halo when distributed */);
/* ---- re-orderings ------------------------------------------------- *
 * Graph = the matrix's own cs_graph (neighbours of i = the columns of row i in stored order).
 * sgm_graph_bfs_order          breadth_first_search(p, g)   src/graph/permutations.f90:22-78
 *                              p(i) = visiting number from vertex 1, -1 if never reached
 * sgm_graph_greedy_coloring    greedy_coloring(colors, g)   permutations.f90:83-157
 * sgm_graph_greedy_color_order greedy_color_ordering(p, ptrs, num_colors, g)  :162-205
 *                              p(i) = new index when vertices are sorted by colour; ptrs
 *                              (num_colors+1 entries, may be NULL) = first index per colour;
 *                              fails where the reference would index out of bounds (a vertex
 *                              not reachable from vertex 1)
 * Outputs are HOST arrays of nrow int32 (1-based values, like the reference's).  The
 * breadth-first numbering runs on the device (level-synchronous, same FIFO order); so does the
 * colouring (option "coloring_pass": parities by union-find for a bipartite graph, else a level sweep
 * that reproduces the sequential pass's order- and tally-dependent choices; the reference's own
 * sequential pass on the host is the last resort and the checker) -- the same colours whichever.
 * On a CSR matrix DISTRIBUTED OVER RANKS (sgm_csr_create_dist; round 6) the three orderings and the two permutations below are
 * collective: the whole index structure is gathered onto every rank once (one grouped exchange, host memory for the length of the
 * call), the orderings run by the single-GPU passes on every rank (the same p / colours everywhere, arrays of the GLOBAL nrow),
 * left_permute cuts this rank's NEW rows out of the gathered matrix and rebuilds the handle's row block (same row partition, new halo
 * plan), right_permute renames this rank's columns without any exchange; p is the global permutation, the same on every rank.  A
 * setup path for matrices one rank can hold; in-process partitions are refused.
 * sgm_mat_left_permute(A, p)   A%left_permute(p)   cs_matrices.f90:471-478: row i -> row p(i)
 * sgm_mat_right_permute(A, p)  A%right_permute(p)  cs_matrices.f90:483-490: column j -> p(j)
 *                              (ELLPACK handles too: ellpack_matrices.f90:601-632)
 *                              device kernels; entries keep their stored order inside a row, so
 *                              row sums are bit-identical to the reference's permuted matrix.
 *                              Preconditioners set up before a permutation must be set up again. */
int sgm_graph_bfs_order(sgm_mat A, int32_t *p_out);
int sgm_graph_greedy_coloring(sgm_mat A, int32_t *colors_out, int32_t *num_colors);
int sgm_graph_greedy_color_order(sgm_mat A, int32_t *p_out, int32_t *ptrs_out, int32_t ptrs_len,
                                 int32_t *num_colors);
int sgm_mat_left_permute(sgm_mat A, const int32_t *p, int where);
int sgm_mat_right_permute(sgm_mat A, const int32_t *p, int where);
/* name of the SpMV kernel variant the matrix runs with under the current options
 * (diagnostics for benches and tests; e.g. "k_csr_sl<W=5>", "k_csr_do<256,1536,CW=1>") */
int sgm_mat_kernel(sgm_mat A, char *buf, int len);
/* bytes by construction: what the handle keeps in HBM (every layout it holds), and what ONE
 * y = A x moves with the kernel the current options select: that kernel's stored format as it
 * reads it (padded slices, codes, row pointers) + every x entry once + every y entry once.
 * bench.py grades the roofline fraction on matvec_bytes, not on the reference layout's bytes. */
int sgm_mat_footprint(sgm_mat A, int64_t *resident_bytes, int64_t *matvec_bytes);
int sgm_mat_destroy(sgm_mat A);

/* ---- vector statements inline in the solvers (SURVEY §2a "dot", "axpy family") ---- *
 * dot_product(a,b)  cg_solvers.f90:131,135,140 ; y = y + alpha*x  cg_solvers.f90:137-138 */
int sgm_dot(int64_t n, const double *a, const double *b, double *result, int where);
int sgm_axpy(int64_t n, double alpha, const double *x, double *y, int where);

/* ---- preconditioners --------------------------------------------------------------- *
 * sgm_jacobi_create <- jacobi() + jacobi_setup      src/solver/jacobi_solvers.f90:23-63
 * sgm_ildu0_create  <- ldu(incomplete,level=0) + sparse_ldu_setup
 *                      src/solver/ldu_solvers.f90:73-130 (pattern :397-440 and factorization
 *                      :275-387 run on the device -- the reference's statements per row, rows
 *                      of one dependency level side by side --; the factors live in HBM, host
 *                      copies of their values are made for sgm_pc_get only);
 *                      on a row-partitioned matrix: ILDU(0) of each part's diagonal block
 *                      (block-Jacobi, no exchange in the apply; iteration counts differ
 *                      from the one-part factorisation)
 * sgm_pc_create     <- jacobi() / ldu(...) as factories: the object before it has seen a matrix (set options, then setup)
 * sgm_pc_setup      <- pc%setup(A): the first one, or again after the values changed
 * sgm_pc_apply      <- pc%solve(A, z, r): jacobi_solve :68-81 / ldu_solve :160-176
 * sgm_pc_get        <- read back idiag / L,D,U for parity checks ("idiag","Lptr","Lnode",
 *                      "Lval","Uptr","Unode","Uval","D"; 1-based like the reference; with "ildu_reorder" they are
 *                      the factors of P A P^T and "perm" is p, row i of A = row p(i) of that matrix)          */
int sgm_pc_create(sgm_pc *out, int32_t kind /* SGM_PC_JACOBI | SGM_PC_ILDU0 */);   /* the factory alone: no matrix seen yet */
int sgm_jacobi_create(sgm_pc *out, sgm_mat A);
int sgm_ildu0_create(sgm_pc *out, sgm_mat A);
int sgm_pc_setup(sgm_pc pc, sgm_mat A);
int sgm_pc_apply(sgm_pc pc, const double *r, double *z, int where);
int sgm_pc_get(sgm_pc pc, const char *name, void *out_host, size_t bytes, size_t *needed);
/* sgm_pc_info: which sweeps serve part `part` (0 on one GPU) of a preconditioner that has been set up, and what an apply
 * costs -- the reference's `ldu()` (ldu_solvers.f90:73-86) factors A in the order it is given, and on a naturally ordered grid
 * the triangular solves of ldu_solve (:160-176, row recurrences :227-236, :254-263) are a dependency chain of nx + ny levels:
 * correct, the reference's semantics, and 50-100 x below the HBM roofline.  This says so BEFORE the solve:
 *   out4[0], out4[1]  dependency levels of L, of U        out4[2]  path: 0 diagonal scaling, 1 row-space sweeps (bandwidth-
 *   bound), 2 strip pipeline, 3 slab pipeline, 4 level walkers        out4[3]  colours of the ordering (0 = A's own order)
 *   est_us  estimated microseconds per apply               path_name  e.g. "strip pipeline, 6323 levels", "row space, 2 levels";
 *   with a colour ordering, followed by "; product of the ordered part: k_csr_sl<W=5>" -- the SpMV kernel of the permuted copy
 * With SGM_TRACE set every sgm_pc_setup prints the same line on stderr.  The remedy for a chain: ldu(reorder = "colour")
 * (option "ildu_reorder"), INTEGRATION.md. */
int sgm_pc_info(sgm_pc pc, int32_t part, int32_t *out4, double *est_us, char *path_name, int len);
int sgm_pc_destroy(sgm_pc pc);

/* ---- solvers ------------------------------------------------------------------------ *
 * sgm_cg_create       <- cg(tolerance)            src/solver/cg_solvers.f90:36-47
 * sgm_bicgstab_create <- bicgstab(tolerance)      src/solver/bicgstab_solvers.f90:37-48
 * sgm_gmres_create    <- NO reference counterpart; GMRES(restart), Arnoldi by low-synchronisation
 *                        classical Gram-Schmidt with re-orthogonalisation (CGS-2: the basis read twice, two
 *                        reductions per step; option "gmres_cgs2" = 0: modified Gram-Schmidt), Givens
 *                        rotations; same tolerance / iterations conventions as cg
 * sgm_solver_setup    <- solver%setup(A): square check, work vectors, iterations = 0
 *                        cg_solvers.f90:52-90
 * sgm_solver_solve    <- solver%solve(A,x,b[,pc]): cg_solve :116-150, cg_solve_pc :155-194,
 *                        bicgstab_solve :124-177, bicgstab_solve_pc :182-237.
 *                        ABSOLUTE tolerance on sqrt(res2), initial guess from x, no
 *                        iteration cap unless sgm_solver_set_max_iter (an extension) is
 *                        called; `iterations` accumulates across solves like the reference.
 *                        The whole loop is device-resident: x and b cross the boundary once.
 *                        A breakdown (NaN res2) ends the loop as it ends the reference's; `converged` is then 0
 *                        (and the call returns SGM_ERR_NOT_CONVERGED when an iteration cap is set).
 * sgm_solver_set_tolerance <- solver%set_params(tolerance) cg_solvers.f90:95-111, bicgstab_solvers.f90:103-119, and any
 *                        later edit of the public field solver%tolerance, which the reference's loops read at every solve
 *                        (cg_solvers.f90:133,175): the next solve on this handle stops at the new ABSOLUTE tolerance;
 *                        work vectors, options and the accumulated `iterations` are kept.  Both Fortran layers and
 *                        sigma_amd push the host object's tolerance in front of every solve.
 * sgm_solver_destroy  <- solver%destroy()          cg_solvers.f90:199-212                 */
int sgm_cg_create(sgm_solver *out, double tolerance);
int sgm_bicgstab_create(sgm_solver *out, double tolerance);
int sgm_gmres_create(sgm_solver *out, double tolerance, int32_t restart);
int sgm_solver_setup(sgm_solver s, sgm_mat A);
int sgm_solver_set_max_iter(sgm_solver s, int64_t max_iter /* <= 0: unbounded */);
int sgm_solver_set_tolerance(sgm_solver s, double tolerance);
int sgm_solver_set_history(sgm_solver s, int64_t capacity); /* record res2 per iteration */
int sgm_solver_solve(sgm_solver s, sgm_mat A, double *x, const double *b, sgm_pc pc_or_null,
                     int where);
int sgm_solver_info(sgm_solver s, int64_t *iterations, double *res2, int32_t *converged,
                    int64_t *last_solve_iterations);
int sgm_solver_get_history(sgm_solver s, double *out_host, int64_t capacity, int64_t *count);
int sgm_solver_destroy(sgm_solver s);

/* ---- Lanczos ------------------------------------------------------------------------- *
 * sgm_lanczos <- lanczos(A, T, Q)  src/eigensolver.f90:27-90 (what eigensolve :160-208 feeds to
 * LAPACK dstev): nsteps Lanczos steps with full re-orthogonalisation.  T_host: 3 x nsteps,
 * column-major (T(2,:) diagonal, T(1,:) = T(3,:) off-diagonal); Q_out (optional): n x nsteps
 * column-major Lanczos vectors.  q1 is the start vector (the reference draws it from a
 * time-seeded RNG); it is normalised inside.  A may be row-partitioned: an in-process partition takes and
 * returns global vectors, a rank of a matrix distributed over processes its owned slice of q1 and Q (n_local x nsteps);
 * the dots are all-reduced, T is the same on every rank.                                    */
int sgm_lanczos(sgm_mat A, int32_t nsteps, const double *q1, double *T_host, double *Q_out, int where);

/* sgm_generalized_lanczos <- generalized_lanczos(A, B, T, Q)  src/eigensolver.f90:95-155: Lanczos for
 * A x = lambda B x.  Every step solves B w = v with `solver_for_B` (set up for B by the caller, like
 * B%set_solver; optional preconditioner), started from w = A q_i as the reference's
 * `call B%solve(w, v)` does (:140).  No re-orthogonalisation (the reference has none here); q1 is
 * normalised in the B-norm (:123-124).  T_host / Q_out as in sgm_lanczos; A and B may be row-partitioned (the same way). */
int sgm_generalized_lanczos(sgm_mat A, sgm_mat B, sgm_solver solver_for_B, sgm_pc pc_or_null, int32_t nsteps,
                            const double *q1, double *T_host, double *Q_out, int where);

/* ---- row-partitioned multi-GPU (nothing in the reference) ---------------- *
 * One process per GPU.  Rank r owns the contiguous global rows
 * [row_starts[r], row_starts[r+1]) of A and the same slice of every vector.
 * sgm_comm_unique_id / sgm_comm_init: RCCL bootstrap (the 128-byte id is broadcast by the
 * host, e.g. over torch.distributed/gloo or MPI).
 * sgm_csr_create_dist: this rank's rows with GLOBAL 1-based column ids; builds the sorted
 * unique halo list, renumbers columns to [owned | halo] and exchanges the send lists.
 * A distributed matrix reads x of length x_len = owned + halo (sgm_mat_info): the owned
 * part is the caller's, the halo part is filled by the library (ncclSend/ncclRecv with
 * the neighbour ranks) in every matvec.  Dots inside the solvers become
 * ncclAllReduce(sum, fp64); row sums stay bit-identical to the 1-GPU result.
 * sgm_csr_create_partitioned: the same partition / halo / reduction machinery with all P
 * row blocks inside ONE process on ONE GPU (exchanges are device gathers): x, y, b are
 * plain global-length vectors.  It exists so that the multi-GPU logic is exercised by
 * `pytest -m gpu` on a single-GPU box.
 * sgm_halo_plan_host: the host-only index work of the two calls above (no HIP call), so it
 * can be checked bit-for-bit without a GPU: sorted unique halo list (global, 1-based) and
 * the renumbered node array (1-based: 1..n_own owned, n_own+1.. halo, in halo-list order). */
int sgm_comm_unique_id(void *id128);
int sgm_comm_init(sgm_comm *out, int rank, int nranks, const void *id128);
int sgm_comm_destroy(sgm_comm c);
/* sgm_comm_attach_halo_comm: a SECOND communicator (its own 128-byte id, broadcast like the first) for the halo send / recv
 * pairs, so that they do not share a queue with the dots' all-reduces -- an A/B switch for the overlap of halo traffic
 * with interior rows on a multi-GPU node.  Optional; without it one communicator carries both.
 * sgm_dist_profile(on) / sgm_dist_profile_read: HIP-event timers around the phases of the row-partitioned path, summed
 * since the last read: ms_out[6] / count_out[6] = { halo gather + send/recv on the communication stream (post -> done),
 * interior-rows kernel(s), what the launch stream then still waits for the halo, boundary-rows kernel(s),
 * per-dot partial -> slot reduction kernels, all-reduces }.  Reading synchronises the library's streams.           */
int sgm_comm_attach_halo_comm(sgm_comm c, const void *id128);
/* sgm_comm_group_selftest: one ncclGroup holding a send / recv pair (rank to itself) and an in-place all-reduce of one double
 * -- the group CG posts per iteration with "dist_halo_fused" = 1 -- on this communicator's transport.  out3 = { the double
 * the pair delivered (42 + rank), the all-reduced 1.0 (= nranks), microseconds of the second such group }.  No reference
 * counterpart (the reference has no communication); a probe that lets a one-GPU box show that the real
 * librccl accepts the mixed group. */
int sgm_comm_group_selftest(sgm_comm c, double *out3);
/* sgm_comm_group_ok: what sgm_comm_init's own probe found (1 / 0).  With more than one rank sgm_comm_init posts, once and
 * collectively, the group CG depends on with "dist_halo_fused" = 1 -- one double to the right neighbour, one from the left and
 * an all-reduce in ONE ncclGroup -- and the ranks agree on the outcome with a plain all-reduce.  0 = some rank's transport
 * refused the group or delivered wrong values: every solve on this communicator then posts the pairs and the all-reduce one
 * after the other (the order "dist_halo_fused" = 2 uses) instead of failing in its first iteration.  No reference counterpart (the reference has no communication). */
int sgm_comm_group_ok(sgm_comm c);
int sgm_dist_profile(int on);
int sgm_dist_profile_read(double *ms_out /* 6 */, int64_t *count_out /* 6 */);
int sgm_csr_create_dist(sgm_mat *out, sgm_comm comm, const int64_t *row_starts /* nranks+1, 0-based */,
                        int64_t nnz_local, const int32_t *ptr_1based_local,
                        const int32_t *node_1based_global, const double *val, int where);
/* sgm_csr_create_dist_rect: rows partitioned by row_starts, x (the columns) by col_starts -- an off-diagonal block
 * of a composite (sparse_matrix_composites.f90:41-162) whose block rows / columns have partitions of their own.
 * matvec reads x as [this rank's col_starts slice | halo].  A composite of such leaves (every leaf on the same
 * communicator, block row i partitioned like block column i) is itself a distributed operator: its vectors are
 * the concatenation of this rank's slices of the block vectors.                                               */
int sgm_csr_create_dist_rect(sgm_mat *out, sgm_comm comm, const int64_t *row_starts, const int64_t *col_starts,
                             int64_t nnz_local, const int32_t *ptr_1based_local,
                             const int32_t *node_1based_global, const double *val, int where);
/* sgm_ell_create_dist: this rank's rows of an ELLPACK matrix (node / val as (max_d, n_local) column-major,
 * GLOBAL 1-based columns, padding as the reference keeps it).  Held as fixed-length CSR rows whose padding
 * slots are stored entries, so the row sums equal ellpack_matvec_add's (ellpack_matrices.f90:640-665). */
int sgm_ell_create_dist(sgm_mat *out, sgm_comm comm, const int64_t *row_starts, int32_t max_d,
                        const int32_t *node_1based_global_colmajor, const double *val_colmajor, int where);
int sgm_csr_create_partitioned(sgm_mat *out, int32_t nparts, const int64_t *row_starts /* nparts+1 */,
                               int32_t nrow, int32_t ncol, int64_t nnz,
                               const int32_t *ptr_1based, const int32_t *node_1based,
                               const double *val /* host arrays */);
/* sgm_csr_create_partitioned_parts: the same in-process partition handed over PART BY PART: for every part its rows as
 * sgm_csr_create_dist takes a rank's (local 1-based row pointers, GLOBAL 1-based columns, values), all host or all device
 * arrays -- for matrices too large to assemble whole on the host (the 7-point 464^3 grid of BASELINE config 5: 8.8 GB).  Same
 * planners and parts as sgm_csr_create_partitioned; no reference counterpart (the reference has one address space:
 * cs_matrices.f90:32-107). */
int sgm_csr_create_partitioned_parts(sgm_mat *out, int32_t nparts, const int64_t *row_starts /* nparts+1 */,
                                     const int64_t *nnz_of_part, const int32_t *const *ptr_1based_local_of_part,
                                     const int32_t *const *node_1based_global_of_part, const double *const *val_of_part,
                                     int where);
int sgm_halo_plan_host(int32_t n_own, int64_t col_begin /* first owned global column, 0-based */,
                       int64_t nnz, const int32_t *node_1based_global,
                       int32_t *node_1based_local_out, int32_t *halo_cols_out /* capacity nnz */,
                       int32_t *n_halo_out);
/* The rest of the host-only planning that sgm_csr_create_dist / _partitioned run (no HIP call; the
 * world_size-2 gloo test drives exactly these, exchanging the lists over gloo instead of RCCL):
 * sgm_dist_plan_host       one rank's requests: want[q] = entries of its (sorted) halo list that
 *                          rank q owns, want_off = their prefix sum (nranks+1: the run of halo
 *                          entries owned by q), req[t] = index of halo entry t in ITS OWNER's local
 *                          numbering (0-based) -- the list that owner gathers from when it sends.
 * sgm_dist_neighbors_host  one rank's neighbour table from the all-gathered want matrix
 *                          (want_all[q*nranks + r] = rank q's want[r]): per neighbour its rank,
 *                          how many entries go to it / come from it, and where the received run
 *                          starts inside this rank's halo region.  Arrays hold up to nranks-1.
 * sgm_partition_links_host every (sender, receiver) link of an in-process partition, as
 *                          sgm_csr_create_partitioned builds them: first call with sender = NULL
 *                          for the sizes (n_links, idx_needed), then with arrays; idx_concat holds
 *                          the senders' gather lists end to end (0-based local indices).
 * sgm_partition_rows_by_nnz contiguous row blocks balanced by the bytes of B_csr they carry
 *                          (12 per stored entry + 20 per row, SURVEY 8d/8e "balanced by nnz"),
 *                          boundaries rounded to multiples of `align` rows (even; 512 keeps the
 *                          sliced kernel's slices whole).  row_starts_out: nparts+1 entries.
 * sgm_mat_halo_nbr         reads the exchange plan of a built matrix back (k < 0: only n_nbrs). */
int sgm_dist_plan_host(int32_t rank, int32_t nranks, const int64_t *row_starts, int32_t n_halo,
                       const int32_t *halo_cols_1based, int32_t *want, int32_t *want_off, int32_t *req);
int sgm_dist_neighbors_host(int32_t rank, int32_t nranks, const int32_t *want_all, int32_t *peer,
                            int32_t *send_count, int32_t *recv_count, int32_t *recv_offset, int32_t *n_nbrs);
int sgm_partition_links_host(int32_t nparts, const int64_t *row_starts, const int32_t *ptr_1based,
                             const int32_t *node_1based, int32_t *n_links, int32_t *sender, int32_t *receiver,
                             int32_t *recv_offset, int32_t *count, int32_t *idx_concat, int64_t idx_capacity,
                             int64_t *idx_needed);
int sgm_partition_rows_by_nnz(int32_t nrow, const int32_t *ptr_1based, int32_t nparts, int32_t align,
                              int64_t *row_starts_out);
/* Two more pieces of host-only index work, exported like the planners above so that the CPU suite (and the sanitizer build,
 * tools/asan) drives the statements the library runs:
 * sgm_ell_degrees_host        degrees(n) of an ELLPACK graph recovered from the padding the reference keeps (ellpack_graphs.f90:164,
 *                             :394-397: the rest of a row repeats the neighbour just added; a row of zeros is empty) -- what
 *                             sgm_ell_create / sgm_ell_create_dist do, since the product's interface carries no degrees
 * sgm_left_permute_rows_host  rows [r0, r1) (0-based) of A%left_permute(p) (cs_matrices.f90:471-478) cut out of the whole matrix:
 *                             what a rank keeps of a permuted matrix distributed over ranks.  lnode == NULL: sizing call. */
int sgm_ell_degrees_host(int32_t n, int32_t max_d, const int32_t *node_1based_colmajor, int32_t *degrees_out);
int sgm_left_permute_rows_host(int32_t n, const int32_t *p_1based, const int32_t *ptr_1based, const int32_t *node_1based,
                               const double *val, int64_t r0, int64_t r1, int32_t *lptr_out, int32_t *lnode_out,
                               double *lval_out, int64_t capacity, int64_t *needed);
int sgm_mat_halo_nbr(sgm_mat A, int32_t part, int32_t k, int32_t *n_nbrs, int32_t *peer, int32_t *send_count,
                     int32_t *recv_count, int32_t *recv_offset, int32_t *send_idx_host, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* SIGMA_HIP_H */
