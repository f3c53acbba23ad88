/*
 * sigma_oracle.c -- CPU restatement of the SiGMA hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP path.  It is never shipped, never
 * linked into libsigma_hip.so, and the product never falls back to it.  Only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg load it
 * (as the checker / the reported CPU baseline).
 *
 * Parity status: PINNED.  Every function below is checked in
 * tests/test_oracle_golden.py against fixtures (tests/golden/*.npz) produced by
 * the REAL reference compiled in place with amdflang (oracle/build_ref.sh +
 * oracle/ref_driver.f90 + oracle/make_golden.py), and against the known answers
 * of the reference's own tests (diffusion_1d: 64 CG iterations, error 0;
 * advection_diffusion_1d: BiCGStab error 4.91e-9).
 * Exception: orc_gmres has NO reference counterpart (the reference
 * has no GMRES) -- "parity unpinned by the reference"; it is pinned only by the
 * analytic solution and by agreement with the BiCGStab oracle.
 *
 * Each function cites the reference file:line (under /root/reference) it follows.
 * Index arrays are 1-based int32 exactly as the Fortran holds them.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction, no
 * reassociation): the reference built for x86-64 without -march has neither.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* Graph / index work (bit-exact)                                           */
/* ------------------------------------------------------------------------ */

/* ll_graph%add_edge (src/graph/formats/ll_graphs.f90:355-370: push unless already
 * connected) followed by cs_graph_build (src/graph/formats/cs_graphs.f90:109-197:
 * count -> prefix sum with ptr(1)=1 -> first-free-slot fill that skips repeats
 * :171-181 -> prune_null_edges when slots stay empty -> max_d).
 * The ll_graph cursor hands the edges over row by row in push order, so the
 * stored order inside a row is first-insertion order.
 * ptr: n+1, node: capacity ne.  Returns the number of stored edges.            */
ORC_API int64_t orc_cs_graph_build(int32_t n, int64_t ne, const int32_t *ei,
                                   const int32_t *ej, int32_t *ptr, int32_t *node,
                                   int32_t *max_d)
{
    int64_t *cnt = calloc((size_t)n + 2, sizeof(int64_t));
    int64_t *start = malloc(((size_t)n + 2) * sizeof(int64_t));
    for (int64_t k = 0; k < ne; k++) cnt[ei[k]]++;
    start[1] = 0;
    for (int32_t i = 1; i <= n; i++) start[i + 1] = start[i] + cnt[i];
    int32_t *tmp = calloc((size_t)(ne > 0 ? ne : 1), sizeof(int32_t));
    /* first-free-slot fill with the duplicate provision (cs_graphs.f90:171-181) */
    for (int64_t k = 0; k < ne; k++) {
        int32_t i = ei[k], j = ej[k];
        for (int64_t l = start[i]; l < start[i + 1]; l++) {
            if (tmp[l] == j) break;
            if (tmp[l] == 0) { tmp[l] = j; break; }
        }
    }
    /* prune_null_edges: compact away slots still 0 */
    int64_t out = 0;
    int32_t md = 0;
    ptr[0] = 1;
    for (int32_t i = 1; i <= n; i++) {
        int32_t d = 0;
        for (int64_t l = start[i]; l < start[i + 1]; l++)
            if (tmp[l] != 0) { node[out++] = tmp[l]; d++; }
        ptr[i] = (int32_t)(out + 1);
        if (d > md) md = d;
    }
    *max_d = md;
    free(cnt); free(start); free(tmp);
    return out;
}

/* max degree after ll_graph de-duplication -- sizes the ELLPACK arrays. */
ORC_API int32_t orc_max_degree(int32_t n, int64_t ne, const int32_t *ei, const int32_t *ej)
{
    int32_t *ptr = malloc(((size_t)n + 1) * sizeof(int32_t));
    int32_t *node = malloc((size_t)(ne > 0 ? ne : 1) * sizeof(int32_t));
    int32_t md = 0;
    orc_cs_graph_build(n, ne, ei, ej, ptr, node, &md);
    free(ptr); free(node);
    return md;
}

/* ellpack_graph_build (src/graph/formats/ellpack_graphs.f90:105-170): degrees,
 * max_d = maxval(degrees), node(max_d,n)=0, then for every edge not yet connected
 * `node(d+1:, i) = j` -- the rest of the row is filled with the newest neighbour, so
 * padding slots repeat the LAST real neighbour (:164, add_edge :394-397).
 * node is column-major (max_d, n): slot k of row i at node[(i-1)*max_d + (k-1)].   */
ORC_API void orc_ellpack_graph_build(int32_t n, int64_t ne, const int32_t *ei,
                                     const int32_t *ej, int32_t max_d, int32_t *node,
                                     int32_t *degrees)
{
    memset(node, 0, (size_t)n * max_d * sizeof(int32_t));
    memset(degrees, 0, (size_t)n * sizeof(int32_t));
    for (int64_t k = 0; k < ne; k++) {
        int32_t i = ei[k], j = ej[k];
        int32_t *row = node + (size_t)(i - 1) * max_d;
        int32_t d = degrees[i - 1];
        int connected = 0;
        for (int32_t l = 0; l < d; l++) if (row[l] == j) { connected = 1; break; }
        if (!connected) {
            for (int32_t l = d; l < max_d; l++) row[l] = j;
            degrees[i - 1] = d + 1;
        }
    }
}

/* csr_matrix_set_value (src/matrix/formats/cs_matrices.f90:840-863): scan the row,
 * overwrite every slot whose node equals j.  A%zero() first (cs_matrices.f90:448). */
ORC_API void orc_csr_set_values(int32_t n, const int32_t *ptr, const int32_t *node,
                                double *val, int64_t ne, const int32_t *ei,
                                const int32_t *ej, const double *ev)
{
    memset(val, 0, (size_t)(ptr[n] - 1) * sizeof(double));
    for (int64_t e = 0; e < ne; e++) {
        int32_t i = ei[e];
        for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++)
            if (node[k - 1] == ej[e]) val[k - 1] = ev[e];
    }
}

/* ellpack_matrix_set_value (src/matrix/formats/ellpack_matrices.f90:444-466): only
 * the first degrees(i) slots are searched; padding slots keep val = 0.          */
ORC_API void orc_ell_set_values(int32_t n, int32_t max_d, const int32_t *node,
                                const int32_t *degrees, double *val, int64_t ne,
                                const int32_t *ei, const int32_t *ej, const double *ev)
{
    memset(val, 0, (size_t)n * max_d * sizeof(double));
    for (int64_t e = 0; e < ne; e++) {
        int32_t i = ei[e];
        const int32_t *row = node + (size_t)(i - 1) * max_d;
        double *vrow = val + (size_t)(i - 1) * max_d;
        for (int32_t k = 0; k < degrees[i - 1]; k++)
            if (row[k] == ej[e]) vrow[k] = ev[e];
    }
}

/* ------------------------------------------------------------------------ */
/* Matrix-vector products                                                   */
/* ------------------------------------------------------------------------ */

/* csr_matvec_add  (src/matrix/formats/cs_matrices.f90:600-622) */
ORC_API void orc_csr_matvec_add(int32_t n, const int32_t *ptr, const int32_t *node,
                                const double *val, const double *x, double *y)
{
    for (int32_t i = 1; i <= n; i++) {
        double z = 0.0;
        for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++) {
            int32_t j = node[k - 1];
            z = z + val[k - 1] * x[j - 1];
        }
        y[i - 1] = y[i - 1] + z;
    }
}

/* ellpack_matvec_add (src/matrix/formats/ellpack_matrices.f90:640-665): ALL max_d
 * slots, padding included. */
ORC_API void orc_ell_matvec_add(int32_t n, int32_t max_d, const int32_t *node,
                                const double *val, const double *x, double *y)
{
    for (int32_t i = 1; i <= n; i++) {
        double z = 0.0;
        const int32_t *row = node + (size_t)(i - 1) * max_d;
        const double *vrow = val + (size_t)(i - 1) * max_d;
        for (int32_t k = 0; k < max_d; k++) {
            int32_t j = row[k];
            z = z + vrow[k] * x[j - 1];
        }
        y[i - 1] = y[i - 1] + z;
    }
}

/* csc_matvec_add applied to a csr_matrix = the transpose product
 * (src/matrix/formats/cs_matrices.f90:627-647, bound as csr matvec_t_add_impl):
 * for j: z = x(j); for k in row j: y(node(k)) = y(node(k)) + val(k)*z */
ORC_API void orc_csr_matvec_t_add(int32_t n, const int32_t *ptr, const int32_t *node,
                                  const double *val, const double *x, double *y)
{
    for (int32_t j = 1; j <= n; j++) {
        double z = x[j - 1];
        for (int32_t k = ptr[j - 1]; k <= ptr[j] - 1; k++) {
            int32_t i = node[k - 1];
            y[i - 1] = y[i - 1] + val[k - 1] * z;
        }
    }
}

/* ellpack_matvec_t_add (src/matrix/formats/ellpack_matrices.f90:670-693): all max_d slots */
ORC_API void orc_ell_matvec_t_add(int32_t n, int32_t max_d, const int32_t *node,
                                  const double *val, const double *x, double *y)
{
    for (int32_t j = 1; j <= n; j++) {
        double z = x[j - 1];
        const int32_t *row = node + (size_t)(j - 1) * max_d;
        const double *vrow = val + (size_t)(j - 1) * max_d;
        for (int32_t k = 0; k < max_d; k++) {
            int32_t i = row[k];
            y[i - 1] = y[i - 1] + vrow[k] * z;
        }
    }
}

/* A generic operator handle for the solvers: fmt 1 = CSR, 2 = ELLPACK. */
typedef struct {
    int32_t fmt, n, max_d;
    const int32_t *ptr, *node;
    const double *val;
} orc_op;

/* linear_operator_matvec (src/linear_operator/linear_operator_interface.f90:185-194):
 * y = 0 ; call matvec_add(x, y) */
static void op_matvec(const orc_op *A, const double *x, double *y)
{
    for (int32_t i = 0; i < A->n; i++) y[i] = 0.0;
    if (A->fmt == 1) orc_csr_matvec_add(A->n, A->ptr, A->node, A->val, x, y);
    else orc_ell_matvec_add(A->n, A->max_d, A->node, A->val, x, y);
}

ORC_API void orc_matvec(int32_t fmt, int32_t n, int32_t max_d, const int32_t *ptr,
                        const int32_t *node, const double *val, const double *x, double *y)
{
    orc_op A = {fmt, n, max_d, ptr, node, val};
    op_matvec(&A, x, y);
}

/* get_value(i,i) through the row scan (cs_matrices.f90:709-724; the ELLPACK version
 * scans the first degrees(i) slots -- padding repeats a real neighbour whose val is 0
 * only in padding, so scanning all slots but keeping the LAST hit would be wrong;
 * we scan real slots only). */
static double op_get_value(const orc_op *A, const int32_t *degrees, int32_t i, int32_t j)
{
    double z = 0.0;
    if (A->fmt == 1) {
        for (int32_t k = A->ptr[i - 1]; k <= A->ptr[i] - 1; k++)
            if (A->node[k - 1] == j) z = A->val[k - 1];
    } else {
        const int32_t *row = A->node + (size_t)(i - 1) * A->max_d;
        const double *vrow = A->val + (size_t)(i - 1) * A->max_d;
        int32_t d = degrees ? degrees[i - 1] : A->max_d;
        for (int32_t k = 0; k < d; k++) if (row[k] == j) z = vrow[k];
    }
    return z;
}

/* ------------------------------------------------------------------------ */
/* dot / axpy statements (inline in the reference solvers)       */
/* ------------------------------------------------------------------------ */
/* Fortran intrinsic dot_product: the summation order is the COMPILER'S choice.  Mode 0 is
 * the plain left-to-right sum; mode 1 is the order a 4-lane vectorising compiler produces
 * (four interleaved partial sums, combined at the end).  Both are valid restatements of
 * `dot_product`; tests use the gap between them to calibrate how far two correct
 * implementations of the same recurrence may drift apart. */
static int g_dot_mode = 0;
ORC_API void orc_set_dot_mode(int mode) { g_dot_mode = mode; }
static double dot(int32_t n, const double *a, const double *b)
{
    if (g_dot_mode == 1) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int32_t i = 0;
        for (; i + 4 <= n; i += 4) {
            s0 = s0 + a[i] * b[i];
            s1 = s1 + a[i + 1] * b[i + 1];
            s2 = s2 + a[i + 2] * b[i + 2];
            s3 = s3 + a[i + 3] * b[i + 3];
        }
        double s = (s0 + s1) + (s2 + s3);
        for (; i < n; i++) s = s + a[i] * b[i];
        return s;
    }
    double s = 0.0;
    for (int32_t i = 0; i < n; i++) s = s + a[i] * b[i];
    return s;
}
ORC_API double orc_dot(int32_t n, const double *a, const double *b) { return dot(n, a, b); }

/* ------------------------------------------------------------------------ */
/* Preconditioners                                                          */
/* ------------------------------------------------------------------------ */
typedef struct {
    int32_t kind;          /* 0 none, 1 jacobi, 2 ildu */
    int32_t n;
    double *idiag;         /* jacobi */
    int32_t *Lptr, *Lnode, *Uptr, *Unode;
    double *Lval, *Uval, *D;
} orc_pc;

/* jacobi_setup (src/solver/jacobi_solvers.f90:37-63): idiag(i) = 1/A%get_value(i,i) */
ORC_API void orc_jacobi_setup(int32_t fmt, int32_t n, int32_t max_d, const int32_t *ptr,
                              const int32_t *node, const double *val,
                              const int32_t *degrees, double *idiag)
{
    orc_op A = {fmt, n, max_d, ptr, node, val};
    for (int32_t i = 1; i <= n; i++) idiag[i - 1] = 1.0 / op_get_value(&A, degrees, i, i);
}

/* jacobi_solve (src/solver/jacobi_solvers.f90:68-81): x = idiag * b */
ORC_API void orc_jacobi_solve(int32_t n, const double *idiag, double *x, const double *b)
{
    for (int32_t i = 0; i < n; i++) x[i] = idiag[i] * b[i];
}

/* lower_triangular_solve / upper_triangular_solve (src/solver/ldu_solvers.f90:208-265) */
static void lower_solve(int32_t n, const int32_t *ptr, const int32_t *node,
                        const double *val, double *x)
{
    for (int32_t i = 1; i <= n; i++) {
        double z = x[i - 1];
        for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++)
            z = z - val[k - 1] * x[node[k - 1] - 1];
        x[i - 1] = z;
    }
}
static void upper_solve(int32_t n, const int32_t *ptr, const int32_t *node,
                        const double *val, double *x)
{
    for (int32_t i = n; i >= 1; i--) {
        double z = x[i - 1];
        for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++)
            z = z - val[k - 1] * x[node[k - 1] - 1];
        x[i - 1] = z;
    }
}

/* ldu_solve (src/solver/ldu_solvers.f90:160-176): x=b; (I+L)^-1; x=x/D; (I+U)^-1 */
ORC_API void orc_ldu_solve(int32_t n, const int32_t *Lptr, const int32_t *Lnode,
                           const double *Lval, const double *D, const int32_t *Uptr,
                           const int32_t *Unode, const double *Uval, double *x,
                           const double *b)
{
    for (int32_t i = 0; i < n; i++) x[i] = b[i];
    lower_solve(n, Lptr, Lnode, Lval, x);
    for (int32_t i = 0; i < n; i++) x[i] = x[i] / D[i];
    upper_solve(n, Uptr, Unode, Uval, x);
}

/* incomplete_ldu_sparsity_pattern (src/solver/ldu_solvers.f90:397-440), level 0:
 * walk A's entries in cursor order (row by row, stored order); i>j goes to gl, j>i
 * to gu (ll_graphs, push order) -> csr L, U.  Counts first, then fill.
 * Lptr/Uptr: n+1.  Returns nnz(L) in *nl and nnz(U) in *nu; call with Lnode==NULL
 * to size. */
ORC_API void orc_ildu_pattern(int32_t n, const int32_t *ptr, const int32_t *node,
                              int32_t *Lptr, int32_t *Lnode, int32_t *Uptr,
                              int32_t *Unode, int64_t *nl, int64_t *nu)
{
    int64_t l = 0, u = 0;
    Lptr[0] = 1; Uptr[0] = 1;
    for (int32_t i = 1; i <= n; i++) {
        for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++) {
            int32_t j = node[k - 1];
            if (i > j) { if (Lnode) Lnode[l] = j; l++; }
            if (j > i) { if (Unode) Unode[u] = j; u++; }
        }
        Lptr[i] = (int32_t)(l + 1);
        Uptr[i] = (int32_t)(u + 1);
    }
    *nl = l; *nu = u;
}

static double csr_get(const int32_t *ptr, const int32_t *node, const double *val,
                      int32_t i, int32_t j)
{
    double z = 0.0;
    for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++) if (node[k - 1] == j) z = val[k - 1];
    return z;
}
static void csr_set(const int32_t *ptr, const int32_t *node, double *val, int32_t i,
                    int32_t j, double z)
{
    for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++) if (node[k - 1] == j) val[k - 1] = z;
}
static void csr_add(const int32_t *ptr, const int32_t *node, double *val, int32_t i,
                    int32_t j, double z)
{
    for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++)
        if (node[k - 1] == j) val[k - 1] = val[k - 1] + z;
}

/* sparse_static_pattern_ldu_factorization (src/solver/ldu_solvers.f90:275-387).
 * Statement-for-statement, including the evaluation order `-Lik * D(k) * Ukj`
 * = ((-Lik)*D(k))*Ukj and the get/set/add_value row scans.
 * NOTE (reference behaviour, kept): U%add_value(i,j,..) / L%add_value(i,j,..) are only
 * ever called for (i,j) inside the static pattern, so no reallocation happens.  */
ORC_API void orc_ildu_factor(int32_t n, const int32_t *ptr, const int32_t *node,
                             const double *val, const int32_t *Lptr,
                             const int32_t *Lnode, double *Lval, const int32_t *Uptr,
                             const int32_t *Unode, double *Uval, double *D)
{
    memset(Lval, 0, (size_t)(Lptr[n] - 1) * sizeof(double));
    memset(Uval, 0, (size_t)(Uptr[n] - 1) * sizeof(double));
    for (int32_t i = 0; i < n; i++) D[i] = 0.0;
    /* copy A into L, D, U (:309-326) */
    for (int32_t i = 1; i <= n; i++)
        for (int32_t k = ptr[i - 1]; k <= ptr[i] - 1; k++) {
            int32_t j = node[k - 1];
            if (i > j) csr_set(Lptr, Lnode, Lval, i, j, val[k - 1]);
            else if (j > i) csr_set(Uptr, Unode, Uval, i, j, val[k - 1]);
            else D[i - 1] = val[k - 1];
        }
    for (int32_t i = 1; i <= n; i++) {
        int32_t dl = Lptr[i] - Lptr[i - 1], du = Uptr[i] - Uptr[i - 1];
        const int32_t *ln = Lnode + (Lptr[i - 1] - 1), *un = Unode + (Uptr[i - 1] - 1);
        for (int32_t ind1 = 0; ind1 < dl; ind1++) {
            int32_t k = ln[ind1];
            double Lik = csr_get(Lptr, Lnode, Lval, i, k);
            double Uki = csr_get(Uptr, Unode, Uval, k, i);
            csr_set(Lptr, Lnode, Lval, i, k, Lik / D[k - 1]);
            Lik = Lik / D[k - 1];
            for (int32_t ind2 = 0; ind2 < dl; ind2++) {
                int32_t j = ln[ind2];
                if (j > k) {
                    double Ukj = csr_get(Uptr, Unode, Uval, k, j);
                    csr_add(Lptr, Lnode, Lval, i, j, -Lik * D[k - 1] * Ukj);
                }
            }
            D[i - 1] = D[i - 1] - Lik * D[k - 1] * Uki;
            for (int32_t ind2 = 0; ind2 < du; ind2++) {
                int32_t j = un[ind2];
                double Ukj = csr_get(Uptr, Unode, Uval, k, j);
                csr_add(Uptr, Unode, Uval, i, j, -Lik * D[k - 1] * Ukj);
            }
        }
        for (int32_t ind2 = 0; ind2 < du; ind2++) {
            int32_t k = un[ind2];
            double Uik = csr_get(Uptr, Unode, Uval, i, k);
            csr_set(Uptr, Unode, Uval, i, k, Uik / D[i - 1]);
        }
    }
}

static void pc_solve(const orc_pc *pc, double *x, const double *b)
{
    if (pc->kind == 1) orc_jacobi_solve(pc->n, pc->idiag, x, b);
    else orc_ldu_solve(pc->n, pc->Lptr, pc->Lnode, pc->Lval, pc->D, pc->Uptr, pc->Unode,
                       pc->Uval, x, b);
}

/* ------------------------------------------------------------------------ */
/* Krylov solvers                                                           */
/* ------------------------------------------------------------------------ */

/* Shared argument block so the ctypes side stays small. */
typedef struct {
    /* operator */
    int32_t fmt, n, max_d;
    const int32_t *ptr, *node;
    const double *val;
    /* preconditioner: kind 0 none / 1 jacobi / 2 ildu */
    int32_t pc_kind;
    const double *idiag;
    const int32_t *Lptr, *Lnode;
    const double *Lval;
    const int32_t *Uptr, *Unode;
    const double *Uval;
    const double *D;
    /* solve */
    double tol;            /* ABSOLUTE tolerance on sqrt(res2) (cg_solvers.f90:133) */
    int64_t max_iter;      /* extension: <= 0 means unbounded, as the reference */
    double *history;       /* optional: res2 after every iteration, capacity hist_cap */
    int64_t hist_cap;
} orc_solve_args;

static void args_to(const orc_solve_args *a, orc_op *A, orc_pc *pc)
{
    A->fmt = a->fmt; A->n = a->n; A->max_d = a->max_d;
    A->ptr = a->ptr; A->node = a->node; A->val = a->val;
    pc->kind = a->pc_kind; pc->n = a->n; pc->idiag = (double *)a->idiag;
    pc->Lptr = (int32_t *)a->Lptr; pc->Lnode = (int32_t *)a->Lnode; pc->Lval = (double *)a->Lval;
    pc->Uptr = (int32_t *)a->Uptr; pc->Unode = (int32_t *)a->Unode; pc->Uval = (double *)a->Uval;
    pc->D = (double *)a->D;
}

/* cg_solve (src/solver/cg_solvers.f90:116-150) and cg_solve_pc (:155-194).
 * Returns the number of iterations; *res2_out = last res2. */
ORC_API int64_t orc_cg(const orc_solve_args *a, double *x, const double *b, double *res2_out)
{
    orc_op A; orc_pc pc; args_to(a, &A, &pc);
    int32_t n = a->n;
    double *p = calloc(n, 8), *q = calloc(n, 8), *r = calloc(n, 8), *z = calloc(n, 8);
    double alpha, beta, res2, dpr;
    int64_t it = 0;

    if (a->pc_kind == 0) {
        op_matvec(&A, x, q);
        for (int32_t i = 0; i < n; i++) r[i] = b[i] - q[i];
        for (int32_t i = 0; i < n; i++) p[i] = r[i];
        res2 = dot(n, r, r);
        while (sqrt(res2) > a->tol) {
            if (a->max_iter > 0 && it >= a->max_iter) break;
            op_matvec(&A, p, q);
            dpr = dot(n, p, q);
            alpha = res2 / dpr;
            for (int32_t i = 0; i < n; i++) x[i] = x[i] + alpha * p[i];
            for (int32_t i = 0; i < n; i++) r[i] = r[i] - alpha * q[i];
            dpr = dot(n, r, r);
            beta = dpr / res2;
            for (int32_t i = 0; i < n; i++) p[i] = r[i] + beta * p[i];
            res2 = dpr;
            if (a->history && it < a->hist_cap) a->history[it] = res2;
            it++;
        }
    } else {
        for (int32_t i = 0; i < n; i++) z[i] = x[i];
        op_matvec(&A, z, q);
        for (int32_t i = 0; i < n; i++) r[i] = b[i] - q[i];
        pc_solve(&pc, z, r);
        for (int32_t i = 0; i < n; i++) p[i] = z[i];
        res2 = dot(n, r, z);
        while (sqrt(res2) > a->tol) {
            if (a->max_iter > 0 && it >= a->max_iter) break;
            op_matvec(&A, p, q);
            dpr = dot(n, p, q);
            alpha = res2 / dpr;
            for (int32_t i = 0; i < n; i++) x[i] = x[i] + alpha * p[i];
            for (int32_t i = 0; i < n; i++) r[i] = r[i] - alpha * q[i];
            pc_solve(&pc, z, r);
            dpr = dot(n, r, z);
            beta = dpr / res2;
            for (int32_t i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
            res2 = dpr;
            if (a->history && it < a->hist_cap) a->history[it] = res2;
            it++;
        }
    }
    if (res2_out) *res2_out = res2;
    free(p); free(q); free(r); free(z);
    return it;
}

/* bicgstab_solve (src/solver/bicgstab_solvers.f90:124-177) and _pc (:182-237). */
ORC_API int64_t orc_bicgstab(const orc_solve_args *a, double *x, const double *b,
                             double *res2_out)
{
    orc_op A; orc_pc pc; args_to(a, &A, &pc);
    int32_t n = a->n;
    double *p = calloc(n, 8), *q = calloc(n, 8), *r = calloc(n, 8), *r0 = calloc(n, 8);
    double *v = calloc(n, 8), *s = calloc(n, 8), *t = calloc(n, 8), *z = calloc(n, 8);
    double alpha, beta, omega, rho, rho_old, res2;
    int64_t it = 0;

    op_matvec(&A, x, q);
    if (a->pc_kind == 0) {
        for (int32_t i = 0; i < n; i++) r0[i] = b[i] - q[i];
    } else {
        for (int32_t i = 0; i < n; i++) z[i] = b[i] - q[i];
        pc_solve(&pc, r0, z);
    }
    for (int32_t i = 0; i < n; i++) r[i] = r0[i];
    rho = 1.0; rho_old = 1.0; alpha = 1.0; omega = 1.0;
    for (int32_t i = 0; i < n; i++) { v[i] = 0.0; p[i] = 0.0; }
    res2 = dot(n, r, r);

    while (sqrt(res2) > a->tol) {
        if (a->max_iter > 0 && it >= a->max_iter) break;
        rho = dot(n, r0, r);
        beta = rho / rho_old * alpha / omega;
        for (int32_t i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - omega * v[i]);
        if (a->pc_kind == 0) {
            op_matvec(&A, p, v);
        } else {
            op_matvec(&A, p, z);
            pc_solve(&pc, v, z);
        }
        alpha = rho / dot(n, r0, v);
        for (int32_t i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
        if (a->pc_kind == 0) {
            op_matvec(&A, s, t);
        } else {
            op_matvec(&A, s, z);
            pc_solve(&pc, t, z);
        }
        omega = dot(n, s, t) / dot(n, t, t);
        if (a->pc_kind == 0 && isnan(omega)) omega = 0.0;   /* :165, plain variant only */
        for (int32_t i = 0; i < n; i++) x[i] = x[i] + alpha * p[i] + omega * s[i];
        for (int32_t i = 0; i < n; i++) r[i] = s[i] - omega * t[i];
        res2 = dot(n, r, r);
        rho_old = rho;
        if (a->history && it < a->hist_cap) a->history[it] = res2;
        it++;
    }
    if (res2_out) *res2_out = res2;
    free(p); free(q); free(r); free(r0); free(v); free(s); free(t); free(z);
    return it;
}

/* GMRES(m), restarted, modified Gram-Schmidt Arnoldi + Givens rotations
 * (Saad, Iterative Methods for Sparse Linear Systems, Alg. 6.9 / 6.11), optional
 * LEFT preconditioning to match bicgstab_solve_pc's convention.
 * NOT IN THE REFERENCE -- parity unpinned by the reference.
 * Conventions borrowed from cg_solve: absolute tolerance on the (preconditioned)
 * residual norm, initial guess taken from x, iterations = number of inner
 * (Arnoldi) steps, counted across restarts; max_iter <= 0 = unbounded. */
/* Arnoldi orthogonalisation of orc_gmres: 0 = modified Gram-Schmidt, 1 = CGS-2 (classical Gram-Schmidt
 * applied twice).  No reference constrains the choice (the reference has no GMRES). */
static int g_gmres_orth = 0;
ORC_API void orc_set_gmres_orth(int mode) { g_gmres_orth = mode; }

ORC_API int64_t orc_gmres(const orc_solve_args *a, int32_t m, double *x, const double *b,
                          double *res_out)
{
    orc_op A; orc_pc pc; args_to(a, &A, &pc);
    int32_t n = a->n;
    double *V = calloc((size_t)n * (m + 1), 8), *w = calloc(n, 8), *tmp = calloc(n, 8);
    double *H = calloc((size_t)(m + 1) * m, 8);        /* H[i + j*(m+1)] */
    double *cs = calloc(m, 8), *sn = calloc(m, 8), *g = calloc(m + 1, 8), *yv = calloc(m, 8);
    double *hc = calloc(m + 1, 8);
    int64_t it = 0;
    double res = 0.0;
    int done = 0;

    while (!done) {
        /* r = M^-1 (b - A x) */
        op_matvec(&A, x, tmp);
        for (int32_t i = 0; i < n; i++) tmp[i] = b[i] - tmp[i];
        if (a->pc_kind) pc_solve(&pc, w, tmp); else memcpy(w, tmp, (size_t)n * 8);
        double beta = sqrt(dot(n, w, w));
        res = beta;
        if (!(beta > a->tol)) break;
        for (int32_t i = 0; i < n; i++) V[i] = w[i] / beta;
        memset(g, 0, (size_t)(m + 1) * 8);
        g[0] = beta;
        int32_t j;
        for (j = 0; j < m; j++) {
            if (a->max_iter > 0 && it >= a->max_iter) { done = 1; break; }
            double *vj = V + (size_t)j * n;
            op_matvec(&A, vj, tmp);
            if (a->pc_kind) pc_solve(&pc, w, tmp); else memcpy(w, tmp, (size_t)n * 8);
            if (g_gmres_orth == 0) {        /* modified Gram-Schmidt */
                for (int32_t i = 0; i <= j; i++) {
                    double *vi = V + (size_t)i * n;
                    double h = dot(n, w, vi);
                    H[i + (size_t)j * (m + 1)] = h;
                    for (int32_t l = 0; l < n; l++) w[l] = w[l] - h * vi[l];
                }
            } else {                        /* CGS-2: classical Gram-Schmidt twice, h = h1 + h2 */
                for (int32_t i = 0; i <= j; i++) H[i + (size_t)j * (m + 1)] = 0.0;
                for (int pass = 0; pass < 2; pass++) {
                    for (int32_t i = 0; i <= j; i++) hc[i] = dot(n, w, V + (size_t)i * n);
                    for (int32_t i = 0; i <= j; i++) {
                        const double *vi = V + (size_t)i * n;
                        for (int32_t l = 0; l < n; l++) w[l] = w[l] - hc[i] * vi[l];
                        H[i + (size_t)j * (m + 1)] = pass ? H[i + (size_t)j * (m + 1)] + hc[i] : hc[i];
                    }
                }
            }
            double hn = sqrt(dot(n, w, w));
            H[(j + 1) + (size_t)j * (m + 1)] = hn;
            double *vn = V + (size_t)(j + 1) * n;
            for (int32_t l = 0; l < n; l++) vn[l] = w[l] / hn;
            /* apply previous rotations to column j */
            for (int32_t i = 0; i < j; i++) {
                double h0 = H[i + (size_t)j * (m + 1)], h1 = H[(i + 1) + (size_t)j * (m + 1)];
                H[i + (size_t)j * (m + 1)] = cs[i] * h0 + sn[i] * h1;
                H[(i + 1) + (size_t)j * (m + 1)] = -sn[i] * h0 + cs[i] * h1;
            }
            double h0 = H[j + (size_t)j * (m + 1)], h1 = H[(j + 1) + (size_t)j * (m + 1)];
            double d = sqrt(h0 * h0 + h1 * h1);
            cs[j] = h0 / d; sn[j] = h1 / d;
            H[j + (size_t)j * (m + 1)] = d;
            H[(j + 1) + (size_t)j * (m + 1)] = 0.0;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            res = fabs(g[j + 1]);
            if (a->history && it < a->hist_cap) a->history[it] = res * res;
            it++;
            if (!(res > a->tol)) { j++; done = 1; break; }
        }
        /* solve the j x j triangular system and update x */
        int32_t k = j;
        for (int32_t i = k - 1; i >= 0; i--) {
            double sacc = g[i];
            for (int32_t l = i + 1; l < k; l++) sacc = sacc - H[i + (size_t)l * (m + 1)] * yv[l];
            yv[i] = sacc / H[i + (size_t)i * (m + 1)];
        }
        for (int32_t i = 0; i < k; i++) {
            double *vi = V + (size_t)i * n;
            for (int32_t l = 0; l < n; l++) x[l] = x[l] + yv[i] * vi[l];
        }
    }
    if (res_out) *res_out = res;
    free(V); free(w); free(tmp); free(H); free(cs); free(sn); free(g); free(yv); free(hc);
    return it;
}

/* lanczos(A, T, Q) (src/eigensolver.f90:27-90) with the start vector handed in instead of
 * the time-seeded random_number (:45-49).  T: 3 x n column-major, Q: nrow x n column-major.
 * `sum(a*b)` is the Fortran intrinsic: order unspecified, restated with dot().           */
ORC_API void orc_lanczos(int32_t fmt, int32_t nrow, int32_t max_d, const int32_t *ptr,
                         const int32_t *node, const double *val, int32_t n, const double *q1,
                         double *T, double *Q)
{
    orc_op A = {fmt, nrow, max_d, ptr, node, val};
    double *w = calloc(nrow, 8);
    double alpha, beta = 0.0;
    memset(T, 0, (size_t)3 * n * 8);
    memset(Q, 0, (size_t)nrow * n * 8);
#define QC(i) (Q + (size_t)((i) - 1) * nrow)
    double nrm = sqrt(dot(nrow, q1, q1));
    for (int32_t l = 0; l < nrow; l++) QC(1)[l] = q1[l] / nrm;
    op_matvec(&A, QC(1), w);
    alpha = dot(nrow, QC(1), w);
    for (int32_t l = 0; l < nrow; l++) w[l] = w[l] - alpha * QC(1)[l];
    beta = sqrt(dot(nrow, w, w));
    for (int32_t l = 0; l < nrow; l++) QC(2)[l] = w[l] / beta;
    T[1] = alpha; T[2] = beta; T[0] = beta;
    for (int32_t i = 2; i <= n - 1; i++) {
        op_matvec(&A, QC(i), w);
        alpha = dot(nrow, QC(i), w);
        for (int32_t l = 0; l < nrow; l++) w[l] = w[l] - alpha * QC(i)[l] - beta * QC(i - 1)[l];
        for (int32_t k = 1; k <= i - 2; k++) {
            double h = dot(nrow, QC(k), w);
            for (int32_t l = 0; l < nrow; l++) w[l] = w[l] - h * QC(k)[l];
        }
        beta = sqrt(dot(nrow, w, w));
        for (int32_t l = 0; l < nrow; l++) QC(i + 1)[l] = w[l] / beta;
        T[3 * (i - 1) + 1] = alpha; T[3 * (i - 1) + 2] = beta; T[3 * (i - 1) + 0] = beta;
    }
    op_matvec(&A, QC(n), w);
    T[3 * (n - 1) + 1] = dot(nrow, QC(n), w);
#undef QC
    free(w);
}

/* generalized_lanczos(A, B, T, Q) (src/eigensolver.f90:95-155): Lanczos for A x = lambda B x;
 * every step solves B w = v with B's solver -- here the unpreconditioned CG restatement with
 * absolute tolerance `tol`, started from the CURRENT content of w (= A q_i, the reference's
 * `call B%solve(w, v)` hands w over as the initial guess).  No re-orthogonalisation (:131-148).
 * The start vector replaces the time-seeded random_number (:120-122); it is normalised in the
 * B-norm like :123-124.  T: 3 x n column-major, Q: nrow x n column-major. */
ORC_API void orc_generalized_lanczos(int32_t nrow, const int32_t *Aptr, const int32_t *Anode, const double *Aval,
                                     const int32_t *Bptr, const int32_t *Bnode, const double *Bval, double tol,
                                     int32_t n, const double *q1, double *T, double *Q)
{
    orc_op A = {1, nrow, 0, Aptr, Anode, Aval}, B = {1, nrow, 0, Bptr, Bnode, Bval};
    orc_solve_args sa;
    memset(&sa, 0, sizeof sa);
    sa.fmt = 1; sa.n = nrow; sa.ptr = Bptr; sa.node = Bnode; sa.val = Bval; sa.tol = tol;
    double *w = calloc(nrow, 8), *v = calloc(nrow, 8), *z = calloc((size_t)nrow * (n + 1), 8);
    double alpha = 0.0, beta = 0.0;
    memset(T, 0, (size_t)3 * n * 8);
    memset(Q, 0, (size_t)nrow * n * 8);
#define QC(i) (Q + (size_t)((i) - 1) * nrow)
#define ZC(i) (z + (size_t)(i) * nrow)              /* z(:, 0:n) */
    op_matvec(&B, q1, w);
    {
        const double nrm = sqrt(dot(nrow, w, q1));
        for (int32_t l = 0; l < nrow; l++) QC(1)[l] = q1[l] / nrm;
    }
    op_matvec(&B, QC(1), ZC(1));
    for (int32_t i = 1; i <= n - 1; i++) {
        op_matvec(&A, QC(i), w);
        for (int32_t l = 0; l < nrow; l++) v[l] = w[l] - beta * ZC(i - 1)[l];
        alpha = dot(nrow, v, QC(i));
        for (int32_t l = 0; l < nrow; l++) v[l] = v[l] - alpha * ZC(i)[l];
        orc_cg(&sa, w, v, NULL);                     /* call B%solve(w, v) */
        beta = sqrt(dot(nrow, w, v));
        for (int32_t l = 0; l < nrow; l++) QC(i + 1)[l] = w[l] / beta;
        for (int32_t l = 0; l < nrow; l++) ZC(i + 1)[l] = v[l] / beta;
        T[3 * (i - 1) + 1] = alpha; T[3 * (i - 1) + 2] = beta; T[3 * (i - 1) + 0] = beta;
    }
    op_matvec(&A, QC(n), v);
    for (int32_t l = 0; l < nrow; l++) v[l] = v[l] - beta * ZC(n)[l];
    T[3 * (n - 1) + 1] = dot(nrow, QC(n), v);       /* the loop index is n after the loop (:151) */
#undef QC
#undef ZC
    free(w); free(v); free(z);
}

/* ------------------------------------------------------------------------ */
/* Timing helpers for bench.py's cpu_baseline leg ("port", 1 thread: the     */
/* reference has no threading, CMakeLists.txt:17-20).                       */
/* ------------------------------------------------------------------------ */
#include <time.h>
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* The same row loop spread over the host's cores (OpenMP, static row blocks; every row still
 * sums left to right, so the result is bit-identical): SURVEY 8(d) "(ii) all cores".  The
 * reference itself has no threading; this is what a maintainer would get from one pragma.
 * Returns seconds per matvec and the thread count used. */
#ifdef _OPENMP
#include <omp.h>
#endif
ORC_API double orc_time_csr_matvec_omp(int32_t n, const int32_t *ptr, const int32_t *node,
                                       const double *val, const double *x, double *y,
                                       int32_t reps, int32_t *threads_used)
{
    int32_t nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    if (threads_used) *threads_used = nt;
    /* A fair all-cores ceiling needs the arrays on the NUMA node of the thread that streams them:
     * private copies are FIRST TOUCHED inside the parallel region with the same static row split
     * the timed loop uses (the caller's numpy arrays were all touched by one thread). */
    const int64_t nnz = (int64_t)ptr[n] - 1;
    int32_t *lptr = malloc(((size_t)n + 1) * 4), *lnode = malloc((size_t)(nnz > 0 ? nnz : 1) * 4);
    double *lval = malloc((size_t)(nnz > 0 ? nnz : 1) * 8), *lx = malloc((size_t)(n > 0 ? n : 1) * 8);
    double *ly = malloc((size_t)(n > 0 ? n : 1) * 8);
    if (!lptr || !lnode || !lval || !lx || !ly) { free(lptr); free(lnode); free(lval); free(lx); free(ly); return -1.0; }
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; i++) {
        lptr[i] = ptr[i];
        if (i == n - 1) lptr[n] = ptr[n];
        for (int32_t k = ptr[i] - 1; k < ptr[i + 1] - 1; k++) { lnode[k] = node[k]; lval[k] = val[k]; }
        lx[i] = x[i];
        ly[i] = 0.0;
    }
    double t0 = 0.0;
    for (int32_t r = -1; r < reps; r++) {           /* r = -1: warm-up */
        if (r == 0) t0 = now_s();
#pragma omp parallel for schedule(static)
        for (int32_t i = 0; i < n; i++) {
            double z = 0.0;
            for (int32_t k = lptr[i] - 1; k < lptr[i + 1] - 1; k++) z = z + lval[k] * lx[lnode[k] - 1];
            ly[i] = 0.0 + z;
        }
    }
    const double sec = (now_s() - t0) / reps;
    memcpy(y, ly, (size_t)n * 8);
    free(lptr); free(lnode); free(lval); free(lx); free(ly);
    return sec;
}

/* reps x (y = A x) with the CSR kernel; returns seconds per matvec. */
ORC_API double orc_time_csr_matvec(int32_t n, const int32_t *ptr, const int32_t *node,
                                   const double *val, const double *x, double *y,
                                   int32_t reps)
{
    orc_op A = {1, n, 0, ptr, node, val};
    op_matvec(&A, x, y);                /* warm-up */
    double t0 = now_s();
    for (int32_t r = 0; r < reps; r++) op_matvec(&A, x, y);
    return (now_s() - t0) / reps;
}

/* ------------------------------------------------------------------------ */
/* Reorderings of the matrix graph (src/graph/permutations.f90) and the      */
/* permutation of a CSR matrix (cs_graphs.f90:499-571 +                      */
/* default_sparse_matrix_kernels.f90:234-277).  All arrays 1-based like the  */
/* reference's; the graph is the cs_graph of the matrix (neighbours of i =   */
/* node(ptr(i) .. ptr(i+1)-1) in stored order, cs_graphs.f90 get_neighbors). */
/* ------------------------------------------------------------------------ */

/* breadth_first_search (permutations.f90:22-78): FIFO from vertex 1; p(i) = visiting
 * number (1..), -1 for vertices the search never reaches.  Returns the count visited. */
ORC_API int32_t orc_bfs_order(int32_t n, const int32_t *ptr, const int32_t *node, int32_t *p)
{
    int32_t *queue = malloc((size_t)(n > 0 ? n : 1) * 4);
    int32_t head = 0, tail = 0, num = 0;
    for (int32_t i = 0; i < n; i++) p[i] = -1;
    if (n > 0) queue[tail++] = 1;
    while (tail > head) {
        num++;
        const int32_t i = queue[head++];
        p[i - 1] = num;
        for (int32_t k = ptr[i - 1]; k < ptr[i]; k++) {
            const int32_t j = node[k - 1];
            if (p[j - 1] == -1) {
                queue[tail++] = j;
                p[j - 1] = 0;
            }
        }
    }
    free(queue);
    return num;
}

/* greedy_coloring (permutations.f90:83-157): vertices in FIFO order from vertex 1; a vertex
 * takes, among the colours already in use that none of its neighbours has, the one with the
 * FEWEST vertices so far (first such in colour order), else a new colour.  colors(i) in
 * 1..used, -1 for unreached vertices.  Returns the number of colours used. */
ORC_API int32_t orc_greedy_coloring(int32_t n, const int32_t *ptr, const int32_t *node,
                                    int32_t *colors)
{
    int32_t d = 0;
    for (int32_t i = 0; i < n; i++) if (ptr[i + 1] - ptr[i] > d) d = ptr[i + 1] - ptr[i];
    int32_t *queue = malloc((size_t)(n > 0 ? n : 1) * 4);
    int32_t *neighbor_colors = calloc((size_t)d + 2, 4), *color_totals = calloc((size_t)d + 2, 4);
    int32_t head = 0, tail = 0, used = 0;
    for (int32_t i = 0; i < n; i++) colors[i] = -1;
    if (n > 0) { queue[tail++] = 1; colors[0] = 0; }
    while (tail > head) {
        for (int32_t k = 0; k <= d; k++) neighbor_colors[k] = 0;
        const int32_t i = queue[head++];
        for (int32_t k = ptr[i - 1]; k < ptr[i]; k++) {
            const int32_t j = node[k - 1];
            const int32_t c = colors[j - 1];
            if (c > 0) neighbor_colors[c - 1]++;
            else if (c == -1) { queue[tail++] = j; colors[j - 1] = 0; }
        }
        int32_t color = 0, min_occupancy = n + 1;
        for (int32_t k = 1; k <= used; k++)
            if (color_totals[k - 1] > 0 && color_totals[k - 1] < min_occupancy &&
                neighbor_colors[k - 1] == 0) {
                color = k;
                min_occupancy = color_totals[k - 1];
            }
        if (color == 0) color = ++used;
        colors[i - 1] = color;
        color_totals[color - 1]++;
    }
    free(queue); free(neighbor_colors); free(color_totals);
    return used;
}

/* greedy_color_ordering (permutations.f90:162-205): p(i) = new index of vertex i when the
 * vertices are sorted by colour (stable in i); ptrs(c) = first new index of colour c
 * (num_colors + 1 entries).  The graph must be connected (the reference indexes ptrs(0)
 * for an unreached vertex).  Returns num_colors, or -1 if some vertex was not reached. */
ORC_API int32_t orc_greedy_color_ordering(int32_t n, const int32_t *ptr, const int32_t *node,
                                          int32_t *p, int32_t *ptrs)
{
    const int32_t nc = orc_greedy_coloring(n, ptr, node, p);
    for (int32_t i = 0; i < n; i++) if (p[i] < 1) return -1;
    for (int32_t c = 0; c <= nc; c++) ptrs[c] = 0;
    for (int32_t i = 0; i < n; i++) ptrs[p[i]]++;            /* ptrs(p(i)+1) += 1 */
    ptrs[0] = 1;
    for (int32_t c = 1; c <= nc; c++) ptrs[c] += ptrs[c - 1];
    int32_t *added = calloc((size_t)nc + 1, 4);
    for (int32_t i = 0; i < n; i++) {
        const int32_t c = p[i];
        p[i] = ptrs[c - 1] + added[c - 1];
        added[c - 1]++;
    }
    free(added);
    return nc;
}

/* cs_matrix%left_permute (cs_matrices.f90:471-478 -> graph_leftperm -> cs_graph_left_permute
 * cs_graphs.f90:499-550): row i becomes row p(i); the entries of a row keep their order. */
ORC_API void orc_csr_left_permute(int32_t n, const int32_t *ptr, const int32_t *node,
                                  const double *val, const int32_t *p, int32_t *ptr2,
                                  int32_t *node2, double *val2)
{
    for (int32_t i = 0; i <= n; i++) ptr2[i] = 0;
    for (int32_t i = 0; i < n; i++) ptr2[p[i]] = ptr[i + 1] - ptr[i];     /* ptr(p(i)+1) */
    ptr2[0] = 1;
    for (int32_t i = 0; i < n; i++) ptr2[i + 1] += ptr2[i];
    for (int32_t i = 0; i < n; i++) {
        const int32_t d = ptr[i + 1] - ptr[i];
        for (int32_t k = 0; k < d; k++) {
            node2[ptr2[p[i] - 1] - 1 + k] = node[ptr[i] - 1 + k];
            val2[ptr2[p[i] - 1] - 1 + k] = val[ptr[i] - 1 + k];
        }
    }
}

/* cs_matrix%right_permute (cs_graph_right_permute cs_graphs.f90:555-571): column j becomes
 * column p(j), in place; values do not move. */
ORC_API void orc_csr_right_permute(int64_t nnz, int32_t *node, const int32_t *p)
{
    for (int64_t k = 0; k < nnz; k++) node[k] = p[node[k] - 1];
}

/* ellpack_matrix%left_permute (ellpack_matrices.f90:601-619 + ellpack_graph_left_permute
 * ellpack_graphs.f90:486-518): column i of node/val(max_d, n) moves to column p(i), degrees too. */
ORC_API void orc_ell_left_permute(int32_t n, int32_t max_d, const int32_t *node, const double *val,
                                  const int32_t *degrees, const int32_t *p, int32_t *node2,
                                  double *val2, int32_t *degrees2)
{
    for (int32_t i = 0; i < n; i++) {
        for (int32_t k = 0; k < max_d; k++) {
            node2[(size_t)(p[i] - 1) * max_d + k] = node[(size_t)i * max_d + k];
            val2[(size_t)(p[i] - 1) * max_d + k] = val[(size_t)i * max_d + k];
        }
        degrees2[p[i] - 1] = degrees[i];
    }
}

/* ellpack_graph_right_permute (ellpack_graphs.f90:523-541): every nonzero neighbour j -> p(j). */
ORC_API void orc_ell_right_permute(int32_t n, int32_t max_d, int32_t *node, const int32_t *p)
{
    for (size_t k = 0; k < (size_t)n * max_d; k++)
        if (node[k] != 0) node[k] = p[node[k] - 1];
}
