/* The C ABI from plain C (C99, no C++ and no Python in the process): BASELINE config C1 -- the reference's own CPU-runnable
 * case, test/solver_test_diffusion_1d.f90:64-115 at n = 10000 -- through include/sigma_hip.h alone.
 *   tridiag(-1, 2, -1) as 1-based CSR arrays (cs_graphs.f90:16-19), f = 2 dx^2, u0 = 0, CG with the reference's default
 *   tolerance 1e-16 (cg_solvers.f90:106), checked against the analytic solution v(i) = i dx (1 - i dx).
 * Build + run (tests/test_gpu_boundary.py does both):
 *   gcc -std=c99 -O2 -I include tools/c_driver.c -L sigma_amd -lsigma_hip -Wl,-rpath,$PWD/sigma_amd -lm -o tools/c_driver
 *   tools/c_driver [n] [dot_order]
 * One line of JSON on stdout; exit code 0 when the solve ran and the answer is right. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sigma_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        const int rc_ = (call);                                                                  \
        if (rc_ != SGM_OK) {                                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sgm_last_error());                     \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    const int32_t n = argc > 1 ? (int32_t)atoi(argv[1]) : 10000;
    const int dot_order = argc > 2 ? atoi(argv[2]) : 0;
    const int64_t nnz = 3 * (int64_t)n - 2;
    int32_t *ptr = (int32_t *)malloc(((size_t)n + 1) * sizeof(int32_t));
    int32_t *node = (int32_t *)malloc((size_t)nnz * sizeof(int32_t));
    double *val = (double *)malloc((size_t)nnz * sizeof(double));
    double *u = (double *)calloc((size_t)n, sizeof(double)), *f = (double *)malloc((size_t)n * sizeof(double));
    double *y = (double *)calloc((size_t)n, sizeof(double));
    if (!ptr || !node || !val || !u || !f || !y) return 3;
    const double dx = 1.0 / (n + 1);
    int64_t k = 0;
    for (int32_t i = 1; i <= n; ++i) {           /* rows in the order the reference's test adds its edges: i-1, i, i+1 */
        ptr[i - 1] = (int32_t)(k + 1);
        if (i > 1) { node[k] = i - 1; val[k++] = -1.0; }
        node[k] = i; val[k++] = 2.0;
        if (i < n) { node[k] = i + 1; val[k++] = -1.0; }
        f[i - 1] = 2.0 * dx * dx;
    }
    ptr[n] = (int32_t)(k + 1);

    sgm_mat A = NULL;
    sgm_solver cg = NULL;
    CHECK(sgm_init(0));
    CHECK(sgm_csr_create(&A, n, n, nnz, ptr, node, val, SGM_HOST));
    CHECK(sgm_mat_matvec(A, f, y, SGM_HOST));                         /* A f: row 2 is -f1 + 2 f2 - f3 = 0 for a constant f */
    CHECK(sgm_cg_create(&cg, 1e-16));
    CHECK(sgm_solver_set_option(cg, "dot_order", dot_order));
    CHECK(sgm_solver_setup(cg, A));
    CHECK(sgm_solver_solve(cg, A, u, f, NULL, SGM_HOST));
    int64_t iterations = 0, last = 0;
    double res2 = 0.0;
    int32_t converged = 0;
    CHECK(sgm_solver_info(cg, &iterations, &res2, &converged, &last));
    double err = 0.0;
    for (int32_t i = 1; i <= n; ++i) {
        const double v = i * dx * (1.0 - i * dx), e = fabs(u[i - 1] - v);
        if (e > err) err = e;
    }
    char kernel[96] = "";
    CHECK(sgm_mat_kernel(A, kernel, (int)sizeof kernel));
    printf("{\"config\": \"C1 tridiag(-1,2,-1) CSR\", \"n\": %d, \"dot_order\": %d, \"iterations\": %lld, \"converged\": %d, "
           "\"sqrt_res2\": %.3e, \"max_err_vs_analytic\": %.3e, \"matvec_row2\": %.3e, \"kernel\": \"%s\"}\n",
           (int)n, dot_order, (long long)iterations, (int)converged, sqrt(res2), err, n > 2 ? y[1] : 0.0, kernel);
    CHECK(sgm_solver_destroy(cg));
    CHECK(sgm_mat_destroy(A));
    free(ptr); free(node); free(val); free(u); free(f); free(y);
    return converged && err <= 1e-9 ? 0 : 1;
}
