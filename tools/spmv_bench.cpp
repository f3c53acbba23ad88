// Tuning harness (not part of the product): times sgm_mat_matvec through the C ABI with HIP
// events for the launch configuration given in SGM_SPMV_CFG.  Build: make -C tools
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/sigma_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sgm_last_error()); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 3162, ny = argc > 2 ? atoi(argv[2]) : 3162;
    const int reps = argc > 3 ? atoi(argv[3]) : 100;
    const int stencil = argc > 4 ? atoi(argv[4]) : 5;      // 5: 2-D nx*ny ; 7: 3-D nx*nx*ny
    int64_t n;
    std::vector<int32_t> ptr, node;
    std::vector<double> val;
    if (stencil == 5) {
        n = (int64_t)nx * ny;
        ptr.resize(n + 1); node.reserve(5 * n); val.reserve(5 * n);
        ptr[0] = 1;
        for (int64_t k = 0; k < n; ++k) {
            const int i = k % nx, j = k / nx;
            if (j > 0) { node.push_back(k - nx + 1); val.push_back(-1); }
            if (i > 0) { node.push_back(k); val.push_back(-1); }
            node.push_back(k + 1); val.push_back(4);
            if (i < nx - 1) { node.push_back(k + 2); val.push_back(-1); }
            if (j < ny - 1) { node.push_back(k + nx + 1); val.push_back(-1); }
            ptr[k + 1] = (int32_t)node.size() + 1;
        }
    } else {
        const int64_t pl = (int64_t)nx * nx;
        n = pl * ny;
        ptr.resize(n + 1); node.reserve(7 * n); val.reserve(7 * n);
        ptr[0] = 1;
        for (int64_t k = 0; k < n; ++k) {
            const int i = k % nx, j = (k / nx) % nx, l = k / pl;
            if (l > 0) { node.push_back(k - pl + 1); val.push_back(-1); }
            if (j > 0) { node.push_back(k - nx + 1); val.push_back(-1); }
            if (i > 0) { node.push_back(k); val.push_back(-1); }
            node.push_back(k + 1); val.push_back(6);
            if (i < nx - 1) { node.push_back(k + 2); val.push_back(-1); }
            if (j < nx - 1) { node.push_back(k + nx + 1); val.push_back(-1); }
            if (l < ny - 1) { node.push_back(k + pl + 1); val.push_back(-1); }
            ptr[k + 1] = (int32_t)node.size() + 1;
        }
    }
    const int64_t nnz = node.size();
    CK(sgm_init(0));
    if (const char *d = getenv("SGM_CSR_DO")) CK(sgm_set_option("csr_offset_dict", atoi(d)));
    if (const char *d = getenv("SGM_CSR_RO")) CK(sgm_set_option("csr_row_owner", atoi(d)));
    sgm_mat A;
    CK(sgm_csr_create(&A, (int32_t)n, (int32_t)n, nnz, ptr.data(), node.data(), val.data(), SGM_HOST));
    std::vector<double> hx(n);
    for (int64_t i = 0; i < n; ++i) hx[i] = sin(0.001 * (i + 1));
    double *x, *y;
    CK(sgm_malloc((void **)&x, n * 8));
    CK(sgm_malloc((void **)&y, n * 8));
    CK(sgm_memcpy(x, hx.data(), n * 8, 0));
    hipStream_t st;
    hipStreamCreate(&st);
    CK(sgm_set_stream(st));
    CK(sgm_set_async(1));
    for (int r = 0; r < 10; ++r) CK(sgm_mat_matvec(A, x, y, SGM_DEVICE));
    hipStreamSynchronize(st);
    std::vector<hipEvent_t> ev(reps + 1);
    for (auto &e : ev) hipEventCreate(&e);
    // SGM_BENCH_FLUSH=1: stream 1 GiB through the caches between matvecs (cold Infinity Cache,
    // the state a matvec finds inside a Krylov iteration)
    const bool flush = getenv("SGM_BENCH_FLUSH") != nullptr;
    void *fl = nullptr;
    if (flush) hipMalloc(&fl, 1ull << 30);
    std::vector<hipEvent_t> ev0(reps);
    for (auto &e : ev0) hipEventCreate(&e);
    hipEventRecord(ev[0], st);
    for (int r = 0; r < reps; ++r) {
        if (flush) hipMemsetAsync(fl, r, 1ull << 30, st);
        hipEventRecord(ev0[r], st);
        CK(sgm_mat_matvec(A, x, y, SGM_DEVICE));
        hipEventRecord(ev[r + 1], st);
    }
    hipStreamSynchronize(st);
    std::vector<float> ms(reps);
    float tot = 0;
    for (int r = 0; r < reps; ++r) { hipEventElapsedTime(&ms[r], ev0[r], ev[r + 1]); tot += ms[r]; }
    std::sort(ms.begin(), ms.end());
    const double bytes = 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n;
    // checksum so that variants can be compared bit for bit
    std::vector<double> hy(n);
    CK(sgm_memcpy(hy.data(), y, n * 8, 1));
    unsigned long long h = 1469598103934665603ull;
    for (int64_t i = 0; i < n; ++i) { unsigned long long b; memcpy(&b, &hy[i], 8); h = (h ^ b) * 1099511628211ull; }
    const char *cfg = getenv("SGM_SPMV_CFG");
    printf("do=%s cfg=%-18s n=%lld nnz=%lld  avg %.2f us  med %.2f us  min %.2f us  -> %.0f GB/s avg, %.0f GB/s best (%.1f%% / %.1f%% of 8 TB/s)  y-hash %016llx\n",
           getenv("SGM_CSR_DO") ? getenv("SGM_CSR_DO") : "1", cfg ? cfg : "default", (long long)n, (long long)nnz, 1e3 * tot / reps, 1e3 * ms[reps / 2], 1e3 * ms[0],
           bytes / (1e-3 * tot / reps) / 1e9, bytes / (1e-3 * ms[0]) / 1e9, bytes / (1e-3 * tot / reps) / 8e10,
           bytes / (1e-3 * ms[0]) / 8e10, h);
    // ---- CG: fixed iteration count, device-resident vectors
    if (getenv("SGM_BENCH_CG")) {
        const int its = atoi(getenv("SGM_BENCH_CG"));
        sgm_solver sv;
        CK(sgm_cg_create(&sv, 1e-300));
        CK(sgm_solver_set_max_iter(sv, its));
        CK(sgm_solver_setup(sv, A));
        double *u, *b;
        CK(sgm_malloc((void **)&u, n * 8));
        CK(sgm_malloc((void **)&b, n * 8));
        std::vector<double> hb(n, 1.0 / n), hz(n, 0.0);
        CK(sgm_memcpy(b, hb.data(), n * 8, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(sgm_memcpy(u, hz.data(), n * 8, 0));
            hipStreamSynchronize(st);
            hipEventRecord(ev[0], st);
            int rc = sgm_solver_solve(sv, A, u, b, nullptr, SGM_DEVICE);
            hipEventRecord(ev[1], st);
            hipStreamSynchronize(st);
            if (rc != 0 && rc != SGM_ERR_NOT_CONVERGED) { fprintf(stderr, "solve: %s\n", sgm_last_error()); return 1; }
            float t; hipEventElapsedTime(&t, ev[0], ev[1]);
            int64_t it, last; double r2; int32_t cv;
            sgm_solver_info(sv, &it, &r2, &cv, &last);
            const double cgb = bytes + 72.0 * n;
            printf("  CG rep %d: %lld its  %.2f us/iter  %.0f iters/s  %.0f GB/s on B_csr+72n (%.1f%% of 8 TB/s)  res2 %.17g\n", rep,
                   (long long)last, 1e3 * t / last, last / (1e-3 * t), cgb * last / (1e-3 * t) / 1e9,
                   cgb * last / (1e-3 * t) / 8e10, r2);
        }
    }
    return 0;
}
