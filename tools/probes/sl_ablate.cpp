// Calibration (not part of the product): which part of the sliced SpMV kernel costs what in
// the streaming regime?  Same access pattern as k_csr_sl on a synthetic stencil; parts are
// switched off by template flags.  ./sl_ablate <n> <W:3|5|7> <nx> <plane> [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

__device__ inline int64_t rowblock_of(int it, int b, int grid) { const int per = grid >> 3; return (int64_t)it * grid + (int64_t)(b & 7) * per + (b >> 3); }

enum { LV = 1, LC = 2, GX = 4, ST = 8, PLAIN = 16, NOMAP = 32, R2 = 64, STPLAIN = 128, XNT = 256 };
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int W, int F>
__global__ __launch_bounds__(256) void k(int32_t n, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict,
                                         const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ int32_t dl[16];
    const int tid = threadIdx.x;
    if (tid < 16) dl[tid] = dict[tid];
    __syncthreads();
    const int64_t nrb = ((int64_t)n + 255) / 256;
    constexpr int R = (F & R2) ? 2 : 1;
    for (int it = 0;; it += R) {
        if ((int64_t)it * gridDim.x >= nrb) break;
        int64_t rb[R]; int32_t row[R]; uint32_t cw[R]; double v[R][W], xv[R][W];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            rb[r] = (F & NOMAP) ? (int64_t)(it + r) * gridDim.x + blockIdx.x : rowblock_of(it + r, blockIdx.x, gridDim.x);
            if (rb[r] >= nrb) rb[r] = nrb - 1;
            row[r] = (int32_t)(rb[r] * 256) + tid;
            cw[r] = 0x76543210u;
            if (F & LC) cw[r] = (F & PLAIN) ? scode[row[r]] : __builtin_nontemporal_load(scode + row[r]);
            const double *vb = sval + rb[r] * (int64_t)(W * 256) + tid;
#pragma unroll
            for (int u = 0; u < W; ++u) v[r][u] = (F & LV) ? ((F & PLAIN) ? vb[u * 256] : __builtin_nontemporal_load(vb + u * 256)) : 1.0 + u;
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const uint32_t c = (cw[r] >> (4 * u)) & 15u;
                xv[r][u] = (F & GX) ? ((F & XNT) ? __builtin_nontemporal_load(x + row[r] + dl[c]) : x[row[r] + dl[c]]) : (double)c;
            }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double z = 0.0;
#pragma unroll
            for (int u = 0; u < W; ++u) z = z + v[r][u] * xv[r][u];
            if ((F & ST) && (F & STPLAIN)) y[row[r]] = z;
            else if (F & ST) __builtin_nontemporal_store(z, y + row[r]);
            else if (z == 1.2345e-300) y[row[r]] = z;
        }
    }
}


// software-pipelined form: the next row block's code word and values are requested before
// the current block's x values are consumed, so a wave always has a full row block in flight
template <int W, int F>
__global__ __launch_bounds__(256) void kp(int32_t n, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict,
                                          const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ int32_t dl[16];
    const int tid = threadIdx.x;
    if (tid < 16) dl[tid] = dict[tid];
    __syncthreads();
    const int64_t nrb = ((int64_t)n + 255) / 256;
    int64_t rb = (F & NOMAP) ? (int64_t)blockIdx.x : rowblock_of(0, blockIdx.x, gridDim.x);
    if (rb >= nrb) return;
    uint32_t cw = __builtin_nontemporal_load(scode + rb * 256 + tid);
    double v[W];
    {
        const double *vb = sval + rb * (int64_t)(W * 256) + tid;
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = __builtin_nontemporal_load(vb + u * 256);
    }
    for (int it = 0;; ++it) {
        const int32_t row = (int32_t)(rb * 256) + tid;
        double xv[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const uint32_t c = (cw >> (4 * u)) & 15u;
            xv[u] = x[row + dl[c]];
        }
        __builtin_amdgcn_sched_barrier(0);
        int64_t rbn = (F & NOMAP) ? (int64_t)(it + 1) * gridDim.x + blockIdx.x : rowblock_of(it + 1, blockIdx.x, gridDim.x);
        const bool more = rbn < nrb;
        uint32_t cwn = 0xffffffffu;
        double vn[W];
        if (more) {
            cwn = __builtin_nontemporal_load(scode + rbn * 256 + tid);
            const double *vb = sval + rbn * (int64_t)(W * 256) + tid;
#pragma unroll
            for (int u = 0; u < W; ++u) vn[u] = __builtin_nontemporal_load(vb + u * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
        double z = 0.0;
#pragma unroll
        for (int u = 0; u < W; ++u) z = z + v[u] * xv[u];
        __builtin_nontemporal_store(z, y + row);
        if (!more) break;
        rb = rbn; cw = cwn;
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = vn[u];
    }
}

template <int W, int F>
void runp(const char *label, int grid, int32_t n, uint32_t *sc, int32_t *dict, double *sv, double *x, double *y, int plane)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) kp<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) kp<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / reps;
    const double bytes = (double)n * (8.0 * W + 4 + 8 + 8);
    printf("%-34s grid %5d: %8.1f us  %6.2f TB/s moved (%.0f MB)\n", label, grid, t * 1e6, bytes / t / 1e12, bytes / 1e6);
}


// two adjacent rows per lane: 16-byte value loads and 16-byte y stores (slices of 512 rows)
template <int W, int F>
__global__ __launch_bounds__(256) void k2(int32_t n, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict,
                                          const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ int32_t dl[16];
    const int tid = threadIdx.x;
    if (tid < 16) dl[tid] = dict[tid];
    __syncthreads();
    const int64_t nrb = ((int64_t)n + 511) / 512;
    for (int it = 0;; ++it) {
        if ((int64_t)it * gridDim.x >= nrb) break;
        int64_t rb = (F & NOMAP) ? (int64_t)it * gridDim.x + blockIdx.x : rowblock_of(it, blockIdx.x, gridDim.x);
        if (rb >= nrb) continue;
        const int32_t row = (int32_t)(rb * 512) + 2 * tid;
        u32x2 cw = {0x76543210u, 0x76543210u};
        if (F & LC) cw = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(scode + row));
        const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + rb * (int64_t)(W * 512)) + tid;
        f64x2 v[W];
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = __builtin_nontemporal_load(vb + u * 256);
        double xa[W], xb[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const uint32_t ca = (cw.x >> (4 * u)) & 15u, cb = (cw.y >> (4 * u)) & 15u;
            xa[u] = (F & GX) ? x[row + dl[ca]] : (double)ca;
            xb[u] = (F & GX) ? x[row + 1 + dl[cb]] : (double)cb;
        }
        f64x2 z = {0.0, 0.0};
#pragma unroll
        for (int u = 0; u < W; ++u) { z.x = z.x + v[u].x * xa[u]; z.y = z.y + v[u].y * xb[u]; }
        if ((F & ST) && (F & STPLAIN)) *reinterpret_cast<f64x2 *>(y + row) = z;
        else if (F & ST) __builtin_nontemporal_store(z, reinterpret_cast<f64x2 *>(y + row));
        else if (z.x == 1.2345e-300) y[row] = z.x;
    }
}

template <int W, int F>
void run2(const char *label, int grid, int32_t n, uint32_t *sc, int32_t *dict, double *sv, double *x, double *y, int plane)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k2<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k2<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / reps;
    const double bytes = (double)n * (8.0 * W + ((F & LC) ? 4 : 0) + ((F & GX) ? 8 : 0) + ((F & ST) ? 8 : 0));
    printf("%-34s grid %5d: %8.1f us  %6.2f TB/s moved (%.0f MB)\n", label, grid, t * 1e6, bytes / t / 1e12, bytes / 1e6);
}


// two adjacent rows per lane AND software-pipelined: the next row block's codes and values are
// requested before the current block's x values are consumed
template <int W, int F>
__global__ __launch_bounds__(256) void k2p(int32_t n, const uint32_t *__restrict__ scode, const int32_t *__restrict__ dict,
                                           const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ y)
{
    __shared__ int32_t dl[16];
    const int tid = threadIdx.x;
    if (tid < 16) dl[tid] = dict[tid];
    __syncthreads();
    const int64_t nrb = ((int64_t)n + 511) / 512;
    int64_t rb = (F & NOMAP) ? (int64_t)blockIdx.x : rowblock_of(0, blockIdx.x, gridDim.x);
    if (rb >= nrb) return;
    u32x2 cw = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(scode + rb * 512 + 2 * tid));
    f64x2 v[W];
    {
        const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + rb * (int64_t)(W * 512)) + tid;
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = __builtin_nontemporal_load(vb + u * 256);
    }
    for (int it = 0;; ++it) {
        const int32_t row = (int32_t)(rb * 512) + 2 * tid;
        double xa[W], xb[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const uint32_t ca = (cw.x >> (4 * u)) & 15u, cb = (cw.y >> (4 * u)) & 15u;
            xa[u] = x[row + dl[ca]];
            xb[u] = x[row + 1 + dl[cb]];
        }
        __builtin_amdgcn_sched_barrier(0);
        const int64_t rbn = (F & NOMAP) ? (int64_t)(it + 1) * gridDim.x + blockIdx.x : rowblock_of(it + 1, blockIdx.x, gridDim.x);
        const bool more = rbn < nrb;
        u32x2 cwn = {0xffffffffu, 0xffffffffu};
        f64x2 vn[W];
        if (more) {
            cwn = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(scode + rbn * 512 + 2 * tid));
            const f64x2 *vb = reinterpret_cast<const f64x2 *>(sval + rbn * (int64_t)(W * 512)) + tid;
#pragma unroll
            for (int u = 0; u < W; ++u) vn[u] = __builtin_nontemporal_load(vb + u * 256);
        }
        __builtin_amdgcn_sched_barrier(0);
        f64x2 z = {0.0, 0.0};
#pragma unroll
        for (int u = 0; u < W; ++u) { z.x = z.x + v[u].x * xa[u]; z.y = z.y + v[u].y * xb[u]; }
        __builtin_nontemporal_store(z, reinterpret_cast<f64x2 *>(y + row));
        if (!more) break;
        rb = rbn; cw = cwn;
#pragma unroll
        for (int u = 0; u < W; ++u) v[u] = vn[u];
    }
}

template <int W, int F>
void run2p(const char *label, int grid, int32_t n, uint32_t *sc, int32_t *dict, double *sv, double *x, double *y, int plane)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k2p<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k2p<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / reps;
    const double bytes = (double)n * (8.0 * W + 4 + 8 + 8);
    printf("%-34s grid %5d: %8.1f us  %6.2f TB/s moved (%.0f MB)\n", label, grid, t * 1e6, bytes / t / 1e12, bytes / 1e6);
}

template <int W, int F>
void run(const char *label, int grid, int32_t n, uint32_t *sc, int32_t *dict, double *sv, double *x, double *y, int plane)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 3; ++r) k<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<W, F><<<grid, 256>>>(n, sc, dict, sv, x + plane, y);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double t = ms * 1e-3 / reps;
    const double bytes = (double)n * (((F & LV) ? 8.0 * W : 0) + ((F & LC) ? 4 : 0) + ((F & GX) ? 8 : 0) + ((F & ST) ? 8 : 0));
    printf("%-34s grid %5d: %8.1f us  %6.2f TB/s moved (%.0f MB)\n", label, grid, t * 1e6, bytes / t / 1e12, bytes / 1e6);
}

template <int W>
void all(int grid, int32_t n, uint32_t *sc, int32_t *dict, double *sv, double *x, double *y, int plane)
{
    run<W, LV>("val only", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | PLAIN>("val only, plain loads", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC>("val+code", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | ST>("val+code+store", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX>("val+code+gather", grid, n, sc, dict, sv, x, y, plane);
    run<W, GX | LC>("code+gather", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX | ST>("all", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX | ST | NOMAP>("all, plain block map", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX | ST | R2>("all, 2 row blocks per pass", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX | ST | PLAIN>("all, plain loads", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | ST | STPLAIN>("val+code+store(plain)", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX | XNT>("val+code+gather(nt)", grid, n, sc, dict, sv, x, y, plane);
    run<W, LV | LC | GX | ST | STPLAIN>("all, plain store", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC>("2rows: val+code", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC | ST>("2rows: val+code+store", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC | ST | STPLAIN>("2rows: val+code+store(plain)", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC | GX>("2rows: val+code+gather", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC | GX | ST>("2rows: all", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC | GX | ST | NOMAP>("2rows: all, plain map", grid, n, sc, dict, sv, x, y, plane);
    run2<W, LC | GX | ST | STPLAIN>("2rows: all, plain store", grid, n, sc, dict, sv, x, y, plane);
    run2p<W, 0>("2rows pipelined", grid, n, sc, dict, sv, x, y, plane);
    run2p<W, NOMAP>("2rows pipelined, plain map", grid, n, sc, dict, sv, x, y, plane);
    runp<W, 0>("all, pipelined", grid, n, sc, dict, sv, x, y, plane);
    runp<W, NOMAP>("all, pipelined, plain map", grid, n, sc, dict, sv, x, y, plane);
}

int main(int argc, char **argv)
{
    const int32_t n = argc > 1 ? atoi(argv[1]) : 10000000;
    const int W = argc > 2 ? atoi(argv[2]) : 5;
    const int nx = argc > 3 ? atoi(argv[3]) : 3162;
    const int plane = argc > 4 ? atoi(argv[4]) : nx;
    const size_t np = ((size_t)n + 511) / 512 * 512;
    uint32_t *sc; int32_t *dict; double *sv, *x, *y;
    hipMalloc(&sc, np * 4); hipMalloc(&dict, 64); hipMalloc(&sv, np * W * 8); hipMalloc(&x, (np + 2 * (size_t)plane) * 8); hipMalloc(&y, np * 8);
    hipMemset(sv, 0, np * W * 8); hipMemset(x, 0, (np + 2 * (size_t)plane) * 8);
    std::vector<int32_t> d(16, 0);
    if (W == 3) { d[0] = -1; d[1] = 0; d[2] = 1; }
    else if (W == 5) { d[0] = -nx; d[1] = -1; d[2] = 0; d[3] = 1; d[4] = nx; }
    else { d[0] = -plane; d[1] = -nx; d[2] = -1; d[3] = 0; d[4] = 1; d[5] = nx; d[6] = plane; }
    hipMemcpy(dict, d.data(), 64, hipMemcpyHostToDevice);
    std::vector<uint32_t> h(np, 0x76543210u);
    hipMemcpy(sc, h.data(), np * 4, hipMemcpyHostToDevice);
    for (int grid : {1024, 2048, argc > 5 ? atoi(argv[5]) : 4096}) {
        if (W == 3) all<3>(grid, n, sc, dict, sv, x, y, plane);
        else if (W == 5) all<5>(grid, n, sc, dict, sv, x, y, plane);
        else all<7>(grid, n, sc, dict, sv, x, y, plane);
    }
    return 0;
}
