// Calibration (not part of the product): what does a plain streaming kernel reach on this
// MI355X?  read-only sum, copy, and a "CSR-shaped" mix (8B+4B streams + 8B write per 5 reads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double f64x2 __attribute__((ext_vector_type(2)));

template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const f64x2 *__restrict__ a, size_t n2, double *out)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        f64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) { f64x2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e-300) out[0] = s;
}
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_copy(const f64x2 *__restrict__ a, f64x2 *__restrict__ b, size_t n2)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        f64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
    }
    for (; i < n2; i += stride) b[i] = a[i];
}

// RPW streamed 16-byte reads per 16-byte streamed write (an SpMV on the 7-point sliced layout reads 68 bytes per 8 written:
// RPW = 8; a whole CG iteration about 3.4 : 1): what a read-mostly MIX reaches, against read-only and copy
template <int RPW>
__global__ __launch_bounds__(256) void k_mix(const f64x2 *__restrict__ a, f64x2 *__restrict__ b, size_t nw)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < nw; i += stride) {
        f64x2 v[RPW];
#pragma unroll
        for (int u = 0; u < RPW; ++u) v[u] = __builtin_nontemporal_load(a + i + u * nw);
        f64x2 s = v[0];
#pragma unroll
        for (int u = 1; u < RPW; ++u) { s.x += v[u].x; s.y += v[u].y; }
        __builtin_nontemporal_store(s, b + i);
    }
}

template <class F>
double timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / reps;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 800ull) << 20;
    const size_t n2 = bytes / 16;
    f64x2 *a, *b; double *out;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&out, 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int grid : {2048, 65536}) {
        double t;
        t = timeit([&] { k_read<false, 1><<<grid, 256>>>(a, n2, out); }, 50);
        printf("read  plain U1 grid %6d: %.1f us  %.0f GB/s\n", grid, t * 1e6, bytes / t / 1e9);
        t = timeit([&] { k_read<false, 4><<<grid, 256>>>(a, n2, out); }, 50);
        printf("read  plain U4 grid %6d: %.1f us  %.0f GB/s\n", grid, t * 1e6, bytes / t / 1e9);
        t = timeit([&] { k_read<true, 4><<<grid, 256>>>(a, n2, out); }, 50);
        printf("read  nt    U4 grid %6d: %.1f us  %.0f GB/s\n", grid, t * 1e6, bytes / t / 1e9);
        t = timeit([&] { k_copy<false, 4><<<grid, 256>>>(a, b, n2 / 2); }, 50);
        printf("copy  plain U4 grid %6d: %.1f us  %.0f GB/s (r+w)\n", grid, t * 1e6, bytes / t / 1e9);
        t = timeit([&] { k_copy<true, 4><<<grid, 256>>>(a, b, n2 / 2); }, 50);
        printf("copy  nt    U4 grid %6d: %.1f us  %.0f GB/s (r+w)\n", grid, t * 1e6, bytes / t / 1e9);
    }
    for (int grid : {2048, 8192}) {
        double t;
        size_t nw = n2 / 8;
        t = timeit([&] { k_mix<8><<<grid, 256>>>(a, b, nw); }, 30);
        printf("mix 8 reads : 1 write grid %6d: %.1f us  %.0f GB/s (r+w)\n", grid, t * 1e6, nw * 16.0 * 9 / t / 1e9);
        nw = n2 / 3;
        t = timeit([&] { k_mix<3><<<grid, 256>>>(a, b, nw); }, 30);
        printf("mix 3 reads : 1 write grid %6d: %.1f us  %.0f GB/s (r+w)\n", grid, t * 1e6, nw * 16.0 * 4 / t / 1e9);
    }
    return 0;
}