// Shader clock during a low-occupancy kernel: cycles (s_memtime) vs 100 MHz wall clock, and the latency of a dependent
// fp64 multiply-subtract chain.   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, long long *t, int n, int waves_active)
{
    double a = out[threadIdx.x], b = 1.0000001, c = 1e-9;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        a = a * b; a = a - c; a = a * b; a = a - c; a = a * b; a = a - c; a = a * b; a = a - c;
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main()
{
    double *out; long long *t, h[2];
    hipMalloc(&out, 1 << 24); hipMalloc(&t, 16); hipMemset(out, 0, 1 << 24);
    const int n = 200000;
    for (int blocks : {1, 13, 26, 256, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, t, n, blocks);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        const double ns = h[1] * 10.0;
        printf("blocks %5d: %lld shader cycles in %.0f ns -> %.2f GHz; %.1f cycles = %.2f ns per dependent fp64 op\n", blocks, h[0], ns,
               h[0] / ns, (double)h[0] / (8.0 * n), ns / (8.0 * n));
    }
    return 0;
}
