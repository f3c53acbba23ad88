// Feasibility probe: 80 dependent tiny kernels, launched one by one vs replayed as a captured hipGraph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_tiny(double *a, int n, int gen, const int *flag) { if (*flag && gen >= *flag) return; int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * 1.0000001 + 1e-9; }
int main()
{
    const int n = 10000, per = 80, reps = 200;
    double *a; int *flag; hipMalloc(&a, n * 8); hipMalloc(&flag, 4); hipMemset(a, 0, n * 8); hipMemset(flag, 0, 4);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto batch = [&]() { for (int k = 0; k < per; ++k) hipLaunchKernelGGL(k_tiny, dim3((n + 255) / 256), dim3(256), 0, st, a, n, k, flag); };
    for (int r = 0; r < 5; ++r) batch();
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) batch();
    hipStreamSynchronize(st);
    double direct = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    batch();
    hipStreamEndCapture(st, &g);
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("instantiate: %s\n", hipGetErrorString(e));
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    double graph = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("direct: %.2f us per kernel; graph: %.2f us per kernel\n", direct / reps / per * 1e6, graph / reps / per * 1e6);
    return 0;
}
