// wave_sum (sgm_internal.hpp: permlane swaps + DPP row rotations) against the __shfl_xor butterfly it replaces: bit for bit,
// on random doubles of mixed magnitude and sign, special values included.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I sigma_amd/csrc -I include tools/probes/wave_sum_probe.cpp -o tools/probes/wave_sum_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "sgm_internal.hpp"

__global__ void k_both(const double *in, double *shfl, double *fast, double *blk)
{
    __shared__ double red[16];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double v = in[i];
    double w = v;
    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    shfl[i] = w;
    fast[i] = sgm::wave_sum(v);
    blk[i] = sgm::block_sum<1024>(v, red);
}

int main()
{
    const int blocks = 512, n = blocks * 1024;
    std::mt19937_64 rng(12345);
    std::vector<double> h(n);
    for (int i = 0; i < n; ++i) {
        const uint64_t r = rng();
        const int e = (int)(r % 80) - 40;
        double v = std::ldexp((double)(int64_t)(rng() >> 11) / 9007199254740992.0, e);
        if (r & (1ull << 40)) v = -v;
        if (i / 1024 == 7 && i % 97 == 0) v = 0.0;
        if (i / 1024 == 8 && i % 1024 == 5) v = INFINITY;
        if (i / 1024 == 9 && i % 1024 == 77) v = NAN;
        h[i] = v;
    }
    double *d = nullptr, *a = nullptr, *b = nullptr, *c = nullptr;
    auto ok = [](hipError_t e) { if (e != hipSuccess) { printf("HIP error: %s\n", hipGetErrorString(e)); exit(2); } };
    ok(hipMalloc(&d, n * 8)); ok(hipMalloc(&a, n * 8)); ok(hipMalloc(&b, n * 8)); ok(hipMalloc(&c, n * 8));
    ok(hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_both, dim3(blocks), dim3(1024), 0, 0, d, a, b, c);
    std::vector<double> ha(n), hb(n), hc(n);
    ok(hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost));
    ok(hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost));
    ok(hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost));
    long bad = 0, badblk = 0;
    for (int i = 0; i < n; ++i) {
        if (std::memcmp(&ha[i], &hb[i], 8) != 0) { if (bad++ < 5) printf("lane %d: shfl %a fast %a\n", i, ha[i], hb[i]); }
        // the block sum: the 16 wave sums (shuffle version, lane 0 of each wave) added in wave order
        const int blk0 = i / 1024 * 1024;
        double s = ha[blk0];
        for (int w = 1; w < 16; ++w) s += ha[blk0 + 64 * w];
        if (std::memcmp(&s, &hc[i], 8) != 0) { if (badblk++ < 5) printf("block sum at %d: want %a got %a\n", i, s, hc[i]); }
    }
    printf("wave_sum vs butterfly: %ld of %d lanes differ; block_sum<1024>: %ld differ\n", bad, n, badblk);
    return bad || badblk ? 1 : 0;
}
