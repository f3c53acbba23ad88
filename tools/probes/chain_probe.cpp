// What one step of the strip chain costs a lone wave, piece by piece: the dependent arithmetic only (registers), with the
// wave-wide DPP shift, with a row-wide one, without the masks.   make -C tools probes/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CTRL>
__device__ inline double dpp_mov(double v, double fill)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int slo = __builtin_amdgcn_update_dpp(__double2loint(fill), lo, CTRL, 0xf, 0xf, false);
    const int shi = __builtin_amdgcn_update_dpp(__double2hiint(fill), hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(shi, slo);
}
template <int V>
__global__ void k(double *out, long long *t, int n)
{
    const int lane = threadIdx.x;
    double prev = out[lane];
    const double cS = 0.3 + 1e-3 * lane, cW = 0.2 - 1e-3 * lane, rhs = 1.0, e = 0.5;
    const uint32_t mS = lane ? 0xffffffffu : 0u, mW = 0xffffffffu;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double left;
            if (V == 0 || V == 3) left = dpp_mov<0x138>(prev, e);          // wave_shr:1
            else if (V == 1) left = dpp_mov<0x111>(prev, e);               // row_shr:1
            else left = prev + e;                                          // no cross-lane move at all (one more add)
            double rS = cS * prev, rW = cW * left;
            if (V != 3) {
                rS = __hiloint2double(__double2hiint(rS) & (int)mS, __double2loint(rS) & (int)mS);
                rW = __hiloint2double(__double2hiint(rW) & (int)mW, __double2loint(rW) & (int)mW);
            }
            double z = rhs - rS;
            z = z - rW;
            prev = z;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = prev;
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
// the same step with three sibling waves in the workgroup that poll an LDS word (s_sleep SL between looks), as the
// helper waves of the strip kernel do; LDSPAD bytes of dynamic LDS keep other workgroups off the CU
template <int SL>
__global__ void k_sib(double *out, long long *t, int n)
{
    __shared__ int done;
    extern __shared__ unsigned char pad[];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (threadIdx.x >= 64) {
        long long polls = 0;
        while (!__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
            if (SL == 1) __builtin_amdgcn_s_sleep(1); else if (SL == 8) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(32);
            ++polls;
        }
        if (lane == 0) t[2 + (threadIdx.x >> 6)] = polls;
        return;
    }
    double prev = out[lane];
    const double cS = 0.3 + 1e-3 * lane, cW = 0.2 - 1e-3 * lane, rhs = 1.0, e = 0.5;
    const uint32_t mS = lane ? 0xffffffffu : 0u, mW = 0xffffffffu;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double left = dpp_mov<0x138>(prev, e);
            double rS = cS * prev, rW = cW * left;
            rS = __hiloint2double(__double2hiint(rS) & (int)mS, __double2loint(rS) & (int)mS);
            rW = __hiloint2double(__double2hiint(rW) & (int)mW, __double2loint(rW) & (int)mW);
            double z = rhs - rS;
            z = z - rW;
            prev = z;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = prev;
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; __hip_atomic_store(&done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
}
template <int SL>
void run_sib(const char *what, double *out, long long *t, int threads)
{
    const int n = 100000;
    long long h[6];
    (void)hipFuncSetAttribute((const void *)k_sib<SL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_sib<SL>, dim3(1), dim3(threads), 96 * 1024, 0, out, t, n); hipDeviceSynchronize(); }
    hipMemcpy(h, t, 48, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles = %5.1f ns per step  (sibling polls per step: %.2f)\n", what, (double)h[0] / (8.0 * n), h[1] * 10.0 / (8.0 * n),
           threads > 64 ? (double)h[3] / (8.0 * n) : 0.0);
}
// the chain as k_trsv_strip2 runs it: records and lane 0's neighbours out of LDS rings a chunk ahead, results into an LDS
// ring, three progress words snapshotted a chunk ahead, one release store per chunk -- with nobody else in the workgroup
// (the rings hold fixed data, the progress words say "everything is there").  PARTS: 0 = registers only (no LDS at all),
// 1 = + record reads, 2 = + result writes, 3 = + neighbour reads, 4 = + progress words and the release store.
typedef double f64x2s __attribute__((ext_vector_type(2)));
template <int PARTS>
__global__ __launch_bounds__(64) void k_lds(double *out, long long *t, int nchunks)
{
    constexpr int CH = 8, RS = 64, ZR = 32;
    extern __shared__ __align__(16) unsigned char dyn[];
    f64x2s *rring = reinterpret_cast<f64x2s *>(dyn);
    double *zring = reinterpret_cast<double *>(dyn + RS * 2048);
    __shared__ double in_ring[512];
    __shared__ int rec_avail, in_avail, out_sent, out_count;
    const int lane = threadIdx.x;
    for (int q = lane; q < RS * 128; q += 64) {
        f64x2s v;
        if ((q / 64) % 2 == 0) { v.x = 0.3 + 1e-3 * (q % 64); v.y = 0.2 - 1e-3 * (q % 64); }
        else { v.x = 1.0; v.y = __longlong_as_double(q % 64 ? -1ll : 0xffffffff00000000ll); }
        rring[q] = v;
    }
    for (int q = lane; q < 512; q += 64) in_ring[q] = 0.5;
    if (lane == 0) { rec_avail = 1 << 30; in_avail = 1 << 30; out_sent = 1 << 30; out_count = 0; }
    __syncthreads();
    f64x2s qa[CH], qb[CH], pa[CH], pb[CH];
    double qe[CH], pe[CH];
    double prev = out[lane];
    int f_rec = 1 << 30, f_in = 1 << 30, f_out = 1 << 30;
    auto snap = [&]() {
        if (PARTS >= 4) {
            f_rec = __hip_atomic_load(&rec_avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            f_in = __hip_atomic_load(&in_avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            f_out = __hip_atomic_load(&out_sent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    auto admit = [&](int tn) {
        if (PARTS >= 4) {
            while (f_rec < tn + CH || f_in < tn + CH || tn + CH - f_out > ZR) { __builtin_amdgcn_s_sleep(1); snap(); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    };
    auto fetch_chunk = [&](int tn, f64x2s (&a)[CH], f64x2s (&b)[CH], double (&e)[CH]) {
        const f64x2s *r = rring + (tn % RS) * 128 + lane;
#pragma unroll
        for (int u = 0; u < CH; ++u) e[u] = PARTS >= 3 ? in_ring[(tn + u) & 511] : 0.5;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (PARTS >= 1) { a[j] = r[j * 128]; b[j] = r[j * 128 + 64]; }
            else { a[j].x = 0.3 + 1e-3 * lane; a[j].y = 0.2 - 1e-3 * lane; b[j].x = 1.0; b[j].y = __longlong_as_double(lane ? -1ll : 0xffffffff00000000ll); }
        }
    };
    auto work_chunk = [&](int t0, const f64x2s (&ca)[CH], const f64x2s (&cb)[CH], const double (&eE)[CH]) {
        double *zw = zring + (t0 % ZR) * 64 + lane;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const f64x2s a = ca[j], b = cb[j];
            const double left = dpp_mov<0x138>(prev, eE[j]);
            const unsigned long long mk = (unsigned long long)__double_as_longlong(b.y);
            const uint32_t mS = (uint32_t)mk, mW = (uint32_t)(mk >> 32);
            const double rS = a.x * prev, rW = a.y * left;
            const double pS = __hiloint2double(__double2hiint(rS) & (int)mS, __double2loint(rS) & (int)mS);
            const double pW = __hiloint2double(__double2hiint(rW) & (int)mW, __double2loint(rW) & (int)mW);
            double z = b.x - pS;
            z = z - pW;
            if (PARTS >= 2) zw[j * 64] = z;
            prev = z;
        }
        if (PARTS >= 4 && lane == 0) __hip_atomic_store(&out_count, t0 + CH, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const long long c0 = clock64(), w0 = wall_clock64();
    snap(); admit(0);
    fetch_chunk(0, qa, qb, qe);
    snap();
    const int S = nchunks * CH;
    for (int t0 = 0; t0 < S; t0 += 2 * CH) {
        admit(t0 + CH);
        fetch_chunk(t0 + CH, pa, pb, pe);
        snap();
        __builtin_amdgcn_sched_barrier(0);
        work_chunk(t0, qa, qb, qe);
        admit(t0 + 2 * CH);
        fetch_chunk(t0 + 2 * CH, qa, qb, qe);
        snap();
        __builtin_amdgcn_sched_barrier(0);
        work_chunk(t0 + CH, pa, pb, pe);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[lane] = prev + (PARTS >= 2 ? zring[lane] : 0.0);
    if (lane == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
template <int PARTS>
void run_lds(const char *what, double *out, long long *t)
{
    const int nchunks = 100000;
    long long h[2];
    const size_t lds = 64 * 2048 + 32 * 512;
    (void)hipFuncSetAttribute((const void *)k_lds<PARTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lds<PARTS>, dim3(1), dim3(64), lds, 0, out, t, nchunks); hipDeviceSynchronize(); }
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles = %5.1f ns per step\n", what, (double)h[0] / (8.0 * nchunks), h[1] * 10.0 / (8.0 * nchunks));
}
template <int V>
void run(const char *what, double *out, long long *t)
{
    const int n = 100000;
    long long h[2];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, t, n); hipDeviceSynchronize(); }
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f cycles = %5.1f ns per step\n", what, (double)h[0] / (8.0 * n), h[1] * 10.0 / (8.0 * n));
}
int main()
{
    double *out; long long *t;
    hipMalloc(&out, 4096); hipMalloc(&t, 64); hipMemset(out, 0, 4096);
    run<0>("wave_shr:1, masks (the chain's step)", out, t);
    run<1>("row_shr:1, masks", out, t);
    run<2>("no cross-lane move (an add instead), masks", out, t);
    run<3>("wave_shr:1, no masks", out, t);
    run_sib<1>("the step, workgroup of 1 wave + 96 KiB LDS", out, t, 64);
    run_sib<1>("the step + 3 siblings polling, s_sleep 1", out, t, 256);
    run_sib<8>("the step + 3 siblings polling, s_sleep 8", out, t, 256);
    run_sib<32>("the step + 3 siblings polling, s_sleep 32", out, t, 256);
    run_sib<8>("the step + 1 sibling polling, s_sleep 8", out, t, 128);
    run_lds<0>("chunked chain, registers only", out, t);
    run_lds<1>("  + records out of the LDS ring", out, t);
    run_lds<2>("  + results into the LDS ring", out, t);
    run_lds<3>("  + lane 0's neighbours out of LDS", out, t);
    run_lds<4>("  + progress words, release store", out, t);
    return 0;
}
