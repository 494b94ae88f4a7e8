// Micro-benchmarks that set the ceiling for the fused kernel (dev tool, not product):
//  mode 0: v_mfma_f64_16x16x4_f64 only (4 independent accumulators per wave)
//  mode 1: same + NV independent v_fma_f64 per MFMA in the SAME wave
//  mode 2: VALU f64 FMA only
//  mode 3: MFMA waves and VALU-f64 waves co-resident (even waves MFMA, odd waves VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = seed + lane, y = seed * 0.5;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    const bool do_mfma = (MODE == 0) || (MODE == 1) || (MODE == 3 && (wave & 1) == 0);
    const bool do_valu = (MODE == 2) || (MODE == 1) || (MODE == 3 && (wave & 1) == 1);
    if (MODE == 3) {
        if (do_mfma) {
            for (int it = 0; it < iters; ++it) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fma(v[i], x, y);
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (do_mfma) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i & 7] = fma(v[i & 7], x, y);
            }
            if (do_mfma) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i & 7] = fma(v[i & 7], x, y);
            }
            if (do_mfma) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i & 7] = fma(v[i & 7], x, y);
            }
            if (do_mfma) a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            if (do_valu) {
#pragma unroll
                for (int i = 0; i < NV; ++i) v[i & 7] = fma(v[i & 7], x, y);
            }
        }
    }
    double s = 0;
    for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char* name, int blocks, int threads, int iters, double mfma_per_it, double fma_per_it)
{
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, NV><<<blocks, threads>>>(d, 10, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NV><<<blocks, threads>>>(d, iters, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double mf = mfma_per_it * iters * waves * 2048.0;     // flops
    const double vf = fma_per_it * iters * waves * 64 * 2.0;
    printf("%-44s blocks %4d x %3d  %8.3f ms  MFMA %7.2f TF  VALU %7.2f TF\n", name, blocks, threads, ms,
           mf / ms / 1e9, vf / ms / 1e9);
    hipFree(d);
}

int main()
{
    const int it = 20000;
    run<0, 0>("mfma only, 1 wave/SIMD", 256, 256, it, 4, 0);
    run<0, 0>("mfma only, 2 waves/SIMD", 256, 512, it, 4, 0);
    run<0, 0>("mfma only, 2 blocks/CU x 4 waves", 512, 256, it, 4, 0);
    run<2, 8>("valu f64 fma only, 1 wave/SIMD", 256, 256, it, 0, 32);
    run<2, 8>("valu f64 fma only, 2 waves/SIMD", 256, 512, it, 0, 32);
    run<1, 4>("same wave: 4 fma per mfma, 1 wave/SIMD", 256, 256, it, 4, 16);
    run<1, 8>("same wave: 8 fma per mfma, 1 wave/SIMD", 256, 256, it, 4, 32);
    run<1, 12>("same wave: 12 fma per mfma, 1 wave/SIMD", 256, 256, it, 4, 48);
    run<1, 16>("same wave: 16 fma per mfma, 1 wave/SIMD", 256, 256, it, 4, 64);
    run<3, 0>("co-resident: mfma waves + valu waves (2/SIMD)", 256, 512, it, 2, 32);  // per-wave avg
    return 0;
}
