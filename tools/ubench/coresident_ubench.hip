// Does gen-like work (LDS reads + integer address math + a few f32 FMAs) on one wave of a SIMD
// slow down back-to-back v_mfma_f64_16x16x4_f64 issued by the other wave of that SIMD?
// 512 threads = 8 waves; waves 0-3 MFMA (one per SIMD), waves 4-7 filler of the given kind.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512, 2) void k(double* out, int iters, double seed, int fill_iters)
{
    __shared__ float lds[8192];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = (float)(seed + i);
    __syncthreads();
    double s = 0;
    if (wave < 4) {
        d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = seed + lane, y = seed * 0.5;
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    } else if (KIND > 0) {
        float f0 = 0, f1 = 0, f2 = 0, f3 = 0;
        double g0 = 0, g1 = 0;
        unsigned idx = lane * 2 + wave;
        for (int it = 0; it < fill_iters; ++it) {
            // gen-like: dependent event read -> two tap reads -> 4 FMAs, integer address math
            const unsigned e = __float_as_uint(lds[idx & 8191]) & 4095u;
            const float2 u0 = *reinterpret_cast<const float2*>(&lds[(e + (lane & 3) * 2) & 8190]);
            const float2 u1 = *reinterpret_cast<const float2*>(&lds[(e + 8 + (lane & 3) * 2) & 8190]);
            if (KIND == 1) {            // f32 FMAs
                f0 = fmaf(u0.x, 1.5f, f0); f1 = fmaf(u0.y, 1.5f, f1);
                f2 = fmaf(u1.x, 1.5f, f2); f3 = fmaf(u1.y, 1.5f, f3);
            } else if (KIND == 2) {     // f64 FMAs
                g0 = fma((double)u0.x, 1.5, g0); g1 = fma((double)u1.y, 1.5, g1);
                g0 = fma((double)u0.y, 1.5, g0); g1 = fma((double)u1.x, 1.5, g1);
            } else {                    // LDS + int only
                idx += __float_as_uint(u0.x) ^ __float_as_uint(u1.y);
            }
            idx = idx * 1664525u + 1013904223u;
        }
        s = f0 + f1 + f2 + f3 + g0 + g1 + idx;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int fill_iters)
{
    const int blocks = 256, iters = 20000;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 512>>>(d, 10, 1e-9, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 512>>>(d, iters, 1e-9, fill_iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = 4.0 * iters * blocks * 4 * 2048.0;
    printf("%-46s fill_iters %6d  %8.3f ms  MFMA %6.2f TF (if MFMA-bound)\n", name, fill_iters, ms, mf / ms / 1e9);
    hipFree(d);
}

int main()
{
    run<0>("mfma waves only (4 of 8 waves)", 0);
    // filler sized to finish before the MFMA waves (so MFMA time is what we measure)
    run<1>("+ gen-like filler waves, f32 FMA", 20000);
    run<1>("+ gen-like filler waves, f32 FMA", 40000);
    run<1>("+ gen-like filler waves, f32 FMA (longer than mfma)", 120000);
    run<2>("+ gen-like filler waves, f64 FMA", 20000);
    run<2>("+ gen-like filler waves, f64 FMA", 40000);
    run<3>("+ LDS+int filler waves", 20000);
    run<3>("+ LDS+int filler waves", 40000);
    return 0;
}
