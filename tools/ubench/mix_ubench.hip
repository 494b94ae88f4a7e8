// What can hide under v_mfma_f64_16x16x4_f64?  Same-wave interleave of NV filler ops per MFMA.
// FILL: 0 none, 1 v_fma_f64, 2 v_fma_f32, 3 v_add_u32 (int), 4 ds_read_b64, 5 v_cvt_f64_f32,
//       6 v_pk_fma_f32, 7 v_exp_f32 (transcendental), 8 v_rcp_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int FILL, int NV, bool MFMA>
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed)
{
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = seed;
    lds[threadIdx.x + 256] = seed;
    __syncthreads();
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = seed + lane, y = seed * 0.5;
    double vd[8];
    float vf[8];
    f2 vp[8];
    unsigned vi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { vd[i] = seed + i; vf[i] = (float)seed + i; vi[i] = lane + i; vp[i] = (f2){vf[i], vf[i]}; }
    const float xf = (float)x, yf = (float)y;
    int idx = lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MFMA) {
                if (m == 0) a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                if (m == 1) a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
                if (m == 2) a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
                if (m == 3) a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int j = i & 7;
                if (FILL == 1) vd[j] = fma(vd[j], x, y);
                if (FILL == 2) vf[j] = fmaf(vf[j], xf, yf);
                if (FILL == 3) vi[j] = vi[j] * 3u + (unsigned)it;
                if (FILL == 4) { vd[j] += lds[(idx + 64 * j) & 1023]; }
                if (FILL == 5) vd[j] = (double)(vf[j] + (float)vd[j]);
                if (FILL == 6) vp[j] = __builtin_elementwise_fma(vp[j], (f2){xf, xf}, (f2){yf, yf});
                if (FILL == 7) vf[j] = __builtin_amdgcn_exp2f(vf[j]);
                if (FILL == 8) vd[j] = __builtin_amdgcn_rcp(vd[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    double s = 0;
    for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    for (int i = 0; i < 8; ++i) s += vd[i] + vf[i] + vi[i] + vp[i][0] + vp[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FILL, int NV, bool MFMA>
void run(const char* name)
{
    const int blocks = 256, threads = 256, iters = 10000;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<FILL, NV, MFMA><<<blocks, threads>>>(d, 10, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<FILL, NV, MFMA><<<blocks, threads>>>(d, iters, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)blocks * threads / 64;
    const double mf = (MFMA ? 4.0 : 0.0) * iters * waves * 2048.0;
    // cycles per MFMA slot at 2.4 GHz nominal, one wave per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (4.0 * iters);
    printf("%-40s NV=%2d mfma=%d  %8.3f ms  MFMA %6.2f TF  %6.1f cyc/slot(@2.4GHz)\n", name, NV, (int)MFMA, ms,
           mf / ms / 1e9, cyc);
    hipFree(d);
}

int main()
{
    run<0, 0, true>("mfma only");
    run<1, 8, false>("v_fma_f64 only");
    run<1, 8, true>("mfma + v_fma_f64");
    run<2, 8, false>("v_fma_f32 only");
    run<2, 8, true>("mfma + v_fma_f32");
    run<2, 16, true>("mfma + v_fma_f32");
    run<2, 24, true>("mfma + v_fma_f32");
    run<3, 8, false>("int mul-add only");
    run<3, 8, true>("mfma + int mul-add");
    run<3, 16, true>("mfma + int mul-add");
    run<4, 8, false>("ds_read_b64+add_f64 only");
    run<4, 4, true>("mfma + ds_read_b64+add_f64");
    run<4, 8, true>("mfma + ds_read_b64+add_f64");
    run<5, 8, false>("cvt f64<->f32 only");
    run<5, 8, true>("mfma + cvt");
    run<6, 8, false>("v_pk_fma_f32 only");
    run<6, 8, true>("mfma + v_pk_fma_f32");
    run<6, 16, true>("mfma + v_pk_fma_f32");
    run<7, 8, false>("v_exp_f32 only");
    run<7, 8, true>("mfma + v_exp_f32");
    run<8, 8, false>("v_rcp_f64 only");
    run<8, 8, true>("mfma + v_rcp_f64");
    return 0;
}
