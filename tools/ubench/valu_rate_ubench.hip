// Issue cost of the VALU instructions the rate epilogues use, on gfx950 (dev tool): 8 independent
// chains per wave, one, two and four waves per SIMD; prints s_memtime ticks per instruction.
//   hipcc --offload-arch=gfx950 -O3 valu_rate_ubench.hip -o valu_rate_ubench && ./valu_rate_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

#define NCH 8
#define BODY(NAME, ASM, CONSTR)                                                                       \
    __global__ __launch_bounds__(1024) void k_##NAME(double* out, long long* cyc, int iters, double a, \
                                                    double b)                                         \
    {                                                                                                 \
        double v[NCH];                                                                                \
        int w[NCH];                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NCH; ++i)                                               \
        {                                                                                             \
            v[i] = 1.0 + threadIdx.x * 1e-3 + i;                                                      \
            w[i] = i + 1;                                                                             \
        }                                                                                             \
        const long long t0 = __builtin_amdgcn_s_memtime();                                            \
        for (int it = 0; it < iters; ++it) {                                                          \
            _Pragma("unroll") for (int rep = 0; rep < 8; ++rep)                                       \
            {                                                                                         \
                _Pragma("unroll") for (int i = 0; i < NCH; ++i) asm volatile(ASM : CONSTR);           \
            }                                                                                         \
        }                                                                                             \
        const long long t1 = __builtin_amdgcn_s_memtime();                                            \
        double s = 0;                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NCH; ++i) s += v[i] + w[i];                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                              \
    }

#define C_VAB "+v"(v[i]) : "v"(a), "v"(b)
#define C_VW "+v"(v[i]), "+v"(w[i]) : "v"(a), "v"(b)
BODY(fma_f64, "v_fma_f64 %0, %0, %1, %2", C_VAB)
BODY(mul_f64, "v_mul_f64 %0, %0, %1", C_VAB)
BODY(add_f64, "v_add_f64 %0, %0, %2", C_VAB)
BODY(max_f64, "v_max_f64 %0, %0, %2", C_VAB)
BODY(rndne_f64, "v_rndne_f64 %0, %0", C_VAB)
BODY(ldexp_f64, "v_ldexp_f64 %0, %0, 1", C_VAB)
BODY(cvt_i32_f64, "v_cvt_i32_f64 %1, %0", C_VW)
BODY(cvt_f64_i32, "v_cvt_f64_i32 %0, %1", C_VW)
BODY(rcp_f64, "v_rcp_f64 %0, %0", C_VAB)
BODY(frexp_mant_f64, "v_frexp_mant_f64 %0, %0", C_VAB)
BODY(frexp_exp_f64, "v_frexp_exp_i32_f64 %1, %0", C_VW)
BODY(cmp_f64, "v_cmp_lt_f64 vcc, %0, %1", C_VAB)
BODY(cmp_cndmask, "v_cmp_lt_f64 vcc, %0, %2\n\tv_cndmask_b32 %1, %1, %1, vcc", C_VW)
BODY(cndmask_b32, "v_cndmask_b32 %1, %1, %1, vcc", C_VW)
BODY(lshl_add_u32, "v_lshl_add_u32 %1, %1, 1, %1", C_VW)
BODY(add_u32, "v_add_u32 %1, %1, %1", C_VW)
BODY(fma_f32, "v_fma_f32 %1, %1, %1, %1", C_VW)
BODY(pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0", C_VAB)
BODY(mov_b32, "v_mov_b32 %1, %1", C_VW)
BODY(and_b32, "v_and_b32 %1, %1, %1", C_VW)
BODY(fract_f64, "v_fract_f64 %0, %0", C_VAB)
BODY(floor_f64, "v_floor_f64 %0, %0", C_VAB)
BODY(log_f32, "v_log_f32 %1, %1", C_VW)
BODY(exp_f32, "v_exp_f32 %1, %1", C_VW)
BODY(rcp_f32, "v_rcp_f32 %1, %1", C_VW)
BODY(cvt_f32_f64, "v_cvt_f32_f64 %1, %0", C_VW)
BODY(cvt_f64_f32, "v_cvt_f64_f32 %0, %1", C_VW)

template <typename K>
void run(const char* name, K kern, int per_iter)
{
    double* out;
    long long* cyc;
    hipMalloc(&out, 8 * 1024 * 512);
    hipMalloc(&cyc, 8 * 1024);
    const int iters = 20000;
    double res[3];
    int q = 0;
    double wall[3];
    for (int threads : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.999999, 1e-7);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.999999, 1e-7);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < 256; ++i) m += h[i];
        m /= 256;
        // SIMD issue time per instruction: 1 wave/SIMD -> per wave; w waves/SIMD -> divide by w
        wall[q] = ms * 1e6 / (iters * 8.0 * NCH * per_iter) / (threads / 256);      // ns
        res[q++] = m / (iters * 8.0 * NCH * per_iter) / (threads / 256);
    }
    printf("%-16s wall ns/instr/SIMD %6.3f %6.3f %6.3f | ", name, wall[0], wall[1], wall[2]);
    printf("%-16s %6.2f (1 wave/SIMD)  %6.2f (2 waves/SIMD)  %6.2f (4 waves/SIMD) s_memtime ticks of SIMD issue time per instruction\n", name, res[0], res[1], res[2]);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
#define R(NAME) run(#NAME, k_##NAME, 1)
    R(fma_f64); R(mul_f64); R(add_f64); R(max_f64); R(rndne_f64); R(ldexp_f64); R(cvt_i32_f64); R(cvt_f64_i32);
    R(rcp_f64); R(frexp_mant_f64); R(frexp_exp_f64); R(cmp_f64); run("cmp+cndmask", k_cmp_cndmask, 2);
    R(cndmask_b32); R(lshl_add_u32); R(add_u32); R(fma_f32); R(pk_fma_f32); R(mov_b32); R(and_b32);
    R(fract_f64); R(floor_f64); R(log_f32); R(exp_f32); R(rcp_f32); R(cvt_f32_f64); R(cvt_f64_f32);
    return 0;
}
