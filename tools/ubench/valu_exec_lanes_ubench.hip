// Issue cost of the VALU / cross-lane instructions the Gibbs rate kernel uses: one wave, 8 independent chains,
// cycles per instruction from s_memtime (shader clock).  dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, double seed, int nlanes)
{
    if ((int)(threadIdx.x & 63) >= nlanes) return;
    double d[8]; float f[8]; int i8[8];
    const unsigned long long msk = __ballot(threadIdx.x & 1);
    asm volatile("s_mov_b64 s[20:21], %0" : : "s"(msk) : "s20", "s21");
    for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x * 1e-3; f[i] = (float)d[i]; i8[i] = (int)threadIdx.x + i; }
    if (OP == 37) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[0]), "v"(f[1]) : "vcc");
    unsigned long long t0 = __builtin_readcyclecounter();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 512; ++it) {
#define ONE(j)                                                                                               \
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[j]));                                  \
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[j]));                                      \
        if (OP == 2) asm volatile("v_max_f64 %0, %0, 0" : "+v"(d[j]));                                       \
        if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(d[j]));                          \
        if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[j]) : "v"(f[j]));                          \
        if (OP == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(f[j]));                                          \
        if (OP == 6) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(f[j]));                                      \
        if (OP == 7) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[j]));                                  \
        if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7]));     \
        if (OP == 9) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(f[j]), "v"(f[(j + 1) & 7]) : "vcc");    \
        if (OP == 10) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[j]), "v"(d[(j + 1) & 7]) : "vcc");   \
        if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(i8[j]) : "v"(i8[(j + 1) & 7])); \
        if (OP == 12) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(i8[j]), "+v"(i8[(j + 4) & 7]));         \
        if (OP == 13) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[j]));                                     \
        if (OP == 14) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[j]));                                       \
        if (OP == 15) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[j]) : "v"(i8[j]));                      \
        if (OP == 16) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[j]));                                         \
        if (OP == 17) asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[j]));                                     \
        if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7]));             \
        if (OP == 19) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[j]));                                  \
        if (OP == 20) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i8[j]) : "v"(d[j]));                        \
        if (OP == 21) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, 0" : "=v"(i8[j]) : "v"(i8[(j + 1) & 7]));     \
        if (OP == 22) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(i8[j]) : "s20");                    \
        if (OP == 23) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(d[j]));                              \
        if (OP == 24) asm volatile("v_log_f32 %0, %0" : "+v"(f[j]));                                         \
        if (OP == 25) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[j]) : "v"(d[(j + 1) & 7]));             \
        if (OP == 26) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7])); \
        if (OP == 27) asm volatile("v_add_f64 %0, %0, s[20:21]" : "+v"(d[j]));                             \
        if (OP == 28) asm volatile("v_fma_f64 %0, %0, s[20:21], %0" : "+v"(d[j]));                         \
        if (OP == 29) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(f[j]));                                  \
        if (OP == 30) asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(d[j]));                                  \
        if (OP == 31) asm volatile("v_cmp_lt_f64 s[22:23], %0, %1" : : "v"(d[j]), "v"(d[(j + 1) & 7]) : "s22", "s23"); \
        if (OP == 32) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "s"(seed));                        \
        if (OP == 33) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7]), "s"(msk)); \
        if (OP == 34) asm volatile("v_mov_b32 %0, %1" : "=v"(i8[j]) : "v"(i8[(j + 1) & 7]));             \
        if (OP == 35) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(i8[j]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "v"(i8[(j + 1) & 7]) : "vcc"); \
        if (OP == 36) asm volatile("v_cmp_lt_f32 s[22:23], %1, %2\n v_cndmask_b32_e64 %0, %0, %3, s[22:23]" : "+v"(i8[j]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "v"(i8[(j + 1) & 7]) : "s22", "s23"); \
        if (OP == 37) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7])); \
        if (OP == 41) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7])); \
        if (OP == 42) asm volatile("v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %1, %1, %0, vcc" : "+v"(i8[j]), "+v"(f[(j + 3) & 7]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "v"(i8[(j + 1) & 7]) : "vcc"); \
        if (OP == 43) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(i8[j]) : "v"(i8[(j + 1) & 7]) : "vcc"); \
        if (OP == 38) asm volatile("v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(i8[j]), "+v"(f[(j + 3) & 7]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "v"(i8[(j + 1) & 7]) : "vcc"); \
        if (OP == 39) asm volatile("v_cmp_lt_f32 s[22:23], %2, %3\n v_cndmask_b32_e64 %0, %0, %4, s[22:23]\n v_cndmask_b32_e64 %1, %1, %4, s[22:23]" : "+v"(i8[j]), "+v"(f[(j + 3) & 7]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "v"(i8[(j + 1) & 7]) : "s22", "s23"); \
        if (OP == 40) asm volatile("v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %0, vcc" : "+v"(i8[j]), "+v"(f[(j + 3) & 7]) : "v"(f[j]), "v"(f[(j + 1) & 7]), "v"(i8[(j + 1) & 7]) : "vcc");
        REP8(ONE) REP8(ONE)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += d[i] + f[i] + i8[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if ((threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int OP> void run(const char* name, int waves, int nlanes = 64)
{
    double* o; unsigned long long* c; unsigned long long hcs[64];
    hipMalloc(&o, 8 * 64 * waves); hipMalloc(&c, 8 * 64);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, o, c, 1.0, nlanes);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64 * waves), 0, 0, o, c, 1.0, nlanes);
    hipMemcpy(hcs, c, 8 * 2 * waves, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0; for (int w = 0; w < waves; ++w) { if (hcs[2 * w] < lo) lo = hcs[2 * w]; if (hcs[2 * w + 1] > hi) hi = hcs[2 * w + 1]; }
    const unsigned long long hc = hi - lo;
    printf("%-22s lanes %d waves/WG %d: %.2f cycles per instruction and SIMD\n", name, nlanes, waves, hc / (512.0 * 16) / (waves > 4 ? waves / 4 : 1));
    hipFree(o); hipFree(c);
}
#define RUN(OP, NAME) run<OP>(NAME, 1); run<OP>(NAME, 4); run<OP>(NAME, 8); run<OP>(NAME, 16);
int main()
{
    for (int nl : {64, 48, 32, 16, 1}) { run<0>("v_fma_f64", 16, nl); run<6>("v_mul_f32", 16, nl); run<5>("v_exp_f32", 16, nl); run<16>("v_rcp_f64", 16, nl); }
    return 0;
}
