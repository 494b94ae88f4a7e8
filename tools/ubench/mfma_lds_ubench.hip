// How fast does the backward-pass MFMA loop of the fused kernels run in isolation?  (dev tool)
// 8 waves per CU, 20 accumulator tiles per wave (160 VGPRs), per 16-bin "tile" 80 MFMAs whose A
// operand comes (mode 1) from LDS with the kernel's ds_read pattern, (mode 0) from registers,
// (mode 2) LDS with a deep software pipeline (all 80 fragments of a tile loaded 8 steps ahead).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int PD>
__global__ __launch_bounds__(512, 2) void k(double* out, int tiles, double seed)
{
    constexpr int KTH = 20, RSH = 322;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* F = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < 16 * RSH; i += 512) F[i] = seed + 1e-9 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int col = lane & 15, grp = lane >> 4;
    d4 G[KTH];
#pragma unroll
    for (int k2 = 0; k2 < KTH; ++k2) G[k2] = (d4){0, 0, 0, 0};
    double rq[4] = {seed, seed * 0.5, seed * 0.25, seed * 0.125};
    const double* fb = F + grp * RSH + col;
    for (int t = 0; t < tiles; ++t) {
        constexpr int NS = 4 * KTH;
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                G[s % KTH] = __builtin_amdgcn_mfma_f64_16x16x4f64(rq[s & 3], rq[s / KTH], G[s % KTH], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            double ar[PD];
#pragma unroll
            for (int s = 0; s < PD; ++s) ar[s] = fb[(4 * (s / KTH)) * RSH + 16 * (s % KTH)];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double a = ar[s % PD];
                if (s + PD < NS) ar[s % PD] = fb[(4 * ((s + PD) / KTH)) * RSH + 16 * ((s + PD) % KTH)];
                G[s % KTH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rq[s / KTH], G[s % KTH], 0, 0, 0);
                if (MODE == 1 && (s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        rq[0] += 1e-12;
    }
    double s = 0;
#pragma unroll
    for (int k2 = 0; k2 < KTH; ++k2) s += G[k2][0] + G[k2][1] + G[k2][2] + G[k2][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int PD>
void run(const char* name)
{
    const int blocks = 256, tiles = 2000;
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 512);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 16 * 322 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, PD>), dim3(blocks), dim3(512), lds, 0, d, tiles, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 8 * tiles * 80 * 2048.0;
        if (rep == 2) printf("%-40s %8.3f ms  %6.1f TFLOP/s\n", name, ms, fl / ms / 1e9);
    }
    hipFree(d);
}

int main()
{
    run<0, 4>("registers only");
    run<1, 4>("LDS A, ring 4, sched_barrier/4");
    run<1, 8>("LDS A, ring 8, sched_barrier/4");
    run<2, 4>("LDS A, ring 4, free scheduling");
    run<2, 8>("LDS A, ring 8, free scheduling");
    run<2, 16>("LDS A, ring 16, free scheduling");
    return 0;
}
