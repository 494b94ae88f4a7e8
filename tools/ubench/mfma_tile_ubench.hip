// Pass-1 tile skeleton of k_fused5 without DMA, epilogue or HBM traffic (dev tool): per tile
// forward (160 MFMAs, A from LDS, Wmat from L2) | barrier | backward (80 MFMAs, A from LDS) |
// barrier.  Shows what the phase structure alone costs against the isolated loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) d2* g_cd2p;

template <int BARRIERS, int EPI>
__global__ __launch_bounds__(512, 2) void k(const double* W, double* out, int tiles, double seed)
{
    constexpr int KTH = 20, RSH = 322, KSH = 80, KS_ALL = 160;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* F = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < 2 * 16 * RSH; i += 512) F[i] = seed + 1e-9 * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    const double* faL = F + col * RSH + grp;
    const double* faH = F + 16 * RSH + col * RSH + grp;
    const double* fb = F + grp * RSH + col;
    const double* wrow = W + (size_t)wave * KS_ALL * 64;
    d4 G[KTH];
#pragma unroll
    for (int i = 0; i < KTH; ++i) G[i] = (d4){0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
        d4 acc0 = (d4){0, 0, 0, 0}, acc1 = acc0;
        {
            const double* wr_s = wrow;
            asm volatile("" : "+s"(wr_s));
            constexpr int PW2 = 4, PA = 4;
            const g_cd2p wr2 = (g_cd2p)wr_s;
            d2 wr[PW2];
            double ar[PA];
#pragma unroll
            for (int s = 0; s < PW2; ++s) wr[s] = wr2[s * 64 + lane];
#pragma unroll
            for (int s = 0; s < PA; ++s) ar[s] = faL[4 * s];
#pragma unroll
            for (int s = 0; s < KS_ALL; ++s) {
                const double a = ar[s % PA];
                const double b = (s & 1) ? wr[(s / 2) % PW2].y : wr[(s / 2) % PW2].x;
                if (s + PA < KS_ALL) ar[s % PA] = (s + PA < KSH) ? faL[4 * (s + PA)] : faH[4 * (s + PA - KSH)];
                if ((s & 1) && (s / 2 + PW2 < KS_ALL / 2)) wr[(s / 2) % PW2] = wr2[(s / 2 + PW2) * 64 + lane];
                if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (BARRIERS) __syncthreads();
        double rr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double x = acc0[r] + acc1[r];
            if (EPI) {                                   // ~60 dependent f64 ops per element, like the rate epilogue
#pragma unroll
                for (int i = 0; i < 30; ++i) x = fma(x, 0.999, 1e-3);
            }
            rr[r] = x;
        }
        {
            constexpr int NS = 4 * KTH, PD = 4;
            double ar[PD];
#pragma unroll
            for (int s = 0; s < PD; ++s) ar[s] = fb[(4 * (s / KTH)) * RSH + 16 * (s % KTH)];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double a = ar[s % PD];
                if (s + PD < NS) ar[s % PD] = fb[(4 * ((s + PD) / KTH)) * RSH + 16 * ((s + PD) % KTH)];
                G[s % KTH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KTH], G[s % KTH], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (BARRIERS) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < KTH; ++i) s += G[i][0] + G[i][1] + G[i][2] + G[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int BARRIERS, int EPI>
void run(const char* name, const double* W, double* d)
{
    const int blocks = 256, tiles = 1000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t lds = 2 * 16 * 322 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<BARRIERS, EPI>), dim3(blocks), dim3(512), lds, 0, W, d, tiles, 1.0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 8 * tiles * 240 * 2048.0;
        if (rep == 2) printf("%-44s %8.3f ms  %6.1f TFLOP/s (%.2f us per tile)\n", name, ms, fl / ms / 1e9, ms * 1e3 / tiles);
    }
}

int main()
{
    double *W, *d;
    (void)hipMalloc(&W, 8 * 160 * 64 * 8);
    (void)hipMemset(W, 0, 8 * 160 * 64 * 8);
    (void)hipMalloc(&d, sizeof(double) * 256 * 512);
    run<0, 0>("fwd + bwd, no barriers, no epilogue", W, d);
    run<1, 0>("fwd | barrier | bwd | barrier", W, d);
    run<1, 1>("fwd | barrier | epilogue-like VALU | bwd | barrier", W, d);
    return 0;
}
