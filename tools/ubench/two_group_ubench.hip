// Prototype of a pass-1 tile loop with the feature tile streamed in K-SLICES through a ring of LDS buffers
// (dev tool, no real math): one wave = one post tile; a workgroup of NW waves shares the slices.  NW = 8 is
// the shape of k_fused5 (one workgroup per CU); NW = 4 runs TWO independent workgroups per CU (4 post tiles
// each, every group streams its own slices), so that one group's epilogue / barrier waits are filled by the
// other group's MFMAs.  Per tile and wave: forward over 5 slices of 8 k-tiles (160 MFMAs, A from LDS),
// a rate-epilogue stand-in (EPI dependent f64 FMAs per element), backward over the 2 slices that are still
// resident (64 MFMAs).  Counted vmcnt waits: the LDS-DMA pieces are the only vector-memory operations.
//   hipcc --offload-arch=gfx950 -O3 two_group_ubench.hip -o two_group_ubench && ./two_group_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int KTS = 8;                                   // k-tiles per slice
constexpr int NSL = 5;                                   // slices per tile (40 k-tiles)
constexpr int RSS = 16 * KTS + 2;                        // row stride (doubles)
constexpr int SLB = ((16 * RSS * 8 + 1023) / 1024) * 1024;   // bytes per slice image
constexpr int NBUF = 4;

template <int NW>
__device__ __forceinline__ void dma_slice(const unsigned char* g, unsigned char* l, int wave, int lane)
{
    typedef __attribute__((address_space(1))) void gvoid;
    typedef __attribute__((address_space(3))) void lvoid;
    constexpr int NCH = SLB / 1024;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += NW) {
        const int c = c0 + wave;
        if (c < NCH)
            __builtin_amdgcn_global_load_lds((gvoid*)(g + (size_t)c * 1024 + lane * 16), (lvoid*)(l + (size_t)c * 1024), 16, 0, 0);
    }
}

template <int NW, int EPI, int SYNC>
__global__ __launch_bounds__(NW * 64, 512 / (NW * 64)) void k(const unsigned char* img, double* out, int tiles, double bval)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, grp = lane >> 4;
    constexpr int GROUPS = 8 / NW;
    const int chunk = blockIdx.x / GROUPS;
    const unsigned char* base = img + (size_t)chunk * tiles * NSL * SLB;
    constexpr int PMIN = (SLB / 1024) / NW;              // DMA pieces per slice every wave issues at least
    d4 G[2 * KTS];
#pragma unroll
    for (int i = 0; i < 2 * KTS; ++i) G[i] = (d4){0, 0, 0, 0};
    // load number n (n = 5 tau + j) carries slice order[j] of tile tau into buffer n % NBUF
    auto issue = [&](const int n) {
        const int tau = n / NSL, j = n % NSL;
        const int sl = (j + 2) % NSL;                    // order 2, 3, 4, 0, 1
        if (tau < tiles) dma_slice<NW>(base + ((size_t)tau * NSL + sl) * SLB, smem + (size_t)(n % NBUF) * SLB, wave, lane);
    };
    // prologue: slices 2, 3, 4 of tile 0 (loads 0, 1, 2); load 3 goes out at the first step
    issue(0);
    issue(1);
    issue(2);
    double sink = 0.0;
    for (int t = 0; t < tiles; ++t) {
        d4 acc0 = (d4){0, 0, 0, 0}, acc1 = acc0;
        // ---- forward over the five slices ----
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int n = t * NSL + j;
            // two younger loads may stay in flight
            if (SYNC) {
                __builtin_amdgcn_s_waitcnt(0x0f70 | (2 * PMIN));
                __builtin_amdgcn_s_barrier();
            }
            // the buffer of the slice consumed one step ago is free: load n + 3 goes there
            // (j = 0: s0(t) over s1(t-1) [done at B1(t-1)], j = 1: s1(t) over s2(t), j = 2: s2(t+1) over s3(t), j = 3: s3(t+1) over s4(t);
            //  j = 4 issues nothing: s4(t+1) has to wait for B0(t))
            if (j < 4) issue(n + 3);
            const double* fa = reinterpret_cast<const double*>(smem + (size_t)(n % NBUF) * SLB) + col * RSS + grp;
            constexpr int PA = 4;
            double ar[PA];
#pragma unroll
            for (int s = 0; s < PA; ++s) ar[s] = fa[4 * s];
#pragma unroll
            for (int s = 0; s < 4 * KTS; ++s) {
                const double a = ar[s % PA];
                if (s + PA < 4 * KTS) ar[s % PA] = fa[4 * (s + PA)];
                if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bval, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bval, acc0, 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- epilogue stand-in: 4 independent chains of EPI dependent f64 FMAs ----
        double rr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[r] = acc0[r] + acc1[r];
#pragma unroll
        for (int i = 0; i < EPI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rr[r] = fma(rr[r], 0.999, 1e-3);
        }
        // ---- backward over the two resident slices (loads 5t+3 = s0, 5t+4 = s1) ----
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = t * NSL + 3 + b;
            if (b == 1) {
                // every wave is done with s0(t): s4(t+1) (load 5t+7) goes over it
                if (SYNC) __builtin_amdgcn_s_barrier();
                issue(n + 3);
            }
            const double* fb = reinterpret_cast<const double*>(smem + (size_t)(n % NBUF) * SLB) + grp * RSS + col;
            constexpr int NS = 4 * KTS, PD = 8;
            double ar[PD];
#pragma unroll
            for (int s = 0; s < PD; ++s) ar[s] = fb[(4 * (s / KTS)) * RSS + 16 * (s % KTS)];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double a = ar[s % PD];
                if (s + PD < NS) ar[s % PD] = fb[(4 * ((s + PD) / KTS)) * RSS + 16 * ((s + PD) % KTS)];
                G[b * KTS + s % KTS] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, rr[s / KTS], G[b * KTS + s % KTS], 0, 0, 0);
                if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        sink += rr[0];
    }
#pragma unroll
    for (int i = 0; i < 2 * KTS; ++i) sink += G[i][0] + G[i][1] + G[i][2] + G[i][3];
    out[blockIdx.x * NW * 64 + threadIdx.x] = sink;
}

template <int NW, int EPI, int SYNC>
void run(const char* name, const unsigned char* img, double* d, int tiles)
{
    constexpr int GROUPS = 8 / NW;
    const int blocks = 256 * GROUPS;
    const size_t lds = (size_t)NBUF * SLB;
    auto kern = k<NW, EPI, SYNC>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), lds, 0, img, d, tiles, 0.37);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * 8 * tiles * (160 + 64) * 2048.0;
        if (rep == 2) printf("%-64s %8.3f ms  %6.1f TFLOP/s  (%s)\n", name, ms, fl / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
}

int main()
{
    const int tiles = 146;
    const size_t bytes = (size_t)256 * tiles * NSL * SLB;         // 256 chunks x 146 tiles x 5 slices: 3.25 GB
    unsigned char* img;
    double* d;
    (void)hipMalloc(&img, bytes);
    (void)hipMalloc(&d, sizeof(double) * 512 * 512);
    {
        // pseudo-random feature values (the MFMA array's power draw depends on the data)
        const size_t n = bytes / 8;
        double* h = (double*)malloc(64 << 20);
        unsigned long long x = 88172645463325252ull;
        for (size_t i = 0; i < (64 << 20) / 8; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
        for (size_t off = 0; off < bytes; off += (64 << 20))
            (void)hipMemcpy(img + off, h, (bytes - off < (64u << 20)) ? bytes - off : (64u << 20), hipMemcpyHostToDevice);
        free(h);
        (void)n;
    }
    run<8, 110, 1>("NW=8 (one workgroup per CU), epilogue 110, slices streamed", img, d, tiles);
    run<4, 110, 1>("NW=4 (two workgroups per CU), epilogue 110, slices streamed", img, d, tiles);
    run<8, 0, 1>("NW=8, no epilogue", img, d, tiles);
    run<4, 0, 1>("NW=4, no epilogue", img, d, tiles);
    run<8, 110, 0>("NW=8, epilogue 110, NO waits/barriers (data race, timing only)", img, d, tiles);
    run<4, 110, 0>("NW=4, epilogue 110, NO waits/barriers (data race, timing only)", img, d, tiles);
    run<8, 110, 1>("NW=8 again", img, d, tiles);
    run<4, 110, 1>("NW=4 again", img, d, tiles);
    return 0;
}
