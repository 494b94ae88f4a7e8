// f64 VALU dependent-issue latency on gfx950: NCH independent v_fma_f64 chains per wave,
// 1 or 2 waves per SIMD.  Prints shader cycles per v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 dp_latency_ubench.hip -o dp_latency_ubench && ./dp_latency_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NCH>
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int iters, double a, double b)
{
    double v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = threadIdx.x * 1e-3 + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) v[i] = __builtin_fma(v[i], a, b);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NCH>
void run(int threads)
{
    double* out; long long* cyc;
    hipMalloc(&out, 8 * 1024 * 512); hipMalloc(&cyc, 8 * 1024);
    const int iters = 2000;
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.999999, 1e-7);
    hipLaunchKernelGGL(k<NCH>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 0.999999, 1e-7);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("chains %2d waves/SIMD %d: %.2f cycles per v_fma_f64 per wave, %.2f cycles per dependent step\n", NCH,
           threads / 256, m / (iters * 16.0 * NCH), m / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int t : {256, 512}) {
        run<1>(t); run<2>(t); run<4>(t); run<8>(t); run<12>(t); run<16>(t);
    }
    return 0;
}
