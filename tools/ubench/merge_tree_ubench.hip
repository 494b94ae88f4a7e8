#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double merge32(double a, double b)
{
    // a' = [a_lo | b_lo], b' = [a_hi | b_hi]
    u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double merge16(double a, double b)
{
    u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
template <int CTRL, int BIT>
__device__ __forceinline__ double merge_dpp(double p, double q, int lane)
{
    const bool up = (lane & BIT) != 0;
    const double keep = up ? q : p, send = up ? p : q;
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(send), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(send), CTRL, 0xf, 0xf, true);
    return keep + __hiloint2double(hi, lo);
}
__global__ void k(const double* __restrict__ x, double* __restrict__ out)
{
    const int lane = threadIdx.x;
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = x[i * 64 + lane];
    double m1[8], m2[4], m3[2];
    for (int i = 0; i < 8; ++i) m1[i] = merge32(v[2 * i], v[2 * i + 1]);
    for (int i = 0; i < 4; ++i) m2[i] = merge16(m1[2 * i], m1[2 * i + 1]);
    for (int i = 0; i < 2; ++i) m3[i] = merge_dpp<0x140, 8>(m2[2 * i], m2[2 * i + 1], lane);
    double r = merge_dpp<0x141, 4>(m3[0], m3[1], lane);
    // quad reduce
    {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), 0xB1, 0xf, 0xf, true);
        r += __hiloint2double(hi, lo);
        lo = __builtin_amdgcn_update_dpp(0, __double2loint(r), 0x4E, 0xf, 0xf, true);       // quad_perm [2,3,0,1]
        hi = __builtin_amdgcn_update_dpp(0, __double2hiint(r), 0x4E, 0xf, 0xf, true);
        r += __hiloint2double(hi, lo);
    }
    out[lane] = r;
}
int main()
{
    double hx[16 * 64], ho[64];
    double *x, *o;
    if (hipMalloc(&x, sizeof(hx)) != hipSuccess || hipMalloc(&o, sizeof(ho)) != hipSuccess) return 2;
    for (int i = 0; i < 16; ++i) for (int l = 0; l < 64; ++l) hx[i * 64 + l] = (i + 1) * 1000.0 + l;
    if (hipMemcpy(x, hx, sizeof(hx), hipMemcpyHostToDevice) != hipSuccess) return 2;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, o);
    if (hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int q = l >> 2;
        const int kk = ((q >> 3) & 1) | ((q >> 2) & 1) << 1 | ((q >> 1) & 1) << 2 | (q & 1) << 3;
        const double want = 64 * (kk + 1) * 1000.0 + 63 * 32;
        if (ho[l] != want) { ++bad; if (bad < 8) printf("lane %d got %f want %f (k %d)\n", l, ho[l], want, kk); }
    }
    printf("bad %d\n", bad);
    return bad != 0;
}
