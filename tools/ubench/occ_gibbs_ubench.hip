#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "../../theano_pyglm_amd/csrc/pglm_kernels.hip.h"
int main()
{
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("sharedMemPerMultiprocessor %zu maxSharedMemoryPerBlock %zu regsPerMultiprocessor %d\n", pr.sharedMemPerMultiprocessor, pr.sharedMemPerBlock, pr.regsPerMultiprocessor);
    for (int kb = 36; kb <= 64; kb += 2) {
        int occ = -1;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_gibbs_rate_cols), hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_gibbs_rate_cols, 256, (size_t)kb * 1024);
        printf("k_gibbs_rate_cols lds %d KB -> %d workgroups per CU (%s)\n", kb, occ, hipGetErrorString(e));
    }
    return 0;
}
