// Diagnostic: resident workgroups per CU the runtime grants each inflate kernel variant.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../include/debig_hip.h"
#include "../debigulator_amd/csrc/inflate_kernel.inc"
#include "../debigulator_amd/csrc/inflate_mw_kernel.inc"
int main()
{
    int n = 0;
    hipFuncAttributes a;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, debig_inflate_kernel, 64, 0);
    hipFuncGetAttributes(&a, (const void *)debig_inflate_kernel);
    printf("debig_inflate_kernel      : %d workgroups/CU, LDS %zu B, %d VGPRs\n", n, a.sharedSizeBytes, a.numRegs);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, debig_inflate_mw_kernel<2>, 128, 0);
    hipFuncGetAttributes(&a, (const void *)debig_inflate_mw_kernel<2>);
    printf("debig_inflate_mw_kernel<2>: %d workgroups/CU, LDS %zu B, %d VGPRs\n", n, a.sharedSizeBytes, a.numRegs);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, debig_inflate_mw_kernel<4>, 256, 0);
    hipFuncGetAttributes(&a, (const void *)debig_inflate_mw_kernel<4>);
    printf("debig_inflate_mw_kernel<4>: %d workgroups/CU, LDS %zu B, %d VGPRs\n", n, a.sharedSizeBytes, a.numRegs);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, debig_inflate_mw_kernel<8>, 512, 0);
    hipFuncGetAttributes(&a, (const void *)debig_inflate_mw_kernel<8>);
    printf("debig_inflate_mw_kernel<8>: %d workgroups/CU, LDS %zu B, %d VGPRs\n", n, a.sharedSizeBytes, a.numRegs);
    return 0;
}
