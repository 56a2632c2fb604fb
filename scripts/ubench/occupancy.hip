// Occupancy (workgroups per CU) the runtime computes for the pipelined backward kernels at given dynamic-LDS sizes, and a
// measured check: N workgroups that spin for a fixed time -- one round if they are all resident, two if not.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include -I quanonet_amd/csrc scripts/ubench/occupancy.hip -o scripts/ubench/occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include "hea_device.hpp"
#include "hea_zyz.hpp"
using namespace qhea;

__global__ __launch_bounds__(320) void spin_kernel(long long ticks, int* sink) {
    extern __shared__ char lds[];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = (int)lds[0];
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int maxlds = 0;
    hipDeviceGetAttribute(&maxlds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0);
    printf("CUs %d, LDS per CU %d\n", cus, maxlds);
    for (size_t lds : {40000ul, 48000ul, 52000ul, 53120ul, 54000ul, 54600ul, 56000ul, 60000ul, 65536ul, 80000ul}) {
        int n1 = 0, n2 = 0;
        hipFuncSetAttribute((const void*)bwd_ztri_kernel<5, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, bwd_ztri_kernel<5, 1, true>, 320, lds);
        hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, spin_kernel, 320, lds);
        // measured: 3 * cus workgroups spinning 20 us each
        int* sink; hipMalloc(&sink, 4 * 3 * cus);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        spin_kernel<<<3 * cus, 320, lds>>>(2000, sink);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spin_kernel<<<3 * cus, 320, lds>>>(2000, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("dyn LDS %6zu: API says %d (ztri compact) / %d (spin) workgroups per CU; 3 x CUs spinning 20 us took %.1f us\n", lds, n1, n2, ms * 1e3);
        hipFree(sink);
    }
    return 0;
}
