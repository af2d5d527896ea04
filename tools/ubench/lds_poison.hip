// Debug helper: fill the LDS of every CU with a 32-bit pattern, so that a kernel which reads LDS it has not written
// (stale data of an earlier workgroup) fails reproducibly instead of depending on what ran on the box before.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_poison(unsigned pattern, int words, unsigned* sink)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
    __syncthreads();
    // keep the block resident for a moment so that the blocks of the grid spread over all CUs
    long long t0 = clock64();
    while (clock64() - t0 < 200000) {}
    if (lds[(threadIdx.x * 17) % words] != pattern && sink) sink[0] = 1;
}

extern "C" int lds_poison(unsigned pattern)
{
    const int bytes = 80 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_poison, dim3(256 * 2 * 4), dim3(256), bytes, 0, pattern, bytes / 4, (unsigned*)nullptr);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
