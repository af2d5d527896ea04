// v_permlane32_swap_b32 on gfx950: which halves does it exchange? (micro test; prints the lane pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out)
{
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[lane] = r.x; out[64 + lane] = r.y;
}
int main()
{
    unsigned* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("first  (in: 100 + lane): lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
    printf("second (in: 200 + lane): lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
    return 0;
}
