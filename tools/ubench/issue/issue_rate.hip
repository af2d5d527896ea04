// Issue rate of one wave64 on a gfx950 SIMD: cycles per instruction for fp32 / fp64 FMA chains, dependent and independent,
// with 1 .. 8 waves resident per SIMD (decides how the lane-per-profile and team kernels should be written).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename T, int ILP>
__global__ __launch_bounds__(256) void k_chain(T* out, long long* cyc, int iters, T a, T b)
{
    T x[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) x[k] = (T)threadIdx.x * (T)1e-3 + (T)k;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < ILP; ++k) x[k] = fma(x[k], a, b);
    }
    const long long t1 = clock64();
    T s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// dependent chain with a VALU compare -> scalar mask logic -> VALU select hop per step (the pattern of `act && wn < w0`), against the
// same number of VALU instructions without the hop (min / max / fma only)
template <int MODE>
__global__ __launch_bounds__(256) void k_hop(double* out, long long* cyc, int iters, double a, double b, double c1, double c2)
{
    double x = (double)threadIdx.x * 1e-3 + 1.0, y = 0.5;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {                       // cmp, cmp, s_and, cndmask x2, fma
                const bool p = x > c1, q = x < c2 + y;
                const double f = fma(x, a, b);
                x = (p && q) ? f : x;
            } else {                               // fma, max, min, add: four dependent VALU instructions, no mask traffic
                x = fmin(fmax(fma(x, a, b), c1), c2) + y;
            }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
static void run_hop(const char* name)
{
    const int blocks = 256, iters = 2000;
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * blocks * 256)); CK(hipMalloc(&cyc, sizeof(long long) * blocks * 4));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_hop<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.999, 0.001, 0.5, 2.0);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_hop<MODE>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 0.999, 0.001, 0.5, 2.0);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-60s %.2f ns per step (1 wave per SIMD)\n", name, ms * 1e6 / ((double)iters * 8));
    CK(hipFree(out)); CK(hipFree(cyc));
}

template <typename T, int ILP>
static void run(const char* name, int waves_per_simd)
{
    // one workgroup of 64 * 4 * waves_per_simd ... a 256-thread block puts one wave on each SIMD of a CU; launch `waves_per_simd`
    // blocks per CU worth of work on 64 CUs only (grid = 64 * waves_per_simd) -- the dispatcher spreads blocks over CUs first,
    // so use a grid of 256 * waves_per_simd blocks (all CUs, waves_per_simd blocks each)
    const int blocks = 256 * waves_per_simd, iters = 2000;
    T* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(T) * blocks * 256)); CK(hipMalloc(&cyc, sizeof(long long) * blocks * 4));
    hipLaunchKernelGGL((k_chain<T, ILP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, (T)0.999, (T)0.001);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_chain<T, ILP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, (T)0.999, (T)0.001);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost));
    double mean = 0; for (long long v : h) mean += (double)v; mean /= (double)h.size();
    const double n_inst = (double)iters * 8 * ILP;
    printf("%-34s waves/SIMD %d: %7.2f clock64 ticks per FMA per wave, kernel %.1f us -> %.2f ns per FMA per wave\n", name, waves_per_simd,
           mean / n_inst, ms * 1e3, ms * 1e6 / n_inst);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main()
{
    run_hop<0>("step = 2 cmp + mask and + select + fma (mask hop)");
    run_hop<1>("step = fma + max + min + add (VALU only)");
    for (int w : {1, 2, 4, 8}) {
        run<float, 1>("fp32 dependent chain", w);
        run<float, 8>("fp32 8 independent chains", w);
        run<double, 1>("fp64 dependent chain", w);
        run<double, 8>("fp64 8 independent chains", w);
    }
    return 0;
}
