// Micro-benchmark: cost of LDS operations in the access patterns of the sweep (64 lanes -> ~21 destination nodes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REPS 512
template <int OP>
__global__ void k(const int* idx, long long* out, double* sink)
{
    __shared__ double buf[256];
    __shared__ unsigned ubuf[256];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { buf[i] = 1e300; ubuf[i] = 0xffffffffu; }
    __syncthreads();
    const int a = idx[lane];
    double v = 1000.0 + lane + threadIdx.x; double acc = 0.0; int ia = a;
    long long t0 = clock64();
#pragma unroll 8
    for (int r = 0; r < REPS; ++r) {
        if (OP == 0) atomicMin(reinterpret_cast<unsigned long long*>(&buf[a]), (unsigned long long)__double_as_longlong(v - r));
        if (OP == 1) atomicAdd(&ubuf[a], 1u);
        if (OP == 2) atomicMin(&ubuf[a], (unsigned)(lane + r));
        if (OP == 3) acc += buf[(a + r) & 255];
        if (OP == 4) acc += __shfl(v, (a + r) & 63);
        if (OP == 5) { ia = __shfl(ia, (ia + r) & 63); }
        if (OP == 6) { acc = fmin(acc + v, v); v = acc * 1.0000001; }
        if (OP == 7) atomicMin(reinterpret_cast<unsigned long long*>(&buf[lane]), (unsigned long long)__double_as_longlong(v - r));
    }
    long long t1 = clock64();
    if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + buf[lane] + ubuf[lane] + ia;
}
template <int OP> void run(const char* name, const int* d_idx, int blocks, int threads)
{
    long long* d_out; double* d_sink;
    hipMalloc(&d_out, sizeof(long long) * blocks * (threads / 64)); hipMalloc(&d_sink, sizeof(double) * blocks * threads);
    k<OP><<<blocks, threads>>>(d_idx, d_out, d_sink); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<OP><<<blocks, threads>>>(d_idx, d_out, d_sink); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * (threads / 64)); hipMemcpy(h.data(), d_out, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double s = 0; for (auto x : h) s += x;
    printf("%-34s blocks %5d x %3d thr: %7.1f cycles / wave-instr (per wave), kernel %.3f ms\n", name, blocks, threads, s / h.size() / REPS, ms);
    hipFree(d_out); hipFree(d_sink);
}
int main()
{
    int h_idx[64]; for (int i = 0; i < 64; ++i) h_idx[i] = i / 3;            // ~21 distinct addresses, runs of 3 (CSC order)
    int* d_idx; hipMalloc(&d_idx, sizeof(h_idx)); hipMemcpy(d_idx, h_idx, sizeof(h_idx), hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 3; ++cfg) {
        int blocks = cfg == 0 ? 1 : 256 * 8, threads = cfg == 2 ? 256 : 64;
        printf("--- %s\n", cfg == 0 ? "one wave alone" : (cfg == 1 ? "8 one-wave blocks per CU" : "8 four-wave blocks per CU"));
        run<0>("ds_min_u64 (21 addr, 3-way)", d_idx, blocks, threads);
        run<7>("ds_min_u64 (64 addr, no conflict)", d_idx, blocks, threads);
        run<1>("ds_add_u32 (21 addr)", d_idx, blocks, threads);
        run<2>("ds_min_u32 (21 addr)", d_idx, blocks, threads);
        run<3>("ds_read_b64 dependent-free", d_idx, blocks, threads);
        run<4>("shfl f64 (2 bpermute)", d_idx, blocks, threads);
        run<5>("shfl i32 dependent chain", d_idx, blocks, threads);
        run<6>("fp64 add+min+mul dependent chain", d_idx, blocks, threads);
    }
    return 0;
}
