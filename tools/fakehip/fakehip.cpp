// TEST TOOL, not product: a stand-in for the part of the HIP runtime libltpl_hip.so's HOST code uses, so that this host code -- argument
// validation, packing into the staging arenas, buffer sizing, copy-in / copy-out, scatter into the caller's arrays -- can run in the
// GPU-less build container under AddressSanitizer / UBSan (tools/fakehip/build.sh links the host-only object of ltpl_hip.hip against this
// file instead of libamdhip64). "Device" memory is zero-initialised heap memory, copies are memcpy (so the sanitizer checks both
// ends of every transfer), streams and events are dummies, and a kernel launch does NOTHING: results are all-zero, i.e. "no action
// set" everywhere. What runs is every line of host code around the launches; what does not run is any device code. The product
// never loads this: it exists under tools/ only, and libltpl_hip.so itself refuses to work without a real device (ltpl_create).
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#include <map>
#include <random>
#include <mutex>
#include <atomic>

namespace {
std::mutex g_mu;
std::map<void*, size_t> g_host; // page-locked allocations (hipPointerGetAttributes; zero-copy outputs live here)
std::map<void*, size_t> g_dev;   // "device" allocations (FAKEHIP_GARBAGE: what a launch scribbles over)
std::atomic<long> g_launches{0};
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local CallCfg g_cfg[8];
thread_local int g_ncfg = 0;
}

extern "C" {

long fakehip_launch_count() { return g_launches; }
// failure injection: the n-th "device" allocation from now on fails (n <= 0: off). Error paths of ltpl_create and of the growing
// staging buffers run under the sanitizers this way.
static std::atomic<long> g_fail_malloc_in{0};
void fakehip_fail_malloc_after(long n) { g_fail_malloc_in = n; }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fakehip"; }

hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int)
{
    std::memset(p, 0, sizeof(*p));
    std::snprintf(p->name, sizeof(p->name), "fakehip (no device)");
    p->multiProcessorCount = 256;
    p->sharedMemPerBlock = 160 * 1024;
    p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    p->warpSize = 64;
    p->totalGlobalMem = (size_t)1 << 34;
    return hipSuccess;
}

hipError_t hipMalloc(void** p, size_t n)
{
    if (g_fail_malloc_in > 0 && --g_fail_malloc_in == 0) { *p = nullptr; return hipErrorOutOfMemory; }
    *p = std::calloc(n ? n : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[*p] = n;
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    { std::lock_guard<std::mutex> lk(g_mu); g_dev.erase(p); }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    *p = std::calloc(n ? n : 1, 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mu);
    g_host[*p] = n;
    return hipSuccess;
}
hipError_t hipHostFree(void* p)
{
    { std::lock_guard<std::mutex> lk(g_mu); g_host.erase(p); }
    std::free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
    std::memset(a, 0, sizeof(*a));
    std::lock_guard<std::mutex> lk(g_mu);
    // (only exact allocation starts are recognised -- all the library asks about)
    if (g_host.count(const_cast<void*>(p))) { a->type = hipMemoryTypeHost; return hipSuccess; }
    return hipErrorInvalidValue;
}

hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind)
{
    for (size_t r = 0; r < h; ++r) std::memcpy(static_cast<char*>(d) + r * dp, static_cast<const char*>(s) + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }

// kernel launch plumbing of the HIP host stubs
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream)
{
    if (g_ncfg < 8) g_cfg[g_ncfg] = CallCfg{grid, block, shmem, stream};
    ++g_ncfg;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream)
{
    --g_ncfg;
    if (g_ncfg >= 0 && g_ncfg < 8) { *grid = g_cfg[g_ncfg].grid; *block = g_cfg[g_ncfg].block; *shmem = g_cfg[g_ncfg].shmem; *stream = g_cfg[g_ncfg].stream; }
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void*, dim3 grid, dim3 block, void**, size_t shmem, hipStream_t)
{
    // launch geometry sanity (what the real runtime would refuse)
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 || shmem > 160 * 1024) {
        std::fprintf(stderr, "fakehip: invalid launch configuration grid (%u,%u,%u) block (%u,%u,%u) shmem %zu\n", grid.x, grid.y, grid.z,
                     block.x, block.y, block.z, shmem);
        return hipErrorInvalidConfiguration;
    }
    ++g_launches;
    // FAKEHIP_GARBAGE=<seed>: a launch overwrites every "device" allocation and every page-locked buffer with random words whose
    // magnitude is small often enough to pass for counts -- the host side must never turn a wrong device result into an out-of-bounds
    // access of its own (it may return nonsense or an error, not corrupt memory)
    static const char* garbage = std::getenv("FAKEHIP_GARBAGE");
    if (garbage) {
        static std::mt19937 rng((unsigned)std::atoi(garbage));
        std::lock_guard<std::mutex> lk(g_mu);
        auto scribble = [&](void* p, size_t n) {
            int* w = static_cast<int*>(p);
            for (size_t i = 0; i + 4 <= n; i += 4) {
                const unsigned r = rng();
                w[i / 4] = (r & 3u) == 0 ? (int)r : (r & 3u) == 1 ? (int)(r >> 8) % 5000 : (r & 3u) == 2 ? -(int)((r >> 8) % 100) : (int)(r >> 8) % 8;
            }
        };
        for (auto& a : g_dev) if (a.second <= ((size_t)64 << 20)) scribble(a.first, a.second);
        for (auto& a : g_host) if (a.second <= ((size_t)64 << 20)) scribble(a.first, a.second);
    }
    return hipSuccess;
}
// hipGraph (the LTPL_TICK_GRAPH form of the single tick): an executable graph is a list of recorded copies that hipGraphLaunch replays
// (kernel nodes do nothing here, like launches)
struct FakeNode { int kind; void* d; const void* s; size_t n; };
struct FakeGraph { std::vector<FakeNode*> nodes; };
hipError_t hipGraphCreate(hipGraph_t* g, unsigned) { *g = reinterpret_cast<hipGraph_t>(new FakeGraph()); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { FakeGraph* fg = reinterpret_cast<FakeGraph*>(g); for (FakeNode* n : fg->nodes) delete n; delete fg; return hipSuccess; }
hipError_t hipGraphAddMemcpyNode1D(hipGraphNode_t* node, hipGraph_t g, const hipGraphNode_t*, size_t, void* d, const void* s, size_t n, hipMemcpyKind)
{
    FakeNode* fn = new FakeNode{0, d, s, n}; reinterpret_cast<FakeGraph*>(g)->nodes.push_back(fn); *node = reinterpret_cast<hipGraphNode_t>(fn); return hipSuccess;
}
hipError_t hipGraphAddKernelNode(hipGraphNode_t* node, hipGraph_t g, const hipGraphNode_t*, size_t, const hipKernelNodeParams* kp)
{
    if (!kp || !kp->func || kp->gridDim.x == 0 || kp->blockDim.x == 0 || kp->sharedMemBytes > 160 * 1024) return hipErrorInvalidValue;
    FakeNode* fn = new FakeNode{1, nullptr, nullptr, 0}; reinterpret_cast<FakeGraph*>(g)->nodes.push_back(fn); *node = reinterpret_cast<hipGraphNode_t>(fn); return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) { *e = reinterpret_cast<hipGraphExec_t>(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipGraphExecMemcpyNodeSetParams1D(hipGraphExec_t, hipGraphNode_t node, void* d, const void* s, size_t n, hipMemcpyKind)
{
    FakeNode* fn = reinterpret_cast<FakeNode*>(node); fn->d = d; fn->s = s; fn->n = n; return hipSuccess;
}
hipError_t hipGraphExecKernelNodeSetParams(hipGraphExec_t, hipGraphNode_t, const hipKernelNodeParams* kp) { return kp && kp->func ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t)
{
    for (FakeNode* n : reinterpret_cast<FakeGraph*>(e)->nodes) { if (n->kind == 0 && n->n) std::memcpy(n->d, n->s, n->n); else if (n->kind == 1) ++g_launches; }
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* dummy = nullptr; return &dummy; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

}  // extern "C"
