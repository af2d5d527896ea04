// ltpl_hip.hip -- MI355X (gfx950 / CDNA4) backend of the graph_ltpl online hot path. C ABI: include/ltpl_hip.h.
//
// Seam (1) lives in paths_team.hpp: a team of NW wave64s plans one scenario (NW = 1 for batches, NW = 4 for single
// ticks). Seam (2) (velocity profiles) and the fused tick live in this file.
// The lattice (< 15 MB for Monteblanco) is uploaded once and is L2 / Infinity-Cache resident afterwards.
// Everything is IEEE fp64 and compiled with -ffp-contract=off so that masks, arg-mins and path costs are
// bit-identical to the NumPy arithmetic of the reference.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/ltpl_hip.h"
#include "planner_host.hpp"
#include "fleet_api.hpp"
#include "capsule.hpp"
#include "layer_grid.hpp"
#include "assembly_records.hpp"

#define WG_THREADS 256
#define PIPELINE_MIN_SCEN 64   // seam (1) alone: batches from this size on use one-wave teams (nw1_min_scen). The fused tick against the batch
                               // pipeline has its own, device-dependent threshold since round 6: ltpl_handle::pipeline_min_scen
#define NUM_WAVES 4
#define MAX_POS 192      // obstacle positions (own + predicted) per scenario
#define MAX_VEH 96       // vehicles per scenario
#define NFILT 4
#define F_PR 0
#define F_DEF 1
#define F_LEFT 2
#define F_RIGHT 3
#define D_PI 3.14159265358979323846

// ---------------------------------------------------------------------------------------------------------------------
// device-side views
// ---------------------------------------------------------------------------------------------------------------------
#define LTPL_SW_PAD 512      // sentinel entries behind the sweep-order edge tables (DevLat::sw_cost / sw_meta)
struct DevLat {
    int L, V, E, S, G;
    int mode;
    int closed;                       // GraphBase.closed; open tracks: the planning range is clamped to the last layer
    double min_plan_horizon, veh_width, sampled_resolution, lat_offset, vel_decrease_lat, veh_length;
    const int* layer_off; const int* rl_idx;
    const double* s_rl; const double* ref_x; const double* ref_y; const double* vel_rl;
    const double* node_x; const double* node_y; const double* vgoal;
    const int* in_ptr; const int* edge_src; const double* edge_cost; const double* edge_len; const int* samp_ptr;
    const double* sx; const double* sy; const double* spsi; const double* slen;
    const double* ssc;                // [S][2] (sin, cos) of spsi, tabulated at ltpl_create: the spline end slopes of a path need no sincos per path
    const double* glob_rl;
    const double* grx; const double* gry;   // x / y columns of glob_rl as contiguous arrays (coalesced scans)
    // derived at ltpl_create
    const int* rng_end;               // [L]     end layer of the planning range that starts in layer l
    const int* layer_ebase;           // [L + 1] first edge INTO layer l (= in_ptr[layer_off[l]]); [L] = E
    const unsigned char* edge_src8;   // [E]     edge_src as bytes (<= 255 nodes per layer)
    const unsigned char* edge_dst8;   // [E]     destination node index inside its layer
    const unsigned char* edge_rank8;  // [E]     rank of the edge among the in-edges of its destination
    const int* layer_degmax;          // [L]     largest in-degree of a node of layer l
    // SWEEP ORDER. Inside every layer transition the sweep, the obstacle mask and the LDS edge bitmap address the edges in a second order:
    // sorted by (in-edge rank, destination node) instead of CSC (destination, source). The lanes of a wave then hold edges into DIFFERENT
    // nodes, so the LDS atomics of the sweep (ds_min_u64 on the destination's frontier slot) meet at most two lanes per address; in CSC
    // order the ~5 (up to 21) in-edges of a node sit in consecutive lanes and every atomic serialises that many times. A transition's
    // edges occupy the same index range [layer_ebase[l], layer_ebase[l + 1]) in both orders.
    const double* sw_cost;            // [E + 1 + LTPL_SW_PAD] edge cost in sweep order; [E] = +inf (sentinel for the lanes beyond a transition)
    const unsigned* sw_meta;          // [same] in-edge rank | source node << 8 | destination node << 16, sweep order; [E] = 0 (the low half
                                      //         is the election key among equal candidates: the reference settles them by source node)
    const int* sw2csc;                // [E]     CSC edge id of sweep position p
    const int* csc2sw;                // [E]     sweep position of CSC edge e
    // track bounds per layer: bound1 = refline + normvec w_right, bound2 = refline - normvec w_left, centre = (b1 + b2) / 2
    const double* b1x; const double* b1y; const double* b2x; const double* b2y; const double* ctx; const double* cty;
    const float4* edge_cap;           // [2 E] IN SWEEP ORDER: capsule of the edge's samples in fp32: (Ax, Ay, ABx, ABy), (1 / |AB|^2, dev, hg2, first sample |
                                      //     #samples << 24 as bit pattern; #samples = 0: look samp_ptr up): chord between the first and the last
                                      //     sample, largest sample distance from it, squared half of the largest gap between consecutive sample
                                      //     projections -- two-sided conservative cull of the obstacle mask, the exact test is fp64
    float cull_slack;                 // bound on the fp32 rounding of the cull's distance (positions, chord, arithmetic)
    // closest-layer grid (layer_grid.hpp): per cell the <= 2 intervals of reference-line layers that can be closest to a point of the cell
    // (first layer, length, first layer, length; length -1 = scan all layers); null: phase 1 scans all layers (LTPL_NO_LAYER_GRID=1)
    const int4* lgrid; double lg_x0, lg_y0, lg_inv; int lg_nx, lg_ny;
    // PATH ASSEMBLY RECORDS (round 5): what the assembly of a path gathers per node / per edge, as ONE record each, so that the chain of
    // dependent global round trips behind the backtrack is node record -> edge record (it was layer_off -> in_ptr -> edge_src8 ->
    // samp_ptr -> sx / sy / ssc: five round trips of ~600 cycles per path on a wave that has nothing else to do meanwhile)
    const int4* node_rec;             // [V] first in-edge (CSC id), then the source nodes of the node's first 12 in-edges as bytes (0xff = none)
    const double* edge_rec;           // [E][LTPL_EDGE_REC] (first sample | #samples << 32 as bit pattern, edge length, x, y of the first sample,
                                      //     x, y of the last sample, sin, cos of the first sample's heading, sin, cos of the last sample's)
};

struct DevPathsIn {
    int n_scen, n_w_last;
    const double* w_last;
    const int* start_layer; const int* start_node; const int* flags; const int* last_action; const int* const_closest;
    const double* psi_s;
    const int* veh_off; const int* pos_off; const double* veh_radius; const double* pos_x; const double* pos_y;
    const int* zone_off; const int* zone_gid;
    const int* n_last; const int* last_layer; const int* last_node;
    // BLOCK -> SCENARIO ORDER (round 6; one-wave batches, null otherwise: block b plans scenario b): order[b] = the scenario of block b, the
    // scenarios sorted by start layer. Blocks are dispatched in index order, so at any time the whole chip works on one region of the track:
    // the waves of a moment share their lattice lines, and the velocity jobs -- numbered in the order their paths complete -- come in runs of
    // similar length, so that the 64 profiles of a lane-kernel wave end together (lane kernel alone 0.265 against 0.282 ms; +2 % ticks/s,
    // profiles/r06r_ab_order.txt). Tried first and NOT kept: the sorted sequence dealt out in eighths, one per XCD (block b runs on XCD b % 8,
    // each XCD with its own 4 MiB L2 and an eighth of the 6 MB lattice to itself) -- 37.8 against 40.1 M ticks/s: the eighths of a track are
    // not equal work (planning ranges of 11 to 29 layers), and the lattice was not missing in L2 to begin with (path kernel alone: unchanged by
    // the order that won). Outputs are indexed by scenario: results do not depend on the order.
    const int* order;
};

// completion signal of the latency path: the LAST block of a launch writes a sequence number into page-locked host memory after
// every block has pushed its (zero-copy) outputs out with a system-scope fence; the host polls that word instead of synchronising
// the stream (launch + poll: 11 us, launch + hipStreamSynchronize: 15 us on the round-2 box, tools/ubench/launch). All null: no signal.
struct DoneSignal { unsigned* host_flag; unsigned* dev_count; unsigned seq; };

// called by ALL threads of every block as the last thing the kernel does
__device__ __forceinline__ void signal_done(const DoneSignal& d)
{
    if (!d.host_flag) return;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd(d.dev_count, 1u);
        if (prev == gridDim.x - 1) {
            *d.dev_count = 0u;                                   // the next launch on the stream starts from zero
            __hip_atomic_store(d.host_flag, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Operand records of the batch velocity stage (|kappa|, element length per path row): an fp64 PAIR, one 16-byte load per row and lane,
// 1 / |kappa| in fp64 -- the reference is IEEE fp64 throughout (VpForwardBackward.py:194-227) and so is every operand, state and output
// of this stage (element-wise vx error against the oracle ~1e-13). -DLTPL_VEL_F32_OPERANDS builds the rounds-3..5 form (fp32 pair,
// v_rcp_f32: +6.5 % ticks/s, element-wise vx error up to 1e-5 -- profiles/r04f_vel_precision.txt); not shipped, not tested.
#ifndef LTPL_VEL_F32_OPERANDS
typedef double ke_scalar;
typedef double ke_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ke_t make_ke(double k, double e) { ke_t v; v.x = k; v.y = e; return v; }
// 1 / |kappa|: hardware reciprocal + ONE Newton step. Measured on gfx950 over 4 M values of |kappa| between 1e-12 and 4e3 1/m
// (tools/ubench/rcp/rcp_f64.hip, profiles/r06c_rcp_f64.txt): v_rcp_f64 alone 4.6e-8 relative, one step 2.2e-15 (20 ulp), two steps correctly
// rounded. The limit speed ay / |kappa| enters results that are compared at 1e-5 (element-wise error of vx against the oracle ~1e-13 with either);
// the second step was two dependent fma on the lane kernels' serial chain: +1.7 % ticks/s without it (same-box A/B, profiles/r06b_ab_bench.txt).
__device__ __forceinline__ double ke_rcp(const ke_t& r)        // (0 -> NaN: treated as +inf by the callers' fmin)
{
    double q = __builtin_amdgcn_rcp(r.x);
    q = fma(fma(-r.x, q, 1.0), q, q);
    return r.x == 0.0 ? (double)INFINITY : q;
}
#else
typedef float ke_scalar;
typedef float2 ke_t;
__device__ __forceinline__ ke_t make_ke(double k, double e) { return make_float2((float)k, (float)e); }
__device__ __forceinline__ double ke_rcp(const ke_t& r) { return (double)__builtin_amdgcn_rcpf(r.x); }     // 1 ulp in fp32, inf on straights
#endif

struct DevPathsOut {
    int cap_nodes, cap_pts;
    int* end_layer; int* closest_obj_index; int* closest_obj_node; int* n_actions;
    int* action_id; int* valid; int* reduced; int* goal_layer; int* n_nodes; int* n_pts; int* n_ties;
    int* nodes; int* node_idx; double* coeff; double* path_param;
    ke_t* vke;                       // optional tiled plane (|kappa|, element length) as fp64 pairs for the batch velocity stage
    double* vxy;                     // optional tiled plane (x, y) of the FOLLOW jobs' path points (tile = follow job index): k_follow_prep
    // job compaction of the batch velocity stage (all nullptr outside the pipeline): every valid path takes a job index
    // from a counter of its class (0 = generic forward-backward profile, 1 = follow); its planes are tiled by JOB, so
    // the lanes of a velocity wave (64 consecutive jobs of one class) are all busy and equally long
    int* job_cnt;                    // [0] generic jobs, [1] follow jobs, [2] largest n_pts of the launch (row chunks beyond it exit at once)
    int2* job_slot;                  // [n_slots_pad] generic jobs -> (slot, n_pts), then [n_scen_pad] follow jobs: one load gives a velocity
                                     // wave slot AND length of its job (two dependent round trips less per block)
    int n_slots_pad;                 // tile index of follow job j = n_slots_pad + j (rows of the blocked planes: cap_pts rounded up to 8)
    DoneSignal done;                 // latency path only (k_paths / k_tick launched for a few scenarios)
};


// Timing / fault-injection switches (LTPL_ABLATE, LTPL_EXP_SKIP, LTPL_LDS_POISON, LTPL_SCRATCH_POISON, LTPL_DEBUG_TIMING, LTPL_DEBUG_OCC)
// only exist in the EXPERIMENT build (-DLTPL_EXPERIMENT -> libltpl_hip_exp.so, tools/ and the stale-LDS test): the release library has
// no code path that skips work, alters memory or instruments kernels on an environment variable's say-so.
#define DBG_SLOTS 64
#ifdef LTPL_EXPERIMENT
__device__ __forceinline__ void dbg_stamp(long long* dbg, int k)
{
    if (dbg && (threadIdx.x & 63) == 0 && blockIdx.x < 256) dbg[(size_t)blockIdx.x * DBG_SLOTS + (threadIdx.x >> 6) * 16 + k] = clock64();
}
#else
__device__ __forceinline__ void dbg_stamp(long long*, int) {}
#endif

// LTPL_DEBUG_TIMING for the lane kernel: `row` selects one of 256 sample rows (generic / follow / unconstrained blocks)
#ifdef LTPL_EXPERIMENT
__device__ __forceinline__ void vl_stamp(long long* dbg, int row, int k)
{
    if (dbg && (threadIdx.x & 63) == 0 && row >= 0 && row < 256) dbg[(size_t)row * DBG_SLOTS + k] = clock64();
}
#else
__device__ __forceinline__ void vl_stamp(long long*, int, int) {}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// wave helpers (wave64)
// ---------------------------------------------------------------------------------------------------------------------
// CROSS-LANE MOVES ON THE VECTOR ALU (round 5). `__shfl_xor` / `__shfl_up` compile to ds_bpermute_b32: an LDS-pipe instruction with an LDS
// round trip of latency per step, in kernels whose LDS pipe is a co-limiter and whose waves spend ~46 % of their cycles in s_waitcnt
// (275 of the 907 static LDS-pipe instructions of the round-4 batch path kernel were ds_bpermute). The reductions and scans below exchange
// through DPP row operations (partner at distance 1, 2, 4, 8 inside a row of 16 lanes) and the gfx950 row / half swaps
// (v_permlane16_swap, v_permlane32_swap: distance 16 and 32) instead: no LDS instruction, no lgkmcnt wait.
// ALL 64 LANES MUST BE ACTIVE at the call (wave-uniform control flow): a DPP move from an inactive lane delivers no data.
// (the round-4 `__shfl_xor` forms: tools/experiments/r05_lost_switches.patch)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the DPP / v_permlane16_swap / v_permlane32_swap reductions below are gfx950 code: build with --offload-arch=gfx950 (the round-4 shuffle forms: tools/experiments/r05_lost_switches.patch)"
#endif
#define DPP_XOR1 0xB1                   // quad_perm:[1,0,3,2]
#define DPP_XOR2 0x4E                   // quad_perm:[2,3,0,1]
#define DPP_HALF_MIRROR 0x141           // lane i <-> 7 - i  of every 8: the partner at distance 4 once the quads are uniform
#define DPP_MIRROR 0x140                // lane i <-> 15 - i of every 16: the partner at distance 8 once the halves of 8 are uniform
#define DPP_ROR8 0x128                  // row_ror:8 = lane i ^ 8
#define DPP_SHL4 0x104                  // row_shl:4 (lane i <- lane i + 4)
#define DPP_SHR(n) (0x110 + (n))        // row_shr:n (lane i <- lane i - n inside the row)
#define DPP_BCAST15 0x142               // lane 15 of every row -> the next row
#define DPP_BCAST31 0x143               // lane 31 -> rows 2 and 3
typedef unsigned lane_pair_t __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v)
{
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
// the value of lane (i ^ M), M = 1, 2, 4, 8, for ANY lane contents (the mirror forms above need uniform sub-groups)
template <int M> __device__ __forceinline__ int xor_i32(int v)
{
    if constexpr (M == 1) return dpp_i32<DPP_XOR1>(v);
    else if constexpr (M == 2) return dpp_i32<DPP_XOR2>(v);
    else if constexpr (M == 4) {           // banks (= quads of a row) 0, 2 take the quad above, banks 1, 3 the quad below
        const int up = __builtin_amdgcn_update_dpp(v, v, DPP_SHL4, 0xf, 0x5, false);
        return __builtin_amdgcn_update_dpp(up, v, DPP_SHR(4), 0xf, 0xa, false);
    } else { static_assert(M == 8, "distance inside a row"); return dpp_i32<DPP_ROR8>(v); }
}
template <int M> __device__ __forceinline__ double xor_f64(double v) { return __hiloint2double(xor_i32<M>(__double2hiint(v)), xor_i32<M>(__double2loint(v))); }
// distance 16 / 32: with both operands = v the swap returns, in every lane i, the pair {v[i], v[i ^ M]} (in an order that depends on the
// lane's half) -- symmetric combinations (min, lexicographic min) need not know which is which
template <int M> __device__ __forceinline__ lane_pair_t swap_pair(unsigned v)
{
    if constexpr (M == 16) return __builtin_amdgcn_permlane16_swap(v, v, false, false);
    else { static_assert(M == 32, "row / half swap"); return __builtin_amdgcn_permlane32_swap(v, v, false, false); }
}
template <int M> __device__ __forceinline__ void swap_pair_f64(double v, double& a, double& b)
{
    const lane_pair_t lo = swap_pair<M>((unsigned)__double2loint(v)), hi = swap_pair<M>((unsigned)__double2hiint(v));
    a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
}

// minimum of a double over the wave (no NaN among the keys), in every lane. The first key decides nearly always, so the callers reduce it
// alone and fall back to wave_min3 only when it is attained more than once.
__device__ __forceinline__ double wave_min_f64(double k)
{
    k = __builtin_fmin(k, dpp_f64<DPP_XOR1>(k));
    k = __builtin_fmin(k, dpp_f64<DPP_XOR2>(k));
    k = __builtin_fmin(k, dpp_f64<DPP_HALF_MIRROR>(k));
    k = __builtin_fmin(k, dpp_f64<DPP_MIRROR>(k));
    double a, b;
    swap_pair_f64<16>(k, a, b); k = __builtin_fmin(a, b);
    swap_pair_f64<32>(k, a, b); k = __builtin_fmin(a, b);
    return k;
}
__device__ __forceinline__ int wave_min_i32(int k)
{
    k = min(k, dpp_i32<DPP_XOR1>(k));
    k = min(k, dpp_i32<DPP_XOR2>(k));
    k = min(k, dpp_i32<DPP_HALF_MIRROR>(k));
    k = min(k, dpp_i32<DPP_MIRROR>(k));
    lane_pair_t r = swap_pair<16>((unsigned)k); k = min((int)r.x, (int)r.y);
    r = swap_pair<32>((unsigned)k); k = min((int)r.x, (int)r.y);
    return k;
}
__device__ __forceinline__ int wave_max_i32(int k)
{
    k = max(k, dpp_i32<DPP_XOR1>(k));
    k = max(k, dpp_i32<DPP_XOR2>(k));
    k = max(k, dpp_i32<DPP_HALF_MIRROR>(k));
    k = max(k, dpp_i32<DPP_MIRROR>(k));
    lane_pair_t r = swap_pair<16>((unsigned)k); k = max((int)r.x, (int)r.y);
    r = swap_pair<32>((unsigned)k); k = max((int)r.x, (int)r.y);
    return k;
}
// maximum over every group of eight lanes (lanes 8 k .. 8 k + 7), in each of its lanes
__device__ __forceinline__ int oct_max_i32(int k)
{
    k = max(k, dpp_i32<DPP_XOR1>(k));
    k = max(k, dpp_i32<DPP_XOR2>(k));
    return max(k, dpp_i32<DPP_HALF_MIRROR>(k));
}
// lexicographic minimum over (k1, k2, idx) / (k, idx): one exchange step at distance M (true partner: the keys differ from lane to lane)
template <int M> __device__ __forceinline__ void min3_step(double& k1, double& k2, int& idx)
{
    if constexpr (M < 16) {
        const double o1 = xor_f64<M>(k1), o2 = xor_f64<M>(k2); const int oi = xor_i32<M>(idx);
        const bool take = (o1 < k1) || (o1 == k1 && ((o2 < k2) || (o2 == k2 && oi < idx)));
        if (take) { k1 = o1; k2 = o2; idx = oi; }
    } else {
        double a1, b1, a2, b2; swap_pair_f64<M>(k1, a1, b1); swap_pair_f64<M>(k2, a2, b2);
        const lane_pair_t ri = swap_pair<M>((unsigned)idx);
        const int ai = (int)ri.x, bi = (int)ri.y;
        const bool take_b = (b1 < a1) || (b1 == a1 && ((b2 < a2) || (b2 == a2 && bi < ai)));
        k1 = take_b ? b1 : a1; k2 = take_b ? b2 : a2; idx = take_b ? bi : ai;
    }
}
template <int M> __device__ __forceinline__ void min2_step(double& k, int& idx)
{
    if constexpr (M < 16) {
        const double o = xor_f64<M>(k); const int oi = xor_i32<M>(idx);
        const bool take = (o < k) || (o == k && oi < idx);
        if (take) { k = o; idx = oi; }
    } else {
        double a, b; swap_pair_f64<M>(k, a, b);
        const lane_pair_t ri = swap_pair<M>((unsigned)idx);
        const int ai = (int)ri.x, bi = (int)ri.y;
        const bool take_b = (b < a) || (b == a && bi < ai);
        k = take_b ? b : a; idx = take_b ? bi : ai;
    }
}
__device__ __forceinline__ void wave_min3(double& k1, double& k2, int& idx)
{
    min3_step<1>(k1, k2, idx); min3_step<2>(k1, k2, idx); min3_step<4>(k1, k2, idx); min3_step<8>(k1, k2, idx);
    min3_step<16>(k1, k2, idx); min3_step<32>(k1, k2, idx);
}
// the same for (key, index) pairs: the closest-point searches carry no second key
__device__ __forceinline__ void wave_min2(double& k, int& idx)
{
    min2_step<1>(k, idx); min2_step<2>(k, idx); min2_step<4>(k, idx); min2_step<8>(k, idx); min2_step<16>(k, idx); min2_step<32>(k, idx);
}
// ... over the lanes that agree in their low bits (lane & (np2 - 1)): the steps at distance >= np2 only (np2 a power of two, uniform)
__device__ __forceinline__ void wave_min2_from(double& k, int& idx, int np2)
{
    if (np2 <= 1) min2_step<1>(k, idx);
    if (np2 <= 2) min2_step<2>(k, idx);
    if (np2 <= 4) min2_step<4>(k, idx);
    if (np2 <= 8) min2_step<8>(k, idx);
    if (np2 <= 16) min2_step<16>(k, idx);
    if (np2 <= 32) min2_step<32>(k, idx);
}

// exclusive prefix sum of an int over the wave (+ the total): Hillis-Steele inside the rows (row_shr, lanes without a source add 0), then
// the row totals through the two broadcast forms
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int& total)
{
    (void)lane;
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, DPP_SHR(1), 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_SHR(2), 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_SHR(4), 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_SHR(8), 0xf, 0xf, true);
    x += __builtin_amdgcn_update_dpp(0, x, DPP_BCAST15, 0xa, 0xf, false);      // rows 1 and 3 += total of the row below
    x += __builtin_amdgcn_update_dpp(0, x, DPP_BCAST31, 0xc, 0xf, false);      // rows 2 and 3 += total of rows 0 and 1
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

// Self-check of the cross-lane helpers above: every helper against a plain loop over the wave's values in LDS. err[0] = number of lanes that
// disagree. Part of the create-time self-test of the product library since round 6 (wave_ops_selftest: the helpers are DPP / permlane-swap code
// that only exists for gfx950 and assumes 64 active lanes); tests/test_gpu_wave_ops.py drives it with its own data through the experiment build.
__global__ __launch_bounds__(64) void k_exp_wave_ops(const double* vals, const int* ivals, int rounds, int* err)
{
    __shared__ double sd[64], sd2[64];
    __shared__ int si[64];
    const int lane = threadIdx.x;
    int bad = 0;
    for (int r = 0; r < rounds; ++r) {
        const double d = vals[(size_t)r * 128 + lane], d2 = vals[(size_t)r * 128 + 64 + lane];
        const int iv = ivals[(size_t)r * 64 + lane];
        sd[lane] = d; sd2[lane] = d2; si[lane] = iv;
        __syncthreads();
        double rmin = INFINITY; int imin = 0x7fffffff, imax = -0x7fffffff - 1, isum = 0, ipre = 0, omax = -0x7fffffff - 1;
        double a1 = INFINITY, a2 = INFINITY; int ai = 0x7fffffff;          // lexicographic minimum over (d, d2, iv)
        double b1 = INFINITY; int bi = 0x7fffffff;                          // ... over (d, iv)
        for (int l = 0; l < 64; ++l) {
            rmin = sd[l] < rmin ? sd[l] : rmin; imin = si[l] < imin ? si[l] : imin; imax = si[l] > imax ? si[l] : imax;
            isum += si[l] & 0xff; if (l < lane) ipre += si[l] & 0xff;
            if ((l >> 3) == (lane >> 3)) omax = si[l] > omax ? si[l] : omax;
            if (sd[l] < a1 || (sd[l] == a1 && (sd2[l] < a2 || (sd2[l] == a2 && si[l] < ai)))) { a1 = sd[l]; a2 = sd2[l]; ai = si[l]; }
            if (sd[l] < b1 || (sd[l] == b1 && si[l] < bi)) { b1 = sd[l]; bi = si[l]; }
        }
        if (wave_min_f64(d) != rmin) ++bad;
        if (wave_min_i32(iv) != imin) ++bad;
        if (wave_max_i32(iv) != imax) ++bad;
        if (oct_max_i32(iv) != omax) ++bad;
        { int tot; const int ex = wave_excl_scan(iv & 0xff, lane, tot); if (ex != ipre || tot != isum) ++bad; }
        { double k1 = d, k2 = d2; int ix = iv; wave_min3(k1, k2, ix); if (k1 != a1 || k2 != a2 || ix != ai) ++bad; }
        { double k = d; int ix = iv; wave_min2(k, ix); if (k != b1 || ix != bi) ++bad; }
        for (int np2 = 1; np2 <= 64; np2 <<= 1) {                            // segment merge of phase 1: lanes with equal (lane & (np2 - 1))
            double k = d; int ix = iv; wave_min2_from(k, ix, np2);
            double c1 = INFINITY; int ci = 0x7fffffff;
            for (int l = lane & (np2 - 1); l < 64; l += np2) if (sd[l] < c1 || (sd[l] == c1 && si[l] < ci)) { c1 = sd[l]; ci = si[l]; }
            if (k != c1 || ix != ci) ++bad;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(err, 1);
}

// LDS hand-over between the lanes of ONE wave: DS operations of a wave execute in order, so only the compiler has to be
// kept from reordering (wavefront-scope fences emit no s_waitcnt vmcnt and do not serialise outstanding global loads)
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double normalize_psi_dev(double psi)
{
    double sgn = (psi > 0.0) ? 1.0 : ((psi < 0.0) ? -1.0 : 0.0);
    double out = sgn * fmod(fabs(psi), 2.0 * D_PI);
    if (out >= D_PI) out -= 2.0 * D_PI;
    else if (out < -D_PI) out += 2.0 * D_PI;
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// the path kernel (seam 1)
// ---------------------------------------------------------------------------------------------------------------------
// Result of the path stage for the calling wave (waves 0..2 own one action slot each).
struct WavePath { int valid; int n_pts; int n_nodes; int name; int reduced; int goal_layer; int end_node; };

#include "paths_team.hpp"

// Waves per SIMD of the one-wave batch kernel: the register budget must hold the kernel WITHOUT register spills (builds of
// this kernel that spilled VGPRs to scratch produced wrong parents on gfx950; __graft_entry__.build() rejects such builds).
// Runtime plan (any lattice): 2 waves per SIMD = 256 VGPRs. Compile-time plan (three register chunks of edges): 4 waves per
// SIMD = 128 VGPRs.
#ifndef LTPL_RT_WAVES
#define LTPL_RT_WAVES 2              // -DLTPL_RT_WAVES=4 builds the spilling variant studied in tools/ubench/spill_study
#endif
template <int NW, class P>
__global__ __launch_bounds__(64 * NW, (NW == 1 ? (P::fixed ? 4 : LTPL_RT_WAVES) : 1)) void k_paths(DevLat lat, DevPathsIn in, DevPathsOut out, TeamLds lp)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ TeamShared ts;
    if constexpr (P::fixed) {
        // compile-time plan classes: the argument structs are read from the kernarg segment phase by phase (karg_reload, paths_team.hpp)
        // instead of being held in scalar registers -- and spilled into vector-register lanes -- from entry to last use; the by-value
        // parameters only define the kernarg layout
        const PathsKArgs* ka = paths_kargs();
        (void)team_paths_body<NW, P, 1>(ka->lat, ka->in, ka->out, ka->lp, smem, ts, nullptr, nullptr, nullptr, nullptr);
        if constexpr (NW != 1) signal_done(ka->out.done);
        return;
    }
    (void)team_paths_body<NW, P>(lat, in, out, lp, smem, ts, nullptr, nullptr, nullptr, nullptr);
    if constexpr (NW != 1) signal_done(out.done);          // (only the four-wave latency form is launched with a completion word)
}
typedef PlanFx<32, 32, 1> PlanA;      // <= 32 nodes per layer, <= 31 layers of planning range (Monteblanco, stock parameters)
typedef PlanFx<32, 40, 1> PlanB;      // <= 32 nodes per layer, <= 39 layers (synthetic C3 oval; zalazone, lvms, modena of the reference's tracks)
typedef PlanFx<48, 32, 1> PlanC;      // <= 48 nodes per layer, <= 31 layers (berlin: 40 nodes per layer -- round 3 ran it on the 204-register PlanRt kernel)
typedef PlanFx<32, 32, NUM_WAVES> PlanA4;   // class A for the four-wave latency kernels (k_paths<4>, k_tick)


// ---------------------------------------------------------------------------------------------------------------------
// velocity stage (seam 2): tph.calc_vel_profile / calc_vel_profile_brake / calc_vel_profile_follow on the device
// ---------------------------------------------------------------------------------------------------------------------
// One wave64 owns one profile. Vectorisable parts (curvature -> lateral-limit speed, run-start detection, final
// sqrt, acceleration) are spread over the lanes; the forward / backward recurrences are inherently sequential scans
// and run on lane 0 over LDS-resident arrays. The recurrence state is w = v^2, which removes the sqrt and the
// divisions from the dependent chain (v_{i+1}^2 = v_i^2 + 2 a_x(v_i) ds_i); comparisons happen in w as well (sqrt is
// monotone). Differences to the reference's v-state arithmetic are rounding-level (~1e-16 relative).
struct DevVelParams {
    double e, inv_e, drag_m, len_veh, v_max;
    int n_axm, ctrl;
    const double* axm;          // [n_axm * 2] in device memory
    double c_p, k_p, k_d, tan_w;
};

// per-wave LDS scratch of the velocity stage; every array holds cap + 1 doubles
struct VelScratch {
    double* w;        // profile state (v^2)
    double* kabs;     // |kappa|
    double* el;       // element lengths
    double* gax;      // per-point longitudinal limit (nullptr -> constant)
    double* gay;      // per-point lateral limit      (nullptr -> constant)
    double* igay;     // 1 / gay (division-free recurrence)
    double* axm;      // LDS copy of ax_max_machines rows [v, ax] (<= 64 rows)
    long long* dbg;
    double* s;        // arc length (cap + 1)
    double* wb;       // ego brake profile (follow)
    double* wc;       // scratch profile (follow)
    unsigned char* start;   // run-start flags
    double* chunk;    // 128 doubles: staging of the global race line for the opponent brake scan
    int cap;
};

// np.interp(v, axm[:, 0], axm[:, 1]) on the LDS copy of the table
__device__ __forceinline__ double interp_axm(double v, const double* t, int n)
{
    if (n == 1 || v <= t[0]) return t[1];
    if (v >= t[2 * (n - 1)]) return t[2 * (n - 1) + 1];
    int j = 0;
    while (j + 1 < n && t[2 * (j + 1)] <= v) ++j;
    const double x0 = t[2 * j], x1 = t[2 * (j + 1)], f0 = t[2 * j + 1], f1 = t[2 * (j + 1) + 1];
    return (f1 - f0) / (x1 - x0) * (v - x0) + f0;
}

// tire share of tph calc_ax_poss: ax_max * (1 - (ay_used / ay_max)^e)^(1/e), 0 when the radicand is not positive.
// EM selects the exponent at compile time (1: e == 1, 2: e == 2, 0: general pow) so that the recurrences below are
// straight-line code: with one wave per profile every taken branch is a pipeline refill nobody hides.
template <int EM>
__device__ __forceinline__ double tire_avail(double q, double ax_max, const DevVelParams& p)
{
    if constexpr (EM == 1) {
        const double rad = 1.0 - q;
        return rad > 0.0 ? ax_max * rad : 0.0;
    } else if constexpr (EM == 2) {
        const double rad = 1.0 - q * q;
        return rad > 0.0 ? ax_max * sqrt(rad) : 0.0;
    } else {
        const double rad = 1.0 - pow(q, p.e);
        return rad > 0.0 ? ax_max * pow(rad, p.inv_e) : 0.0;
    }
}

#define VMODE_ACCEL_FORW 0
#define VMODE_DECEL_FORW 1
#define VMODE_DECEL_BACKW 2

// tph calc_ax_poss in w = v^2; kq = |kappa| / ay_max (so that ay_used / ay_max = w * kq); axm1 = machine limit when the
// table has a single row (AXM1), otherwise the LDS copy of the table is interpolated at v = sqrt(w)
template <int EM, bool AXM1, int MODE>
__device__ __forceinline__ double ax_poss_w(double w, double kq, double ax_max, const DevVelParams& p,
                                            const double* axm_tab, double axm1)
{
    const double q = w * kq;
    double ax = tire_avail<EM>(q, fabs(ax_max), p);
    if constexpr (MODE == VMODE_ACCEL_FORW) {
        const double axm = AXM1 ? axm1 : interp_axm(sqrt(w), axm_tab, p.n_axm);
        ax = ax < axm ? ax : axm;
        return ax - w * p.drag_m;
    } else if constexpr (MODE == VMODE_DECEL_FORW) {
        return -ax - w * p.drag_m;
    } else {
        return ax + w * p.drag_m;
    }
}

// lane l <- lane l - 1 across the whole wave64 (DPP wave_shr:1); lane 0 receives `first`
__device__ __forceinline__ double wave_shr1_f64(double v, double first)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(first), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(first), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_shr1_i32(int v, int first)
{
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

// one sweep of tph __solver_fb_acc_profile on w[0..n). Backward sweeps address the profile, curvature and element
// lengths mirrored but -- restated quirk of the reference solver -- the gg limits unmirrored.
// The recurrence is sequential, but everything except the state is known up front. SYSTOLIC form: the wave handles 64
// steps at a time, lane l holds the operands of step base + l (for exponent 1 and a constant machine limit: the affine
// coefficients); in every iteration ALL lanes evaluate their step on the state handed over by the lane below (DPP
// wave_shr:1, no LDS, no broadcast), so after iteration l lane l holds the final value of its step. One iteration is the
// bare dependent chain: ~a dozen VALU instructions.
template <int EM, bool AXM1, bool GGARR, bool BACK>
__device__ __forceinline__ void fb_sweep(int n, const VelScratch& vs, double cax, double cay, const DevVelParams& p, double vmax2, int lane)
{
    if (n < 2) return;
    // run starts: first index of every run of positive differences of the (mirrored) profile BEFORE the sweep
    for (int i = lane; i < n - 1; i += 64) {
        const int a = BACK ? n - 1 - i : i, b = BACK ? n - 2 - i : i + 1;
        const bool acc = vs.w[b] - vs.w[a] > 0.0;
        bool prev = false;
        if (i > 0) { const int a0 = BACK ? n - i : i - 1, b0 = BACK ? n - 1 - i : i; prev = vs.w[b0] - vs.w[a0] > 0.0; }
        vs.start[i] = (acc && !prev) ? 1 : 0;
    }
    wave_sync_lds();
    constexpr int MODE = BACK ? VMODE_DECEL_BACKW : VMODE_ACCEL_FORW;
    constexpr bool AFFINE = EM == 1 && AXM1;
    const double icay = 1.0 / cay, axm1 = vs.axm[1], dm = p.drag_m;
    double wi = vs.w[BACK ? n - 1 : 0];                   // state entering the chunk (uniform)
    int active = 0;
    for (int base = 0; base < n - 1; base += 64) {
        const int cnt = (n - 1 - base) < 64 ? (n - 1 - base) : 64;
        const int ic = (lane < cnt) ? base + lane : base + cnt - 1;
        const int Pa = BACK ? n - 1 - ic : ic, Pb = BACK ? n - 2 - ic : ic + 1, Ei = BACK ? n - 2 - ic : ic;
        // operands of step ic (the gg limits are indexed by the step, the rest by the mirrored point)
        const double kq_i = vs.kabs[Pa] * (GGARR ? vs.igay[ic] : icay), kq_n = vs.kabs[Pb] * (GGARR ? vs.igay[ic + 1] : icay);
        const double ax_i = GGARR ? fabs(vs.gax[ic]) : fabs(cax), ax_n = GGARR ? fabs(vs.gax[ic + 1]) : fabs(cax);
        const double e_i = vs.el[Ei], wold = vs.w[Pb];
        const int st = vs.start[ic];
        // affine coefficients (state independent):
        //   forward : w' = min(max(T, Z), M),  T = w A1 + B1, Z = w A0, M = w A0 + B2
        //   backward: w' = max(T, Z), then the look-ahead with the limits of the next point (t0 = wn C0 + w, t1 = wn C1 + w + D1)
        const double te = 2.0 * e_i;
        const double A0 = BACK ? 1.0 + te * dm : 1.0 - te * dm, A1 = A0 - te * (ax_i * kq_i), B1 = te * ax_i, B2 = te * axm1;
        const double C0 = te * dm, C1 = C0 - te * (ax_n * kq_n), D1 = te * ax_n;
        double w_in = wi, wnext = wold;
        int act_in = active, act_out = 0;
        // four steps per loop iteration: steps beyond cnt recompute final values from final inputs (idempotent), nothing is stored for them
        auto sys_step = [&]() {
            const bool act = (act_in | st) != 0;
            double wn;
            if constexpr (AFFINE) {
                if constexpr (!BACK) {
                    const double Z = A0 * w_in, T = fma(A1, w_in, B1), M = fma(A0, w_in, B2);
                    wn = fmax(fmin(fmax(T, Z), M), 0.0);
                } else {
                    const double Z = A0 * w_in, T = fma(A1, w_in, B1);
                    wn = fmax(fmax(T, Z), 0.0);
                    const double t0 = fma(C0, wn, w_in), t1 = fma(C1, wn, w_in + D1);
                    wn = fmin(fmax(fmax(t1, t0), 0.0), wn);
                }
            } else {
                const double acur = ax_poss_w<EM, AXM1, MODE>(w_in, kq_i, ax_i, p, vs.axm, axm1);
                wn = w_in + 2.0 * acur * e_i;
                wn = wn < 0.0 ? 0.0 : wn;
                if constexpr (BACK) {
                    const double anext = ax_poss_w<EM, AXM1, MODE>(wn, kq_n, ax_n, p, vs.axm, axm1);
                    double wt = w_in + 2.0 * anext * e_i;
                    wt = wt < 0.0 ? 0.0 : wt;
                    wn = wt < wn ? wt : wn;
                }
            }
            wnext = (act && (wn < wold)) ? wn : wold;
            act_out = (act && !(wn > vmax2)) ? 1 : 0;
            // hand the state to the next lane; lane 0 keeps the state entering the chunk
            w_in = wave_shr1_f64(wnext, wi);
            act_in = wave_shr1_i32(act_out, active);
                };
        for (int it = 0; it < cnt; it += 4) { sys_step(); sys_step(); sys_step(); sys_step(); }
        if (lane < cnt) vs.w[Pb] = wnext;
        wi = readlane_f64(wnext, cnt - 1);
        active = __builtin_amdgcn_readlane(act_out, cnt - 1);
        wave_sync_lds();
    }
}

// tph.calc_vel_profile(closed=False): vs.kabs / el / (gax, gay) hold the inputs, result in vs.w (as v^2)
template <int EM, bool AXM1, bool GGARR>
__device__ __forceinline__ void fb_profile(int n, const VelScratch& vs, double cax, double cay, const DevVelParams& p, double v_max,
                           double v_start, bool has_v_end, double v_end, int lane)
{
    if (v_start < 0.0) v_start = 0.0;
    if (has_v_end && v_end < 0.0) v_end = 0.0;
    const double vmax2 = v_max * v_max;
    for (int i = lane; i < n; i += 64) {
        const double ay = GGARR ? vs.gay[i] : cay;
        double w = ay / vs.kabs[i];                 // ay * radius; kappa == 0 -> inf
        if (!(w < vmax2)) w = vmax2;
        if (i == 0 && w > v_start * v_start) w = v_start * v_start;
        vs.w[i] = w;
    }
    wave_sync_lds();
    dbg_stamp(vs.dbg, 8);
    fb_sweep<EM, AXM1, GGARR, false>(n, vs, cax, cay, p, vmax2, lane);
    dbg_stamp(vs.dbg, 9);
    if (lane == 0 && has_v_end && vs.w[n - 1] > v_end * v_end) vs.w[n - 1] = v_end * v_end;
    wave_sync_lds();
    fb_sweep<EM, AXM1, GGARR, true>(n, vs, cax, cay, p, vmax2, lane);
    dbg_stamp(vs.dbg, 10);
}

// tph.calc_vel_profile_brake on LDS arrays: out[0..n) as v^2, zeros after standstill. Same systolic scheme as fb_sweep.
template <int EM, bool GGARR>
__device__ __forceinline__ void brake_profile(int n, double* out, const VelScratch& vs, double cax, double cay, double v_start,
                              const DevVelParams& p, int lane)
{
    const double icay = 1.0 / cay;
    double w = v_start * v_start;                          // state entering the chunk (uniform)
    int stopped = 0;
    if (lane == 0) out[0] = w;
    for (int base = 0; base < n - 1; base += 64) {
        const int cnt = (n - 1 - base) < 64 ? (n - 1 - base) : 64;
        const int ic = (lane < cnt) ? base + lane : base + cnt - 1;
        const double kq_i = vs.kabs[ic] * (GGARR ? vs.igay[ic] : icay), e_i = vs.el[ic], ax_i = GGARR ? vs.gax[ic] : cax;
        const double te = 2.0 * e_i, axa = fabs(ax_i);
        const double A0 = 1.0 - te * p.drag_m, A1 = A0 + te * (axa * kq_i), B1 = te * axa;
        double w_in = w, w_out = w; int st_in = stopped, st_out = 0;
        double r = 0.0;
        auto sys_step = [&]() {
            if constexpr (EM == 1) r = fmin(fma(A1, w_in, -B1), A0 * w_in);
            else r = w_in + 2.0 * ax_poss_w<EM, true, VMODE_DECEL_FORW>(w_in, kq_i, ax_i, p, vs.axm, 0.0) * e_i;
            st_out = (st_in || (r < 0.0)) ? 1 : 0;
            w_out = st_out ? w_in : r;
            w_in = wave_shr1_f64(w_out, w);
            st_in = wave_shr1_i32(st_out, stopped);
                };
        for (int it = 0; it < cnt; it += 4) { sys_step(); sys_step(); sys_step(); sys_step(); }      // (idempotent beyond cnt)
        if (lane < cnt) out[base + lane + 1] = st_out ? 0.0 : r;
        w = readlane_f64(w_out, cnt - 1);
        stopped = __builtin_amdgcn_readlane(st_out, cnt - 1);
    }
    wave_sync_lds();
}

// out[0] = 0, out[i] = out[i - 1] + src[i - 1] for i < m, in the SEQUENTIAL summation order of np.cumsum (bit-identical),
// executed systolically: lane l adds its element to the partial sum handed over by lane l - 1 (DPP wave_shr:1)
__device__ __forceinline__ void wave_cumsum_seq(const double* src, double* out, int m, int lane)
{
    double carry = 0.0;
    if (lane == 0 && m > 0) out[0] = 0.0;
    for (int base = 0; base < m - 1; base += 64) {
        const int cnt = (m - 1 - base) < 64 ? (m - 1 - base) : 64;
        const double e = src[(lane < cnt) ? base + lane : base + cnt - 1];
        double s_in = carry, s_out = 0.0;
        for (int it = 0; it < cnt; it += 4) {                        // (idempotent beyond cnt)
#pragma unroll
            for (int u = 0; u < 4; ++u) { s_out = s_in + e; s_in = wave_shr1_f64(s_out, carry); }
        }
        if (lane < cnt) out[base + lane + 1] = s_out;
        carry = readlane_f64(s_out, cnt - 1);
    }
    wave_sync_lds();
}

// out[0] = 0, out[i] = sum of src[0 .. i-1] for i < m as a wave-parallel prefix sum (log-step shuffles, 64 elements per round).
// NOT the summation order of np.cumsum: used where the arc length only enters results with a 1e-5 tolerance (distance to the
// object in the batch follow preparation), never where indices are derived from it.
__device__ __forceinline__ void wave_cumsum_par(const double* src, double* out, int m, int lane)
{
    double carry = 0.0;
    if (lane == 0 && m > 0) out[0] = 0.0;
    for (int base = 0; base < m - 1; base += 64) {
        const int i = base + lane;
        const double e = i < m - 1 ? src[i] : 0.0;
        double x = e;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const double y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (i < m - 1) out[i + 1] = carry + x;
        carry += __shfl(x, 63);
    }
    wave_sync_lds();
}

// first index i in [0, m) with pred(i), or m (wave-parallel search; pred is evaluated per lane)
template <typename Pred>
__device__ __forceinline__ int wave_find_first(int m, int lane, Pred pred)
{
    for (int base = 0; base < m; base += 64) {
        const int i = base + lane;
        const unsigned long long b = __ballot(i < m && pred(i));
        if (b) return base + __ffsll((long long)b) - 1;
    }
    return m;
}

__device__ __forceinline__ double angle3pt_dev(double ax, double ay, double bx, double by, double cx, double cy)
{
    double ang = atan2(cy - by, cx - bx) - atan2(ay - by, ax - bx);
    if (ang > D_PI) ang -= 2.0 * D_PI;
    else if (ang <= -D_PI) ang += 2.0 * D_PI;
    return ang;
}

// get_s_coord.py:34-47 picks the neighbour of the closest point by comparing |angle3pt(nb, pos, neighbour)|. The wrapped
// difference of two atan2 values is the angle between u = nb - pos and v = neighbour - pos in [0, pi], and the cosine is
// monotone there: |ang1| > |ang2|  <=>  u.v1 / |v1| < u.v2 / |v2|  (|u| cancels). Two dot products and two square roots instead
// of four atan2 (~600 instructions per projection). Degenerate vectors (pos on a polyline point) take the atan2 form.
// Returns -1 / 0 / +1 for |ang1| < / == / > |ang2|.
__device__ __forceinline__ int angle_order_dev(double nx, double ny, double px, double py, double x1, double y1, double x2, double y2)
{
    const double ux = nx - px, uy = ny - py, v1x = x1 - px, v1y = y1 - py, v2x = x2 - px, v2y = y2 - py;
    const double n1 = v1x * v1x + v1y * v1y, n2 = v2x * v2x + v2y * v2y, nu = ux * ux + uy * uy;
    if (!(n1 > 0.0) || !(n2 > 0.0) || !(nu > 0.0)) {
        const double a1 = fabs(angle3pt_dev(nx, ny, px, py, x1, y1)), a2 = fabs(angle3pt_dev(nx, ny, px, py, x2, y2));
        return a1 > a2 ? 1 : (a1 < a2 ? -1 : 0);
    }
    const double c1 = (ux * v1x + uy * v1y) * sqrt(n2), c2 = (ux * v2x + uy * v2y) * sqrt(n1);
    return c1 < c2 ? 1 : (c1 > c2 ? -1 : 0);
}

// get_s_coord.py:8-99 on a strided polyline in global / LDS memory, all lanes take part; every lane returns s and idx0
__device__ __forceinline__ double get_s_coord_dev(int n, const double* x, const double* y, int stride, const double* s_arr, int s_stride,
                                  double px, double py, bool closed, int lane, int* idx0)
{
    double bd = INFINITY; int nb = 0x7fffffff;
    for (int i = lane; i < n; i += 64) {
        const double dx = x[(size_t)i * stride] - px, dy = y[(size_t)i * stride] - py;
        const double d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; nb = i; }
    }
    wave_min2(bd, nb);
    int i1, i2;
    if (closed) { i1 = nb - 1; if (i1 < 0) i1 += n; i2 = nb + 1; if (i2 > n - 1) i2 = 0; }
    else { i1 = nb - 1 > 0 ? nb - 1 : 0; i2 = nb + 1 < n - 1 ? nb + 1 : n - 1; }
    const double nx = x[(size_t)nb * stride], ny = y[(size_t)nb * stride];
    const double x1 = x[(size_t)i1 * stride], y1 = y[(size_t)i1 * stride];
    const double x2 = x[(size_t)i2 * stride], y2 = y[(size_t)i2 * stride];
    const int ord = angle_order_dev(nx, ny, px, py, x1, y1, x2, y2);
    double ax, ay, bx, by;
    if (ord > 0) { ax = x1; ay = y1; bx = nx; by = ny; } else { ax = nx; ay = ny; bx = x2; by = y2; }
    const double t = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / ((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
    const double fx = ax + t * (bx - ax), fy = ay + t * (by - ay);
    const double ds = sqrt((ax - fx) * (ax - fx) + (ay - fy) * (ay - fy));
    const double s = (ord > 0 ? s_arr[(size_t)i1 * s_stride] : s_arr[(size_t)nb * s_stride]) + ds;
    if (idx0) *idx0 = (ord >= 0) ? i1 : nb;
    return s;
}

// Index pair start of the global race line segment an object is projected on (calc_vel_profile_follow.py:172-176:
// get_s_coord(closed=True, only the first index is used). All 2 x ceil(G / 64) loads of a lane are independent: four points per
// round trip from the contiguous x / y copies instead of one strided load per iteration.
__device__ __forceinline__ int globrl_index_dev(const DevLat& lat, double px, double py, int lane)
{
    const int G = lat.G - 1;
    double bd = INFINITY; int nb = 0x7fffffff;
    for (int i0 = 0; i0 < G; i0 += 256) {
        double xs[4], ys[4]; int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; id[u] = i < G ? i : G - 1; xs[u] = at(lat.grx, id[u]); ys[u] = at(lat.gry, id[u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double dx = xs[u] - px, dy = ys[u] - py, d2 = dx * dx + dy * dy;
            if (d2 < bd) { bd = d2; nb = id[u]; }
        }
    }
    wave_min2(bd, nb);
    int i1 = nb - 1; if (i1 < 0) i1 += G;
    int i2 = nb + 1; if (i2 > G - 1) i2 = 0;
    const int ord = angle_order_dev(at(lat.grx, nb), at(lat.gry, nb), px, py, at(lat.grx, i1), at(lat.gry, i1), at(lat.grx, i2), at(lat.gry, i2));
    return ord >= 0 ? i1 : nb;
}

// Index pair start of the global race line segment ONE LANE's object is projected on (calc_vel_profile_follow.py:172-176; the wave form is
// globrl_index_dev): closest race line point (first minimum, like np.argmin; the points are read with uniform indices, every lane compares
// with its own object), then the neighbour test of get_s_coord (closed = True). Every lane of the wave must call it.
__device__ __forceinline__ int lane_globrl_index(const DevLat& lat, double ox, double oy)
{
    const int G = lat.G - 1;
    double bd = INFINITY; int nb = 0;
#pragma unroll 8
    for (int i = 0; i < G; ++i) {
        const double dx = lat.grx[i] - ox, dy = lat.gry[i] - oy, d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; nb = i; }
    }
    int i1 = nb - 1; if (i1 < 0) i1 += G;
    int i2 = nb + 1; if (i2 > G - 1) i2 = 0;
    const int ord = angle_order_dev(at(lat.grx, nb), at(lat.gry, nb), ox, oy, at(lat.grx, i1), at(lat.gry, i1), at(lat.grx, i2), at(lat.gry, i2));
    return ord >= 0 ? i1 : nb;
}

struct FollowIn { double v_start, v_ego, v_obj, safety_d, obj_dist, obj_x, obj_y; };

// calc_vel_profile_follow.py:78-313. Inputs: vs.kabs[n], vs.el[n_el >= n] (tailing zero), gg; result vs.w (v^2).
// `ext_sync`: when set, the unconstrained profile (:297-307) has been computed by ANOTHER wave of the workgroup into vs.w; the
// function then joins that wave at a workgroup barrier instead of computing it itself (k_tick: wave 3 works for wave 0).
template <int EM, bool AXM1, bool GGARR>
__device__ __forceinline__ void follow_profile(const DevLat& lat, int n, int n_el, const VelScratch& vs, double cax, double cay,
                               const DevVelParams& p, const FollowIn& fi, int lane, int* too_close, int* vel_bound,
                               bool ext_sync = false, bool skip_free = false)
{
    int vb = 1;
    const double control_d = p.c_p * fi.safety_d + p.len_veh;                            // :141
    const double safety_d = fi.safety_d + p.len_veh;                                     // :144
    const int tc = (fi.obj_dist - safety_d) < 0.0;                                       // :147-149
    const double v_max = p.v_max;

    dbg_stamp(vs.dbg, 4);
    brake_profile<EM, GGARR>(n, vs.wb, vs, cax, cay, fi.v_start, p, lane);   // :152-159
    dbg_stamp(vs.dbg, 5);
    // arc length s = [0, cumsum(el[:-1])] (:203) next to the ego stop distance (:162-166)
    wave_cumsum_seq(vs.el, vs.s, n_el, lane);                  // s[i] = sum of el[0 .. i-1], sequential order
    // first point at or below 0.1 m/s on the brake profile; the ego stop distance is the arc length up to it (:162-166)
    int idb = wave_find_first(n, lane, [&](int i) { return !(vs.wb[i] > 0.01); });
    if (idb > n_el) idb = n_el;
    double ego_stop = 0.0;
    if (idb > 0) ego_stop = (idb < n_el) ? vs.s[idb] : vs.s[n_el - 1] + vs.el[n_el - 1];

    // opponent: closest point of the global race line (:172-179), brake scan with ggv [100, 14, 14] (:134,185-199)
    const int G = lat.G - 1;
    const double* grl = lat.glob_rl;
    dbg_stamp(vs.dbg, 6);
    const int idx_s_opp = globrl_index_dev(lat, fi.obj_x, fi.obj_y, lane);
    dbg_stamp(vs.dbg, 7);
    const double vel0 = grl[(size_t)idx_s_opp * 5 + 4];
    const double vel_start = fi.v_obj < vel0 ? fi.v_obj : vel0;                          // :182
    double opp_stop = 0.0, wopp = vel_start * vel_start;
    int stopped = (wopp > 0.01) ? 0 : 1;                                                 // id_brake counts v > 0.1
    for (int c0 = 0; c0 < G && !stopped; c0 += 64) {
        const int i = c0 + lane;
        if (i < G) {
            int j = i + idx_s_opp; if (j >= G) j -= G;
            vs.chunk[lane] = fabs(grl[(size_t)j * 5 + 3]);
            vs.chunk[64 + lane] = grl[(size_t)(j + 1) * 5] - grl[(size_t)j * 5];
        }
        wave_sync_lds();
        if (lane == 0) {
            const int m = (G - c0) < 64 ? (G - c0) : 64;
            for (int k = 0; k < m; ++k) {
                // point c0 + k has v > 0.1: its element length counts towards the stop distance
                opp_stop += vs.chunk[64 + k];
                if (c0 + k + 1 >= G) { stopped = 1; break; }
                const double a = ax_poss_w<EM, true, VMODE_DECEL_FORW>(wopp, vs.chunk[k] * (1.0 / 14.0), 14.0, p, vs.axm, 0.0);
                const double r = wopp + 2.0 * a * vs.chunk[64 + k];
                wopp = r < 0.0 ? 0.0 : r;                                                // standstill: profile stays 0
                if (!(wopp > 0.01)) { stopped = 1; break; }
            }
        }
        stopped = __shfl(stopped, 0);
        wave_sync_lds();
    }
    opp_stop = __shfl(opp_stop, 0);

    dbg_stamp(vs.dbg, 11);
    // characteristic indices (:201-221)
    const double s_stop = fi.obj_dist - safety_d + opp_stop;                             // :206
    int stop_idx = wave_find_first(n_el - 1, lane, [&](int i) { return !(vs.s[i] < s_stop); });   // :208-209
    double v_end = 0.0;
    if (lane == 0) {
        if (s_stop > vs.s[n_el - 1]) {                                                   // :212-221
            const double s_ends = opp_stop - (s_stop - vs.s[n_el - 1]);
            int idx = 0; double summed = 0.0;
            while (summed < s_ends && idx < G) {
                int j = idx + idx_s_opp; if (j >= G) j -= G;
                summed += grl[(size_t)(j + 1) * 5] - grl[(size_t)j * 5];
                ++idx;
            }
            int j = (idx % G) + idx_s_opp; if (j >= G) j -= G;
            v_end = grl[(size_t)j * 5 + 4];
        }
    }
    v_end = __shfl(v_end, 0);
    dbg_stamp(vs.dbg, 12);

    // control velocity (:232-239, get_control_vel :28-75)
    double v_control;
    if (p.ctrl == 0) v_control = (fi.v_obj - p.k_p * (control_d - fi.obj_dist) + p.k_d * (fi.v_obj - fi.v_ego));
    else {
        double a = (control_d - fi.obj_dist) * D_PI / 2 * 1 / p.tan_w;
        const double lo = -D_PI / 2 + 1e-5, hi = D_PI / 2 - 1e-5;
        a = a < lo ? lo : (a > hi ? hi : a);
        v_control = (fi.v_obj - tan(a) * p.k_p + p.k_d * (fi.v_obj - fi.v_ego));
    }
    if (v_control < 0.0) v_control = 0.0;
    if (v_control > v_max) v_control = v_max;

    // vs.wc <- "vx_profile" (:247-294)
    if (ego_stop < s_stop) {
        int idx_c = 0, n_decel = 0; double vcs = fi.v_start;      // vx_control_start
        if (fi.v_start > v_control && stop_idx >= 2) {                                   // :250-258
            const double wctl = v_control * v_control;
            int first = 0x7fffffff;
            for (int i = lane; i < n; i += 64) if (vs.wb[i] <= wctl) { first = i; break; }
            const int di = wave_min_i32(first);                                           // (first index over the lanes: an integer minimum)
            idx_c = (di == 0x7fffffff) ? 0 : di;                                          // np.argmax of an all-False mask
            if (idx_c > stop_idx) idx_c = stop_idx;
            if (idx_c == 0) idx_c = stop_idx;
            n_decel = idx_c + 1 < n ? idx_c + 1 : n;
            vcs = sqrt(vs.wb[n_decel - 1]);
        } else {
            if (!(stop_idx >= 2)) vb = 0;                                                // :260-261
        }
        // SEGMENT 2 (:267-286): FB profile on [idx_c, stop_idx] capped at v_control
        const int m = (stop_idx + 1 < n ? stop_idx + 1 : n) - idx_c;
        if (stop_idx - idx_c > 0) {
            VelScratch sub = vs;
            sub.w = vs.wc + idx_c; sub.kabs = vs.kabs + idx_c; sub.el = vs.el + idx_c;
            sub.gax = vs.gax ? vs.gax + idx_c : nullptr; sub.gay = vs.gay ? vs.gay + idx_c : nullptr;
            sub.igay = vs.igay ? vs.igay + idx_c : nullptr;
            fb_profile<EM, AXM1, GGARR>(m, sub, cax, cay, p, v_control, vcs, true, v_end, lane);
            if (fabs(sqrt(vs.wc[idx_c]) - vcs) > 1.0) vb = 0;
        } else if (stop_idx - idx_c == 0) {
            if (lane == 0) vs.wc[idx_c] = vcs * vcs;
        }
        // vx_decel[:-1] in front, zeros behind (:289)
        for (int i = lane; i < n; i += 64) {
            if (i < n_decel - 1) vs.wc[i] = vs.wb[i];
            else if (i > stop_idx) vs.wc[i] = 0.0;
        }
        wave_sync_lds();
        if (fabs(sqrt(vs.wc[0]) - fi.v_start) > 1.0) vb = 0;                             // :291-292
    } else {
        for (int i = lane; i < n; i += 64) vs.wc[i] = vs.wb[i];                          // :294
        wave_sync_lds();
    }
    // complete profile (:297-307) and intersection (:310)
    if (skip_free) {                                           // LTPL_VEL_FOLLOW_CONTROLLED: the caller intersects (:297-310)
        for (int i = lane; i < n; i += 64) vs.w[i] = vs.wc[i];
        wave_sync_lds();
        *too_close = tc; *vel_bound = vb;
        return;
    }
    if (ext_sync) __syncthreads();                             // the helper wave has written vs.w
    else fb_profile<EM, AXM1, GGARR>(n, vs, cax, cay, p, v_max, fi.v_start, false, 0.0, lane);
    for (int i = lane; i < n; i += 64) { const double a = vs.wc[i], b = vs.w[i]; vs.w[i] = a < b ? a : b; }
    wave_sync_lds();
    *too_close = tc; *vel_bound = vb;
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels of the velocity seam and the fused tick
// ---------------------------------------------------------------------------------------------------------------------
struct DevVelJob {
    int mode, n, n_el, has_v_end;
    int off_kappa, off_el, off_gg, off_out;       // offsets (in doubles) into the pooled job arrays
    double v_start, v_end, v_ego, v_obj, safety_d, obj_dist, obj_x, obj_y;
    // the car of the job (fleet::VelJob, ABI v6): v_max <= 0 / n_axm == 0 -> the launch's parameter set (seam 2, host planner)
    double v_max;
    int axm_off, n_axm;                           // rows [axm_off, axm_off + n_axm) of the launch's stacked machine tables
    int gg_rows, lane_form;                       // gg_rows 1: off_gg holds n caller-supplied [ax, ay] rows (local_gg as a dict, OTH.py:649-666) instead of two
                                                  // constants; lane_form 1 (fleet): operands in the lane plane, the job belongs to a lane kernel
};

// `lite`: only what the forward-backward and brake profiles touch (w, kabs, el, machine table, run flags) -- no arc length, no follow scratch
__device__ __forceinline__ VelScratch carve_vel_scratch(unsigned char* base, int cap, bool with_gg, bool with_xy,
                                                        double** px, double** py, bool lite = false)
{
    VelScratch vs;
    double* d = reinterpret_cast<double*>(base);
    const int c1 = cap + 2;
    vs.w = d; d += c1; vs.kabs = d; d += c1; vs.el = d; d += c1;
    if (!lite) { vs.s = d; d += c1; vs.wb = d; d += c1; vs.wc = d; d += c1; }
    else { vs.s = nullptr; vs.wb = nullptr; vs.wc = nullptr; }
    if (with_gg) { vs.gax = d; d += c1; vs.gay = d; d += c1; vs.igay = d; d += c1; }
    else { vs.gax = nullptr; vs.gay = nullptr; vs.igay = nullptr; }
    if (with_xy) { *px = d; d += c1; *py = d; d += c1; }
    if (!lite) { vs.chunk = d; d += 128; } else vs.chunk = nullptr;
    vs.axm = d; d += 128;
    vs.start = reinterpret_cast<unsigned char*>(d);
    vs.cap = cap;
    vs.dbg = nullptr;
    return vs;
}

static size_t vel_scratch_bytes(int cap, bool with_gg, bool with_xy, bool lite = false)
{
    size_t arrays = (lite ? 3 : 6) + (with_gg ? 3 : 0) + (with_xy ? 2 : 0);
    size_t b = sizeof(double) * (arrays * (size_t)(cap + 2) + (lite ? 128 : 256)) + (size_t)cap + 16;
    return (b + 15) / 16 * 16;
}

// seam (2): one wave per job. GG = false: the job's friction limits are constant along the path (the fleet's jobs: rows 0 of loc_gg hold
// them) -- three LDS arrays less per wave and no per-point loads of the limits. SEL (fleet): the occupancy of these launches is bound by
// the LDS a wave needs and the recurrences are latency chains (~200 cycles per point), so the fleet runs the forward-backward / brake jobs
// (SEL 1: three arrays, "lite" scratch) and the follow jobs (SEL 2: six arrays, job slot 0 of every planner) as two launches; a block
// whose job belongs to the other launch returns at once. SEL 3 (fleet, calls with location dependent friction): the forward-backward
// jobs whose limits are ROWS (DevVelJob::gg_rows) -- the lane kernel that solves a tick's other forward-backward jobs takes constants only.
// `job_stride`: block b works on job b * job_stride.
template <int EM, bool AXM1, bool GG = true, int SEL = 0>
__global__ __launch_bounds__(64) void k_vel_profile(DevLat lat, DevVelParams p_, const DevVelJob* jobs,
                                                    const double* pool, double* out_pool, int* out_flags, int cap,
                                                    long long* dbg, DoneSignal done, int job_stride)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    dbg_stamp(dbg, 0);
    const int j = blockIdx.x * job_stride;
    const DevVelJob jb = jobs[j];
    if (jb.n <= 0) { signal_done(done); return; }          // unused slot of a fleet's job table (fleet_dev.hpp); seam (2) itself rejects empty jobs
    // the car of the job: its own vel_max / machine table (a fleet of different cars), else the launch's
    DevVelParams p = p_;
    if (jb.v_max > 0.0) p.v_max = jb.v_max;
    if (jb.n_axm > 0) { p.n_axm = jb.n_axm; p.axm = p_.axm + 2 * jb.axm_off; }
    const bool follow = jb.mode == LTPL_VEL_FOLLOW || jb.mode == LTPL_VEL_FOLLOW_CONTROLLED;
    if constexpr (SEL == 1) { if (follow) return; }
    if constexpr (SEL == 2) { if (!follow || jb.lane_form) return; }      // (fleet: follow jobs without friction rows run one lane per job, k_fleet_follow_lanes)
    if constexpr (SEL == 3) { if (follow || !jb.gg_rows) return; }
    constexpr bool LITE = SEL == 1 || SEL == 3;
    VelScratch vs = carve_vel_scratch(smem, cap, GG, false, nullptr, nullptr, LITE);
    vs.dbg = dbg;
    const int n = jb.n;
    for (int i = lane; i < n; i += 64) {
        vs.kabs[i] = fabs(pool[jb.off_kappa + i]);
        if constexpr (GG) {
            vs.gax[i] = pool[jb.off_gg + 2 * i];
            const double ay = pool[jb.off_gg + 2 * i + 1];
            vs.gay[i] = ay; vs.igay[i] = 1.0 / ay;
        }
    }
    const double cax = GG ? 1.0 : pool[jb.off_gg], cay = GG ? 1.0 : pool[jb.off_gg + 1];
    for (int i = lane; i < 2 * p.n_axm; i += 64) vs.axm[i] = p.axm[i];
    for (int i = lane; i < jb.n_el; i += 64) vs.el[i] = pool[jb.off_el + i];
    wave_sync_lds();
    dbg_stamp(dbg, 1);
    int too_close = 0, vel_bound = 1;
    if (jb.mode == LTPL_VEL_FB) {
        fb_profile<EM, AXM1, GG>(n, vs, cax, cay, p, p.v_max, jb.v_start, jb.has_v_end != 0, jb.v_end, lane);
    } else if (jb.mode == LTPL_VEL_BRAKE) {
        brake_profile<EM, GG>(n, vs.w, vs, cax, cay, jb.v_start, p, lane);
    } else if constexpr (!LITE) {
        FollowIn fi; fi.v_start = jb.v_start; fi.v_ego = jb.v_ego; fi.v_obj = jb.v_obj; fi.safety_d = jb.safety_d;
        fi.obj_dist = jb.obj_dist; fi.obj_x = jb.obj_x; fi.obj_y = jb.obj_y;
        follow_profile<EM, AXM1, GG>(lat, n, jb.n_el, vs, cax, cay, p, fi, lane, &too_close, &vel_bound, false,
                                     jb.mode == LTPL_VEL_FOLLOW_CONTROLLED);
    }
    dbg_stamp(dbg, 2);
    for (int i = lane; i < n; i += 64) out_pool[jb.off_out + i] = sqrt(vs.w[i]);
    if (lane == 0) { out_flags[2 * j] = too_close; out_flags[2 * j + 1] = vel_bound; }
    dbg_stamp(dbg, 3);
    signal_done(done);
}

struct DevTickVelIn {
    double gg_ax, gg_ay, safety_d, v_max_offset;
    const double* vel_plan; const double* vel_est; const double* pos_est_x; const double* pos_est_y;
    const double* veh_vel;
};
struct DevTickVelOut { double* vx; double* ax; int* vel_bound; int* too_close; };

// per-primitive velocity stage of OnlineTrajectoryHandler.calc_vel_profile (OTH.py:688-941) on a fresh path:
// cut_index_pos = 0, vel_course empty, no brake prefix (the host rejects vel_plan > v_max + 0.1, for which the
// reference itself fails at OTH.py:919 because the prefix it computes is never merged back)
template <int EM, bool AXM1>
__device__ __forceinline__ void tick_vel_stage(const DevLat& lat, const DevPathsIn& in, const DevPathsOut& out, const WavePath& wp,
                               const VelScratch& vs, const double* px, const double* py, const DevVelParams& p,
                               const DevTickVelIn& vin, const DevTickVelOut& vout, int s, int slot, int lane,
                               bool helper_wave)
{
    // vs.kabs already holds |kappa| (k_tick). helper_wave: another wave computes the unconstrained follow profile into vs.w
    const int n = wp.n_pts;
    const double vel_plan = vin.vel_plan[s];
    const double cax = vin.gg_ax, cay = vin.gg_ay;
    // s = [0, cumsum(el[:-1])] (OTH.py:743), one extra entry for get_s_coord's zero-prefixed cumsum (:777)
    wave_cumsum_seq(vs.el, vs.s, n + 1, lane);
    int too_close = 0, vel_bound = 1;
    bool have_follow = false;
    if (wp.name == LTPL_ACT_FOLLOW) {                                                    // OTH.py:763-830
        FollowIn fi; fi.v_start = vel_plan; fi.v_ego = vin.vel_est[s]; fi.safety_d = vin.safety_d;
        const int ci = out.closest_obj_index[s];
        const int v0 = in.veh_off[s];
        if (ci < 0 || ci >= in.veh_off[s + 1] - v0) {
            fi.obj_dist = 0.0; fi.v_obj = 0.0; fi.obj_x = vin.pos_est_x[s]; fi.obj_y = vin.pos_est_y[s];
        } else {
            const int pp = in.pos_off[v0 + ci];
            fi.obj_x = in.pos_x[pp]; fi.obj_y = in.pos_y[pp]; fi.v_obj = vin.veh_vel[v0 + ci];
            const double s_obj = get_s_coord_dev(n, px, py, 1, vs.s, 1, fi.obj_x, fi.obj_y, false, lane, nullptr);
            const double s_sta = get_s_coord_dev(n, px, py, 1, vs.s, 1, vin.pos_est_x[s], vin.pos_est_y[s], false, lane, nullptr);
            fi.obj_dist = s_obj - s_sta;
        }
        // follow_profile rebuilds vs.s as [0, cumsum(el[:-1])] for n_el = n: identical values
        follow_profile<EM, AXM1, false>(lat, n, n, vs, cax, cay, p, fi, lane, &too_close, &vel_bound, helper_wave);
        have_follow = true;
    }
    if (wp.name != LTPL_ACT_FOLLOW || wp.reduced) {                                      // OTH.py:834-923
        if (have_follow) { for (int i = lane; i < n; i += 64) vs.wb[i] = vs.w[i]; wave_sync_lds(); }
        const int rl = lat.rl_idx[wp.goal_layer];
        int dn = wp.end_node - rl; if (dn < 0) dn = -dn;
        const double raceline_offset = (double)dn * lat.lat_offset;
        double v_end; int v_idx;
        if (wp.reduced) {
            v_end = 0.0;
            const double spl_len = vs.s[n - 1];
            int first = 0x7fffffff;
            for (int i = lane; i < n - 1; i += 64) if (!(vs.s[i + 1] < (spl_len - 5.0))) { first = i; break; }
            const int di = wave_min_i32(first);
            v_idx = ((di == 0x7fffffff) ? 0 : di) + 1;
            if (v_idx == 1 && n > 1) v_idx = n;
        } else {
            v_end = lat.vel_rl[wp.goal_layer];
            const double red = v_end * lat.vel_decrease_lat * raceline_offset;
            v_end -= (red < v_end ? red : v_end);
            v_idx = n;
        }
        if (v_idx > 1) fb_profile<EM, AXM1, false>(v_idx, vs, cax, cay, p, p.v_max, vel_plan, true, v_end, lane);
        else { if (lane == 0) vs.w[0] = 0.0; }
        for (int i = (v_idx > 1 ? v_idx : 1) + lane; i < n; i += 64) vs.w[i] = 0.0;         // OTH.py:901-903
        wave_sync_lds();
        vel_bound = fabs(sqrt(vs.w[0]) - vel_plan) < vin.v_max_offset ? 1 : 0;           // OTH.py:906-911
        if (have_follow && n >= 6) {
            // OTH.py:923 compares ROW 5 of both arrays; only column 5 (vx) can differ, so the whole profile switches
            const bool take_follow = sqrt(vs.wb[5]) < sqrt(vs.w[5]);
            if (take_follow) { for (int i = lane; i < n; i += 64) vs.w[i] = vs.wb[i]; wave_sync_lds(); }
        }
    }
    // finalise (OTH.py:925-941): filt_window == 1 is the identity; ax from neighbouring points, -5 at standstill
    double* o_vx = vout.vx + (size_t)slot * out.cap_pts;
    double* o_ax = vout.ax + (size_t)slot * out.cap_pts;
    for (int i = lane; i < n; i += 64) {
        const double wi = vs.w[i];
        const double v = sqrt(wi);
        o_vx[i] = v;
        double a = 0.0;
        if (i < n - 1) {
            a = (vs.w[i + 1] - wi) / (2.0 * (vs.s[i + 1] - vs.s[i]));
            if (fabs(v) <= 1e-8 && fabs(a) <= 1e-8) a = -5.0;
        }
        o_ax[i] = a;
    }
    if (lane == 0) { vout.vel_bound[slot] = vel_bound; vout.too_close[slot] = too_close; }
}

// the arguments of k_tick as they lie in the kernarg segment
struct TickKArgs { PathsKArgs pk; DevVelParams p; DevTickVelIn vin; DevTickVelOut vout; int vel_off, vel_stride, vel_cap; };
static_assert(sizeof(DevVelParams) % 8 == 0 && sizeof(DevTickVelIn) % 8 == 0 && sizeof(DevTickVelOut) % 8 == 0 && alignof(DevVelParams) == 8 &&
              alignof(DevTickVelIn) == 8 && alignof(DevTickVelOut) == 8, "TickKArgs must mirror the kernarg layout of k_tick");
static constexpr size_t KOFF_TP = offsetof(TickKArgs, p), KOFF_TVIN = offsetof(TickKArgs, vin), KOFF_TVOUT = offsetof(TickKArgs, vout);

// body of k_tick. RL = true: every argument struct is a reference INTO THE KERNARG SEGMENT, re-read per stage (karg_reload, paths_team.hpp)
template <int EM, bool AXM1, class P, int RL>
__device__ __forceinline__ void tick_body(const DevLat& lat_, const DevPathsIn& in_, const DevPathsOut& out_, const TeamLds& lp_,
                                          const DevVelParams& p_, const DevTickVelIn& vin_, const DevTickVelOut& vout_,
                                          int vel_off, int vel_stride, int vel_cap, unsigned char* smem, TeamShared& ts, int& sh_follow_n)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    VelScratch vs; double* px = nullptr; double* py = nullptr;
    if (wave < LTPL_MAX_ACTIONS) vs = carve_vel_scratch(smem + vel_off + (size_t)wave * vel_stride, vel_cap, false, true, &px, &py);
    WavePath wp = team_paths_body<NUM_WAVES, P, RL>(lat_, in_, out_, lp_, smem, ts, wave < LTPL_MAX_ACTIONS ? vs.kabs : nullptr,
                                                 wave < LTPL_MAX_ACTIONS ? vs.el : nullptr, px, py);
    // (re-derived from the kernarg segment pointer, not carried across the path search: karg_at, paths_team.hpp)
    const DevLat& lat = *LTPL_KARG_LAT(&lat_); const DevPathsIn& in = *LTPL_KARG_IN(&in_); const DevPathsOut& out = *LTPL_KARG_OUT(&out_);
    const TeamLds& lp = *LTPL_KARG_LP(&lp_);
    // (the small parameter structs of the velocity stage by value: its recurrences must not depend on argument loads)
    const DevVelParams p = *karg_at<RL, DevVelParams, KOFF_TP>(&p_);
    const DevTickVelIn vin = *karg_at<RL, DevTickVelIn, KOFF_TVIN>(&vin_); const DevTickVelOut vout = *karg_at<RL, DevTickVelOut, KOFF_TVOUT>(&vout_);
    const int s = blockIdx.x;
    // The fourth wave has no primitive of its own: it computes the unconstrained profile of the 'follow' slot
    // (calc_vel_profile_follow.py:297-307, independent of the controlled part) while wave 0 runs the brake / segment part.
    if (wave < LTPL_MAX_ACTIONS) {
        for (int i = lane; i < 2 * p.n_axm; i += 64) vs.axm[i] = p.axm[i];
        if (wp.valid) for (int i = lane; i < wp.n_pts; i += 64) vs.kabs[i] = fabs(vs.kabs[i]);
        if (wave == 0 && lane == 0) sh_follow_n = (wp.valid && wp.name == LTPL_ACT_FOLLOW) ? wp.n_pts : 0;
    }
    __syncthreads();
    const int follow_n = sh_follow_n;
    if (wave < LTPL_MAX_ACTIONS) {
        const int slot = s * LTPL_MAX_ACTIONS + wave;
        vs.dbg = lp.dbg;
        if (wp.valid) tick_vel_stage<EM, AXM1>(lat, in, out, wp, vs, px, py, p, vin, vout, s, slot, lane, wave == 0 && follow_n > 0);
        else if (lane == 0) { vout.vel_bound[slot] = 0; vout.too_close[slot] = 0; }
        dbg_stamp(lp.dbg, 11);
        if (!(wave == 0 && follow_n > 0)) __syncthreads();     // wave 0 with a follow slot joins inside follow_profile
    } else {
        if (follow_n > 0) {
            double* dpx; double* dpy;
            VelScratch v0 = carve_vel_scratch(smem + vel_off, vel_cap, false, true, &dpx, &dpy);   // wave 0's arrays
            v0.start = smem + vel_off + (size_t)LTPL_MAX_ACTIONS * vel_stride;                     // own run-start flags
            v0.dbg = nullptr;
            fb_profile<EM, AXM1, false>(follow_n, v0, vin.gg_ax, vin.gg_ay, p, p.v_max, vin.vel_plan[s], false, 0.0, lane);
        }
        __syncthreads();
    }
    signal_done(LTPL_KARG_OUT(&out)->done);
}


template <int EM, bool AXM1, class P>
__global__ __launch_bounds__(WG_THREADS) void k_tick(DevLat lat, DevPathsIn in, DevPathsOut out, TeamLds lp,
                                                     DevVelParams p, DevTickVelIn vin, DevTickVelOut vout,
                                                     int vel_off, int vel_stride, int vel_cap)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ TeamShared ts;
    __shared__ int sh_follow_n;
    if constexpr (P::fixed) {
        // (the by-value parameters only define the kernarg layout: see k_paths)
        const TickKArgs* tk = (const TickKArgs*)(const TickKArgs LTPL_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
        tick_body<EM, AXM1, P, 1>(tk->pk.lat, tk->pk.in, tk->pk.out, tk->pk.lp, tk->p, tk->vin, tk->vout, tk->vel_off, tk->vel_stride, tk->vel_cap,
                                     smem, ts, sh_follow_n);
        return;
    }
    tick_body<EM, AXM1, P, 0>(lat, in, out, lp, p, vin, vout, vel_off, vel_stride, vel_cap, smem, ts, sh_follow_n);
}

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT SINGLE TICK (round 6): k_tick as a RESIDENT kernel behind a mailbox in page-locked memory
// ---------------------------------------------------------------------------------------------------------------------
// One car, one tick at a time (the only use the reference has: OnlineTrajectoryHandler.py:353-366 spends a 0.1 s budget per tick) is a
// LATENCY problem, and the round-5 counters say where a launched tick loses it (profiles/r05D_icache_tick.txt): every launch starts with
// a cold instruction cache (the kernel's ~70 KB come back through L2: 1.1 k line fetches per tick), pays the dispatch of a grid on an
// idle chip and two runtime calls (H2D copy + launch) on the host. This kernel is launched ONCE per handle -- one workgroup of four waves,
// the team of k_tick -- and then serves ticks until it is told to leave or has idled for `idle_limit`:
//   host:   pack the inputs into the page-locked staging buffer (as for k_tick), write the tick's argument block into the mailbox,
//           store-release the next sequence number, spin on the completion word (the DoneSignal the fused tick already carries);
//   kernel: thread 0 polls the sequence word (system-scope acquire loads: they bypass the caches), the others sleep at the barrier;
//           all threads copy the argument block and the packed inputs into device memory (two coalesced passes over PCIe instead of
//           dependent zero-copy reads inside the phases), fence, invalidate the scalar cache (the arguments are read with scalar loads)
//           and run tick_body<.., RL = 2> -- the SAME body as k_tick, its argument block addressed through the kernel's first argument.
// Outputs go straight into page-locked memory as for every call with <= 8 scenarios (LTPL_ZC_OUT). No hipMemcpyAsync, no launch, no
// stream synchronisation on the tick's path. Results are bit-identical to k_tick's (same code, same order of operations).
// A RESIDENT kernel never completes: device-wide synchronisations of the process (hipDeviceSynchronize, hipFree) would wait for it. That
// is why the mode is opt-in (ltpl_create_ex / LTPL_PERSISTENT_TICK), why every other entry point of the handle stops the kernel first
// (persist_stop) and why the kernel leaves BY ITSELF after `idle_limit` (default 250 ms) without a tick; the host notices (`exited`)
// and starts it again with the next tick. Only for lattices whose four-wave kernel has a compile-time LDS plan (PlanA4).
#define PT_CMD_TICK 0u
#define PT_CMD_EXIT 1u
struct TickMailbox {
    // host -> kernel
    unsigned seq;                  // number of the posted tick; written LAST, with release semantics
    unsigned cmd;                  // PT_CMD_*
    unsigned in_bytes;             // packed inputs of the tick: bytes to copy from the staging buffer to its device twin (multiple of 16)
    unsigned pad0;
    // kernel -> host
    unsigned exited;               // 0 while the kernel is resident; its last store: 1
    unsigned n_done;               // ticks served by this residency
    unsigned long long busy_clk;   // wall_clock64 ticks (100 MHz) between "sequence number seen" and "tick done", summed over n_done ticks
    unsigned long long last_clk;   // ... of the last tick
    unsigned long long pad1;
    alignas(16) unsigned char args[2048];      // TickKArgs of the posted tick
};
static_assert(sizeof(TickKArgs) <= 2048 && sizeof(TickKArgs) % 8 == 0, "mailbox argument block");

template <int EM, bool AXM1, class P>
__global__ __launch_bounds__(WG_THREADS) void k_tick_persistent(const TickKArgs* d_args, TickMailbox* mb, const unsigned char* h_in, unsigned char* d_in,
                                                                unsigned start_seq, unsigned long long idle_limit)
{
    static_assert(P::fixed, "the persistent tick reads its arguments like the compile-time plan classes do");
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ TeamShared ts;
    __shared__ int sh_follow_n;
    __shared__ unsigned sh_seq, sh_cmd, sh_in_bytes;
    unsigned last = start_seq, n_done = 0;
    unsigned long long busy = 0;
    for (;;) {
        unsigned long long t_seen = 0;
        if (threadIdx.x == 0) {
            const unsigned long long t0 = wall_clock64();
            unsigned sq, cmd = PT_CMD_TICK;
            for (;;) {
                sq = __hip_atomic_load(&mb->seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (sq != last) { cmd = __hip_atomic_load(&mb->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                if (wall_clock64() - t0 > idle_limit) { cmd = PT_CMD_EXIT; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            t_seen = wall_clock64();
            sh_seq = sq; sh_cmd = cmd;
            sh_in_bytes = __hip_atomic_load(&mb->in_bytes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        if (sh_cmd != PT_CMD_TICK) break;
        last = sh_seq;
        // argument block and packed inputs: page-locked memory -> device memory, 16 bytes per thread and pass. Page-locked memory is not
        // cached on the device (fine-grained: every read crosses PCIe), so ALL loads are issued before the first store -- one PCIe round
        // trip for the block and the first 8 KB of inputs (a C2 tick packs ~2 KB) -- and nothing has to be invalidated for them.
        {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            constexpr unsigned NA = (unsigned)(sizeof(TickKArgs) + 15) / 16;
            static_assert(NA <= WG_THREADS, "one pass over the argument block");
            const u32x4* sa = reinterpret_cast<const u32x4*>(mb->args);
            u32x4* da = reinterpret_cast<u32x4*>(const_cast<TickKArgs*>(d_args));
            const u32x4* si = reinterpret_cast<const u32x4*>(h_in);
            u32x4* di = reinterpret_cast<u32x4*>(d_in);
            const unsigned nq = sh_in_bytes >> 4, t = threadIdx.x;
            const u32x4 zero = {0u, 0u, 0u, 0u};
            const u32x4 a = t < NA ? sa[t] : zero;
            const u32x4 b0 = t < nq ? si[t] : zero, b1 = t + WG_THREADS < nq ? si[t + WG_THREADS] : zero;
            if (t < NA) da[t] = a;
            if (t < nq) di[t] = b0;
            if (t + WG_THREADS < nq) di[t + WG_THREADS] = b1;
            for (unsigned i = t + 2 * WG_THREADS; i < nq; i += WG_THREADS) di[i] = si[i];
            // the copies are consumed by THIS workgroup: stores retired (they write through to L2), then no stale line of the two buffers in
            // the CU's vector cache and no argument of the previous tick in the scalar cache -- no system-scope fence, no L2 write-back
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __builtin_amdgcn_s_dcache_inv();
        }
        tick_body<EM, AXM1, P, 2>(d_args->pk.lat, d_args->pk.in, d_args->pk.out, d_args->pk.lp, d_args->p, d_args->vin, d_args->vout,
                                  *karg_at<2, int, offsetof(TickKArgs, vel_off)>((const int*)nullptr), *karg_at<2, int, offsetof(TickKArgs, vel_stride)>((const int*)nullptr),
                                  *karg_at<2, int, offsetof(TickKArgs, vel_cap)>((const int*)nullptr), smem, ts, sh_follow_n);
        // (tick_body ends with signal_done: outputs fenced out, completion word stored -- the host may already be packing the next tick)
        if (threadIdx.x == 0) {
            const unsigned long long dt = wall_clock64() - t_seen;
            busy += dt; ++n_done;
            __hip_atomic_store(&mb->last_clk, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mb->busy_clk, busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&mb->n_done, n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();                                             // (sh_seq / sh_cmd are rewritten by thread 0 in the next round)
    }
    if (threadIdx.x == 0) __hip_atomic_store(&mb->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------------------------------
// throughput form of the velocity stage: ONE LANE PER PROFILE
// ---------------------------------------------------------------------------------------------------------------------
// The recurrences are sequential per profile, so a wave that owns a single profile runs them with 1/64 of its lanes.
// For batches the velocity stage therefore runs as its own kernel in which every lane of a wave64 integrates its own
// profile: 64 independent dependent-chains per instruction stream. All per-row data a lane touches lives in TILED
// planes: element (slot, row) at ((slot / 64) * cap_pts + row) * 64 + slot % 64, so that the 64 lanes of a wave
// (64 consecutive slots) read / write row i as ONE coalesced 512-byte line. The path kernel writes |kappa| and the
// element lengths of every path into two such planes.
//   job type 0 (one lane per action slot):  'follow' -> the controlled profile (ego brake, opponent stop distance, segment
//            forward-backward profile; calc_vel_profile_follow.py:151-294); other primitives -> the generic
//            forward-backward profile with the race-line end velocity (OTH.py:834-903)
//   job type 1 (one lane per scenario):     the unconstrained profile of a 'follow' slot (calc_vel_profile_follow.py:297-307)
//            -- independent of type 0, so the two halves of the follow mode run on different lanes
// k_vel_final (lane per slot, no recurrence) intersects / selects the profiles, derives ax and writes the outputs.
struct DevVelPrep {             // per-slot scalars produced by k_follow_prep (follow action only)
    double* obj_dist; double* v_obj; double* obj_x; double* obj_y; int* idx_s_opp;
};

struct VelPlanes {              // tiled planes (doubles), tile index = job index: generic jobs [0, n_slots_pad), follow jobs behind
    ke_t* KE;                   // (|kappa|, element length) as an fp64 pair: ONE 16-byte load per row and lane   n_slots_pad + n_scen_pad tiles
    double* P0;                 // type 0 result: follow -> "vx_profile" (:289/:294), else the generic profile     (same size)
    double* P1;                 // type 1 result: unconstrained profile of a follow job                n_scen_pad tiles
    double* P2;                 // ego brake profile (follow)                                           n_scen_pad tiles
    double* P3;                 // segment profile (follow), afterwards the generic profile of a reduced-horizon follow job
    double* XY;                 // (x, y) pairs of the follow jobs' path points (written by the path kernel)   n_scen_pad tiles x 2
    int* flags;                 // per tile: VF_* bits
    int* fseg;                  // per follow job [2]: n_decel (-1: everything from the brake profile), stop_idx -- composition of
                                // "vx_profile" (:289 / :294) from P2 / P3 / zeros, done by k_vel_final (VF_COMPOSE)
    int cap_pts;
    int plane_rows;             // rows of the blocked planes KE / XY (kep_base / kep_row, paths_team.hpp)
};

__device__ __forceinline__ size_t tile_base(int idx, int cap_pts) { return ((size_t)(idx >> 6) * cap_pts) * 64 + (idx & 63); }

struct LaneProf {
    const ke_t* KE;                       // tile-strided: element i at [i * 64]
};

// a / b with the hardware reciprocal and two Newton steps (~1 ulp; the velocity stage is checked to 1e-5 relative). b = 0
// yields NaN, which the callers' `!(w < vmax2)` clamp treats like +inf.
__device__ __forceinline__ double fast_div(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return a * r;
}

// Issue priority of the velocity kernels' waves (s_setprio 0 .. 3) inside a SIMD they share with path waves. EXPERIMENT (round 5, -DLTPL_VEL_PRIO=<n>):
// a velocity wave blocks the slot of a fourth path wave for as long as it lives; with a higher priority it issues whenever it is ready and
// lives shorter. Default 0 = off (the A/B is in DESIGN.md section 9).
#ifndef LTPL_VEL_PRIO
#define LTPL_VEL_PRIO 0
#endif
#define LTPL_VEL_SETPRIO() do { if (LTPL_VEL_PRIO > 0) __builtin_amdgcn_s_setprio(LTPL_VEL_PRIO); } while (0)
#define LCH 8      // rows per register chunk: all loads of a chunk are issued before the chunk's recurrence steps
#ifndef LCHF
#define LCHF 16    // forward sweep of lane_fb_profile: one 8-byte (|kappa|, el) load per row -> 16 rows per chunk in 32 registers
#endif
#ifndef LCHA
#define LCHA 16    // affine case, forward sweep: rows per register chunk
#endif
#ifndef LCHB
#define LCHB 12    // backward sweep: (|kappa|, el) + the fp64 profile state per row
#endif

// DIRECT OUTPUT OF THE GENERIC PROFILES (round 5). The rows of a generic job (every non-follow primitive) used to take a detour: the
// backward sweep rewrote the plane, k_vel_final read plane and element lengths again, took the root, differentiated and wrote vx / ax.
// Now the backward sweep of lane_fb_profile<.., EMIT = true> produces vx / ax of the rows it finalises (row r: v = sqrt(w_r),
// a = (w_r+1 - w_r) / (2 e_r), -5 at standstill, OTH.py:925-941 -- the operations of k_vel_final) and hands a chunk of LCHB rows per lane to
// lane_emit_flush, which transposes through LDS exactly like k_vel_final: lane = job writes its rows, lane = (job of a group of 8, pair of
// rows) stores 16 contiguous bytes of the slot's output row. The sweep runs wave-uniformly in this mode (all 64 lanes take part in every
// flush, lanes that have no rows left contribute none). k_vel_final only serves the follow jobs (compositions, intersections, row-5 choice).
struct LaneEmit {
    double* tbuf;                  // LDS [32 * LE_PITCH]
    int* s_cnt; int* s_rhi;        // LDS [64]: rows of the job's chunk, row of its first value (values run DOWNWARDS from there)
    unsigned long long* s_o;       // LDS [64]: vx row base of every lane's job (0: idle lane), written by the caller
    long long ax_delta;            // vout.ax - vout.vx in doubles: the ax row of a job lies at the same offset of the other array
    double w0;                     // out: v^2 of row 0 after the backward sweep
};
#define LE_PITCH (LCHB + 2)        // doubles per job row in LDS (16-byte aligned pairs); the buffer is sized for the longest chunk
#ifndef LCHC
#define LCHC 8                     // rows per chunk of the CAPPED backward sweep (follow jobs, round 6): three operands per row -- (|kappa|, el), the forward
                                   // value, the cap -- double-buffered next to the chunk's outputs: 8 rows keep the kernel at 256 registers
#endif
template <int NCH>
__device__ __forceinline__ void lane_emit_flush(const LaneEmit& E, int lane, int cnt, int r_hi, const double (&vv)[NCH], const double (&aa)[NCH])
{
    static_assert(NCH % 2 == 0 && NCH <= 16 && NCH <= LCHB, "pairs of rows, eight pieces per job, the LDS buffer of the longest chunk");
    constexpr int PITCH = NCH + 2;
    E.s_cnt[lane] = cnt; E.s_rhi[lane] = r_hi;
    const int q = lane >> 3, piece = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int half = pass & 1;                                      // jobs [32 half, 32 half + 32); passes 0, 1: vx, 2, 3: ax
        if ((lane >> 5) == half) {
#pragma unroll
            for (int c = 0; c < NCH; c += 2) store2(&E.tbuf[(lane & 31) * PITCH + c], pass < 2 ? vv[c] : aa[c], pass < 2 ? vv[c + 1] : aa[c + 1]);
        }
        wave_sync_lds();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jl = k * 8 + q, jj = half * 32 + jl, c = 2 * piece;
            const int cn = E.s_cnt[jj];
            if (c >= cn || c >= NCH) continue;
            const dbl2 v = *reinterpret_cast<const dbl2*>(&E.tbuf[jl * PITCH + c]);     // rows r - c (x) and r - c - 1 (y)
            double* o = reinterpret_cast<double*>(E.s_o[jj]) + (pass < 2 ? 0 : E.ax_delta) + (E.s_rhi[jj] - c);
            if (c + 1 < cn) store2_u(o - 1, v.y, v.x); else o[0] = v.x;
        }
        wave_sync_lds();
    }
}

// tph.calc_vel_profile(closed=False) for one lane: rows [off, off + n) of the path, result into plane D (as v^2).
// Rows are processed in register chunks of LCH; the rows of the NEXT chunk are requested before the current chunk's recurrence
// steps run (double buffer). Two things keep the compiler's s_waitcnt placement tight (measured: a per-step vmcnt(0) -- i.e. a
// full store acknowledgement per row -- made the kernel 8x slower than its instruction count):
//   * the function is force-inlined into the kernel, so the planes are addressed with GLOBAL instructions (an out-of-line copy
//     takes them by reference as generic pointers -> FLAT loads / stores, which always wait on vmcnt(0) and lgkmcnt(0));
//   * a recurrence step is branch-free (selects instead of `if (active)`), so a chunk is one basic block and the waits are
//     counted exactly: the stores of a chunk stay in flight while the next chunk's operands are awaited.
// What the EMIT form puts out per finalised row: the profile's own v^2 (NoCap: the generic jobs), or -- FollowCap, round 6 -- the MINIMUM of it and
// the row of the follow job's "vx_profile" (calc_vel_profile_follow.py:289 / :294: ego brake profile P2 in front of row n_decel - 1, segment
// profile P3 up to stop_idx, zeros behind), i.e. what k_vel_final composed from four planes for every follow job: the unconstrained profile's
// backward sweep now finalises the follow job's rows itself and k_vel_final only serves the (rare) reduced-horizon follow jobs.
struct NoCap { static constexpr bool active = false; __device__ __forceinline__ double at(int) const { return INFINITY; } };
struct FollowCap {
    static constexpr bool active = true;
    const double* P2; const double* P3; int nd, stop_idx;          // nd < 0: every row from the brake profile
    __device__ __forceinline__ double at(int r) const
    {
        const bool from_b = nd < 0 || r < nd - 1;
        const double* src = from_b ? P2 : P3;
        const double v = src[(size_t)r * 64];
        return (!from_b && r > stop_idx) ? 0.0 : v;
    }
};

template <int EM, bool AXM1, bool EMIT = false, class CAP = NoCap>
__device__ __forceinline__ void lane_fb_profile(const LaneProf& L, double* D, int off, int n, double cax, double cay,
                                                const DevVelParams& p, const double* axm_tab, double v_max, double v_start,
                                                bool has_v_end, double v_end, long long* dbg = nullptr, int drow = -1, LaneEmit* em = nullptr, int lane = 0,
                                                const CAP& cap = CAP())
{
    if (v_start < 0.0) v_start = 0.0;
    if (has_v_end && v_end < 0.0) v_end = 0.0;
    const double vmax2 = v_max * v_max, icay = 1.0 / cay, axm1 = axm_tab[1], dm = p.drag_m, axa = fabs(cax);
    const double vend2 = has_v_end ? v_end * v_end : INFINITY;
    const ke_t* KEb = L.KE;                         // row r of this profile: KEb[kep_row(off + r)]
#define KE_AT(r) KEb[kep_row(off + (r))]
    double* Dp = D + (size_t)off * 64;
    const ke_t rec0 = KE_AT(0);
    double kabs_i = (double)rec0.x, e_i = (double)rec0.y;
    double wi = fmin(cay * ke_rcp(rec0), vmax2);
    if (wi > v_start * v_start) wi = v_start * v_start;
    Dp[0] = wi;
    if constexpr (!EMIT) { if (n < 2) return; }
    do {                                   // (EMIT: a lane without a profile -- n < 2 -- skips the forward sweep but takes part in the flushes below)
    if constexpr (EMIT) { if (n < 2) break; }
    // ---- lateral-limit speed + forward sweep (accel_forw) in one pass -------------------------------------------------
    if constexpr (EM == 1 && AXM1) {
        // Affine case (exponent 1, one-row machine table). A velocity wave is alone on its SIMD and issues one instruction every
        // ~13 cycles whether or not it depends on the previous one (measured: 610 cycles per row for ~50 instructions; splitting
        // a chunk into independent coefficient work and a short dependent chain changed nothing), so the step is written for
        // the smallest instruction COUNT:  w' = max(A0 w + te min(max(axa - g w, 0), axm), 0)  with te = 2 e, A0 = 1 - te dm,
        // g = axa |kappa| / ay  -- algebraically the reference's min(tyre, machine) - drag step; the limit speed comes from the
        // fp32 hardware reciprocal of |kappa| (no fp64 division), full chunks run without the `valid` selects (the partial chunk at the end has them).
        constexpr int CA = CAP::active ? LCHC : LCHA;             // rows per register chunk (the capped form carries more state through the sweeps)
        const double axg = axa * icay;
        double orig_p = wi, g_p = kabs_i * axg, e_p = e_i;      // operands of the row in front of the current step
        bool active = false, prev_acc = false;
        ke_t kr[CA], kn[CA];
        const int nst = n - 1;
#pragma unroll
        for (int c = 0; c < CA; ++c) { const int r = 1 + c < n ? 1 + c : n - 1; kr[c] = KE_AT(r); }
        auto step = [&](const ke_t& rec, int i, bool valid) {
            const double w0n = fmin(cay * ke_rcp(rec), vmax2);    // inf on straights
            const bool acc = w0n > orig_p;
            const bool act = active || (acc && !prev_acc);
            const double te = e_p + e_p;
            const double u = fmin(fmax(fma(-g_p, wi, axa), 0.0), axm1);
            const double wn = fmax(fma(te, u, fma(-(te * dm), wi, wi)), 0.0);
            const double wnext = (act && wn < w0n) ? wn : w0n;
            if (valid) {
                Dp[(size_t)(i + 1) * 64] = wnext;
                active = act && !(wn > vmax2); prev_acc = acc;
                wi = wnext; orig_p = w0n; g_p = (double)rec.x * axg; e_p = (double)rec.y;
            }
        };
        int base = 0;
        for (; base + CA <= nst; base += CA) {
#pragma unroll
            for (int c = 0; c < CA; ++c) {                   // operands of the next chunk (clamped rows: harmless re-reads at the end)
                const int r = base + CA + 1 + c < n ? base + CA + 1 + c : n - 1;
                kn[c] = KE_AT(r);
            }
#pragma unroll
            for (int c = 0; c < CA; ++c) step(kr[c], base + c, true);
#pragma unroll
            for (int c = 0; c < CA; ++c) kr[c] = kn[c];
        }
        if (base < nst) {
#pragma unroll
            for (int c = 0; c < CA; ++c) step(kr[c], base + c, base + c < nst);
        }
        if (wi > vend2) { wi = vend2; Dp[(size_t)(n - 1) * 64] = wi; }            // the end-velocity clamp of the last step
        kabs_i = (double)KE_AT(n - 1).x;                             // |kappa| of the last row (start of the backward sweep)
    } else {
        double orig_i = wi;
        bool active = false, prev_acc = false;
        ke_t kr[LCHF], kn[LCHF];
#pragma unroll
        for (int c = 0; c < LCHF; ++c) { const int r = 1 + c < n ? 1 + c : n - 1; kr[c] = KE_AT(r); }
        for (int base = 0; base < n - 1; base += LCHF) {
#pragma unroll
            for (int c = 0; c < LCHF; ++c) {                   // operands of the next chunk (clamped rows: harmless re-reads at the end)
                const int r = base + LCHF + 1 + c < n ? base + LCHF + 1 + c : n - 1;
                kn[c] = KE_AT(r);
            }
#pragma unroll
            for (int c = 0; c < LCHF; ++c) {
                const int i = base + c;
                const bool valid = i < n - 1;
                const double k_c = (double)kr[c].x, e_c = (double)kr[c].y;
                double w0n = fast_div(cay, k_c);
                w0n = (w0n < vmax2) ? w0n : vmax2;
                const bool acc = w0n - orig_i > 0.0;
                const bool act = active || (acc && !prev_acc);
                const double kq_i = kabs_i * icay;
                double wn;
                if constexpr (EM == 1 && AXM1) {
                    const double te = 2.0 * e_i;
                    const double A0 = 1.0 - te * dm, A1 = A0 - te * (axa * kq_i), B1 = te * axa, B2 = te * axm1;
                    wn = fmax(fmin(fmax(fma(A1, wi, B1), A0 * wi), fma(A0, wi, B2)), 0.0);
                } else {
                    const double a = ax_poss_w<EM, AXM1, VMODE_ACCEL_FORW>(wi, kq_i, cax, p, axm_tab, axm1);
                    wn = fmax(wi + 2.0 * a * e_i, 0.0);
                }
                double wnext = (act && wn < w0n) ? wn : w0n;
                const bool act_out = act && !(wn > vmax2);
                wnext = (i + 1 == n - 1 && wnext > vend2) ? vend2 : wnext;
                if (valid) Dp[(size_t)(i + 1) * 64] = wnext;
                active = valid ? act_out : active; prev_acc = valid ? acc : prev_acc;
                orig_i = valid ? w0n : orig_i; wi = valid ? wnext : wi; kabs_i = valid ? k_c : kabs_i; e_i = valid ? e_c : e_i;
            }
#pragma unroll
            for (int c = 0; c < LCHF; ++c) kr[c] = kn[c];
        }
    }
    } while (0);
    vl_stamp(dbg, drow, 8);
    // ---- backward sweep (decel_backw), mirrored indices; with a constant gg the unmirrored-gg quirk is void --------------
    if constexpr (EMIT) {
        // wave-uniform form with direct output (see LaneEmit): every lane walks the chunks of the LONGEST profile of the wave, its own steps
        // masked by `valid`; the plane is only read (forward values), vx / ax of the finalised rows go out through lane_emit_flush
        constexpr int CB = CAP::active ? LCHC : LCHB;             // rows per chunk
        const int nst = n >= 2 ? n - 1 : 0;
        const double axg = axa * icay;
        double orig_p = wi, g_p = kabs_i * axg;                       // affine form: operands of the row above
        double orig_i = wi;                                           // general form
        bool active = false, prev_acc = false;
        ke_t kr[CB], kn[CB]; double wr[CB], wq[CB];
        [[maybe_unused]] double cr[CB], cq[CB];                   // CAP: the cap's rows of the current / the next chunk
        [[maybe_unused]] double fprev = 0.0;                          // CAP: the value put out for the row above
        if constexpr (CAP::active) { if (n >= 1) fprev = fmin(cap.at(n - 1), wi); }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int r = n - 2 - c >= 0 ? n - 2 - c : 0;
            kr[c] = KE_AT(r); wr[c] = Dp[(size_t)r * 64];
            if constexpr (CAP::active) cr[c] = cap.at(r);
        }
        for (int base = 0; __ballot(base < nst) != 0ull; base += CB) {
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const int r = n - 2 - base - CB - c >= 0 ? n - 2 - base - CB - c : 0;
                kn[c] = KE_AT(r); wq[c] = Dp[(size_t)r * 64];
                if constexpr (CAP::active) cq[c] = cap.at(r);
            }
            double vv[CB], aa[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                const bool valid = base + c < nst;
                const double wold = wr[c], e_b = (double)kr[c].y, k_c = (double)kr[c].x;
                double wn; bool acc;
                if constexpr (EM == 1 && AXM1) {
                    acc = wold > orig_p;
                    const double te = 2.0 * e_b, tdm = te * dm, g_n = k_c * axg;
                    const double w1 = fmax(fma(te, fmax(fma(-g_p, wi, axa), 0.0), fma(tdm, wi, wi)), 0.0);
                    wn = fmin(fmax(fma(te, fmax(fma(-g_n, w1, axa), 0.0), fma(tdm, w1, wi)), 0.0), w1);
                } else {
                    acc = wold - orig_i > 0.0;
                    const double kq_i = kabs_i * icay, kq_n = k_c * icay;
                    const double a = ax_poss_w<EM, AXM1, VMODE_DECEL_BACKW>(wi, kq_i, cax, p, axm_tab, axm1);
                    wn = fmax(wi + 2.0 * a * e_b, 0.0);
                    const double a2 = ax_poss_w<EM, AXM1, VMODE_DECEL_BACKW>(wn, kq_n, cax, p, axm_tab, axm1);
                    wn = fmin(fmax(wi + 2.0 * a2 * e_b, 0.0), wn);
                }
                const bool act = active || (acc && !prev_acc);
                const double wnext = (act && wn < wold) ? wn : wold;
                double v = 0.0, a_out = 0.0;
                if (valid) {
                    if constexpr (CAP::active) {
                        const double f = fmin(cr[c], wnext);           // the follow job's row: min("vx_profile", unconstrained profile)
                        v = sqrt(f);
                        a_out = (fprev - f) / (2.0 * e_b);
                        fprev = f;
                    } else {
                        v = sqrt(wnext);
                        a_out = (wi - wnext) / (2.0 * e_b);            // (w_r+1 - w_r) / (2 e_r): the row above is the state before this step
                    }
                    if (fabs(v) <= 1e-8 && fabs(a_out) <= 1e-8) a_out = -5.0;
                    active = act && !(wn > vmax2); prev_acc = acc;
                    wi = wnext; orig_p = wold; orig_i = wold; g_p = k_c * axg; kabs_i = k_c;
                }
                vv[c] = v; aa[c] = a_out;
            }
            const int left = nst - base;
            lane_emit_flush(*em, lane, left < 0 ? 0 : (left < CB ? left : CB), n - 2 - base, vv, aa);
#pragma unroll
            for (int c = 0; c < CB; ++c) { kr[c] = kn[c]; wr[c] = wq[c]; if constexpr (CAP::active) cr[c] = cq[c]; }
        }
        em->w0 = wi;
    } else if constexpr (EM == 1 && AXM1) {
        // same form: w1 = max(A0 w + te max(axa - g_i w, 0), 0), w' = min(max(te dm w1 + w + te max(axa - g_n w1, 0), 0), w1),
        // A0 = 1 + te dm (no machine limit under braking)
        const double axg = axa * icay;
        double orig_p = wi, g_p = kabs_i * axg;
        bool active = false, prev_acc = false;
        ke_t kr[LCHB], kn[LCHB]; double wr[LCHB], wq[LCHB];
        const int nst = n - 1;
#pragma unroll
        for (int c = 0; c < LCHB; ++c) {
            const int r = n - 2 - c >= 0 ? n - 2 - c : 0;
            kr[c] = KE_AT(r); wr[c] = Dp[(size_t)r * 64];
        }
        auto step = [&](const ke_t& rec, double wold, int i, bool valid) {
            const bool acc = wold > orig_p;
            const bool act = active || (acc && !prev_acc);
            const double te = 2.0 * (double)rec.y, tdm = te * dm, g_n = (double)rec.x * axg;
            const double w1 = fmax(fma(te, fmax(fma(-g_p, wi, axa), 0.0), fma(tdm, wi, wi)), 0.0);
            const double wn = fmin(fmax(fma(te, fmax(fma(-g_n, w1, axa), 0.0), fma(tdm, w1, wi)), 0.0), w1);
            const bool take = act && wn < wold;
            const double wnext = take ? wn : wold;
            if (valid) {
                // only rows the sweep lowers are rewritten: an unconditional store per row made the sweep 1.5x slower (gfx9 counts
                // loads and stores in ONE in-order vmcnt, so every wait for prefetched rows also waits for the older stores)
                if (take) Dp[(size_t)(n - 2 - i) * 64] = wn;
                active = act && !(wn > vmax2); prev_acc = acc;
                wi = wnext; orig_p = wold; g_p = g_n;
            }
        };
        int base = 0;
        for (; base + LCHB <= nst; base += LCHB) {
            // rows of the next chunk are not written by this chunk's steps (a step only rewrites its own row n - 2 - i)
#pragma unroll
            for (int c = 0; c < LCHB; ++c) {
                const int r = n - 2 - base - LCHB - c >= 0 ? n - 2 - base - LCHB - c : 0;
                kn[c] = KE_AT(r); wq[c] = Dp[(size_t)r * 64];
            }
#pragma unroll
            for (int c = 0; c < LCHB; ++c) step(kr[c], wr[c], base + c, true);
#pragma unroll
            for (int c = 0; c < LCHB; ++c) { kr[c] = kn[c]; wr[c] = wq[c]; }
        }
        if (base < nst) {
#pragma unroll
            for (int c = 0; c < LCHB; ++c) step(kr[c], wr[c], base + c, base + c < nst);
        }
    } else {
        double orig_i = wi;
        bool active = false, prev_acc = false;
        ke_t kr[LCHB], kn[LCHB]; double wr[LCHB], wq[LCHB];
#pragma unroll
        for (int c = 0; c < LCHB; ++c) {
            const int r = n - 2 - c >= 0 ? n - 2 - c : 0;
            kr[c] = KE_AT(r); wr[c] = Dp[(size_t)r * 64];
        }
        for (int base = 0; base < n - 1; base += LCHB) {
            // rows of the next chunk are not written by this chunk's steps (a step only rewrites its own row n - 2 - i)
#pragma unroll
            for (int c = 0; c < LCHB; ++c) {
                const int r = n - 2 - base - LCHB - c >= 0 ? n - 2 - base - LCHB - c : 0;
                kn[c] = KE_AT(r); wq[c] = Dp[(size_t)r * 64];
            }
#pragma unroll
            for (int c = 0; c < LCHB; ++c) {
                const int i = base + c;
                const bool valid = i < n - 1;
                const double wold = wr[c], e_b = (double)kr[c].y, k_c = (double)kr[c].x;
                const bool acc = wold - orig_i > 0.0;
                const bool act = active || (acc && !prev_acc);
                const double kq_i = kabs_i * icay, kq_n = k_c * icay;
                double wn;
                if constexpr (EM == 1 && AXM1) {
                    const double te = 2.0 * e_b;
                    const double A0 = 1.0 + te * dm, A1 = A0 - te * (axa * kq_i), B1 = te * axa;
                    const double C0 = te * dm, C1 = C0 - te * (axa * kq_n), D1 = te * axa;
                    wn = fmax(fmax(fma(A1, wi, B1), A0 * wi), 0.0);
                    const double t0 = fma(C0, wn, wi), t1 = fma(C1, wn, wi + D1);
                    wn = fmin(fmax(fmax(t1, t0), 0.0), wn);
                } else {
                    const double a = ax_poss_w<EM, AXM1, VMODE_DECEL_BACKW>(wi, kq_i, cax, p, axm_tab, axm1);
                    wn = fmax(wi + 2.0 * a * e_b, 0.0);
                    const double a2 = ax_poss_w<EM, AXM1, VMODE_DECEL_BACKW>(wn, kq_n, cax, p, axm_tab, axm1);
                    wn = fmin(fmax(wi + 2.0 * a2 * e_b, 0.0), wn);
                }
                const bool take = act && wn < wold;
                const double wnext = take ? wn : wold;
                if (valid && take) Dp[(size_t)(n - 2 - i) * 64] = wn;
                const bool act_out = act && !(wn > vmax2);
                active = valid ? act_out : active; prev_acc = valid ? acc : prev_acc;
                orig_i = valid ? wold : orig_i; wi = valid ? wnext : wi; kabs_i = valid ? k_c : kabs_i;
            }
#pragma unroll
            for (int c = 0; c < LCHB; ++c) { kr[c] = kn[c]; wr[c] = wq[c]; }
        }
    }
}

#undef KE_AT

#define VF_BOUND_FOLLOW 1
#define VF_TOO_CLOSE    2
#define VF_HAS_GENERIC  4
#define VF_BOUND_GENERIC 8
#define VF_COMPOSE 16
#define VF_DONE 32                 // the lane kernel wrote the slot's vx / ax / flags itself (round 6: follow jobs of batches): nothing left for k_vel_final

// The CONTROLLED part of calc_vel_profile_follow.py:78-294 for one lane (= one follow job): ego brake profile -> plane P2, opponent stop
// distance on the global race line from index idx_s_opp, characteristic indices, segment profile -> plane P3. "vx_profile" (:289 / :294)
// is then: P2 in front of row n_decel - 1 (all rows when !two_seg), P3 up to stop_idx, zeros behind -- composed by the caller. Shared by the
// batch velocity stage (k_vel_lanes) and the fleet's lane kernel of the follow jobs (k_fleet_follow_lanes, round 5).
struct LaneFollowOut { int vel_bound, too_close, two_seg, n_decel, stop_idx; };
template <int EM, bool AXM1>
__device__ __forceinline__ LaneFollowOut lane_follow_controlled(const DevLat& lat, const LaneProf& L, double* P2, double* P3, int n, double cax, double cay,
                                                                const DevVelParams& p, const double* axm_tab, double v_start, double v_ego, double v_obj,
                                                                double obj_dist, double safety_d_in, int idx_s_opp, long long* dbg, int drow)
{
    const double icay = 1.0 / cay;
    int vel_bound = 1;
    const double control_d = p.c_p * safety_d_in + p.len_veh, safety_d = safety_d_in + p.len_veh;
    const int too_close = (obj_dist - safety_d) < 0.0 ? 1 : 0;
    const double v_max = p.v_max;
    double v_control;
    if (p.ctrl == 0) v_control = (v_obj - p.k_p * (control_d - obj_dist) + p.k_d * (v_obj - v_ego));
    else {
        double a = (control_d - obj_dist) * D_PI / 2 * 1 / p.tan_w;
        const double lo = -D_PI / 2 + 1e-5, hi = D_PI / 2 - 1e-5;
        a = a < lo ? lo : (a > hi ? hi : a);
        v_control = (v_obj - tan(a) * p.k_p + p.k_d * (v_obj - v_ego));
    }
    if (v_control < 0.0) v_control = 0.0;
    if (v_control > v_max) v_control = v_max;
    const double wctl = v_control * v_control;

    // opponent brake distance on the global race line (:169-199); rows are gathered in chunks
    const int G = lat.G - 1;
    const double* grl = lat.glob_rl;
    const double vel0 = grl[(size_t)idx_s_opp * 5 + 4];
    double wopp = fmin(v_obj, vel0); wopp *= wopp;
    double opp_stop = 0.0;
    for (int base = 0; base < G && wopp > 0.01; base += LCH) {
        double kr[LCH], lr[LCH];
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            int j = base + c + idx_s_opp; j = j % G;
            kr[c] = fabs(grl[(size_t)j * 5 + 3]); lr[c] = grl[(size_t)(j + 1) * 5] - grl[(size_t)j * 5];
        }
#pragma unroll
        for (int c = 0; c < LCH; ++c) {
            const int k = base + c;
            if (k < G && wopp > 0.01) {
                opp_stop += lr[c];
                if (k + 1 >= G) wopp = 0.0;
                else {
                    const double a = ax_poss_w<EM, true, VMODE_DECEL_FORW>(wopp, kr[c] * (1.0 / 14.0), 14.0, p, axm_tab, 0.0);
                    const double r = wopp + 2.0 * a * lr[c];
                    wopp = r < 0.0 ? 0.0 : r;
                }
            }
        }
    }
    vl_stamp(dbg, drow, 1);
    // one pass over the path rows: ego brake profile -> P2 (:152-159), ego stop distance (:162-166), first index at
    // or below the control speed (:254), arc length and stop index (:203-209)
    const double s_stop = obj_dist - safety_d + opp_stop;                            // :206
    double ego_stop = 0.0, s_run = 0.0, s_last = 0.0; int first_le = -1, stop_idx = 0;
    {
        double w = v_start * v_start; bool braking = true, counting = true, searching = true;
        double kr[LCH], er[LCH], kn[LCH], en[LCH];
        auto load_rows = [&](int base, double (&k)[LCH], double (&e)[LCH]) {
#pragma unroll
            for (int c = 0; c < LCH; ++c) {
                const int r = base + c < n ? base + c : n - 1;
                const ke_t ke = L.KE[kep_row(r)]; k[c] = (double)ke.x; e[c] = (double)ke.y;
            }
        };
        load_rows(0, kr, er);
        for (int base = 0; base < n; base += LCH) {
            if (base + LCH < n) load_rows(base + LCH, kn, en);         // next chunk in flight during this chunk's steps
#pragma unroll
            for (int c = 0; c < LCH; ++c) {
                const int i = base + c;
                if (i < n) {
                    const double wv = braking ? w : 0.0;
                    P2[(size_t)i * 64] = wv;
                    if (first_le < 0 && wv <= wctl) first_le = i;
                    if (counting) { if (wv > 0.01) ego_stop += er[c]; else counting = false; }
                    if (searching) { if (i < n - 1 && s_run < s_stop) stop_idx = i + 1; else searching = false; }
                    if (i == n - 1) s_last = s_run; // s[n - 1]
                    s_run += er[c];                 // s[i + 1]
                    if (braking && i + 1 < n) {
                        const double kq = kr[c] * icay, e = er[c];
                        double r;
                        if constexpr (EM == 1) {
                            const double te = 2.0 * e, axa = fabs(cax);
                            const double A0 = 1.0 - te * p.drag_m, A1 = A0 + te * (axa * kq);
                            r = fmin(fma(A1, w, -te * axa), A0 * w);
                        } else {
                            r = w + 2.0 * ax_poss_w<EM, true, VMODE_DECEL_FORW>(w, kq, cax, p, axm_tab, 0.0) * e;
                        }
                        if (r < 0.0) braking = false; else w = r;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < LCH; ++c) { kr[c] = kn[c]; er[c] = en[c]; }
        }
    }
    vl_stamp(dbg, drow, 2);
    double v_end = 0.0;
    if (s_stop > s_last) {                                                           // :212-221
        // the stop point lies beyond the path: end velocity = race-line velocity where the opponent's remaining brake
        // distance is used up; the element lengths are fetched LCH at a time (one round trip per chunk, not per step)
        const double s_ends = opp_stop - (s_stop - s_last);
        int idx = 0; double summed = 0.0; bool run = summed < s_ends;
        while (run) {
            double dl[LCH];
#pragma unroll
            for (int c = 0; c < LCH; ++c) {
                int j = idx + c + idx_s_opp; j = j % G;
                dl[c] = grl[(size_t)(j + 1) * 5] - grl[(size_t)j * 5];
            }
#pragma unroll
            for (int c = 0; c < LCH; ++c)
                if (run) { if (summed < s_ends && idx < G) { summed += dl[c]; ++idx; } else run = false; }
            if (run && !(summed < s_ends && idx < G)) run = false;
        }
        int j = (idx % G) + idx_s_opp; if (j >= G) j -= G;
        v_end = grl[(size_t)j * 5 + 4];
    }
    vl_stamp(dbg, drow, 3);
    int idx_c = 0, n_decel = 0; const bool two_seg = ego_stop < s_stop;
    if (two_seg) {                                                                   // :247-292
        double vcs = v_start;
        if (v_start > v_control && stop_idx >= 2) {
            idx_c = first_le < 0 ? 0 : first_le;
            if (idx_c > stop_idx) idx_c = stop_idx;
            if (idx_c == 0) idx_c = stop_idx;
            n_decel = idx_c + 1 < n ? idx_c + 1 : n;
            vcs = sqrt(P2[(size_t)(n_decel - 1) * 64]);
        } else if (!(stop_idx >= 2)) vel_bound = 0;
        const int m = (stop_idx + 1 < n ? stop_idx + 1 : n) - idx_c;
        if (stop_idx - idx_c > 0) {
            lane_fb_profile<EM, AXM1>(L, P3, idx_c, m, cax, cay, p, axm_tab, v_control, vcs, true, v_end);
            if (fabs(sqrt(P3[(size_t)idx_c * 64]) - vcs) > 1.0) vel_bound = 0;
        } else if (stop_idx - idx_c == 0) P3[(size_t)idx_c * 64] = vcs * vcs;
        const double first_v = (n_decel - 1 > 0) ? sqrt(P2[0]) : sqrt(P3[0]);
        if (fabs(first_v - v_start) > 1.0) vel_bound = 0;
    }
    vl_stamp(dbg, drow, 4);
    LaneFollowOut o; o.vel_bound = vel_bound; o.too_close = too_close; o.two_seg = two_seg ? 1 : 0; o.n_decel = n_decel; o.stop_idx = stop_idx;
    return o;
}

// the generic forward-backward profile of a slot (OTH.py:834-903) into plane D; returns its vel_bound flag
template <int EM, bool AXM1>
__device__ __forceinline__ int lane_generic_profile(const DevLat& lat, const DevPathsOut& out, const LaneProf& L, double* D, int slot, int n,
                                    int reduced, double cax, double cay, const DevVelParams& p, const double* axm_tab,
                                    double vel_plan, double v_max_offset)
{
    const int goal = out.goal_layer[slot];
    const int end_node = out.nodes[(size_t)slot * out.cap_nodes + out.n_nodes[slot] - 1];
    int dn = end_node - lat.rl_idx[goal]; if (dn < 0) dn = -dn;
    const double raceline_offset = (double)dn * lat.lat_offset;
    double v_end; int v_idx;
    if (reduced) {
        v_end = 0.0;
        double spl_len = 0.0;
        for (int i = 0; i < n - 1; ++i) spl_len += (double)L.KE[kep_row(i)].y;
        int first = -1; double c = 0.0;
        for (int i = 0; i < n - 1; ++i) { c += (double)L.KE[kep_row(i)].y; if (first < 0 && !(c < (spl_len - 5.0))) first = i; }
        v_idx = (first < 0 ? 0 : first) + 1;
        if (v_idx == 1 && n > 1) v_idx = n;
    } else {
        v_end = lat.vel_rl[goal];
        const double red = v_end * lat.vel_decrease_lat * raceline_offset;
        v_end -= (red < v_end ? red : v_end);
        v_idx = n;
    }
    if (v_idx > 1) lane_fb_profile<EM, AXM1>(L, D, 0, v_idx, cax, cay, p, axm_tab, p.v_max, vel_plan, true, v_end);
    else D[0] = 0.0;
    for (int i = (v_idx > 1 ? v_idx : 1); i < n; ++i) D[(size_t)i * 64] = 0.0;
    return fabs(sqrt(D[0]) - vel_plan) < v_max_offset ? 1 : 0;
}

// The same with DIRECT OUTPUT (LaneEmit): vx / ax of the slot are written from here -- the rows the backward sweep finalises through the
// LDS transposition of lane_emit_flush, the rows it does not own (the profile's last row, the zero tail of a reduced horizon) by the lane
// itself. Called by ALL 64 lanes of the wave (`have` = the lane owns a job); returns the vel_bound flag.
template <int EM, bool AXM1>
__device__ __forceinline__ int lane_generic_profile_emit(const DevLat& lat, const DevPathsOut& out, const LaneProf& L, double* D, bool have, int slot, int n,
                                                         int reduced, double cax, double cay, const DevVelParams& p, const double* axm_tab,
                                                         double vel_plan, double v_max_offset, LaneEmit& E, int lane, double* o_vx)
{
    double v_end = 0.0; int v_idx = 0;
    if (have) {
        const int goal = out.goal_layer[slot];
        const int end_node = out.nodes[(size_t)slot * out.cap_nodes + out.n_nodes[slot] - 1];
        int dn = end_node - lat.rl_idx[goal]; if (dn < 0) dn = -dn;
        const double raceline_offset = (double)dn * lat.lat_offset;
        if (reduced) {
            double spl_len = 0.0;
            for (int i = 0; i < n - 1; ++i) spl_len += (double)L.KE[kep_row(i)].y;
            int first = -1; double c = 0.0;
            for (int i = 0; i < n - 1; ++i) { c += (double)L.KE[kep_row(i)].y; if (first < 0 && !(c < (spl_len - 5.0))) first = i; }
            v_idx = (first < 0 ? 0 : first) + 1;
            if (v_idx == 1 && n > 1) v_idx = n;
        } else {
            v_end = lat.vel_rl[goal];
            const double red = v_end * lat.vel_decrease_lat * raceline_offset;
            v_end -= (red < v_end ? red : v_end);
            v_idx = n;
        }
    }
    const int n_prof = v_idx > 1 ? v_idx : 0;                   // rows of the profile proper (0: the path has none, everything is zero)
    lane_fb_profile<EM, AXM1, true>(L, D, 0, n_prof, cax, cay, p, axm_tab, p.v_max, vel_plan, true, v_end, nullptr, -1, &E, lane);
    if (!have) return 0;
    // rows n_prof - 1 .. n - 1: the profile's last row (its v^2 is in the plane: the backward sweep starts below it) and the zero tail;
    // ax of a row differentiates towards the row above, which is zero (or absent: last row of the path, ax = 0) for all of them
    const double w0 = n_prof >= 2 ? E.w0 : 0.0;
    double* o_ax = o_vx + E.ax_delta;
    for (int i = (n_prof >= 2 ? n_prof - 1 : 0); i < n; ++i) {
        const double w = (n_prof >= 2 && i == n_prof - 1) ? D[(size_t)i * 64] : 0.0;
        const double v = sqrt(w);
        double a = 0.0;
        if (i < n - 1) {
            a = (0.0 - w) / (2.0 * (double)L.KE[kep_row(i)].y);
            if (fabs(v) <= 1e-8 && fabs(a) <= 1e-8) a = -5.0;
        }
        o_vx[i] = v; o_ax[i] = a;
    }
    return fabs(sqrt(w0) - vel_plan) < v_max_offset ? 1 : 0;
}

// amdgpu_waves_per_eu(2): at most 256 registers. A velocity wave shares its SIMD with path waves of 128 registers each (4 x 128 = the whole
// file): one of up to 256 registers waits for TWO of them to retire, one of 257+ for THREE. Same-box A/B of this kernel at 265 registers
// (what the allocator takes when left alone, round 6) against 256 with four values parked in scratch: 38.4 against 40.6 M ticks/s
// (profiles/r06j_ab_follow_emit.txt) -- the footprint of a velocity wave is what the path kernel pays for.
template <int EM, bool AXM1>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_vel_lanes(DevLat lat, DevPathsIn in, DevPathsOut out, DevVelParams p,
                                                  DevTickVelIn vin, DevVelPrep prep, VelPlanes vp, int n_slots, int n_scen,
                                                  int n_blocks0, long long* dbg, DevTickVelOut vout, int emit)
{
    LTPL_VEL_SETPRIO();
    __shared__ __align__(16) double le_tbuf[32 * LE_PITCH];        // direct output of the generic jobs (LaneEmit)
    __shared__ int le_cnt[64], le_rhi[64];
    __shared__ unsigned long long le_o[64];
    // blocks [0, nbG): generic jobs; [nbG, nbG + nbF): follow jobs, controlled part; [nbG + nbF, nbG + 2 nbF): follow jobs,
    // unconstrained profile. Waves beyond the job counters (known only on the device) exit at once.
    __shared__ double axm_tab[128];
    const int lane = threadIdx.x;
    const int nbG = n_blocks0, nbF = (n_scen + 63) / 64;
    const int cntG = out.job_cnt[0], cntF = out.job_cnt[1];
    const int b = blockIdx.x;
    if ((b < nbG && b * 64 >= cntG) || (b >= nbG && ((b - nbG) % nbF) * 64 >= cntF)) return;
    for (int i = lane; i < 2 * p.n_axm; i += 64) axm_tab[i] = p.axm[i];
    __syncthreads();
    const int drow = b < nbG ? (b < 64 ? b : -1) : (b < nbG + nbF ? (b - nbG < 64 ? 64 + b - nbG : -1) : (b - nbG - nbF < 64 ? 128 + b - nbG - nbF : -1));
    vl_stamp(dbg, drow, 0);
    const double cax = vin.gg_ax, cay = vin.gg_ay;
    const int fbase = out.n_slots_pad;
    if (b >= nbG + nbF) {
        // ---- follow jobs, unconstrained profile (calc_vel_profile_follow.py:297-307) -------------------------------------
        const int j = (b - nbG - nbF) * 64 + lane;
        if (j >= cntF) return;
        const int2 js = out.job_slot[fbase + j];
        const int slot = js.x;
        if ((emit & 2) && !out.reduced[slot]) return;                 // (large batches: the follow block below runs this profile itself and puts the job's rows out)
        LaneProf L; L.KE = vp.KE + kep_base(fbase + j, vp.plane_rows);
        lane_fb_profile<EM, AXM1>(L, vp.P1 + tile_base(j, vp.cap_pts), 0, js.y, cax, cay, p, axm_tab, p.v_max,
                                  vin.vel_plan[slot / LTPL_MAX_ACTIONS], false, 0.0, dbg, drow);
        vl_stamp(dbg, drow, 6);
        return;
    }
    const bool emit_generic = (emit & 1) != 0;                       // emit bit 0: generic jobs write their outputs here; bit 1: follow jobs do
    if (b < nbG && emit_generic) {
        // ---- generic jobs (every non-follow primitive, OTH.py:834-941): profile AND outputs, all 64 lanes stay for the output transposition.
        //      Batches only (emit_generic = the launch has >= LTPL_EMIT_MIN_SCEN scenarios): the transposition is ~160 instructions per chunk
        //      of 12 rows on the lane kernel's serial chain -- with a handful of jobs (single ticks on the long-horizon lattice: C5, 1 200
        //      rows) k_vel_final's many short waves do that work faster (measured: +100 us on a 1.5 ms tick) ----
        const int j = b * 64 + lane;
        const bool have = j < cntG;
        const int2 js = have ? out.job_slot[j] : make_int2(0, 0);
        const int slot = js.x, n = have ? js.y : 0;
        {   // longest profile of the launch (k_vel_final skips row chunks beyond it): one atomic per wave
            const int nmax = wave_max_i32(n);
            if (lane == 0) atomicMax(&out.job_cnt[2], nmax);
        }
        LaneProf L; L.KE = vp.KE + kep_base(j, vp.plane_rows);
        double* o_vx = vout.vx + (size_t)slot * out.cap_pts;
        le_o[lane] = have ? (unsigned long long)o_vx : 0ull;
        LaneEmit E; E.tbuf = le_tbuf; E.s_cnt = le_cnt; E.s_rhi = le_rhi; E.s_o = le_o; E.ax_delta = (long long)(vout.ax - vout.vx); E.w0 = 0.0;
        wave_sync_lds();
        const int bound = lane_generic_profile_emit<EM, AXM1>(lat, out, L, vp.P0 + tile_base(j, vp.cap_pts), have, slot, n, have ? out.reduced[slot] : 0,
                                                              cax, cay, p, axm_tab, have ? vin.vel_plan[slot / LTPL_MAX_ACTIONS] : 0.0, vin.v_max_offset, E, lane, o_vx);
        if (have) {
            vout.vel_bound[slot] = bound; vout.too_close[slot] = 0;
            vp.flags[j] = VF_BOUND_FOLLOW | (bound ? VF_BOUND_GENERIC : 0);
        }
        vl_stamp(dbg, drow, 6);
        return;
    }
    if (b >= nbG && (emit & 2)) {
        // ---- follow jobs of LARGE batches (round 6): the controlled part, then the UNCONSTRAINED profile in the same lane, whose backward sweep puts
        //      out min("vx_profile", unconstrained profile) -- the follow job's rows (FollowCap) -- with vx / ax through the same LDS transposition as
        //      the generic jobs. Otherwise the two halves run on two waves and k_vel_final reads four planes per follow job to compose, intersect,
        //      differentiate and transpose: as much wave-time as the lane kernel itself (profiles/r06f_vel_pmc.txt). Same operations on the same
        //      values in the same order: results unchanged bit for bit. Reduced-horizon follow jobs (rare) keep the old route inside this block.
        //      Large batches only (ltpl_handle::follow_emit_min_scen): the lane now runs five sweeps one after the other instead of three next to
        //      two -- neutral on the throughput of 32 768 scenarios per step, 135 against 117 us for a step of 1 024 (profiles/r06z_bench.json).
        const int j = (b - nbG) * 64 + lane;
        const bool have = j < cntF;
        const int tile = fbase + j;                                   // (the planes hold n_scen_pad follow tiles: a lane without a job has one too)
        const int2 js = have ? out.job_slot[tile] : make_int2(0, 0);
        const int slot = js.x, n = have ? js.y : 0;
        {
            const int nmax = wave_max_i32(n);
            if (lane == 0) atomicMax(&out.job_cnt[2], nmax);
        }
        const int s = slot / LTPL_MAX_ACTIONS;
        LaneProf L; L.KE = vp.KE + kep_base(tile, vp.plane_rows);
        double* P2 = vp.P2 + tile_base(j, vp.cap_pts);
        double* P3 = vp.P3 + tile_base(j, vp.cap_pts);
        const double vel_plan = have ? vin.vel_plan[s] : 0.0;
        const int reduced = have ? out.reduced[slot] : 0;
        FollowCap cap; cap.P2 = P2; cap.P3 = P3; cap.nd = -1; cap.stop_idx = 0;
        bool direct = false;
        int o_bound = 0, o_close = 0;
        if (have) {
            const LaneFollowOut fo = lane_follow_controlled<EM, AXM1>(lat, L, P2, P3, n, cax, cay, p, axm_tab, vel_plan, vin.vel_est[s], prep.v_obj[slot],
                                                                      prep.obj_dist[slot], vin.safety_d, prep.idx_s_opp[slot], dbg, drow);
            o_bound = fo.vel_bound; o_close = fo.too_close;
            if (!reduced) { direct = true; cap.nd = fo.two_seg ? fo.n_decel : -1; cap.stop_idx = fo.stop_idx; }
            else {
                // reduced horizon: "vx_profile" materialised in P0, the generic profile on top of it in P3 (OTH.py:834-923), k_vel_final chooses by
                // row 5; the unconstrained profile of such a job comes from its own wave below
                double* P0 = vp.P0 + tile_base(tile, vp.cap_pts);
                int flags = VF_BOUND_FOLLOW | VF_HAS_GENERIC;
                if (fo.too_close) flags |= VF_TOO_CLOSE;
                if (!fo.vel_bound) flags &= ~VF_BOUND_FOLLOW;
                const bool two_seg = fo.two_seg != 0;
                for (int i = 0; i < n; ++i) {
                    const bool from_b = !two_seg || i < fo.n_decel - 1;
                    P0[(size_t)i * 64] = from_b ? P2[(size_t)i * 64] : ((i > fo.stop_idx) ? 0.0 : P3[(size_t)i * 64]);
                }
                if (lane_generic_profile<EM, AXM1>(lat, out, L, P3, slot, n, reduced, cax, cay, p, axm_tab, vel_plan, vin.v_max_offset))
                    flags |= VF_BOUND_GENERIC;
                vp.flags[tile] = flags;
            }
        }
        double* o_vx = vout.vx + (size_t)slot * out.cap_pts;
        le_o[lane] = direct ? (unsigned long long)o_vx : 0ull;
        LaneEmit E; E.tbuf = le_tbuf; E.s_cnt = le_cnt; E.s_rhi = le_rhi; E.s_o = le_o; E.ax_delta = (long long)(vout.ax - vout.vx); E.w0 = 0.0;
        wave_sync_lds();
        // (a lane that puts nothing out walks the sweep with zero rows; its one store -- the start value of row 0 -- goes to its brake plane, not to the
        //  unconstrained plane another wave may be writing for a reduced-horizon job)
        double* D = direct ? vp.P1 + tile_base(j, vp.cap_pts) : P2;
        lane_fb_profile<EM, AXM1, true, FollowCap>(L, D, 0, direct && n >= 2 ? n : 0, cax, cay, p, axm_tab, p.v_max, vel_plan, false, 0.0, nullptr, -1, &E, lane, cap);
        if (direct) {
            // the top row: the backward sweep starts below it; its ax differentiates towards nothing (last row of the path)
            const int i = n - 1;
            const double w = fmin(cap.at(i), D[(size_t)i * 64]);
            o_vx[i] = sqrt(w); o_vx[i + E.ax_delta] = 0.0;
            vout.vel_bound[slot] = o_bound; vout.too_close[slot] = o_close;
            vp.flags[tile] = VF_DONE;
        }
        vl_stamp(dbg, drow, 6);
        return;
    }
    const bool fjob = b >= nbG;                                 // (follow jobs, controlled part; generic jobs of small launches: profile into P0, outputs by k_vel_final)
    const int j = (fjob ? b - nbG : b) * 64 + lane;
    if (j >= (fjob ? cntF : cntG)) return;
    const int tile = fjob ? fbase + j : j;
    const int2 js = out.job_slot[tile];
    const int slot = js.x;
    {
        // longest profile of the launch (k_vel_final skips row chunks beyond it): one atomic per wave
        int nmax = js.y;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(nmax, m); nmax = o > nmax ? o : nmax; }   // (lanes beyond the job count have left: no DPP form here)
        if (lane == 0) atomicMax(&out.job_cnt[2], nmax);
    }
    const int s = slot / LTPL_MAX_ACTIONS;
    const int n = js.y;
    LaneProf L; L.KE = vp.KE + kep_base(tile, vp.plane_rows);
    double* P0 = vp.P0 + tile_base(tile, vp.cap_pts);
    double* P2 = vp.P2 + tile_base(fjob ? j : 0, vp.cap_pts);
    double* P3 = vp.P3 + tile_base(fjob ? j : 0, vp.cap_pts);
    const double vel_plan = vin.vel_plan[s];
    const int name = out.action_id[slot], reduced = out.reduced[slot];
    int flags = VF_BOUND_FOLLOW;

    if (name == LTPL_ACT_FOLLOW) {                                                       // OTH.py:763-830
        // calc_vel_profile_follow.py:78-294 for this lane
        const LaneFollowOut fo = lane_follow_controlled<EM, AXM1>(lat, L, P2, P3, n, cax, cay, p, axm_tab, vel_plan, vin.vel_est[s], prep.v_obj[slot],
                                                                  prep.obj_dist[slot], vin.safety_d, prep.idx_s_opp[slot], dbg, drow);
        if (fo.too_close) flags |= VF_TOO_CLOSE;
        int vel_bound = fo.vel_bound;
        const bool two_seg = fo.two_seg != 0; const int n_decel = fo.n_decel, stop_idx = fo.stop_idx;
        // "vx_profile" (:289 / :294) = brake profile in front, segment profile up to the stop index, zeros behind. Normally only
        // described (two indices) and composed by the row-parallel final kernel; a reduced-horizon follow job needs P3 for its
        // generic profile, so there the composition is materialised in P0 here.
        if (!reduced) { vp.fseg[2 * j] = two_seg ? n_decel : -1; vp.fseg[2 * j + 1] = stop_idx; flags |= VF_COMPOSE; }
        else for (int base = 0; base < n; base += LCH) {
            double a[LCH];
#pragma unroll
            for (int c = 0; c < LCH; ++c) {
                const int i = base + c < n ? base + c : n - 1;
                const bool from_b = !two_seg || i < n_decel - 1;
                a[c] = from_b ? P2[(size_t)i * 64] : ((i > stop_idx) ? 0.0 : P3[(size_t)i * 64]);
            }
#pragma unroll
            for (int c = 0; c < LCH; ++c) if (base + c < n) P0[(size_t)(base + c) * 64] = a[c];
        }
        vl_stamp(dbg, drow, 5);
        if (!vel_bound) flags &= ~VF_BOUND_FOLLOW;
        if (reduced) {                                                                   // OTH.py:834-923 on top of follow
            flags |= VF_HAS_GENERIC;
            if (lane_generic_profile<EM, AXM1>(lat, out, L, P3, slot, n, reduced, cax, cay, p, axm_tab, vel_plan, vin.v_max_offset))
                flags |= VF_BOUND_GENERIC;
        }
    } else {
        if (lane_generic_profile<EM, AXM1>(lat, out, L, P0, slot, n, reduced, cax, cay, p, axm_tab, vel_plan, vin.v_max_offset))
            flags |= VF_BOUND_GENERIC;
    }
    vp.flags[tile] = flags;
    vl_stamp(dbg, drow, 6);
}

// final step of the batch velocity stage, no recurrence: intersection of the two follow profiles
// (calc_vel_profile_follow.py:310), choice between follow and generic profile for reduced horizons (OTH.py:923, row 5),
// vx = sqrt(w), ax from neighbouring points with -5 at standstill (OTH.py:925-941). Rows are independent of each other
// (ax_i only needs w_i, w_i+1 and the element length e_i), so the work is spread over (tile of 64 jobs, chunk of FCH rows).
// Two lane mappings in one wave: the planes are tiled by job, so they are READ with lane = job (one coalesced 512-byte line per
// row); the outputs are rows of a slot, so they are WRITTEN with lane = (job, pair of rows) after a transposition through LDS: one
// store instruction covers 8 jobs x 128 contiguous bytes instead of 64 jobs x 16 bytes on 64 different lines (round 2: 122 us ->
// see DESIGN.md section 6; the stores of the lane = job form kept the address units busy, not HBM).
#define FCH 16
#define FPITCH (FCH + 2)           // doubles per job row in LDS (16-byte aligned pairs, odd multiple of 16 bytes: conflict-free b128 access)
__global__ __launch_bounds__(64) void k_vel_final(DevPathsOut out, DevTickVelIn vin, DevTickVelOut vout, VelPlanes vp, int n_slots,
                                                  int n_scen, int first_block)
{
    LTPL_VEL_SETPRIO();
    __shared__ __align__(16) double tbuf[32 * FPITCH];     // 32 jobs at a time: 4.5 KB, so that many blocks fit next to a resident path kernel
    __shared__ int s_slot[64], s_n[64];
    // blocks [0, nbG): generic jobs, [nbG, nbG + nbF): follow jobs; slots without a path were initialised by the path kernel
    // (round 5: launched for the follow tiles only -- first_block = nbG; the generic jobs' outputs come from the lane kernel's backward sweep)
    const int nbG = (n_slots + 63) / 64;
    const int bx = (int)blockIdx.x + first_block;
    const bool fjob = bx >= nbG;
    const int lane = threadIdx.x;
    const int j = (fjob ? bx - nbG : bx) * 64 + lane;
    const int cnt = out.job_cnt[fjob ? 1 : 0];
    if ((j - lane) >= cnt) return;                                      // whole tile without jobs (uniform)
    const bool have = j < cnt;
    const int tile = fjob ? out.n_slots_pad + j : j;
    const int nmax_all = out.job_cnt[2];                                // longest profile of the launch (lane kernel)
    const int2 js = have ? out.job_slot[tile] : make_int2(0, 0);
    const int slot = js.x;
    // (round 6: a follow job of a batch is finished by the lane kernel -- VF_DONE -- unless its horizon is reduced; such a job counts zero rows here,
    //  and a tile without anything left leaves at once)
    const int tflags = have ? vp.flags[tile] : VF_DONE;
    const int n = (tflags & VF_DONE) ? 0 : js.y;
    if (__ballot(n > 0) == 0ull) return;
    s_slot[lane] = slot; s_n[lane] = n;
    // blockIdx.y strides over the row chunks: a few long-lived blocks per tile instead of one block per (tile, chunk), most of
    // which used to find nothing to do
    for (int base = (int)blockIdx.y * FCH; base < nmax_all; base += (int)gridDim.y * FCH) {
    const bool act = have && base < n;
    if (__ballot(act) == 0ull) continue;                                // uniform
    double vv[FCH], aa[FCH];
    if (act) {
        const int flags = tflags;
        const bool follow = fjob;
        const double* P0 = vp.P0 + tile_base(tile, vp.cap_pts);
        const double* P1 = vp.P1 + tile_base(fjob ? j : 0, vp.cap_pts);
        const double* P2 = vp.P2 + tile_base(fjob ? j : 0, vp.cap_pts);
        const double* P3 = vp.P3 + tile_base(fjob ? j : 0, vp.cap_pts);
        const ke_t* KE = vp.KE + kep_base(tile, vp.plane_rows);
        const bool compose = follow && (flags & VF_COMPOSE);
        const int nd = compose ? vp.fseg[2 * j] : 0, stop_idx = compose ? vp.fseg[2 * j + 1] : 0;
        int vel_bound = follow ? ((flags & VF_BOUND_FOLLOW) ? 1 : 0) : ((flags & VF_BOUND_GENERIC) ? 1 : 0);
        int sel = follow ? 1 : 0;                     // 0: P0, 1: min(P0, P1), 2: P3
        if (follow && (flags & VF_HAS_GENERIC)) {
            vel_bound = (flags & VF_BOUND_GENERIC) ? 1 : 0;
            sel = 2;
            if (n >= 6) {
                const double f5 = fmin(P0[5 * 64], P1[5 * 64]);
                if (sqrt(f5) < sqrt(P3[5 * 64])) sel = 1;
            }
        }
        auto value = [&](int i) {
            const size_t o = (size_t)i * 64;
            if (compose) {                                     // sel == 1: min("vx_profile", unconstrained profile)
                const double a = (nd < 0 || i < nd - 1) ? P2[o] : (i > stop_idx ? 0.0 : P3[o]);
                return fmin(a, P1[o]);
            }
            return sel == 0 ? P0[o] : (sel == 1 ? fmin(P0[o], P1[o]) : P3[o]);
        };
        double w[FCH + 1], er[FCH];
#pragma unroll
        for (int c = 0; c <= FCH; ++c) w[c] = value(base + c < n ? base + c : n - 1);
        // element lengths: two rows per 16-byte load (rows 2m, 2m + 1 are adjacent in the blocked plane and `base` is a multiple of 16;
        // rows beyond n - 1 are unused padding of the block -- never NaN-sensitive: their ax is discarded)
#pragma unroll
        for (int c = 0; c < FCH; c += 2) {
#ifndef LTPL_VEL_F32_OPERANDS
            er[c] = KE[kep_row(base + c)].y; er[c + 1] = KE[kep_row(base + c + 1)].y;
#else
            static_assert(KE_RB >= 2, "pairs of rows in one load");
            const float4 v = *reinterpret_cast<const float4*>(&KE[kep_row(base + c)]);
            er[c] = (double)v.y; er[c + 1] = (double)v.w;
#endif
        }
#pragma unroll
        for (int c = 0; c < FCH; ++c) {
            const int i = base + c;
            const double v = sqrt(w[c]);
            double a = 0.0;
            if (i < n - 1) {
                // the reference divides by 2 (s_i+1 - s_i) with s the running sum of the element lengths; e_i differs from that
                // difference by rounding only (~1e-13 relative, tolerance of ax: 1e-5)
                a = (w[c + 1] - w[c]) / (2.0 * er[c]);
                if (fabs(v) <= 1e-8 && fabs(a) <= 1e-8) a = -5.0;
            }
            vv[c] = v; aa[c] = a;
        }
        if (base == 0) { vout.vel_bound[slot] = vel_bound; vout.too_close[slot] = (flags & VF_TOO_CLOSE) ? 1 : 0; }
    } else {
#pragma unroll
        for (int c = 0; c < FCH; ++c) { vv[c] = 0.0; aa[c] = 0.0; }
    }
    // transposition: lane = job writes its FCH values as one LDS row; lane = (job q of a group of 8, pair of rows) reads 16 bytes
    // and stores them to the slot's row (8-byte aligned 16-byte stores; rows of a slot are contiguous)
    const int q = lane >> 3, piece = lane & 7;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int half = pass & 1;                                      // jobs [32 half, 32 half + 32)
        double* dst_base = pass < 2 ? vout.vx : vout.ax;
        if ((lane >> 5) == half) {
#pragma unroll
            for (int c = 0; c < FCH; c += 2) store2(&tbuf[(lane & 31) * FPITCH + c], pass < 2 ? vv[c] : aa[c], pass < 2 ? vv[c + 1] : aa[c + 1]);
        }
        wave_sync_lds();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int jl = k * 8 + q, jj = half * 32 + jl;
            const int nn = s_n[jj], i = base + 2 * piece;
            if (i >= nn) continue;
            const dbl2 v = *reinterpret_cast<const dbl2*>(&tbuf[jl * FPITCH + 2 * piece]);
            double* o = dst_base + (size_t)s_slot[jj] * out.cap_pts + i;
            if (i + 1 < nn) store2_u(o, v.x, v.y); else o[0] = v.x;
        }
        wave_sync_lds();
    }
    }
}

// follow preparation (projection of the object and of the ego position on the path, OTH.py:774-784; projection of the object
// on the global race line, calc_vel_profile_follow.py:172-176), ONE LANE PER FOLLOW JOB like the lane kernel. Round 2 history: the
// wave-per-job form spent ~1 900 instructions per job on cross-lane reductions and half-empty lanes (47 M instructions per
// 32 768-scenario step, 8 % of the path kernel's -- 118 us of the overlapped step); a lane that scans its own job serially needs
// ~190: the race line is read with scalar loads (uniform index), the path points come from a tiled (x, y) plane the path kernel
// writes for follow jobs (coalesced rows), closest point, arc length at the foot point (a running prefix: np.cumsum's order) and
// the element in front of it are picked up in ONE pass for both query points.
#define PCH 8
__global__ __launch_bounds__(64) void k_follow_prep(DevLat lat, DevPathsIn in, DevPathsOut out, DevTickVelIn vin,
                                                    DevVelPrep prep, VelPlanes vp, int n_slots, long long* dbg)
{
    LTPL_VEL_SETPRIO();
    const int lane = threadIdx.x, cnt = out.job_cnt[1];
    if ((int)blockIdx.x * 64 >= cnt) return;
    const int drow = blockIdx.x < 64 ? 192 + (int)blockIdx.x : -1;      // LTPL_DEBUG_TIMING: rows 192 .. 255 of the stamp table
    vl_stamp(dbg, drow, 0);
    const int j0 = (int)blockIdx.x * 64 + lane;
    const bool have_job = j0 < cnt;
    const int j = have_job ? j0 : cnt - 1;                  // idle lanes repeat the last job (uniform control flow), nothing stored
    const int2 js = out.job_slot[out.n_slots_pad + j];
    const int slot = js.x, s = slot / LTPL_MAX_ACTIONS, n = js.y;
    const int ci = out.closest_obj_index[s], v0 = in.veh_off[s];
    const bool have = !(ci < 0 || ci >= in.veh_off[s + 1] - v0);
    const double ex = vin.pos_est_x[s], ey = vin.pos_est_y[s];
    double ox = ex, oy = ey, vobj = 0.0, odist = 0.0;
    if (have) { const int q = in.pos_off[v0 + ci]; ox = in.pos_x[q]; oy = in.pos_y[q]; vobj = vin.veh_vel[v0 + ci]; }
    vl_stamp(dbg, drow, 1);
    const int idx = lane_globrl_index(lat, ox, oy);
    vl_stamp(dbg, drow, 2);
    if (__ballot(have) != 0ull) {
        const double* xy = vp.XY + 2 * kep_base(j, vp.plane_rows);         // pairs: row r of this lane's job at xy + 2 * kep_row(r)
        const ke_t* KE = vp.KE + kep_base(out.n_slots_pad + j, vp.plane_rows);
        int nmax = n;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(nmax, m); nmax = o > nmax ? o : nmax; }   // (lanes beyond the job count have left: no DPP form here)
        // one pass: closest path point of the object (o) and of the ego position (e), arc length in front of it and the element before
        double bo = INFINITY, be = INFINITY, so = 0.0, se = 0.0, po = 0.0, pe = 0.0, run = 0.0, e_prev = 0.0;
        int no = 0, ne = 0;
        dbl2 pr[PCH], pn[PCH]; ke_t kr[PCH], kn[PCH];
        auto load_rows = [&](int base, dbl2 (&p)[PCH], ke_t (&k)[PCH]) {
#pragma unroll
            for (int c = 0; c < PCH; ++c) {
                const int r = base + c < n ? base + c : n - 1;
                p[c] = *reinterpret_cast<const dbl2*>(xy + 2 * kep_row(r)); k[c] = KE[kep_row(r)];
            }
        };
        load_rows(0, pr, kr);
        for (int base = 0; base < nmax; base += PCH) {
            load_rows(base + PCH, pn, kn);
#pragma unroll
            for (int c = 0; c < PCH; ++c) {
                const int i = base + c;
                const bool valid = i < n;
                const double x = pr[c].x, y = pr[c].y, e = (double)kr[c].y;
                const double ao = (x - ox) * (x - ox) + (y - oy) * (y - oy), ae = (x - ex) * (x - ex) + (y - ey) * (y - ey);
                if (valid && ao < bo) { bo = ao; no = i; so = run; po = e_prev; }
                if (valid && ae < be) { be = ae; ne = i; se = run; pe = e_prev; }
                if (valid) { run += e; e_prev = e; }
            }
#pragma unroll
            for (int c = 0; c < PCH; ++c) { pr[c] = pn[c]; kr[c] = kn[c]; }
        }
        vl_stamp(dbg, drow, 3);
        // get_s_coord.py:34-99 (closed = False) for one query: arc length of the foot point
        auto foot = [&](int nb, double s_nb, double e_before, double px, double py) {
            const int i1 = nb - 1 > 0 ? nb - 1 : 0, i2 = nb + 1 < n - 1 ? nb + 1 : n - 1;
            const dbl2 pN = *reinterpret_cast<const dbl2*>(xy + 2 * kep_row(nb)), p1 = *reinterpret_cast<const dbl2*>(xy + 2 * kep_row(i1)),
                       p2 = *reinterpret_cast<const dbl2*>(xy + 2 * kep_row(i2));
            const int ord = angle_order_dev(pN.x, pN.y, px, py, p1.x, p1.y, p2.x, p2.y);
            double ax, ay, bx, by;
            if (ord > 0) { ax = p1.x; ay = p1.y; bx = pN.x; by = pN.y; } else { ax = pN.x; ay = pN.y; bx = p2.x; by = p2.y; }
            const double t = ((px - ax) * (bx - ax) + (py - ay) * (by - ay)) / ((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
            const double fx = ax + t * (bx - ax), fy = ay + t * (by - ay);
            const double ds = sqrt((ax - fx) * (ax - fx) + (ay - fy) * (ay - fy));
            return (ord > 0 ? (nb > 0 ? s_nb - e_before : 0.0) : s_nb) + ds;        // s[i1] = s[nb] - el[nb - 1]; s[0] = 0
        };
        const double s_obj = foot(no, so, po, ox, oy), s_sta = foot(ne, se, pe, ex, ey);
        if (have) odist = s_obj - s_sta;
        vl_stamp(dbg, drow, 4);
    }
    if (have_job) { prep.obj_dist[slot] = odist; prep.v_obj[slot] = vobj; prep.obj_x[slot] = ox; prep.obj_y[slot] = oy; prep.idx_s_opp[slot] = idx; }
    vl_stamp(dbg, drow, 5);
}

// ---------------------------------------------------------------------------------------------------------------------
// compact trajectory export (ltpl_tick_batch_compact): rows [s, x, y, psi, kappa, vx, ax] of the valid slots, back to back
// ---------------------------------------------------------------------------------------------------------------------
// rows kept per slot and their exclusive prefix (one block; n_slots <= a few hundred thousand)
__global__ __launch_bounds__(1024) void k_compact_offsets(const int* valid, const int* n_pts, int n_slots, int max_rows,
                                                          int* rows_out, long long* off_out, long long* total)
{
    __shared__ long long part[1024];
    const int t = threadIdx.x, per = (n_slots + 1023) / 1024;
    const int a = t * per, b = a + per < n_slots ? a + per : n_slots;
    long long sum = 0;
    for (int i = a; i < b; ++i) {
        int r = valid[i] ? n_pts[i] : 0;
        if (max_rows > 0 && r > max_rows) r = max_rows;
        rows_out[i] = r; sum += r;
    }
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const long long v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    long long run = part[t] - sum;
    for (int i = a; i < b; ++i) { off_out[i] = run; run += rows_out[i]; }
    if (t == 1023) *total = part[1023];
}

// one wave per slot: s = [0, cumsum(el[:-1])] (OTH.py:743, wave-parallel prefix), then the seven columns of every kept row
__global__ __launch_bounds__(64) void k_compact_rows(DevPathsOut out, DevTickVelOut vout, const int* rows_kept, const long long* off,
                                                     double* dst, long long capacity_rows)
{
    const int slot = blockIdx.x, lane = threadIdx.x;
    const int n = rows_kept[slot];
    if (n <= 0) return;
    const long long o = off[slot];
    if (o + n > capacity_rows) return;
    const double* pp = out.path_param + (size_t)slot * out.cap_pts * 5;
    const double* vx = vout.vx + (size_t)slot * out.cap_pts;
    const double* ax = vout.ax + (size_t)slot * out.cap_pts;
    double carry = 0.0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const double e = (i < n) ? pp[(size_t)i * 5 + 4] : 0.0;
        double x = e;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const double y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (i < n) {
            double* r = dst + (size_t)(o + i) * 7;
            r[0] = carry + (x - e);
            r[1] = pp[(size_t)i * 5]; r[2] = pp[(size_t)i * 5 + 1]; r[3] = pp[(size_t)i * 5 + 2]; r[4] = pp[(size_t)i * 5 + 3];
            r[5] = vx[i]; r[6] = ax[i];
        }
        carry += __shfl(x, 63);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// object ingestion: ObjectListInterface.process_object_list (ObjectListInterface.py:75-153)
// ---------------------------------------------------------------------------------------------------------------------
// One lane per object: closest centre-line point (first minimum), the neighbour with the larger opening angle
// (get_s_coord.py:34-47,93-96 with closed = True), closest of the 50 np.linspace points between the two bracketing
// points, comparison of the squared distances to the interpolated bounds with the squared track width
// (check_inside_bounds.py:26-56); constant-velocity prediction and radius (ObjectListInterface.py:117-133).
__global__ __launch_bounds__(64) void k_process_objects(DevLat lat, int n_obj, double dt, const double* ox, const double* oy,
                                                        const double* oth, const double* ov, const double* olen,
                                                        int* on_track, double* pred_x, double* pred_y, double* radius)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n_obj) return;
    const double px = ox[k], py = oy[k];
    const int L = lat.L;
    int nb = 0; double best = INFINITY;
    for (int l = 0; l < L; ++l) {
        const double dx = lat.ctx[l] - px, dy = lat.cty[l] - py;
        const double d2 = dx * dx + dy * dy;
        if (d2 < best) { best = d2; nb = l; }
    }
    int i1 = nb - 1; if (i1 < 0) i1 += L;
    int i2 = nb + 1; if (i2 > L - 1) i2 = 0;
    const double ang1 = fabs(angle3pt_dev(lat.ctx[nb], lat.cty[nb], px, py, lat.ctx[i1], lat.cty[i1]));
    const double ang2 = fabs(angle3pt_dev(lat.ctx[nb], lat.cty[nb], px, py, lat.ctx[i2], lat.cty[i2]));
    const int a = ang1 >= ang2 ? i1 : nb, b = ang1 >= ang2 ? nb : i2;
    const double cax = lat.ctx[a], cay = lat.cty[a], cbx = lat.ctx[b], cby = lat.cty[b];
    // np.linspace(start, stop, 50): y_i = i * ((stop - start) / 49) + start, y_49 = stop
    const double sx = (cbx - cax) / 49.0, sy = (cby - cay) / 49.0;
    int bi = 0; double bd = INFINITY;
    for (int i = 0; i < 50; ++i) {
        const double qx = i == 49 ? cbx : (double)i * sx + cax, qy = i == 49 ? cby : (double)i * sy + cay;
        const double dx = qx - px, dy = qy - py;
        const double d2 = dx * dx + dy * dy;
        if (d2 < bd) { bd = d2; bi = i; }
    }
    auto lin = [&](double A, double B) { return bi == 49 ? B : (double)bi * ((B - A) / 49.0) + A; };
    const double l1x = lin(lat.b1x[a], lat.b1x[b]), l1y = lin(lat.b1y[a], lat.b1y[b]);
    const double l2x = lin(lat.b2x[a], lat.b2x[b]), l2y = lin(lat.b2y[a], lat.b2y[b]);
    const double d_track_2 = (l1x - l2x) * (l1x - l2x) + (l1y - l2y) * (l1y - l2y);
    const double d_b1_2 = (l1x - px) * (l1x - px) + (l1y - py) * (l1y - py);
    const double d_b2_2 = (l2x - px) * (l2x - px) + (l2y - py) * (l2y - py);
    on_track[k] = !(d_b1_2 > d_track_2 || d_b2_2 > d_track_2) ? 1 : 0;
    pred_x[k] = px - sin(oth[k]) * ov[k] * dt;
    pred_y[k] = py + cos(oth[k]) * ov[k] * dt;
    radius[k] = olen[k] / 2.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;
struct TickLayout;
static void free_resident(TickLayout* t);
static int self_test(struct ltpl_handle* h, const ltpl_lattice_desc* d);

struct ltpl_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    DevLat lat{};
    TeamLds lp1{}, lp4{};            // LDS plans of the path kernel: one wave / four waves per scenario
    int batch_nw = 1;                // waves per scenario used for batches (LTPL_BATCH_NW)
    int plan_class = 0;              // LDS plan of the one-wave batch kernel: 0 = runtime (PlanRt), 1 = PlanA, 2 = PlanB, 3 = PlanC
    int plan_class4 = 0;             // LDS plan of the four-wave kernels: 0 = runtime, 1 = PlanA4
    int long_horizon = 0;            // 1: parent tables in global memory (PlanRtG), velocity stage always through the lane kernels
    int nw1_min_scen = PIPELINE_MIN_SCEN;   // calls with at least this many scenarios use one-wave teams
    // ltpl_tick_batch & co: calls with at least this many scenarios run the batch PIPELINE (one wave per scenario + the three velocity kernels),
    // smaller ones the FUSED tick kernel (one four-wave workgroup per scenario, one launch). Rounds 1-5: 64. Round 6, measured on the MI355X
    // with resident inputs (tools/c4_fused_ab.py, profiles/r06h_c4_fused_ab.txt; us per step, fused / pipeline): 128 scenarios 81 / 110,
    // 256: 81 / 109, 512: 92 / 111, 1024: 152 / 117 -- while every workgroup of the fused kernel finds a compute unit of its own (or shares
    // one with a single neighbour) a step costs ONE tick's latency, the pipeline pays four dependent launches. BASELINE config C4 shards
    // 1024 scenarios over 8 GPUs: 128 per GPU. Default: more than 2 workgroups per compute unit (fewer where the fused kernel's LDS
    // footprint lets fewer be resident) -> pipeline; LTPL_PIPELINE_MIN_SCEN=<n> overrides (the GPU test-suite pins 64: its batches of
    // 64 .. 256 scenarios are there to exercise the pipeline).
    int pipeline_min_scen = PIPELINE_MIN_SCEN;
    // pipeline launches with at least this many scenarios finish their follow jobs inside the lane kernel (k_vel_lanes, emit bit 1); smaller ones
    // keep the two halves of a follow job on two waves + k_vel_final: a shorter chain, which is what a small batch's step is made of.
    // LTPL_FOLLOW_EMIT_MIN_SCEN overrides (the GPU test-suite sets 256 so that its batches exercise the form the headline runs on).
    int follow_emit_min_scen = 8192;
    int scen_order = 1;              // batches of >= LTPL_SCEN_ORDER_MIN_SCEN scenarios are planned in the order of their start layers (DevPathsIn::order);
                                     // LTPL_NO_SCEN_ORDER=1: in the caller's order (identical results: tests, same-box A/B)
    void* d_par = nullptr; size_t d_par_cap = 0;         // parent-table slabs of the long-horizon mode
    ltpl_caps caps{};
    std::vector<void*> dev_allocs;
    // staging
    void* h_in = nullptr; size_t h_in_cap = 0;
    void* d_in = nullptr; size_t d_in_cap = 0;
    void* h_out = nullptr; size_t h_out_cap = 0;
    void* d_out = nullptr; size_t d_out_cap = 0;
    struct TickLayout* resident = nullptr;   // device-resident batch of ltpl_batch_upload
    long long* d_dbg = nullptr;              // LTPL_DEBUG_TIMING=1: cycle stamps
    void* d_planes = nullptr; size_t d_planes_cap = 0;   // tiled profile planes of the lane kernel
    // second buffer set + stream of the device-resident batch: the velocity kernels of step k overlap the path kernel of
    // step k + 1 (ltpl_batch_run)
    // Round 3: THREE buffer sets and TWO velocity streams. With two sets the path kernel of step r + 2 waited for the velocity kernels of
    // step r, and those take a whole step when they share the chip with a path kernel (its four waves per SIMD hold the register file: a
    // velocity wave only gets on a SIMD when a path wave retires) -- the velocity chain was the critical path (0.99 ms per step against
    // 0.89 ms of path kernel). Chains of consecutive steps now run next to each other on alternating streams and the path kernel only
    // waits for the chain three steps back: 0.947 ms per step (33.4 -> 34.4 M ticks/s).
#ifndef LTPL_PIPE_SETS
#define LTPL_PIPE_SETS 3               // (A/B on one box: 3, 4, 6 sets with two velocity streams all 34.3-34.4 M ticks/s; three or four streams: 33.4 M)
#endif
#ifndef LTPL_VEL_STREAMS
#define LTPL_VEL_STREAMS 2
#endif
    static constexpr int PIPE_SETS = LTPL_PIPE_SETS, VEL_STREAMS = LTPL_VEL_STREAMS;
    struct TickLayout* resident_x[PIPE_SETS - 1] = {};    // sets 1 .. (set 0 = `resident` on d_out / d_planes)
    void* d_out_x[PIPE_SETS - 1] = {}; size_t d_out_x_cap[PIPE_SETS - 1] = {};
    void* d_planes_x[PIPE_SETS - 1] = {}; size_t d_planes_x_cap[PIPE_SETS - 1] = {};
    hipStream_t vel_stream[VEL_STREAMS] = {};
    hipEvent_t ev_paths[PIPE_SETS] = {}, ev_vel[PIPE_SETS] = {};
    // second stream for the path kernels of odd steps (LTPL_PATH_STREAMS=2): the drain of one path kernel -- its longest scenarios on a
    // mostly idle chip -- overlaps the ramp of the next one
    hipStream_t path_stream2 = nullptr; hipEvent_t ev_begin = nullptr; int path_streams = 1;
    int last_set = 0;
    std::vector<hipEvent_t> ev_step;          // timing events around the path kernel of every step of the last timed run
    float last_paths_ms = 0.0f; int last_paths_n = 0;
    // host copy of the per-layer / per-node tables for the planner state machine (planner_core.hpp); empty when the
    // descriptor came without raceline / node_psi columns
    ltplp::HostLat hostlat; bool has_hostlat = false;
    int scratch_poison_on = 0; unsigned scratch_poison_word = 0;   // LTPL_SCRATCH_POISON (testing)
    int final_y = 8;                                                // row-chunk blocks per tile of k_vel_final (LTPL_FINAL_Y; A/B on one box: 2: 1.09, 4: 1.07, 8 / 22: 1.06 ms per step)
    int exp_skip = 0;                                               // LTPL_EXP_SKIP (timing experiments only): 1 prep, 2 lanes, 4 final kernel not launched
    int force_fused = 0, no_overlap = 0;                            // LTPL_FORCE_FUSED, LTPL_NO_OVERLAP (measurement switches)
    int poll_sync_every = 4096, poll_query = 0;
    int poll = 0;                    // LTPL_POLL=1: small zero-copy calls complete through a polled word in page-locked memory instead of a stream
                                     // synchronisation. Measured on the drop-in tick: p50 -4 us, but p99 +8 us with occasional 250 us outliers
                                     // (the runtime retires the launch concurrently with the next call) -- off by default, p99 is the metric
    unsigned* h_flag = nullptr; unsigned* d_done_cnt = nullptr; unsigned done_seq = 0; unsigned polled_calls = 0;
    int zc_in = 0;                   // small calls: kernels read their inputs straight from the page-locked staging buffer (no H2D copy)
    int zc_out = 0;                  // small calls: kernels write their outputs straight into the page-locked host buffer (no D2H copy)
    std::vector<int> rng_end_host;   // planning range end per start layer, -1 = no planning range (end of an open track)
    std::vector<int> sw2csc_host;    // CSC edge id of every sweep position (DevLat::sw2csc)
    int n_planners = 0;              // live ltpl_planner objects that compute through this handle (ltpl_destroy refuses while > 0)
    // LTPL_TICK_GRAPH=1 (round 5, measurement switch, off by default): the single fused tick -- H2D copy of the packed inputs -> k_tick
    // [-> D2H copy of the outputs] -- submitted as ONE hipGraph launch instead of two or three stream calls. The executable graph is
    // built at the first tick and re-parameterised per tick (the pointers into the staging buffers move with the tick's counts).
    int tick_graph = 0;
    // PERSISTENT SINGLE TICK (round 6, k_tick_persistent): opt-in by ltpl_create_ex(LTPL_CREATE_PERSISTENT_TICK) or LTPL_PERSISTENT_TICK=1
    struct PersistTick {
        int enabled = 0;                    // requested AND the lattice's four-wave kernel has a compile-time LDS plan
        int running = 0, variant = -1; size_t lds = 0;
        TickMailbox* mb = nullptr;          // page-locked
        TickKArgs* d_args = nullptr;        // device copy of the argument block (the resident kernel's "kernarg segment")
        hipStream_t stream = nullptr;
        const void* h_in = nullptr; void* d_in = nullptr;      // staging buffers the resident kernel was started on
        unsigned seq = 0;
        double idle_ms = 250.0;             // LTPL_PERSIST_IDLE_MS: the kernel leaves by itself after this long without a tick
        unsigned long long ticks = 0, launches = 0;            // served ticks, kernel starts (1 + restarts after idling out / other entry points)
        unsigned long long busy_clk = 0, busy_n = 0;           // device-side time of completed residencies (wall_clock64 ticks), their ticks
    } pt;
    hipGraph_t tg_graph = nullptr; hipGraphExec_t tg_exec = nullptr;
    hipGraphNode_t tg_n_in = nullptr, tg_n_k = nullptr, tg_n_out = nullptr;
    const void* tg_func = nullptr; int tg_has_out = 0;
};

#define HIP_TRY(h, call)                                                                                              \
    do {                                                                                                              \
        hipError_t e_ = (call);                                                                                       \
        if (e_ != hipSuccess) {                                                                                       \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                             \
            return LTPL_ERR_HIP;                                                                                      \
        }                                                                                                             \
    } while (0)

static const void* paths_kernel_of(const ltpl_handle* h, int nw)
{
    if (nw == 1) {
        if (h->long_horizon) return reinterpret_cast<const void*>(k_paths<1, PlanRtG>);
        switch (h->plan_class) {
            case 1: return reinterpret_cast<const void*>(k_paths<1, PlanA>);
            case 2: return reinterpret_cast<const void*>(k_paths<1, PlanB>);
            case 3: return reinterpret_cast<const void*>(k_paths<1, PlanC>);
            default: return reinterpret_cast<const void*>(k_paths<1, PlanRt>);
        }
    }
    if (h->long_horizon) return reinterpret_cast<const void*>(k_paths<NUM_WAVES, PlanRtG>);
    return h->plan_class4 == 1 ? reinterpret_cast<const void*>(k_paths<NUM_WAVES, PlanA4>)
                               : reinterpret_cast<const void*>(k_paths<NUM_WAVES, PlanRt>);
}

// (mangled) symbol prefix of that kernel in the code object: what a profile is matched against (ltpl_paths_kernel_symbol)
static const char* paths_kernel_symbol_of(const ltpl_handle* h, int nw)
{
    if (nw == 1) {
        if (h->long_horizon) return "_Z7k_pathsILi1E7PlanRtGE";
        switch (h->plan_class) {
            case 1: return "_Z7k_pathsILi1E6PlanFxILi32ELi32ELi1EEE";
            case 2: return "_Z7k_pathsILi1E6PlanFxILi32ELi40ELi1EEE";
            case 3: return "_Z7k_pathsILi1E6PlanFxILi48ELi32ELi1EEE";
            default: return "_Z7k_pathsILi1E6PlanRtE";
        }
    }
    if (h->long_horizon) return "_Z7k_pathsILi4E7PlanRtGE";
    return h->plan_class4 == 1 ? "_Z7k_pathsILi4E6PlanFxILi32ELi32ELi4EEE" : "_Z7k_pathsILi4E6PlanRtE";
}
extern "C" const char* ltpl_paths_kernel_symbol(const ltpl_handle* h, int32_t team_waves)
{
    return h ? paths_kernel_symbol_of(h, team_waves == 4 ? NUM_WAVES : 1) : "";
}

// the path kernel with `nw` waves per scenario in the LDS plan class chosen for the lattice at ltpl_create
static int launch_paths(ltpl_handle* h, int nw, int n_scen, hipStream_t st, const DevPathsIn& di, const DevPathsOut& dout,
                        unsigned* mask_out = nullptr)
{
    TeamLds lp = nw == 1 ? h->lp1 : h->lp4;
    lp.mask_out = mask_out;
    if (h->long_horizon) {
        const size_t need = (size_t)lp.par_glob_stride * (size_t)n_scen;
        if (need > h->d_par_cap) {
            HIP_TRY(h, hipDeviceSynchronize());          // an earlier launch may still use the old slabs
            if (h->d_par) (void)hipFree(h->d_par);
            h->d_par = nullptr; h->d_par_cap = 0;
            HIP_TRY(h, hipMalloc(&h->d_par, need));
            h->d_par_cap = need;
        }
        lp.par_glob = static_cast<unsigned char*>(h->d_par);
    }
    const dim3 grid(n_scen), block(nw == 1 ? 64 : WG_THREADS);
    if (nw == 1) {
        if (h->long_horizon) hipLaunchKernelGGL((k_paths<1, PlanRtG>), grid, block, lp.total, st, h->lat, di, dout, lp);
        else switch (h->plan_class) {
            case 1: hipLaunchKernelGGL((k_paths<1, PlanA>), grid, block, lp.total, st, h->lat, di, dout, lp); break;
            case 2: hipLaunchKernelGGL((k_paths<1, PlanB>), grid, block, lp.total, st, h->lat, di, dout, lp); break;
            case 3: hipLaunchKernelGGL((k_paths<1, PlanC>), grid, block, lp.total, st, h->lat, di, dout, lp); break;
            default: hipLaunchKernelGGL((k_paths<1, PlanRt>), grid, block, lp.total, st, h->lat, di, dout, lp); break;
        }
    } else if (h->long_horizon) hipLaunchKernelGGL((k_paths<NUM_WAVES, PlanRtG>), grid, block, lp.total, st, h->lat, di, dout, lp);
    else if (h->plan_class4 == 1) hipLaunchKernelGGL((k_paths<NUM_WAVES, PlanA4>), grid, block, lp.total, st, h->lat, di, dout, lp);
    else hipLaunchKernelGGL((k_paths<NUM_WAVES, PlanRt>), grid, block, lp.total, st, h->lat, di, dout, lp);
    HIP_TRY(h, hipGetLastError());
    return LTPL_OK;
}

static void dbg_report(ltpl_handle* h, const char* what, int n_blocks)
{
    if (!h->d_dbg) return;
    std::vector<long long> v((size_t)256 * DBG_SLOTS);
    if (hipMemcpy(v.data(), h->d_dbg, v.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    int nb = n_blocks < 256 ? n_blocks : 256;
    fprintf(stderr, "[ltpl dbg] %s: mean cycles between stamps over %d blocks (wave: d01 d12 ...)\n", what, nb);
    for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "[ltpl dbg]   wave %d:", w);
        for (int k = 0; k + 1 < 16; ++k) {
            double acc = 0; int cnt = 0;
            for (int b = 0; b < nb; ++b) {
                long long a = v[(size_t)b * DBG_SLOTS + w * 16 + k], c = v[(size_t)b * DBG_SLOTS + w * 16 + k + 1];
                if (a > 0 && c > a) { acc += (double)(c - a); ++cnt; }
            }
            fprintf(stderr, " %9.0f", cnt ? acc / cnt : 0.0);
        }
        fprintf(stderr, "\n");
    }
    (void)hipMemset(h->d_dbg, 0, v.size() * sizeof(long long));
}



static void dbg_report_lanes(ltpl_handle* h)
{
    if (!h->d_dbg) return;
    std::vector<long long> v((size_t)256 * DBG_SLOTS);
    if (hipMemcpy(v.data(), h->d_dbg, v.size() * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess) return;
    const char* names[3] = {"generic", "follow (controlled)", "follow (unconstrained)"};
    {
        double acc = 0; int cnt = 0;
        for (int r = 0; r < 64; ++r) { const long long* row = &v[(size_t)(128 + r) * DBG_SLOTS]; if (row[0] > 0 && row[8] > row[0]) { acc += (double)(row[8] - row[0]); ++cnt; } }
        fprintf(stderr, "[ltpl dbg] k_vel_lanes unconstrained profile: forward sweep %9.0f cycles (%d rows)\n", cnt ? acc / cnt : 0.0, cnt);
    }
    {
        fprintf(stderr, "[ltpl dbg] k_follow_prep mean cycles between stamps 0-1-2-3-4-5 (inputs | race line scan | path scan | foot points | sums):");
        for (int k = 0; k < 5; ++k) {
            double acc = 0; int cnt = 0;
            for (int r = 0; r < 64; ++r) { const long long* row = &v[(size_t)(192 + r) * DBG_SLOTS]; if (row[k] > 0 && row[k + 1] > row[k]) { acc += (double)(row[k + 1] - row[k]); ++cnt; } }
            fprintf(stderr, " %8.0f", cnt ? acc / cnt : 0.0);
        }
        fprintf(stderr, "\n");
    }
    for (int g = 0; g < 3; ++g) {
        fprintf(stderr, "[ltpl dbg] k_vel_lanes %-24s mean cycles between stamps 0-1-2-3-4-5-6, total:", names[g]);
        double tot = 0; int tc = 0;
        for (int k = 0; k < 6; ++k) {
            double acc = 0; int cnt = 0;
            for (int r = 0; r < 64; ++r) {
                const long long* row = &v[(size_t)(g * 64 + r) * DBG_SLOTS];
                int k2 = k + 1; while (k2 < 7 && row[k2] == 0) ++k2;
                if (row[k] > 0 && k2 < 7 && row[k2] > row[k] && (k2 == k + 1)) { acc += (double)(row[k2] - row[k]); ++cnt; }
            }
            fprintf(stderr, " %9.0f", cnt ? acc / cnt : 0.0);
        }
        for (int r = 0; r < 64; ++r) { const long long* row = &v[(size_t)(g * 64 + r) * DBG_SLOTS]; if (row[0] > 0 && row[6] > row[0]) { tot += (double)(row[6] - row[0]); ++tc; } }
        fprintf(stderr, " | %9.0f (%d rows)\n", tc ? tot / tc : 0.0, tc);
    }
    (void)hipMemset(h->d_dbg, 0, v.size() * sizeof(long long));
}

template <typename T>
static int upload(ltpl_handle* h, const T* src, size_t n, const T** dst)
{
    void* p = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    HIP_TRY(h, hipMalloc(&p, bytes));
    h->dev_allocs.push_back(p);
    if (n) HIP_TRY(h, hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return LTPL_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// planning-range statistics over all start layers (sizes the LDS plan and the output capacities)
static int horizon_stats(const ltpl_lattice_desc* d, int* hmax, int* ehmax, int* nhmax, int* ptsmax, int* kmax,
                         int* degmax, int* etmax, std::vector<int>* rng_end, std::string* why)
{
    const int L = d->num_layers;
    std::vector<int> edges_into(L, 0), maxsamp_into(L, 0);
    *kmax = 0; *degmax = 0;
    for (int l = 0; l < L; ++l) {
        int K = d->layer_node_off[l + 1] - d->layer_node_off[l];
        if (K > *kmax) *kmax = K;
        for (int v = d->layer_node_off[l]; v < d->layer_node_off[l + 1]; ++v) {
            int deg = d->in_ptr[v + 1] - d->in_ptr[v];
            if (deg > *degmax) *degmax = deg;
            edges_into[l] += deg;
            for (int e = d->in_ptr[v]; e < d->in_ptr[v + 1]; ++e) {
                if (!(d->edge_cost[e] >= 0.0)) { *why = "negative or NaN edge cost"; return LTPL_ERR_INVALID_ARG; }
                int ns = d->samp_ptr[e + 1] - d->samp_ptr[e];
                if (ns < 2) { *why = "edge with fewer than 2 samples"; return LTPL_ERR_INVALID_ARG; }
                if (ns > maxsamp_into[l]) maxsamp_into[l] = ns;
                int pl = (l + L - 1) % L;
                if (d->edge_src[e] < 0 || d->edge_src[e] >= d->layer_node_off[pl + 1] - d->layer_node_off[pl]) {
                    *why = "edge_src out of range"; return LTPL_ERR_INVALID_ARG;
                }
            }
        }
    }
    *hmax = *ehmax = *nhmax = *ptsmax = 0; *etmax = 0;
    for (int l = 0; l < L; ++l) if (edges_into[l] > *etmax) *etmax = edges_into[l];
    rng_end->assign((size_t)L, 0);
    for (int sl = 0; sl < L; ++sl) {
        int el;
        if (d->plan_horizon_mode == 0) {                             // gen_local_node_template.py:104-124
            double des = d->s_raceline[sl] + d->min_plan_horizon;
            if (des > d->s_raceline[L - 1]) { if (d->closed) des -= d->s_raceline[L - 1]; else des = d->s_raceline[L - 1]; }
            int lo = 0, hi = L;
            while (lo < hi) { int mid = (lo + hi) / 2; if (d->s_raceline[mid] < des) lo = mid + 1; else hi = mid; }
            el = lo;
        } else if (d->closed) el = (sl + (int)d->min_plan_horizon) % L;
        else el = std::max(sl + (int)d->min_plan_horizon, L - 1);    // :131-133 (sic: max)
        if (el >= L) {
            if (!d->closed) { (*rng_end)[(size_t)sl] = -1; continue; }   // the reference fails for this start layer (GraphBase.py:889)
            *why = "planning horizon runs past the last layer (track shorter than the horizon?)"; return LTPL_ERR_UNSUPPORTED;
        }
        (*rng_end)[(size_t)sl] = el;
        int H = el - sl; if (H < 0) H = L - sl + el;
        if (!d->closed) {
            if (H <= 0) { (*rng_end)[(size_t)sl] = -1; continue; }       // last layer of an open track: empty planning range
        } else if (H <= 0 || H >= L - 1) { *why = "planning range covers the whole track; not supported"; return LTPL_ERR_UNSUPPORTED; }
        int eh = 0, nh = d->layer_node_off[sl + 1] - d->layer_node_off[sl], pts = 1;
        for (int j = 1; j <= H; ++j) {
            int b = (sl + j) % L;
            eh += edges_into[b]; nh += d->layer_node_off[b + 1] - d->layer_node_off[b];
            pts += maxsamp_into[b] - 1;
        }
        if (H + 1 > *hmax) *hmax = H + 1;
        if (eh > *ehmax) *ehmax = eh;
        if (nh > *nhmax) *nhmax = nh;
        if (pts > *ptsmax) *ptsmax = pts;
    }
    return LTPL_OK;
}

// "No C++ exception crosses the ABI" (include/ltpl_hip.h): every entry point that allocates is a function-try-block; an exception
// (std::bad_alloc / std::length_error of a host container in practice) becomes LTPL_ERR_EXCEPTION with the message in the handle's
// (or the thread's create-time) error string.
static int abi_caught(std::string* err, const char* what) noexcept
{
    try { (err ? *err : g_create_error) = std::string("C++ exception caught at the ABI: ") + what; } catch (...) {}
    return LTPL_ERR_EXCEPTION;
}
static std::string* abi_err_of(const ltpl_handle* h) { return h ? const_cast<std::string*>(&h->err) : nullptr; }
static std::string* abi_err_of(const ltpl_planner* p) { return p ? const_cast<std::string*>(&p->P.err) : nullptr; }
#define LTPL_ABI_CATCH(errptr) \
    catch (const std::exception& e) { return abi_caught(errptr, e.what()); } \
    catch (...) { return abi_caught(errptr, "unknown exception"); }

extern "C" int ltpl_version(void) { return LTPL_ABI_VERSION; }

static void persist_stop(ltpl_handle* h);      // (the resident single-tick kernel leaves: defined next to ltpl_tick_batch)

extern "C" const char* ltpl_last_error(const ltpl_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int ltpl_destroy(ltpl_handle* h)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (h->n_planners > 0) { h->err = "ltpl_destroy: planners created on this handle are still alive (destroy them first)"; return LTPL_ERR_INVALID_ARG; }
    (void)hipSetDevice(h->device);
    persist_stop(h);
    if (h->pt.stream) (void)hipStreamDestroy(h->pt.stream);
    if (h->pt.mb) (void)hipHostFree(h->pt.mb);
    if (h->pt.d_args) (void)hipFree(h->pt.d_args);
    for (void* p : h->dev_allocs) (void)hipFree(p);
    if (h->d_in) (void)hipFree(h->d_in);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->d_dbg) (void)hipFree(h->d_dbg);
    if (h->d_planes) (void)hipFree(h->d_planes);
    if (h->d_par) (void)hipFree(h->d_par);
    for (int i = 0; i < ltpl_handle::PIPE_SETS - 1; ++i) {
        if (h->d_planes_x[i]) (void)hipFree(h->d_planes_x[i]);
        if (h->d_out_x[i]) (void)hipFree(h->d_out_x[i]);
        free_resident(h->resident_x[i]);
    }
    for (int i = 0; i < ltpl_handle::VEL_STREAMS; ++i) if (h->vel_stream[i]) (void)hipStreamDestroy(h->vel_stream[i]);
    if (h->path_stream2) (void)hipStreamDestroy(h->path_stream2);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    for (int i = 0; i < ltpl_handle::PIPE_SETS; ++i) { if (h->ev_paths[i]) (void)hipEventDestroy(h->ev_paths[i]); if (h->ev_vel[i]) (void)hipEventDestroy(h->ev_vel[i]); }
    for (hipEvent_t e : h->ev_step) (void)hipEventDestroy(e);
    if (h->h_in) (void)hipHostFree(h->h_in);
    if (h->h_out) (void)hipHostFree(h->h_out);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->d_done_cnt) (void)hipFree(h->d_done_cnt);
    if (h->tg_exec) (void)hipGraphExecDestroy(h->tg_exec);
    if (h->tg_graph) (void)hipGraphDestroy(h->tg_graph);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    free_resident(h->resident);
    delete h;
    return LTPL_OK;
}

extern "C" int ltpl_create(const ltpl_lattice_desc* d, int device, ltpl_handle** out_handle)
{
    return ltpl_create_ex(d, device, 0u, out_handle);
}

extern "C" int ltpl_create_ex(const ltpl_lattice_desc* d, int device, uint32_t flags, ltpl_handle** out_handle)
try {
    g_create_error.clear();
    if (flags & ~(uint32_t)LTPL_CREATE_PERSISTENT_TICK) { g_create_error = "ltpl_create_ex: unknown flag"; return LTPL_ERR_INVALID_ARG; }
    if (!d || !out_handle) { g_create_error = "null argument"; return LTPL_ERR_INVALID_ARG; }
    if (d->num_layers < 4 || d->num_nodes < 1 || d->num_edges < 1) { g_create_error = "empty lattice"; return LTPL_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "no HIP device visible"; return LTPL_ERR_NO_DEVICE; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) { g_create_error = "device index out of range"; return LTPL_ERR_NO_DEVICE; }

    int hmax, ehmax, nhmax, ptsmax, kmax, degmax, etmax;
    std::vector<int> rng_end;
    int rc = horizon_stats(d, &hmax, &ehmax, &nhmax, &ptsmax, &kmax, &degmax, &etmax, &rng_end, &g_create_error);
    if (rc) return rc;
    if (hmax < 2) { g_create_error = "no start layer with a planning range"; return LTPL_ERR_INVALID_ARG; }
    if (etmax > 65535) { g_create_error = "more than 65535 edges in one layer transition"; return LTPL_ERR_CAPACITY; }
    if (kmax > 255 || degmax > 127) { g_create_error = "more than 255 nodes per layer or 127 in-edges per node"; return LTPL_ERR_CAPACITY; }

    ltpl_handle* h = new ltpl_handle();
    h->device = device;
    h->rng_end_host = rng_end;
    struct Guard { ltpl_handle* h; ~Guard() { if (h) ltpl_destroy(h); } } guard{h};     // error returns and exceptions release the handle
    auto fail = [&](int code) { g_create_error = h->err; return code; };
    if (hipSetDevice(device) != hipSuccess) { h->err = "hipSetDevice failed"; return fail(LTPL_ERR_HIP); }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { h->err = "hipStreamCreate failed"; return fail(LTPL_ERR_HIP); }

    DevLat& L = h->lat;
    L.L = d->num_layers; L.V = d->num_nodes; L.E = d->num_edges; L.S = d->num_samples; L.G = d->num_glob_rl;
    L.mode = d->plan_horizon_mode; L.closed = d->closed ? 1 : 0; L.min_plan_horizon = d->min_plan_horizon; L.veh_width = d->veh_width;
    L.sampled_resolution = d->sampled_resolution; L.lat_offset = d->lat_offset;
    L.vel_decrease_lat = d->vel_decrease_lat; L.veh_length = d->veh_length;
#define UP(field, src, n) if ((rc = upload(h, src, (size_t)(n), &L.field)) != LTPL_OK) return fail(rc)
    UP(layer_off, d->layer_node_off, L.L + 1); UP(rl_idx, d->raceline_index, L.L);
    UP(s_rl, d->s_raceline, L.L); UP(ref_x, d->refline_x, L.L); UP(ref_y, d->refline_y, L.L);
    UP(vel_rl, d->vel_raceline, L.L);
    UP(node_x, d->node_x, L.V); UP(node_y, d->node_y, L.V); UP(vgoal, d->vgoal_cost, L.V);
    UP(in_ptr, d->in_ptr, L.V + 1); UP(edge_src, d->edge_src, L.E);
    UP(edge_cost, d->edge_cost, L.E);
    UP(edge_len, d->edge_len, L.E); UP(samp_ptr, d->samp_ptr, L.E + 1);
    UP(sx, d->samp_x, L.S); UP(sy, d->samp_y, L.S); UP(spsi, d->samp_psi, L.S); UP(slen, d->samp_len, L.S);
    UP(glob_rl, d->glob_rl, (size_t)L.G * 5);
    {
        std::vector<double> sc(2 * (size_t)L.S + 2, 0.0);
        for (int k = 0; k < L.S; ++k) { sc[2 * (size_t)k] = sin(d->samp_psi[k]); sc[2 * (size_t)k + 1] = cos(d->samp_psi[k]); }
        UP(ssc, sc.data(), 2 * (size_t)L.S + 2);
    }
    {
        std::vector<double> gx((size_t)L.G), gy((size_t)L.G);
        for (int k = 0; k < L.G; ++k) { gx[(size_t)k] = d->glob_rl[(size_t)k * 5 + 1]; gy[(size_t)k] = d->glob_rl[(size_t)k * 5 + 2]; }
        UP(grx, gx.data(), L.G); UP(gry, gy.data(), L.G);
    }
    {
        // derived tables: planning range per start layer, first edge into every layer, byte-wide edge sources
        std::vector<int> ebase((size_t)L.L + 1);
        for (int l = 0; l <= L.L; ++l) ebase[(size_t)l] = d->in_ptr[d->layer_node_off[l]];
        std::vector<unsigned char> src8((size_t)L.E + 16, (unsigned char)0xff);   // padded: path assembly reads 16 bytes at a time
        for (int e = 0; e < L.E; ++e) src8[(size_t)e] = (unsigned char)d->edge_src[e];
        std::vector<unsigned char> dst8((size_t)L.E);
        for (int l = 0; l < L.L; ++l)
            for (int v = d->layer_node_off[l]; v < d->layer_node_off[l + 1]; ++v)
                for (int e = d->in_ptr[v]; e < d->in_ptr[v + 1]; ++e) dst8[(size_t)e] = (unsigned char)(v - d->layer_node_off[l]);
        UP(rng_end, rng_end.data(), L.L); UP(layer_ebase, ebase.data(), L.L + 1); UP(edge_src8, src8.data(), L.E + 16);
        UP(edge_dst8, dst8.data(), L.E);
        std::vector<unsigned char> rank8((size_t)L.E);
        for (int v = 0; v < L.V; ++v)
            for (int e = d->in_ptr[v]; e < d->in_ptr[v + 1]; ++e) rank8[(size_t)e] = (unsigned char)(e - d->in_ptr[v]);
        UP(edge_rank8, rank8.data(), L.E);
        {
            // path assembly records (see DevLat; csrc/assembly_records.hpp)
            std::vector<int32_t> nrec; std::vector<double> erec;
            ltplrec::build(L.V, L.E, d->in_ptr, d->edge_src, d->edge_len, d->samp_ptr, d->samp_x, d->samp_y, d->samp_psi, nrec, erec);
            const int4* nr = nullptr;
            if ((rc = upload(h, reinterpret_cast<const int4*>(nrec.data()), (size_t)L.V, &nr)) != LTPL_OK) return fail(rc);
            L.node_rec = nr;
            UP(edge_rec, erec.data(), (size_t)L.E * LTPL_EDGE_REC);
        }
        std::vector<int> ldeg((size_t)L.L, 0);
        for (int l = 0; l < L.L; ++l)
            for (int v = d->layer_node_off[l]; v < d->layer_node_off[l + 1]; ++v)
                ldeg[(size_t)l] = std::max(ldeg[(size_t)l], d->in_ptr[v + 1] - d->in_ptr[v]);
        UP(layer_degmax, ldeg.data(), L.L);
        // sweep order (see DevLat): per transition sorted by (rank, destination); cost / meta with the sentinel at index E
        h->sw2csc_host.resize((size_t)L.E);
        std::vector<int> csc2sw((size_t)L.E);
        for (int l = 0; l < L.L; ++l) {
            const int e0 = ebase[(size_t)l], e1 = ebase[(size_t)l + 1];
            for (int e = e0; e < e1; ++e) h->sw2csc_host[(size_t)e] = e;
            std::stable_sort(h->sw2csc_host.begin() + e0, h->sw2csc_host.begin() + e1, [&](int a, int b) {
                if (rank8[(size_t)a] != rank8[(size_t)b]) return rank8[(size_t)a] < rank8[(size_t)b];
                return dst8[(size_t)a] < dst8[(size_t)b];
            });
        }
        // (LTPL_SW_PAD sentinel entries behind the last edge: the sweep's prefetch reads whole 64-edge chunks from a transition's first edge
        //  on and replaces what lies beyond the transition by the sentinel in registers -- behind the LAST transition it reads these)
        std::vector<double> swc((size_t)L.E + 1 + LTPL_SW_PAD, (double)INFINITY);
        std::vector<unsigned> swm((size_t)L.E + 1 + LTPL_SW_PAD, 0u);
        for (int p = 0; p < L.E; ++p) {
            const int e = h->sw2csc_host[(size_t)p];
            csc2sw[(size_t)e] = p; swc[(size_t)p] = d->edge_cost[e];
            swm[(size_t)p] = (unsigned)rank8[(size_t)e] | ((unsigned)src8[(size_t)e] << 8) | ((unsigned)dst8[(size_t)e] << 16);
        }
        swc[(size_t)L.E] = INFINITY;
        UP(sw_cost, swc.data(), L.E + 1 + LTPL_SW_PAD); UP(sw_meta, swm.data(), L.E + 1 + LTPL_SW_PAD);
        UP(sw2csc, h->sw2csc_host.data(), L.E); UP(csc2sw, csc2sw.data(), L.E);
        if (d->normvec_x && d->normvec_y && d->width_right && d->width_left) {
            // ObjectListInterface.py:71-72 and check_inside_bounds.py:27 (same operations, no contraction)
            std::vector<double> b1x((size_t)L.L), b1y((size_t)L.L), b2x((size_t)L.L), b2y((size_t)L.L), cx((size_t)L.L), cy((size_t)L.L);
            for (int l = 0; l < L.L; ++l) {
                b1x[(size_t)l] = d->refline_x[l] + d->normvec_x[l] * d->width_right[l]; b1y[(size_t)l] = d->refline_y[l] + d->normvec_y[l] * d->width_right[l];
                b2x[(size_t)l] = d->refline_x[l] - d->normvec_x[l] * d->width_left[l]; b2y[(size_t)l] = d->refline_y[l] - d->normvec_y[l] * d->width_left[l];
                cx[(size_t)l] = (b1x[(size_t)l] + b2x[(size_t)l]) / 2; cy[(size_t)l] = (b1y[(size_t)l] + b2y[(size_t)l]) / 2;
            }
            UP(b1x, b1x.data(), L.L); UP(b1y, b1y.data(), L.L); UP(b2x, b2x.data(), L.L); UP(b2y, b2y.data(), L.L);
            UP(ctx, cx.data(), L.L); UP(cty, cy.data(), L.L);
        }
        // capsule per edge (two-sided cull of the obstacle mask, paths_team.hpp phase 2; construction and its conservativeness
        // argument in capsule.hpp)
        std::vector<float4> cap((size_t)L.E * 2), cap_sw((size_t)L.E * 2);
        L.cull_slack = ltplcap::build(L.E, d->samp_ptr, d->samp_x, d->samp_y, L.S, reinterpret_cast<float*>(cap.data()));
        for (int p = 0; p < L.E; ++p) {                         // the mask phase walks a transition in sweep order
            const size_t e = (size_t)h->sw2csc_host[(size_t)p];
            cap_sw[2 * (size_t)p] = cap[2 * e]; cap_sw[2 * (size_t)p + 1] = cap[2 * e + 1];
        }
        UP(edge_cap, cap_sw.data(), (size_t)L.E * 2);
        // closest-layer grid of phase 1 (layer_grid.hpp: construction and conservativeness argument; tests/test_layer_grid.py)
        L.lgrid = nullptr; L.lg_x0 = L.lg_y0 = L.lg_inv = 0.0; L.lg_nx = L.lg_ny = 0;
        if (!getenv("LTPL_NO_LAYER_GRID")) {
            const ltplgrid::Grid g = ltplgrid::build(L.L, d->refline_x, d->refline_y);
            if (g.nx > 0 && g.ny > 0) {
                UP(lgrid, reinterpret_cast<const int4*>(g.cells.data()), (size_t)g.nx * g.ny);
                L.lg_x0 = g.x0; L.lg_y0 = g.y0; L.lg_inv = g.inv_cell; L.lg_nx = g.nx; L.lg_ny = g.ny;
            }
        }
    }
#undef UP

    auto make_plan = [&](int nw, TeamLds* lp, bool par_in_global) {
        lp->kpad = (int)align_up((size_t)kmax, 4);
        lp->hmax = hmax + 1;
        lp->etmax = etmax;
        lp->words_blocked = (ehmax + 31) / 32 + 1;
        lp->words_zone = (nhmax + 31) / 32 + 1;
        lp->n_path_bufs = nw < LTPL_MAX_ACTIONS ? nw : LTPL_MAX_ACTIONS;
        size_t off = 0;
        lp->off_dist = (int)off; off += sizeof(double) * NFILT * 2 * lp->kpad;
        lp->off_cnt = (int)off; off += sizeof(unsigned) * NFILT * lp->kpad;
        lp->off_widx = (int)off; off += sizeof(unsigned) * NFILT * lp->kpad; off = align_up(off, 16);
        lp->off_dumin = (int)off; off += sizeof(double) * NFILT * lp->kpad;
        lp->path_stride = (int)align_up(sizeof(double) * 7 * lp->hmax + sizeof(int) * 2 * (lp->hmax + 1), 16);
        if (lp->n_path_bufs == 1) {
            // one-wave teams: the path scratch aliases the sweep's frontier / election arrays (dead during path assembly)
            lp->off_path = lp->off_dist;
            if ((size_t)lp->path_stride > off - (size_t)lp->off_dist) off = (size_t)lp->off_dist + (size_t)lp->path_stride;
        } else {
            lp->off_path = (int)off; off += (size_t)lp->path_stride * lp->n_path_bufs;
        }
        lp->off_best = (int)off; off += sizeof(int) * NFILT * lp->hmax; off = align_up(off, 16);
        lp->off_blocked = (int)off; off += sizeof(unsigned) * lp->words_blocked; off = align_up(off, 16);
        lp->off_zone = (int)off; off += sizeof(unsigned) * lp->words_zone; off = align_up(off, 16);
        const size_t par_bytes = sizeof(uchar2) * NPAR * (size_t)lp->hmax * lp->kpad;
        lp->off_par = (int)off; if (!par_in_global) { off += par_bytes; off = align_up(off, 16); }
        lp->ref_lds = (!par_in_global && sizeof(double) * 2 * (size_t)d->num_layers <= par_bytes) ? 1 : 0;
        lp->par_glob = nullptr; lp->par_glob_stride = par_in_global ? (long long)align_up(par_bytes, 256) : 0;
        lp->off_lay = (int)off; off += sizeof(int) * 4 * (size_t)lp->hmax;
        lp->off_pos_layer = (int)off; off += sizeof(short) * MAX_POS; off = align_up(off, 16);
        lp->off_pos_veh = (int)off; off += MAX_POS; off = align_up(off, 16);
        // shell list of the obstacle mask (phase 2): 128 entries of 8 bytes per wave (runtime plans: own storage)
        lp->shell_cap = 128; lp->off_shell = (int)off; off += (size_t)8 * 128 * nw; off = align_up(off, 16);
        lp->total = (int)off;
        lp->mask_out = nullptr;
        lp->ablate = 0; lp->poison_on = 0; lp->poison = 0u; lp->dbg = nullptr;
#ifdef LTPL_EXPERIMENT
        lp->ablate = getenv("LTPL_ABLATE") ? atoi(getenv("LTPL_ABLATE")) : 0;
        lp->poison_on = getenv("LTPL_LDS_POISON") ? 1 : 0;
        lp->poison = lp->poison_on ? (unsigned)strtoul(getenv("LTPL_LDS_POISON"), nullptr, 0) : 0u;
#endif
    };
    make_plan(1, &h->lp1, false); make_plan(NUM_WAVES, &h->lp4, false);
    const int lds_limit = 150 * 1024;
    if (h->lp4.total > lds_limit || h->lp1.total > lds_limit || getenv("LTPL_FORCE_LONG_HORIZON")) {
        // long planning horizons: the parent tables move to global memory, the path scratch stays in LDS
        h->long_horizon = 1;
        make_plan(1, &h->lp1, true); make_plan(NUM_WAVES, &h->lp4, true);
    }
    // compile-time plan classes of the one-wave batch kernel: same arrays, hot offsets taken from the policy (PlanFx)
    auto make_fixed_plan = [&](auto plan_tag, TeamLds* lp) {       // *lp holds the runtime plan of the same team size
        typedef decltype(plan_tag) PL;
        lp->kpad = PL::c_kpad; lp->hmax = PL::c_hmax; lp->n_path_bufs = PL::c_n_path_bufs;
        lp->off_dist = PL::c_off_dist; lp->off_cnt = PL::c_off_cnt; lp->off_widx = PL::c_off_widx; lp->off_dumin = PL::c_off_dumin;
        lp->path_stride = PL::c_path_stride; lp->off_path = PL::c_off_path; lp->off_best = PL::c_off_best;
        lp->off_par = PL::c_off_par; lp->off_lay = PL::c_off_lay;
        // the shell list of phase 2 lives in the frontier / election arrays (not written before phase 4)
        lp->off_shell = PL::c_off_dist;
        lp->shell_cap = (PL::c_end_elect - PL::c_off_dist) / 8 / (PL::c_n_path_bufs == 1 ? 1 : NUM_WAVES);
        size_t off = (size_t)PL::c_fixed_end;
        lp->off_blocked = (int)off; off += sizeof(unsigned) * lp->words_blocked; off = align_up(off, 16);
        lp->off_zone = (int)off; off += sizeof(unsigned) * lp->words_zone; off = align_up(off, 16);
        lp->ref_lds = (sizeof(double) * 2 * (size_t)d->num_layers <= (size_t)PL::c_par_bytes) ? 1 : 0;
        // the per-position tables are dead once phase 3 is over, the parent table is not written before phase 4: they
        // share its storage (behind the staged reference line) when they fit
        const size_t ref_bytes = lp->ref_lds ? align_up(sizeof(double) * 2 * (size_t)d->num_layers, 16) : 0;
        const size_t pos_bytes = align_up(sizeof(short) * MAX_POS, 16) + align_up((size_t)MAX_POS, 16);
        if (ref_bytes + pos_bytes <= (size_t)PL::c_par_bytes) {
            lp->off_pos_layer = lp->off_par + (int)ref_bytes;
            lp->off_pos_veh = lp->off_pos_layer + (int)align_up(sizeof(short) * MAX_POS, 16);
        } else {
            lp->off_pos_layer = (int)off; off += sizeof(short) * MAX_POS; off = align_up(off, 16);
            lp->off_pos_veh = (int)off; off += MAX_POS; off = align_up(off, 16);
        }
        lp->total = (int)off;
    };
    if (!getenv("LTPL_NO_FIXED_PLAN") && !h->long_horizon) {
        if (kmax <= PlanA::c_kpad && hmax + 1 <= PlanA::c_hmax && d->num_layers >= PlanA::c_hmax) {
            h->plan_class = 1; make_fixed_plan(PlanA(), &h->lp1);
            h->plan_class4 = 1; make_fixed_plan(PlanA4(), &h->lp4);
        }
        else if (kmax <= PlanB::c_kpad && hmax + 1 <= PlanB::c_hmax && d->num_layers >= PlanB::c_hmax) { h->plan_class = 2; make_fixed_plan(PlanB(), &h->lp1); }
        else if (kmax <= PlanC::c_kpad && hmax + 1 <= PlanC::c_hmax && d->num_layers >= PlanC::c_hmax) { h->plan_class = 3; make_fixed_plan(PlanC(), &h->lp1); }
    }
    if (const char* e = getenv("LTPL_BATCH_NW")) h->batch_nw = atoi(e) == 4 ? 4 : 1;
    h->zc_out = 1;
    if (const char* e = getenv("LTPL_ZC_OUT")) h->zc_out = atoi(e);
    if (const char* e = getenv("LTPL_ZC_IN")) h->zc_in = atoi(e);
    if (const char* e = getenv("LTPL_POLL")) h->poll = atoi(e);
    if (const char* e = getenv("LTPL_TICK_GRAPH")) h->tick_graph = atoi(e);
    if (const char* e = getenv("LTPL_POLL_SYNC_EVERY")) h->poll_sync_every = atoi(e);
    if (const char* e = getenv("LTPL_POLL_QUERY")) h->poll_query = atoi(e);
    // the persistent single tick: requested by flag or environment, engaged where the four-wave kernel has a compile-time LDS plan
    {
        const char* e = getenv("LTPL_PERSISTENT_TICK");
        const bool want = (flags & LTPL_CREATE_PERSISTENT_TICK) != 0u || (e && atoi(e) != 0);
        h->pt.enabled = (want && h->plan_class4 == 1 && !h->long_horizon && h->zc_out) ? 1 : 0;
        if (const char* m = getenv("LTPL_PERSIST_IDLE_MS")) { const double v = atof(m); if (v > 0.0) h->pt.idle_ms = v; }
    }
    if (h->poll || h->pt.enabled) {
        if (hipHostMalloc(reinterpret_cast<void**>(&h->h_flag), 64, hipHostMallocDefault) != hipSuccess ||
            hipMalloc(reinterpret_cast<void**>(&h->d_done_cnt), 64) != hipSuccess ||
            hipMemset(h->d_done_cnt, 0, 64) != hipSuccess) { h->err = "cannot allocate the completion word"; return fail(LTPL_ERR_HIP); }
        *h->h_flag = 0u;
    }
    // environment switches are read ONCE here: getenv() in a per-tick entry point costs microseconds in a process with a large environment
#ifdef LTPL_EXPERIMENT
    if (const char* e = getenv("LTPL_SCRATCH_POISON")) { h->scratch_poison_on = 1; h->scratch_poison_word = (unsigned)strtoul(e, nullptr, 0); }
    if (const char* e = getenv("LTPL_EXP_SKIP")) h->exp_skip = atoi(e);
#endif
    h->force_fused = getenv("LTPL_FORCE_FUSED") ? 1 : 0;
    h->no_overlap = getenv("LTPL_NO_OVERLAP") ? 1 : 0;
    if (const char* e = getenv("LTPL_FINAL_Y")) h->final_y = atoi(e) > 0 ? atoi(e) : 8;
    if (const char* e = getenv("LTPL_NW1_MIN_SCEN")) h->nw1_min_scen = atoi(e) > 0 ? atoi(e) : PIPELINE_MIN_SCEN;
#ifdef LTPL_EXPERIMENT
    if (getenv("LTPL_DEBUG_TIMING")) {
        if (hipMalloc(reinterpret_cast<void**>(&h->d_dbg), sizeof(long long) * 256 * DBG_SLOTS) == hipSuccess) {
            (void)hipMemset(h->d_dbg, 0, sizeof(long long) * 256 * DBG_SLOTS);
            h->lp1.dbg = h->d_dbg; h->lp4.dbg = h->d_dbg;
        }
    }
#endif
    if (h->lp4.total > lds_limit || h->lp1.total > lds_limit) {
        h->err = "planning horizon too large: the path scratch of the sweep does not fit in LDS (" + std::to_string(h->lp4.total) + " B > 150 KiB)";
        return fail(LTPL_ERR_CAPACITY);
    }
    if (h->lp4.total > 48 * 1024 || h->lp1.total > 48 * 1024) {
        if (hipFuncSetAttribute(paths_kernel_of(h, 1), hipFuncAttributeMaxDynamicSharedMemorySize,
                                h->lp1.total) != hipSuccess ||
            hipFuncSetAttribute(paths_kernel_of(h, NUM_WAVES), hipFuncAttributeMaxDynamicSharedMemorySize,
                                h->lp4.total) != hipSuccess) { h->err = "cannot raise dynamic LDS limit"; return fail(LTPL_ERR_HIP); }
    }
#ifdef LTPL_EXPERIMENT
    if (getenv("LTPL_DEBUG_OCC"))
#else
    if (false)
#endif
    {
        int nb1 = -1, nb4 = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb1, paths_kernel_of(h, 1), 64, (size_t)h->lp1.total);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb4, paths_kernel_of(h, NUM_WAVES), WG_THREADS, (size_t)h->lp4.total);
        fprintf(stderr, "[ltpl occ] k_paths<1> (plan class %d): %d blocks/CU at %d B LDS; k_paths<4>: %d blocks/CU at %d B LDS\n", h->plan_class, nb1, h->lp1.total, nb4, h->lp4.total);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { h->err = "hipGetDeviceProperties failed"; return fail(LTPL_ERR_HIP); }
    // capacity in rows rounded up to 16: a slot's row block of the fp64 outputs (vx, ax: 8 B per row; path_param: 40 B) then starts on a
    // 128-byte line, so the 128-byte chunks the final velocity kernel writes are whole sectors (measured write traffic was 1.8x the data)
    h->caps.max_path_nodes = hmax; h->caps.max_path_pts = (int)align_up((size_t)ptsmax, 16); h->caps.max_horizon_edges = ehmax;
    h->caps.device = device; h->caps.num_cus = prop.multiProcessorCount; h->caps.lds_bytes_paths = h->lp1.total;
    {
        // fused tick vs pipeline (see pipeline_min_scen): workgroups of the fused kernel that fit one compute unit by LDS (its footprint is
        // the four-wave path plan + the velocity scratch of three primitives), at most two counted
        const size_t lds_tick = (size_t)h->lp4.total + vel_scratch_bytes(h->caps.max_path_pts, false, true) * LTPL_MAX_ACTIONS + (size_t)h->caps.max_path_pts + 32;
        const int per_cu = lds_tick > 0 && 160 * 1024 / lds_tick >= 2 ? 2 : 1;
        h->pipeline_min_scen = (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * per_cu + 1;
        if (h->pipeline_min_scen < PIPELINE_MIN_SCEN) h->pipeline_min_scen = PIPELINE_MIN_SCEN;
        if (const char* e = getenv("LTPL_PIPELINE_MIN_SCEN")) { if (atoi(e) > 0) h->pipeline_min_scen = atoi(e); }
        if (const char* e = getenv("LTPL_FOLLOW_EMIT_MIN_SCEN")) { if (atoi(e) > 0) h->follow_emit_min_scen = atoi(e); }
        if (getenv("LTPL_NO_SCEN_ORDER")) h->scen_order = 0;
    }
    if (d->raceline_x && d->raceline_y && d->node_psi) {
        std::string why;
        h->has_hostlat = h->hostlat.init(d, h->caps.max_path_nodes, h->caps.max_path_pts, &why) == LTPL_OK;
    }
    if (!getenv("LTPL_NO_SELFTEST")) {
        if ((rc = self_test(h, d)) != LTPL_OK) return fail(rc);
    }
    guard.h = nullptr;
    *out_handle = h;
    return LTPL_OK;
} LTPL_ABI_CATCH(nullptr)

extern "C" int ltpl_get_caps(const ltpl_handle* h, ltpl_caps* caps)
{
    if (!h || !caps) return LTPL_ERR_INVALID_ARG;
    *caps = h->caps;
    return LTPL_OK;
}

// ---- staging arena ---------------------------------------------------------------------------------------------------
struct Arena {
    size_t size = 0;
    size_t add(size_t bytes) { size_t o = size; size = align_up(size + bytes, 256); return o; }   // arrays start on cache-line multiples
};

// The device-resident batch of ltpl_batch_upload points into the staging buffers (d_in / d_out / d_planes). Every other
// entry point reuses (and may reallocate) them, so it drops the resident batch first: a later ltpl_batch_run / _download then
// fails with "no resident batch" instead of reading overwritten or freed memory.
static void drop_resident(ltpl_handle* h, bool keep_persistent_tick = false)
{
    // (a resident tick kernel reads the staging buffers too -- and would keep a device-wide synchronisation waiting: it leaves first)
    if (!keep_persistent_tick) persist_stop(h);
    if (h->resident || h->resident_x[0]) { persist_stop(h); (void)hipDeviceSynchronize(); }
    free_resident(h->resident); h->resident = nullptr;
    for (int i = 0; i < ltpl_handle::PIPE_SETS - 1; ++i) { free_resident(h->resident_x[i]); h->resident_x[i] = nullptr; }
}

static int ensure(ltpl_handle* h, void** hp, size_t* hcap, void** dp, size_t* dcap, size_t need)
{
    if (need > *hcap) {
        if (*hp) (void)hipHostFree(*hp);
        *hp = nullptr; *hcap = 0;
        size_t cap = align_up(need + need / 2, 4096);
        HIP_TRY(h, hipHostMalloc(hp, cap, hipHostMallocDefault));
        *hcap = cap;
    }
    if (need > *dcap) {
        if (*dp) (void)hipFree(*dp);
        *dp = nullptr; *dcap = 0;
        size_t cap = align_up(need + need / 2, 4096);
        HIP_TRY(h, hipMalloc(dp, cap));
        *dcap = cap;
    }
    return LTPL_OK;
}

struct InLayout {
    size_t w_last, start_layer, start_node, flags, last_action, const_closest, psi_s, veh_off, pos_off, veh_radius,
        pos_x, pos_y, zone_off, zone_gid, n_last, last_layer, last_node, order, total;
    int n_veh, n_pos, n_zone;
};

static int validate_and_layout(ltpl_handle* h, const ltpl_paths_in* in, InLayout* lo)
{
    if (!in || in->n_scen < 1) { h->err = "n_scen < 1"; return LTPL_ERR_INVALID_ARG; }
    const int n = in->n_scen;
    if (in->n_w_last < 0 || in->n_w_last > LTPL_MAX_LAST_NODES - 1) { h->err = "n_w_last out of range"; return LTPL_ERR_INVALID_ARG; }
    if (in->veh_off[0] != 0 || in->zone_off[0] != 0) { h->err = "offset arrays must start at 0"; return LTPL_ERR_INVALID_ARG; }
    lo->n_veh = in->veh_off[n]; lo->n_zone = in->zone_off[n];
    if (lo->n_veh < 0 || lo->n_zone < 0) { h->err = "negative offsets"; return LTPL_ERR_INVALID_ARG; }
    if (in->pos_off[0] != 0) { h->err = "pos_off must start at 0"; return LTPL_ERR_INVALID_ARG; }
    lo->n_pos = in->pos_off[lo->n_veh];
    for (int s = 0; s < n; ++s) {
        const int sl = in->start_layer[s];
        if (sl < 0 || sl >= h->lat.L) { h->err = "start_layer out of range"; return LTPL_ERR_INVALID_ARG; }
        if (h->rng_end_host[(size_t)sl] < 0) { h->err = "start layer without a planning range (end of an open track)"; return LTPL_ERR_INVALID_ARG; }
        const int nv = in->veh_off[s + 1] - in->veh_off[s];
        if (nv < 0 || nv > MAX_VEH) { h->err = "more than 96 vehicles in one scenario"; return LTPL_ERR_CAPACITY; }
        const int np = in->pos_off[in->veh_off[s + 1]] - in->pos_off[in->veh_off[s]];
        if (np < 0 || np > MAX_POS) { h->err = "more than 192 obstacle positions in one scenario"; return LTPL_ERR_CAPACITY; }
        if (in->n_last[s] < 0 || in->n_last[s] > LTPL_MAX_LAST_NODES) { h->err = "n_last out of range"; return LTPL_ERR_INVALID_ARG; }
        for (int v = in->veh_off[s]; v < in->veh_off[s + 1]; ++v)
            if (in->pos_off[v + 1] - in->pos_off[v] < 1) { h->err = "vehicle without position"; return LTPL_ERR_INVALID_ARG; }
    }
    for (int i = 0; i < lo->n_zone; ++i)
        if (in->zone_gid[i] < 0 || in->zone_gid[i] >= h->lat.V) { h->err = "zone node id out of range"; return LTPL_ERR_INVALID_ARG; }
    Arena a;
    lo->w_last = a.add(sizeof(double) * (size_t)(in->n_w_last + 1));
    lo->start_layer = a.add(sizeof(int) * (size_t)n); lo->start_node = a.add(sizeof(int) * (size_t)n);
    lo->flags = a.add(sizeof(int) * (size_t)n); lo->last_action = a.add(sizeof(int) * (size_t)n);
    lo->const_closest = a.add(sizeof(int) * (size_t)n); lo->psi_s = a.add(sizeof(double) * (size_t)n);
    lo->veh_off = a.add(sizeof(int) * (size_t)(n + 1)); lo->pos_off = a.add(sizeof(int) * (size_t)(lo->n_veh + 1));
    lo->veh_radius = a.add(sizeof(double) * (size_t)(lo->n_veh + 1));
    lo->pos_x = a.add(sizeof(double) * (size_t)(lo->n_pos + 1)); lo->pos_y = a.add(sizeof(double) * (size_t)(lo->n_pos + 1));
    lo->zone_off = a.add(sizeof(int) * (size_t)(n + 1)); lo->zone_gid = a.add(sizeof(int) * (size_t)(lo->n_zone + 1));
    lo->n_last = a.add(sizeof(int) * (size_t)n);
    lo->last_layer = a.add(sizeof(int) * (size_t)n * LTPL_MAX_LAST_NODES);
    lo->last_node = a.add(sizeof(int) * (size_t)n * LTPL_MAX_LAST_NODES);
    lo->order = a.add(sizeof(int) * (size_t)n);
    lo->total = a.size;
    return LTPL_OK;
}

#ifndef LTPL_SCEN_ORDER_MIN_SCEN
#define LTPL_SCEN_ORDER_MIN_SCEN 2048       // batches from this size on plan their scenarios in the order of their start layers (DevPathsIn::order)
#endif
static void pack_in(const ltpl_paths_in* in, const InLayout& lo, unsigned char* hb, const unsigned char* db, DevPathsIn* di, bool scen_order = true)
{
    const int n = in->n_scen;
#define CP(field, src, count, T) do { if ((count) > 0) memcpy(hb + lo.field, src, sizeof(T) * (size_t)(count)); } while (0)
    CP(w_last, in->w_last_edges, in->n_w_last, double);
    CP(start_layer, in->start_layer, n, int); CP(start_node, in->start_node, n, int); CP(flags, in->flags, n, int);
    CP(last_action, in->last_action, n, int); CP(const_closest, in->const_closest, n, int); CP(psi_s, in->psi_s, n, double);
    CP(veh_off, in->veh_off, n + 1, int); CP(pos_off, in->pos_off, lo.n_veh + 1, int);
    CP(veh_radius, in->veh_radius, lo.n_veh, double); CP(pos_x, in->pos_x, lo.n_pos, double); CP(pos_y, in->pos_y, lo.n_pos, double);
    CP(zone_off, in->zone_off, n + 1, int); CP(zone_gid, in->zone_gid, lo.n_zone, int);
    CP(n_last, in->n_last, n, int); CP(last_layer, in->last_layer, n * LTPL_MAX_LAST_NODES, int);
    CP(last_node, in->last_node, n * LTPL_MAX_LAST_NODES, int);
#undef CP
    di->n_scen = n; di->n_w_last = in->n_w_last;
    di->w_last = reinterpret_cast<const double*>(db + lo.w_last);
    di->start_layer = reinterpret_cast<const int*>(db + lo.start_layer);
    di->start_node = reinterpret_cast<const int*>(db + lo.start_node);
    di->flags = reinterpret_cast<const int*>(db + lo.flags);
    di->last_action = reinterpret_cast<const int*>(db + lo.last_action);
    di->const_closest = reinterpret_cast<const int*>(db + lo.const_closest);
    di->psi_s = reinterpret_cast<const double*>(db + lo.psi_s);
    di->veh_off = reinterpret_cast<const int*>(db + lo.veh_off);
    di->pos_off = reinterpret_cast<const int*>(db + lo.pos_off);
    di->veh_radius = reinterpret_cast<const double*>(db + lo.veh_radius);
    di->pos_x = reinterpret_cast<const double*>(db + lo.pos_x);
    di->pos_y = reinterpret_cast<const double*>(db + lo.pos_y);
    di->zone_off = reinterpret_cast<const int*>(db + lo.zone_off);
    di->zone_gid = reinterpret_cast<const int*>(db + lo.zone_gid);
    di->order = nullptr;
    if (n >= LTPL_SCEN_ORDER_MIN_SCEN && scen_order) {
        // counting sort of the scenarios by start layer (stable: equal layers keep the caller's order)
        int* ord = reinterpret_cast<int*>(hb + lo.order);
        int lmax = 0;
        for (int i = 0; i < n; ++i) { const int l = in->start_layer[i]; if (l > lmax) lmax = l; }
        std::vector<int> cnt((size_t)(lmax > 0 ? lmax : 0) + 2, 0);
        for (int i = 0; i < n; ++i) { const int l = in->start_layer[i]; ++cnt[(size_t)(l > 0 ? l : 0) + 1]; }
        for (size_t l = 1; l < cnt.size(); ++l) cnt[l] += cnt[l - 1];
        for (int i = 0; i < n; ++i) { const int l = in->start_layer[i]; ord[(size_t)cnt[(size_t)(l > 0 ? l : 0)]++] = i; }
        di->order = reinterpret_cast<const int*>(db + lo.order);
    }
    di->n_last = reinterpret_cast<const int*>(db + lo.n_last);
    di->last_layer = reinterpret_cast<const int*>(db + lo.last_layer);
    di->last_node = reinterpret_cast<const int*>(db + lo.last_node);
}

struct OutLayout {
    size_t end_layer, closest_obj_index, closest_obj_node, n_actions, action_id, valid, reduced, goal_layer, n_nodes,
        n_pts, n_ties, nodes, node_idx, coeff, path_param, total;
};

static void layout_out(int n, int cap_nodes, int cap_pts, OutLayout* lo)
{
    Arena a; const size_t A = LTPL_MAX_ACTIONS;
    lo->end_layer = a.add(sizeof(int) * (size_t)n); lo->closest_obj_index = a.add(sizeof(int) * (size_t)n);
    lo->closest_obj_node = a.add(sizeof(int) * (size_t)n * 2); lo->n_actions = a.add(sizeof(int) * (size_t)n);
    lo->action_id = a.add(sizeof(int) * n * A); lo->valid = a.add(sizeof(int) * n * A);
    lo->reduced = a.add(sizeof(int) * n * A); lo->goal_layer = a.add(sizeof(int) * n * A);
    lo->n_nodes = a.add(sizeof(int) * n * A); lo->n_pts = a.add(sizeof(int) * n * A); lo->n_ties = a.add(sizeof(int) * n * A);
    lo->nodes = a.add(sizeof(int) * n * A * (size_t)cap_nodes); lo->node_idx = a.add(sizeof(int) * n * A * (size_t)cap_nodes);
    lo->coeff = a.add(sizeof(double) * n * A * (size_t)cap_nodes * 8);
    lo->path_param = a.add(sizeof(double) * n * A * (size_t)cap_pts * 5);
    lo->total = a.size;
}

static void bind_out(unsigned char* db, const OutLayout& lo, int cap_nodes, int cap_pts, DevPathsOut* d)
{
    d->cap_nodes = cap_nodes; d->cap_pts = cap_pts;
    d->end_layer = reinterpret_cast<int*>(db + lo.end_layer);
    d->closest_obj_index = reinterpret_cast<int*>(db + lo.closest_obj_index);
    d->closest_obj_node = reinterpret_cast<int*>(db + lo.closest_obj_node);
    d->n_actions = reinterpret_cast<int*>(db + lo.n_actions);
    d->action_id = reinterpret_cast<int*>(db + lo.action_id); d->valid = reinterpret_cast<int*>(db + lo.valid);
    d->reduced = reinterpret_cast<int*>(db + lo.reduced); d->goal_layer = reinterpret_cast<int*>(db + lo.goal_layer);
    d->n_nodes = reinterpret_cast<int*>(db + lo.n_nodes); d->n_pts = reinterpret_cast<int*>(db + lo.n_pts);
    d->n_ties = reinterpret_cast<int*>(db + lo.n_ties); d->nodes = reinterpret_cast<int*>(db + lo.nodes);
    d->node_idx = reinterpret_cast<int*>(db + lo.node_idx); d->coeff = reinterpret_cast<double*>(db + lo.coeff);
    d->path_param = reinterpret_cast<double*>(db + lo.path_param);
    d->vke = nullptr; d->vxy = nullptr; d->job_cnt = nullptr; d->job_slot = nullptr; d->n_slots_pad = 0;
    d->done.host_flag = nullptr; d->done.dev_count = nullptr; d->done.seq = 0u;
}

static void scatter_out(const unsigned char* hb, const OutLayout& lo, int n, ltpl_paths_out* out)
{
    const size_t A = LTPL_MAX_ACTIONS, cn = (size_t)out->cap_nodes, cp = (size_t)out->cap_pts;
    memcpy(out->end_layer, hb + lo.end_layer, sizeof(int) * (size_t)n);
    memcpy(out->closest_obj_index, hb + lo.closest_obj_index, sizeof(int) * (size_t)n);
    memcpy(out->closest_obj_node, hb + lo.closest_obj_node, sizeof(int) * (size_t)n * 2);
    memcpy(out->n_actions, hb + lo.n_actions, sizeof(int) * (size_t)n);
    memcpy(out->action_id, hb + lo.action_id, sizeof(int) * n * A);
    memcpy(out->valid, hb + lo.valid, sizeof(int) * n * A);
    memcpy(out->reduced, hb + lo.reduced, sizeof(int) * n * A);
    memcpy(out->goal_layer, hb + lo.goal_layer, sizeof(int) * n * A);
    memcpy(out->n_nodes, hb + lo.n_nodes, sizeof(int) * n * A);
    memcpy(out->n_pts, hb + lo.n_pts, sizeof(int) * n * A);
    memcpy(out->n_ties, hb + lo.n_ties, sizeof(int) * n * A);
    memcpy(out->nodes, hb + lo.nodes, sizeof(int) * n * A * cn);
    memcpy(out->node_idx, hb + lo.node_idx, sizeof(int) * n * A * cn);
    memcpy(out->coeff, hb + lo.coeff, sizeof(double) * n * A * cn * 8);
    memcpy(out->path_param, hb + lo.path_param, sizeof(double) * n * A * cp * 5);
}

// testing only (LTPL_SCRATCH_POISON=<hex word>): fills the private-segment arena of the stream with a pattern before the
// path kernel runs, so that a reload of a register spill slot the lane never stored shows up as a parity failure
#ifdef LTPL_EXPERIMENT
__global__ __launch_bounds__(64) void k_scratch_poison(unsigned pattern, unsigned* sink)
{
    volatile unsigned a[256];
    for (int i = 0; i < 256; ++i) a[i] = pattern;
    unsigned acc = 0;
    for (int i = 0; i < 256; i += 37) acc += a[i];
    if (acc == 0x12345u && sink) *sink = acc;
}
#endif

static void scratch_poison(ltpl_handle* h)
{
#ifdef LTPL_EXPERIMENT
    if (h->scratch_poison_on)
        hipLaunchKernelGGL(k_scratch_poison, dim3(256 * 64), dim3(64), 0, h->stream, h->scratch_poison_word, (unsigned*)nullptr);
#else
    (void)h;
#endif
}

// completion of a small zero-copy call. The polled word is written by the kernel's last block (signal_done); a launch that
// never signals (fault) is picked up by the stream synchronisation the wait falls back to after 20 ms.
static DoneSignal next_done_signal(ltpl_handle* h)
{
    DoneSignal d; d.host_flag = h->h_flag; d.dev_count = h->d_done_cnt; d.seq = ++h->done_seq;
    if (d.seq == 0u) d.seq = ++h->done_seq;                    // 0 is the initial value of the word
    return d;
}
static int wait_done(ltpl_handle* h, unsigned seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0x3fffu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) != seq) { h->err = "kernel finished without its completion signal"; return LTPL_ERR_HIP; }
            break;
        }
    }
    if (h->poll_query) (void)hipStreamQuery(h->stream);       // non-blocking: lets the runtime retire the launch's bookkeeping
    // let the runtime retire its bookkeeping of the launches now and then (LTPL_POLL_SYNC_EVERY calls, 0 = never)
    if (h->poll_sync_every > 0 && (++h->polled_calls % (unsigned)h->poll_sync_every) == 0u) HIP_TRY(h, hipStreamSynchronize(h->stream));
    return LTPL_OK;
}

static int plan_paths_impl(ltpl_handle* h, const ltpl_paths_in* in, ltpl_paths_out* out, int force_nw, uint8_t* blocked = nullptr);

extern "C" int ltpl_plan_paths(ltpl_handle* h, const ltpl_paths_in* in, ltpl_paths_out* out)
try {
    return plan_paths_impl(h, in, out, 0);
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_plan_paths_mask(ltpl_handle* h, const ltpl_paths_in* in, ltpl_paths_out* out, int32_t team_waves, uint8_t* blocked)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!blocked || !(team_waves == 0 || team_waves == 1 || team_waves == NUM_WAVES)) { h->err = "ltpl_plan_paths_mask: bad argument"; return LTPL_ERR_INVALID_ARG; }
    return plan_paths_impl(h, in, out, team_waves, blocked);
} LTPL_ABI_CATCH(abi_err_of(h))

// force_nw: 0 = choose by batch size, 1 / NUM_WAVES = one-wave batch kernel / four-wave latency kernel (self-test, diagnostics);
// blocked: [n_scen * E] bytes or nullptr -- the obstacle x edge mask as the kernel computed it (ltpl_plan_paths_mask)
static int plan_paths_impl(ltpl_handle* h, const ltpl_paths_in* in, ltpl_paths_out* out, int force_nw, uint8_t* blocked)
{
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!out) { h->err = "null output"; return LTPL_ERR_INVALID_ARG; }
    LTPL_PROF(prof_pack, "plan_paths.validate+pack");
    HIP_TRY(h, hipSetDevice(h->device));
    drop_resident(h);
    InLayout li;
    int rc = validate_and_layout(h, in, &li);
    if (rc) return rc;
    if (out->cap_nodes < h->caps.max_path_nodes || out->cap_pts < h->caps.max_path_pts) {
        h->err = "output capacity below ltpl_caps.max_path_nodes / max_path_pts"; return LTPL_ERR_CAPACITY;
    }
    OutLayout lo;
    layout_out(in->n_scen, out->cap_nodes, out->cap_pts, &lo);
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, li.total))) return rc;
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, lo.total))) return rc;
    DevPathsIn di; DevPathsOut dout;
    const bool zci = h->zc_in && in->n_scen <= 8;
    pack_in(in, li, static_cast<unsigned char*>(h->h_in), static_cast<const unsigned char*>(zci ? h->h_in : h->d_in), &di, h->scen_order != 0);
    // small calls (latency path): the output slab is the page-locked host buffer itself (device-accessible under unified
    // addressing): the kernel's stores cross PCIe as posted writes, no D2H copy is enqueued
    const bool zc = h->zc_out && in->n_scen <= 8;
    bind_out(static_cast<unsigned char*>(zc ? h->h_out : h->d_out), lo, out->cap_nodes, out->cap_pts, &dout);
    const int nw = force_nw ? force_nw : ((in->n_scen >= h->nw1_min_scen && h->batch_nw == 1) ? 1 : NUM_WAVES);
    const bool polled = zc && h->poll && !h->d_dbg && nw != 1;      // (the one-wave batch form carries no completion word)
    if (polled) dout.done = next_done_signal(h);
    prof_pack.stop();
    LTPL_PROF(prof_enq, "plan_paths.enqueue");
    {
        LTPL_PROF(prof_h2d, "plan_paths.enqueue.h2d");
        if (!zci) HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, li.total, hipMemcpyHostToDevice, h->stream));
    }
    scratch_poison(h);
    unsigned* d_mask = nullptr;
    const size_t mask_words = (size_t)((nw == 1 ? h->lp1 : h->lp4).words_blocked + 2);
    struct MaskGuard { unsigned* p = nullptr; ~MaskGuard() { if (p) (void)hipFree(p); } } mask_guard;
    if (blocked) {
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&d_mask), sizeof(unsigned) * mask_words * (size_t)in->n_scen));
        mask_guard.p = d_mask;
    }
    {
        LTPL_PROF(prof_l, "plan_paths.enqueue.launch");
        if ((rc = launch_paths(h, nw, in->n_scen, h->stream, di, dout, d_mask))) return rc;
    }
    if (!zc) HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, lo.total, hipMemcpyDeviceToHost, h->stream));
    prof_enq.stop();
    LTPL_PROF(prof_sync, "plan_paths.sync");
    if (polled) { if ((rc = wait_done(h, dout.done.seq))) return rc; }
    else HIP_TRY(h, hipStreamSynchronize(h->stream));
    prof_sync.stop();
    dbg_report(h, "k_paths", in->n_scen);
    if (blocked) {
        // per scenario: first edge of the planning range, its layer count, one bit per edge of the range in edge-id order (wraps at E)
        std::vector<unsigned> hm(mask_words * (size_t)in->n_scen);
        HIP_TRY(h, hipMemcpy(hm.data(), d_mask, sizeof(unsigned) * hm.size(), hipMemcpyDeviceToHost));
        const size_t E = (size_t)h->lat.E;
        memset(blocked, 0, E * (size_t)in->n_scen);
        for (int s = 0; s < in->n_scen; ++s) {
            const unsigned* w = hm.data() + mask_words * (size_t)s;
            const size_t e_base = w[0];
            for (size_t i = 0; i < (mask_words - 2) * 32 && i < E; ++i)
                if ((w[2 + (i >> 5)] >> (i & 31)) & 1u) blocked[(size_t)s * E + (size_t)h->sw2csc_host[(e_base + i) % E]] = 1;   // bitmap in sweep order
        }
    }
    LTPL_PROF(prof_sc, "plan_paths.scatter");
    scatter_out(static_cast<const unsigned char*>(h->h_out), lo, in->n_scen, out);
    return LTPL_OK;
}

extern "C" int ltpl_process_objects(ltpl_handle* h, const ltpl_objects_in* in, ltpl_objects_out* out)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!in || !out || in->n_obj < 1 || !in->x || !in->y || !in->theta || !in->v || !in->length ||
        !out->on_track || !out->pred_x || !out->pred_y || !out->radius) { h->err = "null argument or n_obj < 1"; return LTPL_ERR_INVALID_ARG; }
    if (!h->lat.ctx) { h->err = "the lattice was created without track bounds (normvec / width_right / width_left)"; return LTPL_ERR_UNSUPPORTED; }
    HIP_TRY(h, hipSetDevice(h->device));
    drop_resident(h);
    const size_t n = (size_t)in->n_obj;
    Arena ain, aout;
    const size_t o_x = ain.add(8 * n), o_y = ain.add(8 * n), o_t = ain.add(8 * n), o_v = ain.add(8 * n), o_l = ain.add(8 * n);
    const size_t o_px = aout.add(8 * n), o_py = aout.add(8 * n), o_r = aout.add(8 * n), o_on = aout.add(4 * n);
    int rc;
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, ain.size))) return rc;
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, aout.size))) return rc;
    unsigned char* hb = static_cast<unsigned char*>(h->h_in); unsigned char* db = static_cast<unsigned char*>(h->d_in);
    unsigned char* dob = static_cast<unsigned char*>(h->d_out);
    memcpy(hb + o_x, in->x, 8 * n); memcpy(hb + o_y, in->y, 8 * n); memcpy(hb + o_t, in->theta, 8 * n);
    memcpy(hb + o_v, in->v, 8 * n); memcpy(hb + o_l, in->length, 8 * n);
    HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, ain.size, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_process_objects, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, h->lat, in->n_obj, in->dt,
                       reinterpret_cast<const double*>(db + o_x), reinterpret_cast<const double*>(db + o_y),
                       reinterpret_cast<const double*>(db + o_t), reinterpret_cast<const double*>(db + o_v),
                       reinterpret_cast<const double*>(db + o_l), reinterpret_cast<int*>(dob + o_on),
                       reinterpret_cast<double*>(dob + o_px), reinterpret_cast<double*>(dob + o_py), reinterpret_cast<double*>(dob + o_r));
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, aout.size, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const unsigned char* ho = static_cast<const unsigned char*>(h->h_out);
    memcpy(out->pred_x, ho + o_px, 8 * n); memcpy(out->pred_y, ho + o_py, 8 * n); memcpy(out->radius, ho + o_r, 8 * n);
    memcpy(out->on_track, ho + o_on, 4 * n);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

// --- velocity seam / fused tick ---------------------------------------------------------------------------------------
// kernel variant = (exponent mode, single-row machine table): compile-time specialisations of the recurrences
static int vel_variant(const ltpl_vel_params* vp)
{
    const int em = vp->dyn_model_exp == 1.0 ? 1 : (vp->dyn_model_exp == 2.0 ? 2 : 0);
    return em * 2 + (vp->n_ax_max_machines == 1 ? 1 : 0);
}
typedef void (*vel_kernel_t)(DevLat, DevVelParams, const DevVelJob*, const double*, double*, int*, int, long long*, DoneSignal, int);
typedef void (*tick_kernel_t)(DevLat, DevPathsIn, DevPathsOut, TeamLds, DevVelParams, DevTickVelIn, DevTickVelOut, int, int, int);
typedef void (*lanes_kernel_t)(DevLat, DevPathsIn, DevPathsOut, DevVelParams, DevTickVelIn, DevVelPrep, VelPlanes, int, int, int, long long*, DevTickVelOut, int);
#ifndef LTPL_EMIT_MIN_SCEN
#define LTPL_EMIT_MIN_SCEN 256        // launches with at least this many scenarios write the generic jobs' vx / ax from the lane kernel (k_vel_lanes)
#endif
static lanes_kernel_t lanes_kernel_of(int v)
{
    switch (v) {
        case 0: return k_vel_lanes<0, false>; case 1: return k_vel_lanes<0, true>;
        case 2: return k_vel_lanes<1, false>; case 3: return k_vel_lanes<1, true>;
        case 4: return k_vel_lanes<2, false>; default: return k_vel_lanes<2, true>;
    }
}
static vel_kernel_t vel_kernel_of(int v)
{
    switch (v) {
        case 0: return k_vel_profile<0, false>; case 1: return k_vel_profile<0, true>;
        case 2: return k_vel_profile<1, false>; case 3: return k_vel_profile<1, true>;
        case 4: return k_vel_profile<2, false>; default: return k_vel_profile<2, true>;
    }
}
template <int SEL>
static vel_kernel_t vel_kernel_const_of(int v)          // constant friction limits per job (fleet); SEL: see k_vel_profile
{
    switch (v) {
        case 0: return k_vel_profile<0, false, false, SEL>; case 1: return k_vel_profile<0, true, false, SEL>;
        case 2: return k_vel_profile<1, false, false, SEL>; case 3: return k_vel_profile<1, true, false, SEL>;
        case 4: return k_vel_profile<2, false, false, SEL>; default: return k_vel_profile<2, true, false, SEL>;
    }
}
template <int SEL>
static vel_kernel_t vel_kernel_rows_of(int v)           // friction limits as rows per job (fleet, local_gg as a dict): the interpolating machine-table form only
{
    switch (v >> 1) {
        case 0: return k_vel_profile<0, false, true, SEL>; case 1: return k_vel_profile<1, false, true, SEL>;
        default: return k_vel_profile<2, false, true, SEL>;
    }
}
static tick_kernel_t tick_kernel_of(int v, bool plan_a = false)
{
    if (plan_a)
        switch (v) {
            case 0: return k_tick<0, false, PlanA4>; case 1: return k_tick<0, true, PlanA4>;
            case 2: return k_tick<1, false, PlanA4>; case 3: return k_tick<1, true, PlanA4>;
            case 4: return k_tick<2, false, PlanA4>; default: return k_tick<2, true, PlanA4>;
        }
    switch (v) {
        case 0: return k_tick<0, false, PlanRt>; case 1: return k_tick<0, true, PlanRt>;
        case 2: return k_tick<1, false, PlanRt>; case 3: return k_tick<1, true, PlanRt>;
        case 4: return k_tick<2, false, PlanRt>; default: return k_tick<2, true, PlanRt>;
    }
}

static int make_vel_params(ltpl_handle* h, const ltpl_vel_params* vp, const double* d_axm, DevVelParams* p)
{
    if (!vp || vp->n_ax_max_machines < 1 || !vp->ax_max_machines) { h->err = "ax_max_machines missing"; return LTPL_ERR_INVALID_ARG; }
    if (vp->n_ax_max_machines > 64) { h->err = "ax_max_machines with more than 64 rows"; return LTPL_ERR_CAPACITY; }
    if (!(vp->dyn_model_exp > 0.0) || !(vp->m_veh > 0.0)) { h->err = "invalid vehicle parameters"; return LTPL_ERR_INVALID_ARG; }
    p->e = vp->dyn_model_exp; p->inv_e = 1.0 / vp->dyn_model_exp; p->drag_m = vp->drag_coeff / vp->m_veh;
    p->len_veh = vp->len_veh; p->v_max = vp->v_max; p->n_axm = vp->n_ax_max_machines; p->ctrl = vp->follow_control_type;
    p->axm = d_axm; p->c_p = vp->c_p; p->k_p = vp->k_p; p->k_d = vp->k_d; p->tan_w = vp->tan_w;
    return LTPL_OK;
}

extern "C" int ltpl_vel_profile(ltpl_handle* h, const ltpl_vel_params* vp, int n_jobs, const ltpl_vel_job* jobs,
                                ltpl_vel_result* results)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (n_jobs < 1 || !jobs || !results) { h->err = "no jobs"; return LTPL_ERR_INVALID_ARG; }
    LTPL_PROF(prof_pack, "vel_profile.validate+pack");
    HIP_TRY(h, hipSetDevice(h->device));
    drop_resident(h);
    // pooled layout: [axm table][jobs][kappa | el | gg per job] ; outputs: [flags][vx per job]
    Arena ain, aout;
    const size_t o_axm = ain.add(sizeof(double) * 2 * (size_t)(vp ? vp->n_ax_max_machines : 1));
    const size_t o_jobs = ain.add(sizeof(DevVelJob) * (size_t)n_jobs);
    const size_t o_pool = ain.add(0);
    size_t pool_doubles = 0, out_doubles = 0; int cap = 0;
    std::vector<DevVelJob> dj((size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        const ltpl_vel_job& jb = jobs[j];
        if (jb.n < 1 || !jb.kappa || !jb.loc_gg || !results[j].vx) { h->err = "job without data"; return LTPL_ERR_INVALID_ARG; }
        if (jb.mode == LTPL_VEL_FOLLOW || jb.mode == LTPL_VEL_FOLLOW_CONTROLLED) { if (jb.n_el < jb.n) { h->err = "follow: el_lengths shorter than kappa"; return LTPL_ERR_INVALID_ARG; } }
        else if (jb.mode == LTPL_VEL_FB || jb.mode == LTPL_VEL_BRAKE) {
            if (jb.n_el != jb.n - 1) { h->err = "kappa must have the length of el_lengths + 1"; return LTPL_ERR_INVALID_ARG; }
        } else { h->err = "unknown velocity mode"; return LTPL_ERR_INVALID_ARG; }
        DevVelJob& d = dj[(size_t)j];
        d.mode = jb.mode; d.n = jb.n; d.n_el = jb.n_el; d.has_v_end = jb.has_v_end;
        d.off_kappa = (int)pool_doubles; pool_doubles += (size_t)jb.n;
        d.off_el = (int)pool_doubles; pool_doubles += (size_t)jb.n_el;
        d.off_gg = (int)pool_doubles; pool_doubles += 2 * (size_t)jb.n;
        d.off_out = (int)out_doubles; out_doubles += (size_t)jb.n;
        d.v_start = jb.v_start; d.v_end = jb.v_end; d.v_ego = jb.v_ego; d.v_obj = jb.v_obj; d.safety_d = jb.safety_d;
        d.obj_dist = jb.obj_dist; d.obj_x = jb.obj_x; d.obj_y = jb.obj_y;
        const int need = jb.n_el > jb.n ? jb.n_el : jb.n;
        if (need > cap) cap = need;
    }
    ain.add(sizeof(double) * pool_doubles);
    const size_t o_flags = aout.add(sizeof(int) * 2 * (size_t)n_jobs);
    const size_t o_vx = aout.add(sizeof(double) * out_doubles);
    const size_t lds = vel_scratch_bytes(cap, true, false);
    if (lds > 150 * 1024) { h->err = "velocity profile too long for the LDS-resident solver"; return LTPL_ERR_CAPACITY; }
    int rc;
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, ain.size))) return rc;
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, aout.size))) return rc;
    unsigned char* hb = static_cast<unsigned char*>(h->h_in);
    const bool zci = h->zc_in && n_jobs <= 16;
    unsigned char* db = static_cast<unsigned char*>(zci ? h->h_in : h->d_in);
    DevVelParams p;
    if ((rc = make_vel_params(h, vp, reinterpret_cast<const double*>(db + o_axm), &p))) return rc;
    memcpy(hb + o_axm, vp->ax_max_machines, sizeof(double) * 2 * (size_t)vp->n_ax_max_machines);
    memcpy(hb + o_jobs, dj.data(), sizeof(DevVelJob) * (size_t)n_jobs);
    double* pool = reinterpret_cast<double*>(hb + o_pool);
    for (int j = 0; j < n_jobs; ++j) {
        const ltpl_vel_job& jb = jobs[j]; const DevVelJob& d = dj[(size_t)j];
        memcpy(pool + d.off_kappa, jb.kappa, sizeof(double) * (size_t)jb.n);
        if (jb.n_el > 0) memcpy(pool + d.off_el, jb.el_lengths, sizeof(double) * (size_t)jb.n_el);
        memcpy(pool + d.off_gg, jb.loc_gg, sizeof(double) * 2 * (size_t)jb.n);
    }
    vel_kernel_t kern = vel_kernel_of(vel_variant(vp));
    {
        // the dynamic-LDS limit of a kernel is process-wide state: raised when a call needs more, never lowered, not re-set per tick
        static std::atomic<size_t> g_set[16][8];
        std::atomic<size_t>& cur = g_set[h->device & 15][vel_variant(vp) & 7];
        if (lds > 48 * 1024 && lds > cur.load(std::memory_order_relaxed)) {
            HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            cur = lds;
        }
    }
    prof_pack.stop();
    LTPL_PROF(prof_enq, "vel_profile.enqueue");
    if (!zci) HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, ain.size, hipMemcpyHostToDevice, h->stream));
    const bool zc = h->zc_out && n_jobs <= 16;
    unsigned char* dob = static_cast<unsigned char*>(zc ? h->h_out : h->d_out);
    const bool polled = zc && h->poll && !h->d_dbg;
    DoneSignal done; done.host_flag = nullptr; done.dev_count = nullptr; done.seq = 0u;
    if (polled) done = next_done_signal(h);
    hipLaunchKernelGGL(kern, dim3(n_jobs), dim3(64), lds, h->stream, h->lat, p,
                       reinterpret_cast<const DevVelJob*>(db + o_jobs), reinterpret_cast<const double*>(db + o_pool),
                       reinterpret_cast<double*>(dob + o_vx), reinterpret_cast<int*>(dob + o_flags), cap, h->lp4.dbg, done, 1);
    HIP_TRY(h, hipGetLastError());
    if (!zc) HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, aout.size, hipMemcpyDeviceToHost, h->stream));
    prof_enq.stop();
    LTPL_PROF(prof_sync, "vel_profile.sync");
    if (polled) { if ((rc = wait_done(h, done.seq))) return rc; }
    else HIP_TRY(h, hipStreamSynchronize(h->stream));
    prof_sync.stop();
    dbg_report(h, "k_vel_profile", n_jobs);
    LTPL_PROF(prof_sc, "vel_profile.scatter");
    const unsigned char* ho = static_cast<const unsigned char*>(h->h_out);
    const int* flags = reinterpret_cast<const int*>(ho + o_flags);
    const double* vx = reinterpret_cast<const double*>(ho + o_vx);
    for (int j = 0; j < n_jobs; ++j) {
        memcpy(results[j].vx, vx + dj[(size_t)j].off_out, sizeof(double) * (size_t)jobs[j].n);
        results[j].too_close = flags[2 * j]; results[j].vel_bound = flags[2 * j + 1];
    }
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

// ---- fused tick ------------------------------------------------------------------------------------------------------
struct TickLayout {
    InLayout in; size_t axm, vel_plan, vel_est, pos_x, pos_y, veh_vel, in_total;
    OutLayout out; size_t vx, ax, vel_bound, too_close, out_total;
    int n_scen, cap_nodes, cap_pts;
    DevPathsIn di; DevPathsOut dout; DevVelParams p; DevTickVelIn dvin; DevTickVelOut dvout;
    int vel_off, vel_stride, vel_cap; size_t lds;
    int variant;
    // two-kernel batch pipeline (n_scen >= pipeline_min_scen)
    bool pipeline; size_t prep_odist, prep_vobj, prep_ox, prep_oy, prep_idx; size_t planes_bytes;
    DevVelPrep dprep; int prep_off, prep_stride;
    VelPlanes vp{}; int n_slots_pad = 0, n_scen_pad = 0;
};

static int tick_prepare(ltpl_handle* h, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin, int cap_nodes, int cap_pts,
                        TickLayout* t)
{
    int rc = validate_and_layout(h, in, &t->in);
    if (rc) return rc;
    if (!vin || !vin->params) { h->err = "velocity inputs missing"; return LTPL_ERR_INVALID_ARG; }
    if (cap_nodes < h->caps.max_path_nodes || cap_pts < h->caps.max_path_pts) { h->err = "output capacity too small"; return LTPL_ERR_CAPACITY; }
    const int n = in->n_scen;
    for (int s = 0; s < n; ++s)
        if (vin->vel_plan[s] > vin->params->v_max + 0.1) {
            h->err = "vel_plan > v_max + 0.1: the brake-prefix branch is not part of the fused tick (the reference raises at OTH.py:919)";
            return LTPL_ERR_UNSUPPORTED;
        }
    t->n_scen = n; t->cap_nodes = cap_nodes; t->cap_pts = cap_pts;
    t->variant = vel_variant(vin->params);
    Arena a; a.size = t->in.total;
    t->axm = a.add(sizeof(double) * 2 * (size_t)vin->params->n_ax_max_machines);
    t->vel_plan = a.add(sizeof(double) * (size_t)n); t->vel_est = a.add(sizeof(double) * (size_t)n);
    t->pos_x = a.add(sizeof(double) * (size_t)n); t->pos_y = a.add(sizeof(double) * (size_t)n);
    t->veh_vel = a.add(sizeof(double) * (size_t)(t->in.n_veh + 1));
    t->in_total = a.size;
    layout_out(n, cap_nodes, cap_pts, &t->out);
    Arena b; b.size = t->out.total;
    t->vx = b.add(sizeof(double) * (size_t)n * LTPL_MAX_ACTIONS * (size_t)cap_pts);
    t->ax = b.add(sizeof(double) * (size_t)n * LTPL_MAX_ACTIONS * (size_t)cap_pts);
    t->vel_bound = b.add(sizeof(int) * (size_t)n * LTPL_MAX_ACTIONS);
    t->too_close = b.add(sizeof(int) * (size_t)n * LTPL_MAX_ACTIONS);
    t->pipeline = h->long_horizon || (n >= h->pipeline_min_scen && !h->force_fused);
    t->prep_odist = b.add(sizeof(double) * (size_t)n * LTPL_MAX_ACTIONS);
    t->prep_vobj = b.add(sizeof(double) * (size_t)n * LTPL_MAX_ACTIONS);
    t->prep_ox = b.add(sizeof(double) * (size_t)n * LTPL_MAX_ACTIONS);
    t->prep_oy = b.add(sizeof(double) * (size_t)n * LTPL_MAX_ACTIONS);
    t->prep_idx = b.add(sizeof(int) * (size_t)n * LTPL_MAX_ACTIONS);
    t->out_total = b.size;
    t->n_slots_pad = (int)align_up((size_t)n * LTPL_MAX_ACTIONS, 64); t->n_scen_pad = (int)align_up((size_t)n, 64);
    // tiled planes by job: K, E, P0 for generic + follow jobs, P1, P2, P3 for follow jobs; flags, job table and counters behind
    {
        const size_t tiles = (size_t)t->n_slots_pad + (size_t)t->n_scen_pad;
        const size_t plane_rows = align_up((size_t)cap_pts, 8);
        // (the operand plane holds plane_rows = cap_pts rounded up to 8 records of sizeof(ke_t) = 16 bytes per tile)
        t->planes_bytes = t->pipeline ? sizeof(double) * (2 * plane_rows * tiles + (size_t)cap_pts * (tiles + 3 * (size_t)t->n_scen_pad) + 2 * plane_rows * (size_t)t->n_scen_pad)
                                            + sizeof(int) * (3 * tiles + 16 + 2 * (size_t)t->n_scen_pad) : 0;
    }
    t->prep_off = 0; t->prep_stride = 0;
    t->vel_cap = h->caps.max_path_pts;
    t->vel_stride = (int)vel_scratch_bytes(t->vel_cap, false, true);
    t->vel_off = h->lp4.total;
    t->lds = (size_t)h->lp4.total + (size_t)t->vel_stride * LTPL_MAX_ACTIONS + align_up((size_t)t->vel_cap + 16, 16);
    if (!t->pipeline && t->lds > 150 * 1024) { h->err = "fused tick exceeds the LDS budget"; return LTPL_ERR_CAPACITY; }
    return LTPL_OK;
}

// device-side output pointers of a tick layout for one buffer set (output slab + tiled planes)
static void tick_bind_outputs(TickLayout* t, unsigned char* dob, double* planes)
{
    bind_out(dob, t->out, t->cap_nodes, t->cap_pts, &t->dout);
    t->dvout.vx = reinterpret_cast<double*>(dob + t->vx); t->dvout.ax = reinterpret_cast<double*>(dob + t->ax);
    t->dvout.vel_bound = reinterpret_cast<int*>(dob + t->vel_bound);
    t->dvout.too_close = reinterpret_cast<int*>(dob + t->too_close);
    t->dprep.obj_dist = reinterpret_cast<double*>(dob + t->prep_odist);
    t->dprep.v_obj = reinterpret_cast<double*>(dob + t->prep_vobj);
    t->dprep.obj_x = reinterpret_cast<double*>(dob + t->prep_ox);
    t->dprep.obj_y = reinterpret_cast<double*>(dob + t->prep_oy);
    t->dprep.idx_s_opp = reinterpret_cast<int*>(dob + t->prep_idx);
    if (t->pipeline && planes) {
        const size_t tiles = (size_t)t->n_slots_pad + (size_t)t->n_scen_pad;
        const size_t per_all = (size_t)t->cap_pts * tiles, per_scen = (size_t)t->cap_pts * (size_t)t->n_scen_pad;
        const size_t plane_rows = align_up((size_t)t->cap_pts, 8);
        static_assert(sizeof(ke_t) <= 2 * sizeof(double), "operand plane: two doubles per row and tile");
        t->vp.KE = reinterpret_cast<ke_t*>(planes); t->vp.P0 = planes + 2 * plane_rows * tiles;
        t->vp.P1 = t->vp.P0 + per_all; t->vp.P2 = t->vp.P1 + per_scen; t->vp.P3 = t->vp.P2 + per_scen;
        t->vp.XY = t->vp.P3 + per_scen;
        int* ints = reinterpret_cast<int*>(t->vp.XY + 2 * plane_rows * (size_t)t->n_scen_pad);
        t->vp.flags = ints; t->dout.job_slot = reinterpret_cast<int2*>(ints + tiles); t->dout.job_cnt = ints + 3 * tiles; t->vp.fseg = ints + 3 * tiles + 16;
        t->dout.n_slots_pad = t->n_slots_pad;
        t->vp.cap_pts = t->cap_pts; t->vp.plane_rows = (int)plane_rows;
        t->dout.vke = t->vp.KE; t->dout.vxy = t->vp.XY;
    }
}

static int tick_pack(ltpl_handle* h, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin, TickLayout* t,
                     unsigned char* hb, unsigned char* db, unsigned char* dob)
{
    pack_in(in, t->in, hb, db, &t->di, h->scen_order != 0);
    const int n = in->n_scen;
    memcpy(hb + t->axm, vin->params->ax_max_machines, sizeof(double) * 2 * (size_t)vin->params->n_ax_max_machines);
    memcpy(hb + t->vel_plan, vin->vel_plan, sizeof(double) * (size_t)n);
    memcpy(hb + t->vel_est, vin->vel_est, sizeof(double) * (size_t)n);
    memcpy(hb + t->pos_x, vin->pos_est_x, sizeof(double) * (size_t)n);
    memcpy(hb + t->pos_y, vin->pos_est_y, sizeof(double) * (size_t)n);
    if (t->in.n_veh > 0) memcpy(hb + t->veh_vel, vin->veh_vel, sizeof(double) * (size_t)t->in.n_veh);
    int rc = make_vel_params(h, vin->params, reinterpret_cast<const double*>(db + t->axm), &t->p);
    if (rc) return rc;
    t->dvin.gg_ax = vin->gg_ax; t->dvin.gg_ay = vin->gg_ay; t->dvin.safety_d = vin->safety_d;
    t->dvin.v_max_offset = vin->v_max_offset;
    t->dvin.vel_plan = reinterpret_cast<const double*>(db + t->vel_plan);
    t->dvin.vel_est = reinterpret_cast<const double*>(db + t->vel_est);
    t->dvin.pos_est_x = reinterpret_cast<const double*>(db + t->pos_x);
    t->dvin.pos_est_y = reinterpret_cast<const double*>(db + t->pos_y);
    t->dvin.veh_vel = reinterpret_cast<const double*>(db + t->veh_vel);
    if (t->pipeline && t->planes_bytes > h->d_planes_cap) {
        if (h->d_planes) (void)hipFree(h->d_planes);
        h->d_planes = nullptr; h->d_planes_cap = 0;
        HIP_TRY(h, hipMalloc(&h->d_planes, t->planes_bytes));
        h->d_planes_cap = t->planes_bytes;
    }
    tick_bind_outputs(t, dob, static_cast<double*>(h->d_planes));
    return LTPL_OK;
}

static int tick_launch_paths(ltpl_handle* h, const TickLayout& t, hipStream_t st)
{
    if (t.dout.job_cnt) HIP_TRY(h, hipMemsetAsync(t.dout.job_cnt, 0, 3 * sizeof(int), st));
    return launch_paths(h, (t.n_scen >= h->nw1_min_scen && h->batch_nw == 1) ? 1 : NUM_WAVES, t.n_scen, st, t.di, t.dout);
}

static int tick_launch_vel(ltpl_handle* h, const TickLayout& t, hipStream_t st, hipEvent_t ev_after_prep = nullptr)
{
    // slots without a path: vel_bound = too_close = 0 (the job kernels only touch slots that own a job)
    HIP_TRY(h, hipMemsetAsync(t.dvout.vel_bound, 0, sizeof(int) * (size_t)t.n_scen * LTPL_MAX_ACTIONS, st));
    HIP_TRY(h, hipMemsetAsync(t.dvout.too_close, 0, sizeof(int) * (size_t)t.n_scen * LTPL_MAX_ACTIONS, st));
#ifdef LTPL_EXPERIMENT
    if (!(h->exp_skip & 1))
#endif
    hipLaunchKernelGGL(k_follow_prep, dim3((t.n_scen + 63) / 64), dim3(64), 0, st, h->lat, t.di, t.dout,
                       t.dvin, t.dprep, t.vp, t.n_scen, h->lp4.dbg);
    HIP_TRY(h, hipGetLastError());
    if (ev_after_prep) HIP_TRY(h, hipEventRecord(ev_after_prep, st));
    const int n_slots = t.n_scen * LTPL_MAX_ACTIONS;
    const int nb0 = (n_slots + 63) / 64, nb1 = (t.n_scen + 63) / 64;
    const int emit_generic = t.n_scen >= LTPL_EMIT_MIN_SCEN ? 1 : 0;
    const int emit = emit_generic | ((emit_generic && t.n_scen >= h->follow_emit_min_scen) ? 2 : 0);
#ifdef LTPL_EXPERIMENT
    if (!(h->exp_skip & 2))
#endif
    hipLaunchKernelGGL(lanes_kernel_of(t.variant), dim3(nb0 + 2 * nb1), dim3(64), 0, st, h->lat, t.di, t.dout,
                       t.p, t.dvin, t.dprep, t.vp, n_slots, t.n_scen, nb0, h->lp4.dbg, t.dvout, emit);
    HIP_TRY(h, hipGetLastError());
#ifdef LTPL_EXPERIMENT
    if (!(h->exp_skip & 4))
#endif
    hipLaunchKernelGGL(k_vel_final, dim3(emit_generic ? nb1 : nb0 + nb1, h->final_y), dim3(64), 0, st, t.dout, t.dvin, t.dvout, t.vp, n_slots, t.n_scen,
                       emit_generic ? nb0 : 0);
    HIP_TRY(h, hipGetLastError());
    return LTPL_OK;
}

static int tick_launch(ltpl_handle* h, const TickLayout& t, hipEvent_t* ev = nullptr)
{
    // ev (optional, 4 events): recorded before the path kernel, after it, after the follow preparation, after the lane kernels
    if (ev) HIP_TRY(h, hipEventRecord(ev[0], h->stream));
    if (t.pipeline) {
        int rc = tick_launch_paths(h, t, h->stream);
        if (rc) return rc;
        if (ev) HIP_TRY(h, hipEventRecord(ev[1], h->stream));
        if ((rc = tick_launch_vel(h, t, h->stream, ev ? ev[2] : nullptr))) return rc;
        if (ev) HIP_TRY(h, hipEventRecord(ev[3], h->stream));
        return LTPL_OK;
    }
    hipLaunchKernelGGL(tick_kernel_of(t.variant, h->plan_class4 == 1), dim3(t.n_scen), dim3(WG_THREADS), t.lds, h->stream, h->lat, t.di, t.dout, h->lp4, t.p,
                       t.dvin, t.dvout, t.vel_off, t.vel_stride, t.vel_cap);
    HIP_TRY(h, hipGetLastError());
    if (ev) { HIP_TRY(h, hipEventRecord(ev[1], h->stream)); HIP_TRY(h, hipEventRecord(ev[2], h->stream)); HIP_TRY(h, hipEventRecord(ev[3], h->stream)); }
    return LTPL_OK;
}

static void tick_scatter(const unsigned char* hb, const TickLayout& t, ltpl_paths_out* out, ltpl_tick_vel_out* vout)
{
    scatter_out(hb, t.out, t.n_scen, out);
    const size_t A = LTPL_MAX_ACTIONS, n = (size_t)t.n_scen;
    if (out->cap_pts == t.cap_pts) {
        memcpy(vout->vx, hb + t.vx, sizeof(double) * n * A * (size_t)t.cap_pts);
        memcpy(vout->ax, hb + t.ax, sizeof(double) * n * A * (size_t)t.cap_pts);
    }
    memcpy(vout->vel_bound, hb + t.vel_bound, sizeof(int) * n * A);
    memcpy(vout->too_close, hb + t.too_close, sizeof(int) * n * A);
}

static int tick_set_lds_limit(ltpl_handle* h, size_t lds, int variant)
{
    if (h->long_horizon) return LTPL_OK;                  // the fused kernel is not used
    static std::atomic<size_t> g_set[16][8][2];           // (process-wide limit: raised on demand, never lowered, not re-set per tick)
    std::atomic<size_t>& cur = g_set[h->device & 15][variant & 7][h->plan_class4 == 1 ? 1 : 0];
    if (lds > 48 * 1024 && lds > cur.load(std::memory_order_relaxed)) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(tick_kernel_of(variant, h->plan_class4 == 1)), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        cur = lds;
    }
    return LTPL_OK;
}

// The fused single tick as one hipGraph launch (LTPL_TICK_GRAPH=1): copy node (packed inputs, H2D) -> kernel node (k_tick) -> copy node
// (outputs, D2H; absent with zero-copy outputs). Built once; every tick re-parameterises the nodes of the executable graph (sizes and the
// pointers into the staging buffers follow the tick's counts) and launches it. A change of topology or kernel rebuilds the graph.
static int tick_launch_graph(ltpl_handle* h, TickLayout& t, bool zc)
{
    const void* func = reinterpret_cast<const void*>(tick_kernel_of(t.variant, h->plan_class4 == 1));
    int vel_off = t.vel_off, vel_stride = t.vel_stride, vel_cap = t.vel_cap;
    void* kargs[] = {&h->lat, &t.di, &t.dout, &h->lp4, &t.p, &t.dvin, &t.dvout, &vel_off, &vel_stride, &vel_cap};
    hipKernelNodeParams kp{};
    kp.func = const_cast<void*>(func); kp.gridDim = dim3(t.n_scen); kp.blockDim = dim3(WG_THREADS); kp.sharedMemBytes = (unsigned)t.lds;
    kp.kernelParams = kargs; kp.extra = nullptr;
    const int has_out = zc ? 0 : 1;
    bool ok = h->tg_exec && h->tg_func == func && h->tg_has_out == has_out;
    if (ok) {
        ok = hipGraphExecMemcpyNodeSetParams1D(h->tg_exec, h->tg_n_in, h->d_in, h->h_in, t.in_total, hipMemcpyHostToDevice) == hipSuccess &&
             hipGraphExecKernelNodeSetParams(h->tg_exec, h->tg_n_k, &kp) == hipSuccess &&
             (!has_out || hipGraphExecMemcpyNodeSetParams1D(h->tg_exec, h->tg_n_out, h->h_out, h->d_out, t.out_total, hipMemcpyDeviceToHost) == hipSuccess);
        if (!ok) (void)hipGetLastError();
    }
    if (!ok) {
        if (h->tg_exec) { (void)hipGraphExecDestroy(h->tg_exec); h->tg_exec = nullptr; }
        if (h->tg_graph) { (void)hipGraphDestroy(h->tg_graph); h->tg_graph = nullptr; }
        HIP_TRY(h, hipGraphCreate(&h->tg_graph, 0));
        HIP_TRY(h, hipGraphAddMemcpyNode1D(&h->tg_n_in, h->tg_graph, nullptr, 0, h->d_in, h->h_in, t.in_total, hipMemcpyHostToDevice));
        HIP_TRY(h, hipGraphAddKernelNode(&h->tg_n_k, h->tg_graph, &h->tg_n_in, 1, &kp));
        if (has_out) HIP_TRY(h, hipGraphAddMemcpyNode1D(&h->tg_n_out, h->tg_graph, &h->tg_n_k, 1, h->h_out, h->d_out, t.out_total, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipGraphInstantiate(&h->tg_exec, h->tg_graph, nullptr, nullptr, 0));
        h->tg_func = func; h->tg_has_out = has_out;
    }
    HIP_TRY(h, hipGraphLaunch(h->tg_exec, h->stream));
    return LTPL_OK;
}

// ---- the persistent single tick (k_tick_persistent): host side --------------------------------------------------------------------------
typedef void (*ptick_kernel_t)(const TickKArgs*, TickMailbox*, const unsigned char*, unsigned char*, unsigned, unsigned long long);
static ptick_kernel_t ptick_kernel_of(int v)
{
    switch (v) {
        case 0: return k_tick_persistent<0, false, PlanA4>; case 1: return k_tick_persistent<0, true, PlanA4>;
        case 2: return k_tick_persistent<1, false, PlanA4>; case 3: return k_tick_persistent<1, true, PlanA4>;
        case 4: return k_tick_persistent<2, false, PlanA4>; default: return k_tick_persistent<2, true, PlanA4>;
    }
}

// Tell the resident kernel to leave and wait until it has (every entry point that reuses the staging buffers or synchronises the device
// calls this first; ltpl_destroy too). Cheap when nothing is resident.
static void persist_stop(ltpl_handle* h)
{
    ltpl_handle::PersistTick& P = h->pt;
    if (!P.running) return;
    __atomic_store_n(&P.mb->cmd, PT_CMD_EXIT, __ATOMIC_RELAXED);
    if (++P.seq == 0u) ++P.seq;
    __atomic_store_n(&P.mb->seq, P.seq, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(P.stream);                     // the kernel sees the command within one poll (or had idled out already)
    P.busy_clk += P.mb->busy_clk; P.busy_n += P.mb->n_done;
    P.running = 0;
}

static int persist_start(ltpl_handle* h, const TickLayout& t, unsigned start_seq)
{
    ltpl_handle::PersistTick& P = h->pt;
    if (!P.stream) HIP_TRY(h, hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
    if (!P.mb) {
        HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&P.mb), sizeof(TickMailbox), hipHostMallocDefault));
        memset(P.mb, 0, sizeof(TickMailbox));
    }
    if (!P.d_args) HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&P.d_args), 2048));
    const void* func = reinterpret_cast<const void*>(ptick_kernel_of(t.variant));
    if (t.lds > 48 * 1024) HIP_TRY(h, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t.lds));
    P.mb->exited = 0; P.mb->n_done = 0; P.mb->busy_clk = 0; P.mb->cmd = PT_CMD_TICK;
    __atomic_store_n(&P.mb->seq, start_seq, __ATOMIC_RELEASE);
    const unsigned long long idle_limit = (unsigned long long)(P.idle_ms * 1e5);           // wall_clock64 counts at 100 MHz
    hipLaunchKernelGGL(ptick_kernel_of(t.variant), dim3(1), dim3(WG_THREADS), t.lds, P.stream, (const TickKArgs*)P.d_args, P.mb,
                       static_cast<const unsigned char*>(h->h_in), static_cast<unsigned char*>(h->d_in), start_seq, idle_limit);
    HIP_TRY(h, hipGetLastError());
    P.running = 1; P.variant = t.variant; P.lds = t.lds; P.h_in = h->h_in; P.d_in = h->d_in; P.seq = start_seq; ++P.launches;
    return LTPL_OK;
}

// one tick through the resident kernel: staging buffers packed as for k_tick (zero-copy outputs, polled completion word)
static int persist_tick(ltpl_handle* h, TickLayout& t)
{
    ltpl_handle::PersistTick& P = h->pt;
    // the kernel idled out since the last tick (its last store: `exited`), or serves another kernel variant / other buffers: (re)start
    if (P.running && __atomic_load_n(&P.mb->exited, __ATOMIC_ACQUIRE) != 0u) {
        HIP_TRY(h, hipStreamSynchronize(P.stream));
        P.busy_clk += P.mb->busy_clk; P.busy_n += P.mb->n_done; P.running = 0;
    }
    if (P.running && (P.variant != t.variant || P.lds != t.lds || P.h_in != h->h_in || P.d_in != h->d_in)) persist_stop(h);
    int rc;
    if (!P.running && (rc = persist_start(h, t, P.seq))) return rc;
    TickKArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.pk.lat = h->lat; ka.pk.in = t.di; ka.pk.out = t.dout; ka.pk.lp = h->lp4;
    ka.p = t.p; ka.vin = t.dvin; ka.vout = t.dvout; ka.vel_off = t.vel_off; ka.vel_stride = t.vel_stride; ka.vel_cap = t.vel_cap;
    memcpy(P.mb->args, &ka, sizeof(ka));
    P.mb->in_bytes = (unsigned)align_up(t.in_total, 16);
    P.mb->cmd = PT_CMD_TICK;
    if (++P.seq == 0u) ++P.seq;
    __atomic_store_n(&P.mb->seq, P.seq, __ATOMIC_RELEASE);
    const unsigned want = t.dout.done.seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1;; ++spins) {
        if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) == want) break;
        if ((spins & 0xfffu) != 0u) continue;
        if (__atomic_load_n(&P.mb->exited, __ATOMIC_ACQUIRE) != 0u) {
            // the kernel left (idle limit) between our check and our post: it cannot have seen this tick -- unless the completion word says so
            HIP_TRY(h, hipStreamSynchronize(P.stream));
            P.busy_clk += P.mb->busy_clk; P.busy_n += P.mb->n_done; P.running = 0;
            if (__atomic_load_n(h->h_flag, __ATOMIC_ACQUIRE) == want) break;
            const unsigned posted = P.seq;
            if ((rc = persist_start(h, t, posted - 1u))) return rc;    // the new kernel starts "one behind" ...
            P.seq = posted;
            __atomic_store_n(&P.mb->seq, posted, __ATOMIC_RELEASE);  // ... and finds the tick that is still in the mailbox
            continue;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2000)) {
            __atomic_store_n(&P.mb->cmd, PT_CMD_EXIT, __ATOMIC_RELAXED);
            h->err = "persistent tick: no completion within 2 s"; return LTPL_ERR_HIP;
        }
    }
    ++P.ticks;
    return LTPL_OK;
}

extern "C" int ltpl_tick_persistent_stop(ltpl_handle* h)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    persist_stop(h);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_tick_persistent_stats(ltpl_handle* h, ltpl_persistent_stats* st)
try {
    if (!h || !st) return LTPL_ERR_INVALID_ARG;
    const ltpl_handle::PersistTick& P = h->pt;
    st->enabled = P.enabled; st->resident = (P.running && P.mb && __atomic_load_n(&P.mb->exited, __ATOMIC_ACQUIRE) == 0u) ? 1 : 0;
    st->ticks = (long long)P.ticks; st->launches = (long long)P.launches;
    unsigned long long clk = P.busy_clk, n = P.busy_n;
    if (P.running && P.mb) { clk += P.mb->busy_clk; n += P.mb->n_done; }
    st->device_us_mean = n ? (double)clk / (double)n / 100.0 : 0.0;
    st->device_us_last = (P.mb ? (double)P.mb->last_clk : 0.0) / 100.0;
    st->idle_ms = P.idle_ms;
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_tick_batch(ltpl_handle* h, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin, ltpl_paths_out* out,
                               ltpl_tick_vel_out* vout)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!out || !vout) { h->err = "null output"; return LTPL_ERR_INVALID_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    const bool persistent = h->pt.enabled && in && in->n_scen == 1 && h->zc_out && !h->d_dbg;
    drop_resident(h, persistent);
    TickLayout t;
    int rc = tick_prepare(h, in, vin, out->cap_nodes, out->cap_pts, &t);
    if (rc) return rc;
    if (persistent && !t.pipeline) {
        // (growing a staging buffer frees the old one: the resident kernel leaves first -- persist_tick starts it again on the new buffers)
        if (t.in_total > h->h_in_cap || t.in_total > h->d_in_cap || t.out_total > h->h_out_cap || t.out_total > h->d_out_cap) persist_stop(h);
        if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, align_up(t.in_total, 16)))) return rc;
        if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, t.out_total))) return rc;
        if ((rc = tick_pack(h, in, vin, &t, static_cast<unsigned char*>(h->h_in), static_cast<unsigned char*>(h->d_in),
                            static_cast<unsigned char*>(h->h_out)))) return rc;
        t.dout.done = next_done_signal(h);
        if ((rc = persist_tick(h, t))) return rc;
        tick_scatter(static_cast<const unsigned char*>(h->h_out), t, out, vout);
        return LTPL_OK;
    }
    persist_stop(h);
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, t.in_total))) return rc;
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, t.out_total))) return rc;
    // single ticks (fused kernel): outputs straight into the page-locked buffer, completion through the polled word (see plan_paths_impl)
    const bool zc = h->zc_out && !t.pipeline && in->n_scen <= 8;
    const bool polled = zc && h->poll && !h->d_dbg;
    if ((rc = tick_pack(h, in, vin, &t, static_cast<unsigned char*>(h->h_in), static_cast<unsigned char*>(h->d_in),
                        static_cast<unsigned char*>(zc ? h->h_out : h->d_out)))) return rc;
    if (polled) t.dout.done = next_done_signal(h);
    if ((rc = tick_set_lds_limit(h, t.lds, t.variant))) return rc;
    if (h->tick_graph && !t.pipeline) {
        if ((rc = tick_launch_graph(h, t, zc))) return rc;
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, t.in_total, hipMemcpyHostToDevice, h->stream));
        if ((rc = tick_launch(h, t))) return rc;
        if (!zc) HIP_TRY(h, hipMemcpyAsync(h->h_out, h->d_out, t.out_total, hipMemcpyDeviceToHost, h->stream));
    }
    if (polled) { if ((rc = wait_done(h, t.dout.done.seq))) return rc; }
    else HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (t.pipeline) dbg_report_lanes(h); else dbg_report(h, "k_tick", in->n_scen);
    tick_scatter(static_cast<const unsigned char*>(h->h_out), t, out, vout);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" void* ltpl_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void ltpl_host_free(void* p) { if (p) (void)hipHostFree(p); }

extern "C" int ltpl_tick_batch_compact(ltpl_handle* h, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin, ltpl_traj_out* out)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!out || !out->rows || !out->action_id || !out->n_rows || !out->vel_bound || !out->reduced || !out->row_off || out->capacity_rows < 0) {
        h->err = "null output"; return LTPL_ERR_INVALID_ARG;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    drop_resident(h);
    TickLayout t;
    int rc = tick_prepare(h, in, vin, h->caps.max_path_nodes, h->caps.max_path_pts, &t);
    if (rc) return rc;
    const size_t n_slots = (size_t)in->n_scen * LTPL_MAX_ACTIONS;
    // device scratch behind the regular output slab: rows kept, offsets, total, packed rows
    Arena b; b.size = t.out_total;
    const size_t o_rows = b.add(sizeof(int) * n_slots), o_off = b.add(sizeof(long long) * n_slots), o_tot = b.add(sizeof(long long));
    const size_t o_pack = b.add(sizeof(double) * 7 * (size_t)out->capacity_rows);
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, t.in_total))) return rc;
    // host staging only for the small per-slot arrays (+ the packed rows when the caller's buffer is not page-locked)
    hipPointerAttribute_t attr; bool pinned = false;
    if (hipPointerGetAttributes(&attr, out->rows) == hipSuccess) pinned = attr.type == hipMemoryTypeHost;
    else (void)hipGetLastError();
    const size_t small_bytes = sizeof(int) * n_slots * 4 + sizeof(long long) * (n_slots + 1) + 64;
    const size_t h_need = small_bytes + (pinned ? 0 : sizeof(double) * 7 * (size_t)out->capacity_rows);
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, b.size > h_need ? b.size : h_need))) return rc;
    if ((rc = tick_pack(h, in, vin, &t, static_cast<unsigned char*>(h->h_in), static_cast<unsigned char*>(h->d_in),
                        static_cast<unsigned char*>(h->d_out)))) return rc;
    if ((rc = tick_set_lds_limit(h, t.lds, t.variant))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, t.in_total, hipMemcpyHostToDevice, h->stream));
    if ((rc = tick_launch(h, t))) return rc;
    unsigned char* dob = static_cast<unsigned char*>(h->d_out);
    int* d_rows = reinterpret_cast<int*>(dob + o_rows); long long* d_off = reinterpret_cast<long long*>(dob + o_off);
    long long* d_tot = reinterpret_cast<long long*>(dob + o_tot); double* d_pack = reinterpret_cast<double*>(dob + o_pack);
    hipLaunchKernelGGL(k_compact_offsets, dim3(1), dim3(1024), 0, h->stream, t.dout.valid, t.dout.n_pts, (int)n_slots, out->max_rows, d_rows, d_off, d_tot);
    hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)n_slots), dim3(64), 0, h->stream, t.dout, t.dvout, d_rows, d_off, d_pack, (long long)out->capacity_rows);
    HIP_TRY(h, hipGetLastError());
    // small arrays first (they tell how many packed bytes there are)
    unsigned char* hb = static_cast<unsigned char*>(h->h_out);
    int* h_rows = reinterpret_cast<int*>(hb); int* h_act = h_rows + n_slots; int* h_vb = h_act + n_slots; int* h_red = h_vb + n_slots;
    long long* h_off = reinterpret_cast<long long*>(h_red + n_slots + 2); long long* h_tot = h_off + n_slots;
    HIP_TRY(h, hipMemcpyAsync(h_rows, d_rows, sizeof(int) * n_slots, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h_act, t.dout.action_id, sizeof(int) * n_slots, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h_vb, t.dvout.vel_bound, sizeof(int) * n_slots, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h_red, t.dout.reduced, sizeof(int) * n_slots, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h_off, d_off, sizeof(long long) * n_slots, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h_tot, d_tot, sizeof(long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const long long total = *h_tot;
    out->total_rows = total;
    if (total > out->capacity_rows) { h->err = "ltpl_traj_out.capacity_rows too small (total_rows holds the need)"; return LTPL_ERR_CAPACITY; }
    double* h_pack = pinned ? out->rows : reinterpret_cast<double*>(hb + small_bytes);
    if (total > 0) {
        HIP_TRY(h, hipMemcpyAsync(h_pack, d_pack, sizeof(double) * 7 * (size_t)total, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (!pinned) memcpy(out->rows, h_pack, sizeof(double) * 7 * (size_t)total);
    }
    memcpy(out->n_rows, h_rows, sizeof(int) * n_slots); memcpy(out->action_id, h_act, sizeof(int) * n_slots);
    memcpy(out->vel_bound, h_vb, sizeof(int) * n_slots); memcpy(out->reduced, h_red, sizeof(int) * n_slots);
    memcpy(out->row_off, h_off, sizeof(long long) * n_slots);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

// device-resident batch (benchmarks): inputs stay in HBM, the fused kernel is replayed on the handle's stream
extern "C" int ltpl_batch_upload(ltpl_handle* h, const ltpl_paths_in* in, const ltpl_tick_vel_in* vin, int32_t cap_nodes,
                                 int32_t cap_pts)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    drop_resident(h);
    TickLayout* t = new TickLayout();
    int rc = tick_prepare(h, in, vin, cap_nodes, cap_pts, t);
    if (rc) { delete t; return rc; }
    if ((rc = ensure(h, &h->h_in, &h->h_in_cap, &h->d_in, &h->d_in_cap, t->in_total))) { delete t; return rc; }
    if ((rc = ensure(h, &h->h_out, &h->h_out_cap, &h->d_out, &h->d_out_cap, t->out_total))) { delete t; return rc; }
    if ((rc = tick_pack(h, in, vin, t, static_cast<unsigned char*>(h->h_in), static_cast<unsigned char*>(h->d_in),
                        static_cast<unsigned char*>(h->d_out)))) { delete t; return rc; }
    if ((rc = tick_set_lds_limit(h, t->lds, t->variant))) { delete t; return rc; }
    HIP_TRY(h, hipMemcpyAsync(h->d_in, h->h_in, t->in_total, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->resident = t;
    h->last_set = 0;
    if (t->pipeline && !h->no_overlap) {
        // further buffer sets for the software pipeline of ltpl_batch_run
        for (int i = 0; i < ltpl_handle::VEL_STREAMS; ++i) if (!h->vel_stream[i]) {
            // LTPL_VEL_STREAM_PRIO=1 (experiment): the velocity streams at the device's highest stream priority (their waves are dispatched
            // ahead of the path kernel's whenever a slot frees up)
            static const bool hi = getenv("LTPL_VEL_STREAM_PRIO") && atoi(getenv("LTPL_VEL_STREAM_PRIO")) != 0;
            // LTPL_VEL_CUS=<n> (round-5 experiment): the velocity streams confined to n compute units by a CU mask. The velocity waves are few,
            // long-lived (a serial recurrence over the rows of a profile) and hold 197 VGPRs: next to one of them a SIMD keeps two path waves
            // instead of four. Confined, they run among themselves on a corner of the chip. 0 / unset: no mask.
            static const int vel_cus = getenv("LTPL_VEL_CUS") ? atoi(getenv("LTPL_VEL_CUS")) : 0;
            bool made = false;
            if (vel_cus > 0) {
                const int ncu = h->caps.num_cus > 0 ? h->caps.num_cus : 256;
                std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
                for (int c = 0; c < vel_cus && c < ncu; ++c) mask[(size_t)c >> 5] |= 1u << (c & 31);
                made = hipExtStreamCreateWithCUMask(&h->vel_stream[i], (uint32_t)mask.size(), mask.data()) == hipSuccess;
                if (!made) { (void)hipGetLastError(); h->vel_stream[i] = nullptr; }
            }
            if (made) continue;
            if (hi) { int lo_p = 0, hi_p = 0; (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p); HIP_TRY(h, hipStreamCreateWithPriority(&h->vel_stream[i], hipStreamNonBlocking, hi_p)); }
            else HIP_TRY(h, hipStreamCreateWithFlags(&h->vel_stream[i], hipStreamNonBlocking));
        }
        {
            const char* ps = getenv("LTPL_PATH_STREAMS");
            h->path_streams = (ps && atoi(ps) == 2) ? 2 : 1;
            if (h->path_streams == 2 && !h->path_stream2) HIP_TRY(h, hipStreamCreateWithFlags(&h->path_stream2, hipStreamNonBlocking));
            if (!h->ev_begin) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_begin, hipEventDisableTiming));
        }
        for (int i = 0; i < ltpl_handle::PIPE_SETS; ++i) {
            if (!h->ev_paths[i]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_paths[i], hipEventDisableTiming));
            if (!h->ev_vel[i]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_vel[i], hipEventDisableTiming));
        }
        for (int i = 0; i < ltpl_handle::PIPE_SETS - 1; ++i) {
            if (t->out_total > h->d_out_x_cap[i]) {
                if (h->d_out_x[i]) (void)hipFree(h->d_out_x[i]);
                h->d_out_x[i] = nullptr; h->d_out_x_cap[i] = 0;
                HIP_TRY(h, hipMalloc(&h->d_out_x[i], t->out_total)); h->d_out_x_cap[i] = t->out_total;
            }
            if (t->planes_bytes > h->d_planes_x_cap[i]) {
                if (h->d_planes_x[i]) (void)hipFree(h->d_planes_x[i]);
                h->d_planes_x[i] = nullptr; h->d_planes_x_cap[i] = 0;
                HIP_TRY(h, hipMalloc(&h->d_planes_x[i], t->planes_bytes)); h->d_planes_x_cap[i] = t->planes_bytes;
            }
            TickLayout* tx = new TickLayout(*t);
            tick_bind_outputs(tx, static_cast<unsigned char*>(h->d_out_x[i]), static_cast<double*>(h->d_planes_x[i]));
            h->resident_x[i] = tx;
        }
    }
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_batch_run(ltpl_handle* h, int reps, float* ms_total)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!h->resident) { h->err = "no resident batch: call ltpl_batch_upload first"; return LTPL_ERR_INVALID_ARG; }
    if (reps < 1) { h->err = "reps < 1"; return LTPL_ERR_INVALID_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_total) {
        HIP_TRY(h, hipEventCreate(&e0)); HIP_TRY(h, hipEventCreate(&e1));
        HIP_TRY(h, hipEventRecord(e0, h->stream));
    }
    if (h->resident_x[0] && ms_total) {
        while (h->ev_step.size() < 2 * (size_t)reps) { hipEvent_t e; HIP_TRY(h, hipEventCreate(&e)); h->ev_step.push_back(e); }
    }
    h->last_paths_ms = 0.0f; h->last_paths_n = 0;
    if (h->resident_x[0]) {
        // software pipeline over steps: path kernel of step r on `stream`, velocity kernels of step r on `vel_stream[r % VEL_STREAMS]` (alternating:
        // the chains of consecutive steps run next to each other), PIPE_SETS buffer sets; the path kernel of step r + PIPE_SETS waits
        // until the velocity kernels of step r released its set
        constexpr int K = ltpl_handle::PIPE_SETS;
        const bool two = h->path_streams == 2 && h->path_stream2;
        if (two) {      // the second path stream starts behind everything queued on `stream` so far
            HIP_TRY(h, hipEventRecord(h->ev_begin, h->stream));
            HIP_TRY(h, hipStreamWaitEvent(h->path_stream2, h->ev_begin, 0));
        }
        for (int r = 0; r < reps; ++r) {
            const int set = r % K;
            const TickLayout& T = set ? *h->resident_x[set - 1] : *h->resident;
            hipStream_t sp = (two && (r & 1)) ? h->path_stream2 : h->stream;
            if (r >= K) HIP_TRY(h, hipStreamWaitEvent(sp, h->ev_vel[set], 0));
            if (ms_total) HIP_TRY(h, hipEventRecord(h->ev_step[2 * (size_t)r], sp));
            int rc = tick_launch_paths(h, T, sp);
            if (rc) return rc;
            if (ms_total) HIP_TRY(h, hipEventRecord(h->ev_step[2 * (size_t)r + 1], sp));
            HIP_TRY(h, hipEventRecord(h->ev_paths[set], sp));
            hipStream_t sv = h->vel_stream[r % ltpl_handle::VEL_STREAMS];
            HIP_TRY(h, hipStreamWaitEvent(sv, h->ev_paths[set], 0));
            if ((rc = tick_launch_vel(h, T, sv))) return rc;
            HIP_TRY(h, hipEventRecord(h->ev_vel[set], sv));
            h->last_set = set;
        }
        // everything joins `stream` again (the caller's events / synchronisation are on it)
        for (int i = 0; i < K && i < reps; ++i) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_vel[i], 0));
    } else {
        for (int r = 0; r < reps; ++r) { int rc = tick_launch(h, *h->resident); if (rc) return rc; }
    }
    if (ms_total) {
        HIP_TRY(h, hipEventRecord(e1, h->stream));
        HIP_TRY(h, hipEventSynchronize(e1));
        HIP_TRY(h, hipEventElapsedTime(ms_total, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (h->resident_x[0]) {
            for (int r = 0; r < reps; ++r) {
                float ms = 0.0f;
                HIP_TRY(h, hipEventElapsedTime(&ms, h->ev_step[2 * (size_t)r], h->ev_step[2 * (size_t)r + 1]));
                h->last_paths_ms += ms;
            }
            h->last_paths_n = reps;
        }
    } else {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_batch_last_paths_ms(ltpl_handle* h, float* ms_avg)
{
    if (!h || !ms_avg) return LTPL_ERR_INVALID_ARG;
    *ms_avg = h->last_paths_n > 0 ? h->last_paths_ms / (float)h->last_paths_n : 0.0f;
    return LTPL_OK;
}

extern "C" int ltpl_batch_run_profile(ltpl_handle* h, int reps, float* ms_kernels)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!h->resident) { h->err = "no resident batch: call ltpl_batch_upload first"; return LTPL_ERR_INVALID_ARG; }
    if (reps < 1 || !ms_kernels) { h->err = "reps < 1 or null output"; return LTPL_ERR_INVALID_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) HIP_TRY(h, hipEventCreate(&ev[i]));
    ms_kernels[0] = ms_kernels[1] = ms_kernels[2] = 0.0f;
    for (int r = 0; r < reps; ++r) {
        int rc = tick_launch(h, *h->resident, ev);
        if (rc) return rc;
        HIP_TRY(h, hipEventSynchronize(ev[3]));
        for (int k = 0; k < 3; ++k) { float ms = 0.0f; HIP_TRY(h, hipEventElapsedTime(&ms, ev[k], ev[k + 1])); ms_kernels[k] += ms; }
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_batch_download(ltpl_handle* h, ltpl_paths_out* out, ltpl_tick_vel_out* vout)
try {
    if (!h) return LTPL_ERR_INVALID_ARG;
    if (!h->resident) { h->err = "no resident batch"; return LTPL_ERR_INVALID_ARG; }
    if (!out || !vout || out->cap_nodes != h->resident->cap_nodes || out->cap_pts != h->resident->cap_pts) {
        h->err = "output capacities differ from ltpl_batch_upload"; return LTPL_ERR_INVALID_ARG;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const void* src = (h->resident_x[0] && h->last_set >= 1) ? h->d_out_x[h->last_set - 1] : h->d_out;
    HIP_TRY(h, hipMemcpyAsync(h->h_out, src, h->resident->out_total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    tick_scatter(static_cast<const unsigned char*>(h->h_out), *h->resident, out, vout);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

static void free_resident(TickLayout* t) { delete t; }

// ---------------------------------------------------------------------------------------------------------------------
// offline lattice build: every candidate edge of a track in one launch, lane = edge
// ---------------------------------------------------------------------------------------------------------------------
struct DevOfflineIn {
    int n_edges, cap; double step, kmax_turn;
    const double* sx; const double* sy; const double* spsi; const double* ex; const double* ey; const double* epsi;
    const double* kmax_vel; const int* rl_edge; const double* given;
};
struct DevOfflineOut { int* n_samples; int* valid; double* coeff; double* length; double* kavg; double* krange; double* samples; };

__global__ __launch_bounds__(64) void k_offline_edges(DevOfflineIn in, DevOfflineOut out)
{
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e >= in.n_edges) return;
    double cx[4], cy[4];
    if (in.rl_edge[e]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { cx[k] = in.given[(size_t)e * 8 + k]; cy[k] = in.given[(size_t)e * 8 + 4 + k]; }
    } else {
        // tph.calc_splines on two points with heading constraints (gen_edges.py:80-84): cubic Hermite segment, end slopes scaled
        // by the chord length (psi = 0 is north, hence + pi / 2)
        const double x0 = in.sx[e], y0 = in.sy[e], x1 = in.ex[e], y1 = in.ey[e];
        const double el = sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0));
        const double tx0 = cos(in.spsi[e] + D_PI / 2) * el, ty0 = sin(in.spsi[e] + D_PI / 2) * el;
        const double tx1 = cos(in.epsi[e] + D_PI / 2) * el, ty1 = sin(in.epsi[e] + D_PI / 2) * el;
        cx[0] = x0; cx[1] = tx0; cx[2] = 3.0 * (x1 - x0) - 2.0 * tx0 - tx1; cx[3] = -2.0 * (x1 - x0) + tx0 + tx1;
        cy[0] = y0; cy[1] = ty0; cy[2] = 3.0 * (y1 - y0) - 2.0 * ty0 - ty1; cy[3] = -2.0 * (y1 - y0) + ty0 + ty1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { out.coeff[(size_t)e * 8 + k] = cx[k]; out.coeff[(size_t)e * 8 + 4 + k] = cy[k]; }
    auto px = [&](double t) { return cx[0] + cx[1] * t + cx[2] * (t * t) + cx[3] * (t * t * t); };
    auto py = [&](double t) { return cy[0] + cy[1] * t + cy[2] * (t * t) + cy[3] * (t * t * t); };
    // tph.calc_spline_lengths: polyline over 15 uniform-t points
    double len = 0.0;
    {
        double lx = px(0.0), ly = py(0.0);
        for (int i = 1; i < 15; ++i) {
            const double t = i == 14 ? 1.0 : (double)i * (1.0 / 14.0);
            const double qx = px(t), qy = py(t);
            len += sqrt((qx - lx) * (qx - lx) + (qy - ly) * (qy - ly));
            lx = qx; ly = qy;
        }
    }
    const int n = (int)ceil(len / in.step) + 1;                  // tph.interp_splines(stepsize_approx, incl_last_point=True)
    out.n_samples[e] = n;
    if (n > in.cap || n < 2) { out.valid[e] = 0; out.length[e] = 0.0; out.kavg[e] = 0.0; out.krange[e] = 0.0; return; }
    double* S = out.samples + (size_t)e * in.cap * 5;
    const double lin_step = len / (double)(n - 1);
    const double kmax_v = in.kmax_vel[e], kmax_t = in.kmax_turn;
    bool ok = true; double sum_abs = 0.0, kmin = INFINITY, kmax = -INFINITY, slen = 0.0, lx = 0.0, ly = 0.0;
    for (int i = 0; i < n; ++i) {
        const bool last = i == n - 1;
        const double t = last ? 1.0 : ((double)i * lin_step) / len;
        double x = px(t), y = py(t);
        if (last) { x = ((cx[0] + cx[1]) + cx[2]) + cx[3]; y = ((cy[0] + cy[1]) + cy[2]) + cy[3]; }
        const double xd = cx[1] + 2.0 * cx[2] * t + 3.0 * cx[3] * (t * t), yd = cy[1] + 2.0 * cy[2] * t + 3.0 * cy[3] * (t * t);
        const double xdd = 2.0 * cx[2] + 6.0 * cx[3] * t, ydd = 2.0 * cy[2] + 6.0 * cy[3] * t;
        const double psi = normalize_psi_dev(atan2(yd, xd) - D_PI / 2);
        const double q = xd * xd + yd * yd;
        const double kap = (xd * ydd - yd * xdd) / (q * sqrt(q));
        double* r = S + (size_t)i * 5;
        r[0] = x; r[1] = y; r[2] = psi; r[3] = kap; r[4] = 0.0;
        if (i > 0) { const double el = sqrt((x - lx) * (x - lx) + (y - ly) * (y - ly)); S[(size_t)(i - 1) * 5 + 4] = el; slen += el; }
        lx = x; ly = y;
        const double ka = fabs(kap);
        ok = ok && ka <= kmax_t && ka <= kmax_v;
        sum_abs += ka; kmin = kap < kmin ? kap : kmin; kmax = kap > kmax ? kap : kmax;
    }
    out.valid[e] = (ok || in.rl_edge[e]) ? 1 : 0;
    out.length[e] = slen; out.kavg[e] = sum_abs / (double)n; out.krange[e] = fabs(kmax - kmin);
}

extern "C" int ltpl_offline_edges(int device, const ltpl_offline_edges_in* in, ltpl_offline_edges_out* out)
try {
    g_create_error.clear();
    if (!in || !out || in->n_edges < 1 || in->cap_samples < 2 || !(in->stepsize_approx > 0.0)) { g_create_error = "offline edges: invalid argument"; return LTPL_ERR_INVALID_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "no HIP device visible"; return LTPL_ERR_NO_DEVICE; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev || hipSetDevice(device) != hipSuccess) { g_create_error = "device index out of range"; return LTPL_ERR_NO_DEVICE; }
    const size_t n = (size_t)in->n_edges, cap = (size_t)in->cap_samples;
    Arena ai, ao;
    const size_t o_sx = ai.add(8 * n), o_sy = ai.add(8 * n), o_sp = ai.add(8 * n), o_ex = ai.add(8 * n), o_ey = ai.add(8 * n), o_ep = ai.add(8 * n),
                 o_kv = ai.add(8 * n), o_rl = ai.add(4 * n), o_gc = ai.add(64 * n);
    const size_t q_ns = ao.add(4 * n), q_va = ao.add(4 * n), q_co = ao.add(64 * n), q_le = ao.add(8 * n), q_ka = ao.add(8 * n), q_kr = ao.add(8 * n),
                 q_sa = ao.add(8 * 5 * cap * n);
    unsigned char *di = nullptr, *dobuf = nullptr;
    auto fail = [&](const char* why) { g_create_error = why; if (di) (void)hipFree(di); if (dobuf) (void)hipFree(dobuf); return LTPL_ERR_HIP; };
    if (hipMalloc(reinterpret_cast<void**>(&di), ai.size) != hipSuccess) return fail("offline edges: hipMalloc failed");
    if (hipMalloc(reinterpret_cast<void**>(&dobuf), ao.size) != hipSuccess) return fail("offline edges: hipMalloc failed");
    auto up = [&](size_t off, const void* src, size_t bytes) { return hipMemcpy(di + off, src, bytes, hipMemcpyHostToDevice) == hipSuccess; };
    if (!up(o_sx, in->start_x, 8 * n) || !up(o_sy, in->start_y, 8 * n) || !up(o_sp, in->start_psi, 8 * n) || !up(o_ex, in->end_x, 8 * n) ||
        !up(o_ey, in->end_y, 8 * n) || !up(o_ep, in->end_psi, 8 * n) || !up(o_kv, in->kappa_max_vel, 8 * n) || !up(o_rl, in->raceline_edge, 4 * n) ||
        !up(o_gc, in->given_coeff, 64 * n)) return fail("offline edges: upload failed");
    DevOfflineIn d; d.n_edges = in->n_edges; d.cap = in->cap_samples; d.step = in->stepsize_approx; d.kmax_turn = in->kappa_max_turn;
    d.sx = reinterpret_cast<double*>(di + o_sx); d.sy = reinterpret_cast<double*>(di + o_sy); d.spsi = reinterpret_cast<double*>(di + o_sp);
    d.ex = reinterpret_cast<double*>(di + o_ex); d.ey = reinterpret_cast<double*>(di + o_ey); d.epsi = reinterpret_cast<double*>(di + o_ep);
    d.kmax_vel = reinterpret_cast<double*>(di + o_kv); d.rl_edge = reinterpret_cast<int*>(di + o_rl); d.given = reinterpret_cast<double*>(di + o_gc);
    DevOfflineOut o; o.n_samples = reinterpret_cast<int*>(dobuf + q_ns); o.valid = reinterpret_cast<int*>(dobuf + q_va);
    o.coeff = reinterpret_cast<double*>(dobuf + q_co); o.length = reinterpret_cast<double*>(dobuf + q_le); o.kavg = reinterpret_cast<double*>(dobuf + q_ka);
    o.krange = reinterpret_cast<double*>(dobuf + q_kr); o.samples = reinterpret_cast<double*>(dobuf + q_sa);
    hipLaunchKernelGGL(k_offline_edges, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, d, o);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return fail("offline edges: kernel failed");
    auto down = [&](void* dst, size_t off, size_t bytes) { return hipMemcpy(dst, dobuf + off, bytes, hipMemcpyDeviceToHost) == hipSuccess; };
    if (!down(out->n_samples, q_ns, 4 * n) || !down(out->valid, q_va, 4 * n) || !down(out->coeff, q_co, 64 * n) || !down(out->length, q_le, 8 * n) ||
        !down(out->kappa_avg, q_ka, 8 * n) || !down(out->kappa_range, q_kr, 8 * n) || !down(out->samples, q_sa, 8 * 5 * cap * n)) return fail("offline edges: download failed");
    (void)hipFree(di); (void)hipFree(dobuf);
    return LTPL_OK;
} LTPL_ABI_CATCH(nullptr)

// ---------------------------------------------------------------------------------------------------------------------
// self-test at ltpl_create: the one-wave batch kernel and the four-wave latency kernel are two differently scheduled builds
// of the same source (different register budgets, LDS plans and synchronisation); 64 probe scenarios derived from the
// lattice itself (an obstacle on the race line four layers ahead: mask, three filters, tie-break rounds) must come out
// IDENTICAL from both, otherwise the handle is refused. Guards against a toolchain that miscompiles one of them
//  for lattices no parity test has seen. LTPL_NO_SELFTEST=1 skips it.
// ---------------------------------------------------------------------------------------------------------------------
// the cross-lane helpers of the kernels (reductions, scans, lexicographic minima: DPP row operations + v_permlane16/32_swap) against plain loops,
// on 16 waves of values with many ties -- one launch of k_exp_wave_ops at create time
static int wave_ops_selftest(ltpl_handle* h)
{
    const int rounds = 16;
    std::vector<double> vals((size_t)rounds * 128); std::vector<int> ivals((size_t)rounds * 64);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { x = x * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(x >> 33); };
    for (size_t i = 0; i < vals.size(); ++i) vals[i] = (double)(next() % 23u) * 0.25 - 2.0;          // few distinct values: ties in every reduction
    for (size_t i = 0; i < ivals.size(); ++i) ivals[i] = (int)(next() % 200u) - 60;
    double* dv = nullptr; int* di = nullptr; int* de = nullptr;
    struct Guard { void** p[3]; ~Guard() { for (void** q : p) if (*q) (void)hipFree(*q); } } guard{{reinterpret_cast<void**>(&dv), reinterpret_cast<void**>(&di), reinterpret_cast<void**>(&de)}};
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&dv), sizeof(double) * vals.size()));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&di), sizeof(int) * ivals.size()));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&de), sizeof(int)));
    HIP_TRY(h, hipMemcpy(dv, vals.data(), sizeof(double) * vals.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(di, ivals.data(), sizeof(int) * ivals.size(), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemset(de, 0, sizeof(int)));
    hipLaunchKernelGGL(k_exp_wave_ops, dim3(1), dim3(64), 0, h->stream, dv, di, rounds, de);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    int bad = 0;
    HIP_TRY(h, hipMemcpy(&bad, de, sizeof(int), hipMemcpyDeviceToHost));
    if (bad) { h->err = "self-test failed: the wave reductions / scans of the kernels disagree with plain loops on this device"; return LTPL_ERR_UNSUPPORTED; }
    return LTPL_OK;
}

static int self_test(ltpl_handle* h, const ltpl_lattice_desc* d)
{
    {
        const int rc = wave_ops_selftest(h);
        if (rc) return rc;
    }
    const int n = 64, L = d->num_layers, A = LTPL_MAX_ACTIONS;
    std::vector<int> sl(n), sn(n), flags(n, LTPL_FLAG_ACTION_SETS), la(n, LTPL_ACT_NONE), cc(n, -1), veh_off(n + 1), pos_off(n + 1),
        zone_off(n + 1, 0), zone(1, 0), n_last(n, 0), ll((size_t)n * LTPL_MAX_LAST_NODES, -1), ln((size_t)n * LTPL_MAX_LAST_NODES, -1);
    std::vector<double> psi(n, 0.0), rad(n, 2.5), px(n), py(n), w(1, 0.0);
    for (int i = 0; i < n; ++i) {
        sl[i] = (int)(((long long)i * L) / n);
        while (h->rng_end_host[(size_t)sl[i]] < 0 && sl[i] > 0) --sl[i];     // open tracks: the last layer(s) have no planning range
        sn[i] = d->raceline_index[sl[i]];
        const int ol = (sl[i] + 4) % L, g = d->layer_node_off[ol] + d->raceline_index[ol];
        px[i] = d->node_x[g]; py[i] = d->node_y[g];
        veh_off[i] = i; pos_off[i] = i;
        if (i % 4 == 3) flags[i] |= LTPL_FLAG_OBJ_BESIDES;            // constant-segment template: follow + left + right on `default`
    }
    veh_off[n] = n; pos_off[n] = n;
    ltpl_paths_in in; memset(&in, 0, sizeof(in));
    in.n_scen = n; in.n_w_last = 0; in.w_last_edges = w.data(); in.start_layer = sl.data(); in.start_node = sn.data();
    in.flags = flags.data(); in.last_action = la.data(); in.const_closest = cc.data(); in.psi_s = psi.data();
    in.veh_off = veh_off.data(); in.pos_off = pos_off.data(); in.veh_radius = rad.data(); in.pos_x = px.data(); in.pos_y = py.data();
    in.zone_off = zone_off.data(); in.zone_gid = zone.data(); in.n_last = n_last.data(); in.last_layer = ll.data(); in.last_node = ln.data();
    const int cn = h->caps.max_path_nodes, cp = h->caps.max_path_pts;
    struct Out {
        std::vector<int> end_layer, coi, con, n_actions, action_id, valid, reduced, goal_layer, n_nodes, n_pts, n_ties, nodes, node_idx;
        std::vector<double> coeff, pp; ltpl_paths_out o;
    };
    auto make = [&](Out& O) {
        O.end_layer.assign(n, 0); O.coi.assign(n, 0); O.con.assign((size_t)n * 2, 0); O.n_actions.assign(n, 0);
        for (auto* v : {&O.action_id, &O.valid, &O.reduced, &O.goal_layer, &O.n_nodes, &O.n_pts, &O.n_ties}) v->assign((size_t)n * A, 0);
        O.nodes.assign((size_t)n * A * cn, 0); O.node_idx.assign((size_t)n * A * cn, 0);
        O.coeff.assign((size_t)n * A * cn * 8, 0.0); O.pp.assign((size_t)n * A * cp * 5, 0.0);
        memset(&O.o, 0, sizeof(O.o));
        O.o.cap_nodes = cn; O.o.cap_pts = cp; O.o.end_layer = O.end_layer.data(); O.o.closest_obj_index = O.coi.data();
        O.o.closest_obj_node = O.con.data(); O.o.n_actions = O.n_actions.data(); O.o.action_id = O.action_id.data();
        O.o.valid = O.valid.data(); O.o.reduced = O.reduced.data(); O.o.goal_layer = O.goal_layer.data(); O.o.n_nodes = O.n_nodes.data();
        O.o.n_pts = O.n_pts.data(); O.o.n_ties = O.n_ties.data(); O.o.nodes = O.nodes.data(); O.o.node_idx = O.node_idx.data();
        O.o.coeff = O.coeff.data(); O.o.path_param = O.pp.data();
    };
    Out a, b; make(a); make(b);
    int rc = plan_paths_impl(h, &in, &a.o, 1);
    if (rc) return rc;
    if ((rc = plan_paths_impl(h, &in, &b.o, NUM_WAVES))) return rc;
    bool same = a.n_actions == b.n_actions && a.action_id == b.action_id && a.valid == b.valid && a.reduced == b.reduced &&
                a.n_nodes == b.n_nodes && a.n_pts == b.n_pts && a.n_ties == b.n_ties && a.coi == b.coi;
    int n_paths = 0;
    for (size_t slot = 0; same && slot < (size_t)n * A; ++slot) {
        if (!a.valid[slot]) continue;
        ++n_paths;
        for (int i = 0; i < a.n_nodes[slot]; ++i) same = same && a.nodes[slot * cn + i] == b.nodes[slot * cn + i];
    }
    if (!same) { h->err = "self-test failed: the one-wave batch kernel and the four-wave kernel disagree on the probe scenarios"; return LTPL_ERR_HIP; }
    if (n_paths == 0) { h->err = "self-test failed: no path found on any probe scenario"; return LTPL_ERR_HIP; }
    return LTPL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// planner (ABI v3): the OnlineTrajectoryHandler state machine of planner_core.hpp on top of the kernels above
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct HipCompute : ltplp::Compute {
    ltpl_handle* h;
    // a planner holds on to its lattice handle: the handle counts its planners and ltpl_destroy refuses while one is alive
    explicit HipCompute(ltpl_handle* handle) : h(handle) { ++h->n_planners; }
    ~HipCompute() override { --h->n_planners; }
    int plan_paths(const ltpl_paths_in* in, ltpl_paths_out* out) override { return ltpl_plan_paths(h, in, out); }
    int vel_profile(const ltpl_vel_params* p, int n, const ltpl_vel_job* jobs, ltpl_vel_result* res) override
    {
        return ltpl_vel_profile(h, p, n, jobs, res);
    }
    const char* last_error() override { return h->err.c_str(); }
};
}

extern "C" int ltpl_const_segment_test(const ltpl_handle* h, const double* seg, int32_t n_rows, const double* pos_est, int32_t n_veh,
                                       const double* veh_x, const double* veh_y, const double* veh_radius, int32_t* flags_out,
                                       int32_t* closest_out)
try {
    if (!h || !flags_out || !closest_out || n_veh < 0 || n_rows < 0) return LTPL_ERR_INVALID_ARG;
    if (!h->has_hostlat) return LTPL_ERR_UNSUPPORTED;
    int in_const, besides, closest;
    ltplp::const_segment_test(h->hostlat, n_rows > 0 ? seg : nullptr, n_rows, pos_est, n_veh, veh_x, veh_y, veh_radius, &in_const, &besides, &closest);
    *flags_out = (in_const ? LTPL_FLAG_OBJ_IN_CONST : 0) | (besides ? LTPL_FLAG_OBJ_BESIDES : 0);
    *closest_out = closest;
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_edge_capsules(int32_t n_edges, const int32_t* samp_ptr, const double* samp_x, const double* samp_y, int32_t n_samples,
                                  float* capsules_out, float* slack_out)
try {
    if (n_edges < 0 || !samp_ptr || !samp_x || !samp_y || !capsules_out || !slack_out) return LTPL_ERR_INVALID_ARG;
    *slack_out = ltplcap::build(n_edges, samp_ptr, samp_x, samp_y, n_samples, capsules_out);
    return LTPL_OK;
} LTPL_ABI_CATCH(nullptr)

extern "C" int ltpl_layer_grid(int32_t n_layers, const double* ref_x, const double* ref_y, double* origin_cell, int32_t* dims, int32_t* cells,
                               int32_t cap_cells)
try {
    if (n_layers < 1 || !ref_x || !ref_y || !origin_cell || !dims) return LTPL_ERR_INVALID_ARG;
    const ltplgrid::Grid g = ltplgrid::build(n_layers, ref_x, ref_y);
    origin_cell[0] = g.x0; origin_cell[1] = g.y0; origin_cell[2] = g.inv_cell;
    dims[0] = g.nx; dims[1] = g.ny;
    if (cells) {
        if ((size_t)cap_cells < (size_t)g.nx * g.ny) return LTPL_ERR_CAPACITY;
        memcpy(cells, g.cells.data(), sizeof(int32_t) * g.cells.size());
    }
    return LTPL_OK;
} LTPL_ABI_CATCH(nullptr)

extern "C" int ltpl_assembly_records(int32_t n_nodes, int32_t n_edges, const int32_t* in_ptr, const int32_t* edge_src, const double* edge_len,
                                     const int32_t* samp_ptr, const double* samp_x, const double* samp_y, const double* samp_psi,
                                     int32_t* node_rec_out, double* edge_rec_out)
try {
    if (n_nodes < 1 || n_edges < 1 || !in_ptr || !edge_src || !edge_len || !samp_ptr || !samp_x || !samp_y || !samp_psi || !node_rec_out ||
        !edge_rec_out) return LTPL_ERR_INVALID_ARG;
    std::vector<int32_t> nrec; std::vector<double> erec;
    ltplrec::build(n_nodes, n_edges, in_ptr, edge_src, edge_len, samp_ptr, samp_x, samp_y, samp_psi, nrec, erec);
    memcpy(node_rec_out, nrec.data(), sizeof(int32_t) * nrec.size());
    memcpy(edge_rec_out, erec.data(), sizeof(double) * erec.size());
    return LTPL_OK;
} LTPL_ABI_CATCH(nullptr)

extern "C" int ltpl_raceline_s(const ltpl_handle* h, double x, double y, double* s_out)
try {
    if (!h || !s_out) return LTPL_ERR_INVALID_ARG;
    if (!h->has_hostlat) return LTPL_ERR_UNSUPPORTED;
    *s_out = ltplp::raceline_s(h->hostlat, x, y);
    return LTPL_OK;
} LTPL_ABI_CATCH(abi_err_of(h))

extern "C" int ltpl_planner_create(ltpl_handle* h, const ltpl_planner_config* cfg, ltpl_planner** out)
try {
    g_create_error.clear();
    if (!h) { g_create_error = "planner: null handle"; return LTPL_ERR_INVALID_ARG; }
    if (!h->has_hostlat) { g_create_error = "planner: the lattice was created without raceline_x / raceline_y / node_psi"; return LTPL_ERR_UNSUPPORTED; }
    return ltplp::api_create(new HipCompute(h), h->hostlat, cfg, out, &g_create_error);
} LTPL_ABI_CATCH(nullptr)
extern "C" int ltpl_planner_destroy(ltpl_planner* p) { delete p; return LTPL_OK; }
extern "C" int ltpl_planner_get_caps(const ltpl_planner* p, ltpl_planner_caps* c) try { return ltplp::api_get_caps(p, c); } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" const char* ltpl_planner_last_error(const ltpl_planner* p) { return p ? p->P.err.c_str() : g_create_error.c_str(); }
extern "C" int ltpl_planner_set_start(ltpl_planner* p, int32_t scen, double x, double y, double heading, double vel,
                                      double max_heading_offset, int32_t* in_track, int32_t* cor_heading)
try {
    if (!p || !in_track || !cor_heading) return LTPL_ERR_INVALID_ARG;
    return p->P.set_start(scen, x, y, heading, vel, max_heading_offset, in_track, cor_heading);
} LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_calc_paths(ltpl_planner* p, const ltpl_planner_paths_in* in) try { return ltplp::api_calc_paths(p, in); } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_calc_paths_begin(ltpl_planner* p, const ltpl_planner_paths_in* in) try { return ltplp::api_calc_paths_begin(p, in); } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_calc_paths_finish(ltpl_planner* p, const int32_t* zo, const int32_t* zg) try { return ltplp::api_calc_paths_finish(p, zo, zg); } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_get_ref_idx(ltpl_planner* p, const double* px, const double* py) try { return (p && px && py) ? p->P.get_ref_idx(px, py) : LTPL_ERR_INVALID_ARG; } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_calc_vel_profile(ltpl_planner* p, const ltpl_planner_vel_in* in) try { return ltplp::api_calc_vel_profile(p, in); } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_get_paths(const ltpl_planner* p, int32_t scen, ltpl_planner_paths_view* v) try { return ltplp::api_get_paths(p, scen, v); } LTPL_ABI_CATCH(abi_err_of(p))
extern "C" int ltpl_planner_get_trajectories(const ltpl_planner* p, int32_t scen, ltpl_planner_traj_view* v) try { return ltplp::api_get_trajectories(p, scen, v); } LTPL_ABI_CATCH(abi_err_of(p))

#ifdef LTPL_EXPERIMENT
// experiment build only: the cross-lane helpers against plain loops on `rounds` waves of caller-provided values (vals: rounds x 128
// doubles, ivals: rounds x 64 ints, host memory); *n_bad = lanes that disagreed in some round
extern "C" int ltpl_exp_wave_ops_check(int32_t device, const double* vals, const int32_t* ivals, int32_t rounds, int32_t* n_bad)
try {
    if (!vals || !ivals || !n_bad || rounds <= 0) return LTPL_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return LTPL_ERR_HIP;
    double* dv = nullptr; int* di = nullptr; int* de = nullptr;
    int rc = LTPL_OK, zero = 0;
    if (hipMalloc(&dv, sizeof(double) * 128 * (size_t)rounds) != hipSuccess || hipMalloc(&di, sizeof(int) * 64 * (size_t)rounds) != hipSuccess ||
        hipMalloc(&de, sizeof(int)) != hipSuccess) rc = LTPL_ERR_HIP;
    if (!rc && (hipMemcpy(dv, vals, sizeof(double) * 128 * (size_t)rounds, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(di, ivals, sizeof(int) * 64 * (size_t)rounds, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(de, &zero, sizeof(int), hipMemcpyHostToDevice) != hipSuccess)) rc = LTPL_ERR_HIP;
    if (!rc) {
        hipLaunchKernelGGL(k_exp_wave_ops, dim3(1), dim3(64), 0, 0, dv, di, rounds, de);
        if (hipGetLastError() != hipSuccess || hipMemcpy(n_bad, de, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = LTPL_ERR_HIP;
    }
    (void)hipFree(dv); (void)hipFree(di); (void)hipFree(de);
    return rc;
} LTPL_ABI_CATCH(nullptr)
#endif

#ifdef LTPL_EXPERIMENT
// experiment build only: heading_atan2 (path re-sampling) for n caller-provided (y, x) pairs (host memory)
__global__ void k_exp_heading(const double* y, const double* x, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = heading_atan2(y[i], x[i]);
}
__global__ void k_exp_sincos(const double* x, double* sn, double* cs, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) heading_sincos(x[i], &sn[i], &cs[i]);
}
// experiment build only: heading_sincos (start heading of a scenario) for n caller-provided angles (host memory)
extern "C" int ltpl_exp_heading_sincos(int32_t device, const double* x, double* sn, double* cs, int32_t n)
try {
    if (!x || !sn || !cs || n <= 0) return LTPL_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return LTPL_ERR_HIP;
    double* d = nullptr;
    int rc = LTPL_OK;
    const size_t b = sizeof(double) * (size_t)n;
    if (hipMalloc(&d, 3 * b) != hipSuccess) return LTPL_ERR_HIP;
    if (hipMemcpy(d, x, b, hipMemcpyHostToDevice) != hipSuccess) rc = LTPL_ERR_HIP;
    if (!rc) {
        hipLaunchKernelGGL(k_exp_sincos, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(sn, d + n, b, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(cs, d + 2 * (size_t)n, b, hipMemcpyDeviceToHost) != hipSuccess) rc = LTPL_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
} LTPL_ABI_CATCH(nullptr)
extern "C" int ltpl_exp_heading_atan2(int32_t device, const double* y, const double* x, double* out, int32_t n)
try {
    if (!y || !x || !out || n <= 0) return LTPL_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return LTPL_ERR_HIP;
    double* d = nullptr;
    int rc = LTPL_OK;
    const size_t b = sizeof(double) * (size_t)n;
    if (hipMalloc(&d, 3 * b) != hipSuccess) return LTPL_ERR_HIP;
    if (hipMemcpy(d, y, b, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d + n, x, b, hipMemcpyHostToDevice) != hipSuccess) rc = LTPL_ERR_HIP;
    if (!rc) {
        hipLaunchKernelGGL(k_exp_heading, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, n);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d + 2 * (size_t)n, b, hipMemcpyDeviceToHost) != hipSuccess) rc = LTPL_ERR_HIP;
    }
    (void)hipFree(d);
    return rc;
} LTPL_ABI_CATCH(nullptr)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// fleet (ABI v5): planners with device-resident state
// ---------------------------------------------------------------------------------------------------------------------
#include "fleet_dev.hpp"
