// paths_team.hpp -- seam (1) on the device: a TEAM of NW wave64s plans one scenario (= one call of the reference's
// main_online_path_gen, graph_ltpl/online_graph/src/main_online_path_gen.py:11-334). Included by ltpl_hip.hip.
//
//   NW = 1  throughput form: one wave per scenario, ~10 KB of LDS, up to 16 scenarios resident per CU; every latency
//           chain of one scenario is covered by the other resident scenarios.
//   NW = 4  latency form: four waves per scenario (single-tick path, fused with the velocity stage in k_tick).
//
// Phases (reference rows of SURVEY.md section 8a in brackets):
//   1  closest reference-line layer per obstacle position, lexicographic wave min-reduction              [M1]
//   2  obstacle x edge-sample collision mask. Transition-major: the sample arrays of a layer transition are
//      streamed ONCE (coalesced, unrolled for memory-level parallelism) and tested against every obstacle position
//      whose 3-layer window contains the transition; result = one bit per horizon edge in LDS              [M2]
//   3  closest object / node and action-template choice (uniform, computed redundantly per wave)        [M3, T1]
//   4  layered min-plus sweeps. Per layer transition the edges (cost, source) are staged ONCE into LDS by
//      coalesced loads and shared by all filters; blocked edges are marked in the sign bit of the staged cost.
//      Lane = destination node, private min over its in-edges (CSC => no atomics); filters are node / edge
//      predicates (planning_range, default, overtake_left, overtake_right), not graph copies. overtake_left /
//      overtake_right equal `default` in front of the object layer, so they branch off the `default` sweep there.
//      Goal nodes are evaluated lazily: only for the layer a path actually ends in                        [F1, S1]
//   5  horizon back-off / reduced-horizon logic on the per-layer reachability table                         [S2]
//   6  backtrack, gather, tridiagonal C2 spline solve, re-sampling, heading / curvature                 [G1, P1-P3]
#pragma once

// 16-byte stores of two doubles; the _u form only promises 8-byte alignment (rows of 5 doubles)
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef double dbl2_u __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void store2(double* p, double a, double b) { dbl2 v; v.x = a; v.y = b; *reinterpret_cast<dbl2*>(p) = v; }
__device__ __forceinline__ void store2_u(double* p, double a, double b) { dbl2_u v; v.x = a; v.y = b; *reinterpret_cast<dbl2_u*>(p) = v; }

// 1 / b with the hardware reciprocal and two Newton steps (~1 ulp) instead of the IEEE division sequence (~25 instructions):
// used where the result only enters spline coefficients / samples (1e-5 tolerance), never where indices are derived
__device__ __forceinline__ double fast_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}

// sin and cos of a heading (round 5: the start heading of a scenario's constant path segment; the library's sincos is ~185 vector
// instructions with its large-argument reduction): k = rint(x 2 / pi), r = x - k pi / 2 in two steps, the fdlibm kernel polynomials on
// |r| <= pi / 4, quadrant by k. Error <= 2.3e-16 for |x| <= 4 pi (checked in tests/test_gpu_wave_ops.py); the caller falls back to sincos beyond.
__device__ __forceinline__ void heading_sincos(double x, double* sn, double* cs)
{
    const double k = rint(x * 0.63661977236758138);
    double r = fma(-k, 1.5707963267948966, x);
    r = fma(-k, 6.123233995736766e-17, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06); ps = fma(z, ps, -1.98412698298579493134e-04); ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07); pc = fma(z, pc, 2.48015872894767294178e-05); pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double a = (q & 1) ? c : s, b = (q & 1) ? s : c;           // sin <- (s, c, -s, -c)[q], cos <- (c, -s, -c, s)[q]
    *sn = (q & 2) ? -a : a;
    *cs = ((q + 1) & 2) ? -b : b;
}

// q^(-3/2), q > 0 (curvature of a path sample: cross / |tangent|^3): hardware reciprocal square root + two Newton steps, cubed -- 10 vector
// instructions (round 5); q * sqrt(q) through the library's correctly rounded sqrt and a reciprocal was ~26. Relative error a few 1e-16.
__device__ __forceinline__ double rsqrt_cubed(double q)
{
    double r = __builtin_amdgcn_rsq(q);
    const double h = 0.5 * q;
    r = r * fma(-(h * r), r, 1.5);
    r = r * fma(-(h * r), r, 1.5);
    return (r * r) * r;
}

// atan2(y, x) for the heading of a path sample (round 5; the library routine is ~80 vector instructions per row block of the re-sampling):
// t = min(|x|, |y|) / max(|x|, |y|) is reduced by k pi / 8 (k = 0, 1, 2) WITHOUT a second division --
//   z = (t - c) / (1 + t c) = (mn - c mx) / (mx + c mn),  c = tan(k pi / 8),  |z| <= tan(pi / 16) = 0.199
// -- and atan(z) is its Taylor series through z^17 (remainder < z^19 / 19 = 2.5e-15); then the octant, quadrant and sign. Agrees with
// atan2 to a few 1e-15 rad (tests/test_gpu_wave_ops.py checks it over all octants and on the axes); not for (0, 0).
__device__ __forceinline__ double heading_atan2(double y, double x)
{
    const double ay = fabs(y), ax = fabs(x);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool k1 = mn > 0.19891236737965800 * mx, k2 = mn > 0.66817863791929891 * mx;     // tan(pi / 16), tan(3 pi / 16)
    const double c = k2 ? 1.0 : (k1 ? 0.41421356237309503 : 0.0);                             // tan(k pi / 8)
    const double off = k2 ? 0.78539816339744831 : (k1 ? 0.39269908169872414 : 0.0);           // k pi / 8
    const double z = fma(-c, mx, mn) * fast_rcp(fma(c, mn, mx));
    const double w = z * z;
    double p = fma(w, 1.0 / 17.0, -1.0 / 15.0);
    p = fma(w, p, 1.0 / 13.0); p = fma(w, p, -1.0 / 11.0); p = fma(w, p, 1.0 / 9.0); p = fma(w, p, -1.0 / 7.0); p = fma(w, p, 1.0 / 5.0);
    p = fma(w, p, -1.0 / 3.0);
    double r = off + fma(z, w * p, z);
    if (ay > ax) r = 1.57079632679489662 - r;
    if (x < 0.0) r = D_PI - r;
    return copysign(r, y);
}

// (|kappa|, e) / (x, y) planes of the batch velocity stage: tiled by job and BLOCKED by rows -- element (job, row) lives at
//   ((job / 64 * plane_rows / KE_RB + row / KE_RB) * 64 + job % 64) * KE_RB + row % KE_RB           (plane_rows: a multiple of 8)
// so that the path kernel (lane = row of ONE job) writes KE_RB rows = KE_RB * 16 contiguous bytes per block instead of one element on each of
// 64 cache lines (measured in round 2: the scattered plane stores were 45 us of the 1.1 ms launch), while the velocity kernels (lane = job)
// still find the rows of a chunk in the lines they already touched. KE_RB = 8 with 16-byte records: one 128-byte line per (job, 8 rows);
// a row access of a velocity wave then touches 64 lines (KE_RB = 1: the 8 lines of a fully coalesced kilobyte).
#ifndef LTPL_KE_RB
#define LTPL_KE_RB 8
#endif
#define KE_RB LTPL_KE_RB
static_assert(KE_RB == 1 || KE_RB == 2 || KE_RB == 4 || KE_RB == 8, "rows per block of the operand planes");
// (32-bit element indices: a plane holds < 2^31 elements for any batch that fits the device)
__device__ __forceinline__ unsigned kep_base(int job, int plane_rows)
{
    return ((unsigned)(job >> 6) * ((unsigned)plane_rows / KE_RB) * 64u + (unsigned)(job & 63)) * KE_RB;
}
__device__ __forceinline__ unsigned kep_row(int r) { return ((unsigned)r / KE_RB) * (64u * KE_RB) + ((unsigned)r % KE_RB); }

// element i of a lattice array (0 <= i, array < 4 GB): the byte offset is formed in 32 bits, which lets the compiler address the
// element as scalar base + 32-bit lane offset instead of building a 64-bit address per lane
template <class T>
__device__ __forceinline__ const T& at(const T* p, int i)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + (unsigned)((unsigned)i * (unsigned)sizeof(T)));
}

struct TeamLds {                 // dynamic-LDS plan (byte offsets), computed once per lattice on the host
    int kpad, hmax, etmax;
    int words_blocked, words_zone;
    int off_pos_layer;           // short[MAX_POS]   closest reference-line layer per position, -1 = gated out
    int off_pos_veh;             // uchar[MAX_POS]   vehicle of a position
    int off_blocked;             // u32[words_blocked]
    int off_zone;                // u32[words_zone]
    int off_dist;                // double[NFILT][2][kpad]
    int off_par;                 // parent entries [NPAR][hmax][kpad], table = par_tab(filter); PlanRt: 2 bytes (source node, in-edge rank |
                                 // tie bit 0x80), PlanFx: 1 byte (source node | tie bit 0x80; the in-edge is looked up at path assembly)
    int off_best;                // int[NFILT][hmax]  -1 unreachable, -2 reachable (goal not evaluated), >= 0 goal node | tie << 30
    int off_cnt;                 // u32[NFILT][kpad]  number of in-edges that attain the minimum
    int off_widx;                // u32[NFILT][kpad]  election key (source node << 8 | in-edge rank, elect_key) of the first
    int off_dumin;               // double[NFILT][kpad]  smallest predecessor distance among tying edges (exact tie-break, rare)
    int off_lay;                 // int4[hmax]  per horizon layer j: first node id, #nodes, first / past-last edge INTO it
    int ref_lds;                 // 1: the reference line (x, y interleaved) is staged in the `par` region during phase 1
    int off_path;                // path scratch, `n_path_bufs` buffers of `path_stride` bytes
    int path_stride, n_path_bufs;
    int total;
    // long planning horizons (PlanRtG): the parent tables live in global memory, `par_glob_stride` bytes per scenario
    unsigned char* par_glob; long long par_glob_stride;
    // diagnostics (ltpl_plan_paths_mask; nullptr otherwise): the obstacle x edge mask of phase 2 as the kernel computed it, per scenario
    // `words_blocked + 2` words: first edge of the planning range (global edge id), layers of the planning range, then one bit per
    // edge of the planning range in edge-id order
    unsigned* mask_out;
    int off_shell, shell_cap;    // uint2[shell_cap] per wave: shell list of the obstacle mask (phase 2); fixed plans: in the frontier / election arrays
    // EXPERIMENT BUILD ONLY (-DLTPL_EXPERIMENT, libltpl_hip_exp.so; never read by the release library):
    int ablate;                  // LTPL_ABLATE, timing only: 1 = skip the mask, 2 = skip the sweeps, 4 = skip path assembly, 8 = launch cost only
    int poison_on; unsigned poison;   // LTPL_LDS_POISON=<hex word>: fill the team's LDS before phase 0, so that a read of LDS the
                                 // scenario has not written itself shows up as a parity failure instead of depending on stale data
    long long* dbg;              // LTPL_DEBUG_TIMING: cycle stamps per phase
};
// The timing / fault-injection switches exist in the experiment build only: the release library contains no code path that skips
// work or alters LDS on an environment variable's say-so.
#ifdef LTPL_EXPERIMENT
#define LTPL_ABLATED(lp, bits) (((lp).ablate & (bits)) != 0)
#define LTPL_POISON_ON(lp) ((lp).poison_on != 0)
#else
#define LTPL_ABLATED(lp, bits) false
#define LTPL_POISON_ON(lp) false
#endif

// a value every lane holds alike, moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int4 uniform_i4(const int4 v)
{
    return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y), __builtin_amdgcn_readfirstlane(v.z),
                     __builtin_amdgcn_readfirstlane(v.w));
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// KERNEL ARGUMENTS RE-READ FROM THE KERNARG SEGMENT (round 4). The four argument structs hold ~90 pointers and scalars; passed by
// value the compiler loads ALL of them into scalar registers at kernel entry, keeps them live to their last use and -- the register
// file holds ~100 -- parks most of them in lanes of two vector registers: 257 v_writelane at entry and a v_readlane (a VECTOR
// instruction in a kernel that is bound by vector-instruction issue) in front of nearly every use (1 046 static reloads in the one-wave
// kernel; 18 of the 57 vector instructions of the mask's per-batch block, ~190 in the path assembly). With RL = true the body reads
// the structs through a pointer into the kernarg segment (constant address space -> s_load_dword, scalar memory pipe, no vector
// instruction, no register held across phases) that is passed through an empty asm statement at every phase boundary: the compiler
// cannot merge loads across such a point, so every phase loads what IT needs and drops it afterwards.
#define LTPL_AS4 __attribute__((address_space(4)))
template <int RL, class T>
__device__ __forceinline__ const T* karg_reload(const T* p)
{
    if constexpr (RL) {
        unsigned long long v = (unsigned long long)p;
        asm volatile("" : "+s"(v));                                   // uniform: the address is kernarg base + constant
        return (const T*)(const T LTPL_AS4*)v;                        // (address space inferred back through the cast: scalar loads)
    } else return p;
}
// A value read from the kernarg segment that a LOOP uses: passed through an empty asm statement it is an ordinary scalar value for the
// register allocator. (An s_load from the kernarg segment is REMATERIALISABLE: under register pressure the allocator re-issues the load
// at the use instead of keeping the value -- seen inside the innermost loop of the serial sweep form, a scalar-memory round trip per edge.)
template <class T> __device__ __forceinline__ T pin_sgpr(T v) { asm volatile("" : "+s"(v)); return v; }
// (pointers: every pointer of the argument structs is a device-memory address; behind the asm statement the compiler no longer knows
//  where it came from and would address it with FLAT instructions -- the cast pair states the address space again)
template <class T> __device__ __forceinline__ T* pin_sgpr(T* p)
{
    unsigned long long v = (unsigned long long)p;                    // (through the integer: a pointer-to-pointer cast pair is folded away)
    asm volatile("" : "+s"(v));
    return (T*)(T __attribute__((address_space(1)))*)v;
}

// the four leading kernel arguments of k_paths / k_tick as they lie in the kernarg segment (all 8-byte aligned, sizes multiples of 8)
struct PathsKArgs { DevLat lat; DevPathsIn in; DevPathsOut out; TeamLds lp; };
static_assert(sizeof(DevLat) % 8 == 0 && sizeof(DevPathsIn) % 8 == 0 && sizeof(DevPathsOut) % 8 == 0 && sizeof(TeamLds) % 8 == 0 &&
              alignof(DevLat) == 8 && alignof(DevPathsIn) == 8 && alignof(DevPathsOut) == 8 && alignof(TeamLds) == 8,
              "PathsKArgs must mirror the kernarg layout of (DevLat, DevPathsIn, DevPathsOut, TeamLds)");
__device__ __forceinline__ const PathsKArgs* paths_kargs()
{
    return (const PathsKArgs*)(const PathsKArgs LTPL_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
}

// One of the four structs, RE-DERIVED from the kernarg segment pointer (RL = true only: the body is then known to run on the leading
// arguments of k_paths / k_tick). karg_reload launders a pointer that has to stay alive between the phases -- four pointers = eight
// scalar registers that the compiler parked in vector lanes around every phase (52 of the kernel's lane moves were attributed to that
// asm statement, -gline-tables-only build); the segment pointer itself is a kernel input the hardware provides.
// RL = 2 (round 6, the PERSISTENT single-tick kernel k_tick_persistent): the argument block is not the kernel's own kernarg segment -- a
// resident kernel receives new arguments every tick -- but a copy in device memory whose ADDRESS is the kernel's first argument: one more
// (cached) scalar load per derivation, everything else as RL = 1. The caller invalidates the scalar cache between ticks (s_dcache_inv).
template <int RL, class T, size_t OFF>
__device__ __forceinline__ const T* karg_at(const T* p)
{
    if constexpr (RL == 2) {
        unsigned long long seg = (unsigned long long)(const char LTPL_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(seg));
        const unsigned long long v = *(const unsigned long long LTPL_AS4*)seg;     // first kernel argument = address of the argument block
        return (const T*)(const T LTPL_AS4*)(v + OFF);
    } else if constexpr (RL == 1) {
        unsigned long long v = (unsigned long long)(const char LTPL_AS4*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(v));
        return (const T*)(const T LTPL_AS4*)(v + OFF);
    } else return p;
}
// (constants, not offsetof in the macros: the kernel bodies #define lat / in / out / lp)
static constexpr size_t KOFF_LAT = offsetof(PathsKArgs, lat), KOFF_IN = offsetof(PathsKArgs, in), KOFF_OUT = offsetof(PathsKArgs, out),
                        KOFF_LP = offsetof(PathsKArgs, lp);
#define LTPL_KARG_LAT(p) karg_at<RL, DevLat, KOFF_LAT>(p)
#define LTPL_KARG_IN(p) karg_at<RL, DevPathsIn, KOFF_IN>(p)
#define LTPL_KARG_OUT(p) karg_at<RL, DevPathsOut, KOFF_OUT>(p)
#define LTPL_KARG_LP(p) karg_at<RL, TeamLds, KOFF_LP>(p)

// Parent tables: `default` and `overtake_left` are never both needed beyond the object layer (the templates either use
// `default` itself or branch left / right off its prefix), so they share one table; three tables serve four filters.
#define NPAR 3
__device__ __forceinline__ constexpr int par_tab(int f) { return f == F_PR ? 0 : (f == F_RIGHT ? 2 : 1); }

// LDS plan policies. The sweep's hot arrays (frontiers, election words, reachability, parents, layer table) are addressed
// through a policy: PlanRt reads every offset from the host-computed TeamLds (any lattice); PlanFx<KPAD, HM, NW> fixes the
// row pitch, the table height and therewith all hot offsets at compile time, so the address arithmetic folds into
// instruction offsets (fewer live registers: the one-wave batch kernel then needs no register spills). The variable-size
// arrays (edge mask, zone mask, per-position tables) keep runtime offsets in both policies.
constexpr int plan_align16(int x) { return (x + 15) / 16 * 16; }
struct PlanRt {
    static constexpr bool fixed = false;
    static constexpr bool par_global = false;     // parent tables in LDS
    static constexpr int par_entry = 2;           // bytes per parent entry
    static constexpr int ch1 = 4;                 // register chunks of 64 edges per layer transition (one-wave team)
    static constexpr bool tail_pf = false;        // (see PlanFx)
#define LTPL_PLAN_FIELD(name) static __device__ __forceinline__ int name(const TeamLds& lp) { return lp.name; }
    LTPL_PLAN_FIELD(kpad) LTPL_PLAN_FIELD(hmax) LTPL_PLAN_FIELD(off_dist) LTPL_PLAN_FIELD(off_cnt) LTPL_PLAN_FIELD(off_widx)
    LTPL_PLAN_FIELD(off_dumin) LTPL_PLAN_FIELD(off_best) LTPL_PLAN_FIELD(off_par) LTPL_PLAN_FIELD(off_lay)
    LTPL_PLAN_FIELD(off_path) LTPL_PLAN_FIELD(path_stride) LTPL_PLAN_FIELD(n_path_bufs)
#undef LTPL_PLAN_FIELD
};
template <int KPAD, int HM, int NW>
struct PlanFx {
    static constexpr bool fixed = true;
    static constexpr bool par_global = false;
    static constexpr int par_entry = 1;           // KPAD <= 127: the source node and the tie bit share one byte
    // 192 edges in registers: fits 128 VGPRs without spills in the layer loops (4 waves per SIMD). A FOURTH chunk for plan class B (C3's 245 edges
    // per transition without a tail, round 6): 31 registers parked with scratch accesses inside the layer loops, 17.0 against 17.9 M ticks/s
    // on C3 (profiles/r06d_c3_ch4.txt) -- not kept.
    static constexpr int ch1 = 3;
    // The edges of a transition beyond the register image (`tail_edges`) are read from global memory inside the layer step. Plan class B
    // (HM = 40: the C3 oval with 245 edges per transition, lvms up to 330) meets them on EVERY layer: there the first 64 tail edges are
    // requested at the top of the layer step and arrive while the register chunks are processed (round 5; C3: two dependent global round
    // trips per layer less). Class A / C lattices (Monteblanco: 42 % of the transitions have a tail) keep their register budget.
    static constexpr bool tail_pf = HM > 32 && NW == 1;
    static constexpr int c_kpad = KPAD, c_hmax = HM;
    static constexpr int c_n_path_bufs = NW < LTPL_MAX_ACTIONS ? NW : LTPL_MAX_ACTIONS;
    static constexpr int c_off_dist = 0;
    static constexpr int c_off_cnt = c_off_dist + 8 * NFILT * 2 * KPAD;
    static constexpr int c_off_widx = c_off_cnt + 4 * NFILT * KPAD;
    static constexpr int c_off_dumin = plan_align16(c_off_widx + 4 * NFILT * KPAD);
    static constexpr int c_end_elect = c_off_dumin + 8 * NFILT * KPAD;
    static constexpr int c_path_stride = plan_align16(8 * 7 * HM + 4 * 2 * (HM + 1));
    // one-wave teams: the path scratch aliases the frontier / election arrays (dead during path assembly)
    static constexpr int c_off_path = c_n_path_bufs == 1 ? c_off_dist : c_end_elect;
    static constexpr int c_after_path = c_n_path_bufs == 1
        ? (c_end_elect > c_off_dist + c_path_stride ? c_end_elect : c_off_dist + c_path_stride)
        : c_end_elect + c_path_stride * c_n_path_bufs;
    static constexpr int c_off_best = c_after_path;
    static constexpr int c_off_par = plan_align16(c_off_best + 4 * NFILT * HM);
    static constexpr int c_par_bytes = par_entry * NPAR * HM * KPAD;
    static constexpr int c_off_lay = plan_align16(c_off_par + c_par_bytes);
    static constexpr int c_fixed_end = c_off_lay + 16 * HM;
#define LTPL_PLAN_FIELD(name) static __host__ __device__ __forceinline__ constexpr int name(const TeamLds&) { return c_##name; }
    LTPL_PLAN_FIELD(kpad) LTPL_PLAN_FIELD(hmax) LTPL_PLAN_FIELD(off_dist) LTPL_PLAN_FIELD(off_cnt) LTPL_PLAN_FIELD(off_widx)
    LTPL_PLAN_FIELD(off_dumin) LTPL_PLAN_FIELD(off_best) LTPL_PLAN_FIELD(off_par) LTPL_PLAN_FIELD(off_lay)
    LTPL_PLAN_FIELD(off_path) LTPL_PLAN_FIELD(path_stride) LTPL_PLAN_FIELD(n_path_bufs)
#undef LTPL_PLAN_FIELD
};

// Where the rows of team_backtrack_all live: one-wave teams with compile-time plans keep the path scratch ON the frontier / election arrays
// (dead during path assembly); the tie-break array behind the scratch stays free -- LTPL_MAX_ACTIONS rows of LTPL_BT_PITCH bytes go there.
#define LTPL_BT_PITCH 64
template <class P, int NW, bool F = P::fixed> struct BtPlace { static constexpr int off = -1; };
template <class P, int NW> struct BtPlace<P, NW, true> {
    static constexpr bool ok = NW == 1 && P::c_n_path_bufs == 1 && !P::par_global && P::par_entry == 1 && P::c_hmax <= LTPL_BT_PITCH &&
                               P::c_off_dumin >= P::c_off_path + P::c_path_stride &&
                               P::c_off_dumin + LTPL_MAX_ACTIONS * LTPL_BT_PITCH <= P::c_end_elect;
    static constexpr int off = ok ? P::c_off_dumin : -1;
};

// runtime plan with the parent tables in a per-scenario slab of global memory (L2 resident): planning horizons whose
// tables do not fit in LDS next to the path scratch (e.g. 600 layers at 0.5 m layer spacing)
struct PlanRtG : PlanRt { static constexpr bool par_global = true; };

template <class P>
__device__ __forceinline__ unsigned char* par_base(const TeamLds& lp, unsigned char* smem)
{
    if constexpr (P::par_global) return lp.par_glob + (size_t)blockIdx.x * (size_t)lp.par_glob_stride;
    else return smem + P::off_par(lp);
}

template <class P>
__device__ __forceinline__ void par_store(unsigned char* par, size_t idx, int src, int rank, int tie)
{
    if constexpr (P::par_entry == 2) reinterpret_cast<uchar2*>(par)[idx] = make_uchar2((unsigned char)src, (unsigned char)(rank | (tie ? 0x80 : 0)));
    else par[idx] = (unsigned char)(src | (tie ? 0x80 : 0));
}

struct TeamShared {
    int closest_idx, cl, cn, have_cn;          // written by wave 0 in phase 3
    int start_ok[NFILT];
    // previous-solution cost discount (gen_local_node_template.py:154-162), filled in phase 0: pair i of the node list applies
    // to the transition into the layer at distance fac_j[i] from the start layer (-1: outside the planning range)
    int fac_j[LTPL_MAX_LAST_NODES], fac_src[LTPL_MAX_LAST_NODES], fac_dst[LTPL_MAX_LAST_NODES];
    double fac[LTPL_MAX_LAST_NODES];
    int fac_jmax;
    double psi_sc[2];                          // (sin, cos) of the scenario's start heading psi_s (LTPL_FLAG_HAS_PSI_S), phase 0
    int grid_miss[4];                          // phase 1, per wave: some position of the wave lies outside the closest-layer grid / in a full-scan cell
};

// uniform per-scenario state, computed redundantly by every wave (scalar registers)
struct Scen {
    int s, sl, sn, flags, el, H;
    int e_base, n_base, NH;
    int veh0, n_veh, pos0, n_pos;
    int n_fac;
};

__device__ __forceinline__ bool team_node_removed(const unsigned* zone_bits, int V, const Scen& sc, int cl, int cn,
                                                  int f, int layer, int n, int gid)
{
    int nl = gid - sc.n_base; if (nl < 0) nl += V;
    if (zone_bits[nl >> 5] & (1u << (nl & 31))) return true;
    if (f == F_LEFT && layer == cl && n >= cn) return true;      // main_online_path_gen.py:148-152
    if (f == F_RIGHT && layer == cl && n < cn) return true;      // main_online_path_gen.py:155-159
    return false;
}

template <int NW>
__device__ __forceinline__ void team_sync()
{
    if constexpr (NW == 1) {
        // one wave: DS operations of a wave execute in order, so only the compiler has to be kept from reordering
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        // LDS-only workgroup barrier: the team only exchanges data through LDS, so outstanding GLOBAL loads (the edge
        // prefetch of the next layer) must not be drained here -- __syncthreads() would wait for vmcnt(0) as well
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// Election key of an edge among the edges that attain a node's minimum: the reference settles equal candidates in CSC order, i.e. by
// source node (the in-edges of a node are sorted by source); key = source << 8 | in-edge rank, the smallest key wins. (The lanes hold the
// edges in SWEEP order -- see DevLat -- so a lane or edge index says nothing about that order.)
// Sweep-order edge word (DevLat::sw_meta): in-edge rank | source node << 8 | destination node << 16 -- the low half IS the election key.
__device__ __forceinline__ unsigned elect_key(unsigned meta) { return meta & 0xffffu; }
__device__ __forceinline__ int sw_src(unsigned meta) { return (int)((meta >> 8) & 255u); }
__device__ __forceinline__ int sw_dst(unsigned meta) { return (int)(meta >> 16); }

// register image of one edge of the NEXT layer transition (raw prefetched values, consumed one layer later)
struct EdgeRegs { double c; unsigned meta; };     // meta: sweep-order edge word (elect_key / sw_src / sw_dst)

// goal node of layer j for filter f from the frontier distances in `dcur` (virtual goal edges, GraphBase.py:188-194):
// lexicographic min over (dist + vgoal, dist, node); returns node | (tie << 30) or -1
__device__ __forceinline__ int team_goal(const DevLat& lat, const double* dcur, int v0, int Kb, int lane)
{
    double g1 = INFINITY, g2 = INFINITY; int gn = 0x7fffffff;
    for (int n = lane; n < Kb; n += 64) {
        const double bestc = dcur[n];
        if (bestc < INFINITY) {
            const double tot = bestc + at(lat.vgoal, v0 + n);
            if (tot < g1 || (tot == g1 && (bestc < g2 || (bestc == g2 && n < gn)))) { g1 = tot; g2 = bestc; gn = n; }
        }
    }
    // the total alone first: a unique minimum (the normal case) needs no tie-break and carries no tie flag
    const double mt = wave_min_f64(g1);
    if (!(mt < INFINITY)) return -1;
    const unsigned long long eq = __ballot(g1 == mt);
    if ((eq & (eq - 1ull)) == 0ull) return __builtin_amdgcn_readlane(gn, __ffsll((long long)eq) - 1);
    double m1 = g1, m2 = g2; int mn = gn;
    wave_min3(m1, m2, mn);
    return mn | (1 << 30);                                     // (two or more lanes attain the total: the flag of the round-3 form)
}

// Serial form of one sweep layer for filter f (lane = destination node, private loop over its in-edges in CSC order):
// dcur[n] = min over in-edges (u, n) of dprev[u] + cost(u, n); strict '<' updates; among exact ties the predecessor with
// the smaller dprev[u], then the smaller node id wins (= the order in which Dijkstra settles them). Used by the rare
// re-sweep of reduced-horizon paths; the hot sweep is the edge-parallel form in team_paths_body.
// what the serial form reads of the lattice / the LDS plan, copied into registers in front of its loops (see SweepK)
struct SerialK {
    const int* in_ptr; const unsigned char* edge_src8; const double* edge_cost; const int* csc2sw; int E, V;
    const unsigned* blocked_bits; const unsigned* zone_bits;
};
__device__ __forceinline__ SerialK serial_k(const DevLat& lat, const TeamLds& lp, unsigned char* smem)
{
    return SerialK{pin_sgpr(lat.in_ptr), pin_sgpr(lat.edge_src8), pin_sgpr(lat.edge_cost), pin_sgpr(lat.csc2sw), pin_sgpr(lat.E), pin_sgpr(lat.V),
                   reinterpret_cast<const unsigned*>(smem + pin_sgpr(lp.off_blocked)), reinterpret_cast<const unsigned*>(smem + pin_sgpr(lp.off_zone))};
}

__device__ __forceinline__ void team_serial_node(const SerialK& K, const Scen& sc, const unsigned* blocked_bits, int f, int n,
                                                 int v, const double* dprev, int fac_src, int fac_dst, double fac,
                                                 double& bestc, int& bsrc, int& bk, int& tie)
{
    double bestdu = INFINITY;
    bestc = INFINITY; bk = 0; bsrc = 0; tie = 0;
    const int e0 = at(K.in_ptr, v), e1 = at(K.in_ptr, v + 1);
    for (int e = e0; e < e1; ++e) {
        const int src = at(K.edge_src8, e);
        double c = at(K.edge_cost, e);
        if (f != F_PR) {
            int el_ = at(K.csc2sw, e) - sc.e_base; if (el_ < 0) el_ += K.E;       // the edge bitmap is indexed in sweep order
            if ((blocked_bits[el_ >> 5] >> (el_ & 31)) & 1u) continue;
        }
        const double du = dprev[src];
        if (!(du < INFINITY)) continue;
        if (src == fac_src && n == fac_dst) c *= fac;
        const double cand = du + c;
        if (cand < bestc) { bestc = cand; bestdu = du; bk = e - e0; bsrc = src; tie = 0; }
        else if (cand == bestc) {
            tie = 1;
            if (du < bestdu) { bestdu = du; bk = e - e0; bsrc = src; }
        }
    }
}

template <class P>
__device__ __forceinline__ bool team_relax_layer(const SerialK& K, const Scen& sc, const TeamLds& lp,
                                                 int cl, int cn, int f, int j, int b, int v0, int Kb,
                                                 const double* dprev, double* dcur, unsigned char* par, size_t row, int lane,
                                                 int fac_src, int fac_dst, double fac)
{
    const unsigned* blocked_bits = K.blocked_bits;
    const unsigned* zone_bits = K.zone_bits;
    bool any = false;
    for (int n = lane; n < Kb; n += 64) {
        const int v = v0 + n;
        double bestc = INFINITY; int bk = 0, bsrc = 0, tie = 0;
        if (!team_node_removed(zone_bits, K.V, sc, cl, cn, f, b, n, v))
            team_serial_node(K, sc, blocked_bits, f, n, v, dprev, fac_src, fac_dst, fac, bestc, bsrc, bk, tie);
        dcur[n] = bestc;
        par_store<P>(par, row + n, bsrc, bk, tie);
        any = any || (bestc < INFINITY);
    }
    for (int n = Kb + lane; n < P::kpad(lp); n += 64) dcur[n] = INFINITY;
    return __ballot(any) != 0ull;
}

// cost discount along the previous solution (gen_local_node_template.py:154-162) for the transition j-1 -> j: pair i of
// the node list (factor w_last_edges[i]) applies to whatever transition its two nodes span -- normally i = j - 1 (the list
// starts at the start node, OTH.py:393), but the seam accepts any alignment (GraphBase.factor_edge_cost, GraphBase.py:478-512)
__device__ __forceinline__ void team_factor(int L, const int* ll, const int* ln, const double* w_last, const Scen& sc, int j, int b,
                                            int& fac_src, int& fac_dst, double& fac)
{
    fac_src = -1; fac_dst = -1; fac = 1.0;
    int pb = b - 1; if (pb < 0) pb += L;
    for (int i = 0; i < sc.n_fac; ++i)
        if (ll[i] == pb && ll[i + 1] == b) { fac_src = ln[i]; fac_dst = ln[i + 1]; fac = w_last[i]; break; }
}

#define LTPL_RESWEEP_ATTR __forceinline__
// Re-sweep of one filter up to layer J straight from global memory (reduced-horizon paths only: the goal node of a
// layer in front of the planning horizon is needed). Parents are rewritten with identical values.
template <class P>
__device__ LTPL_RESWEEP_ATTR void team_resweep(const DevLat& lat, const DevPathsIn& in, const Scen& sc, const TeamLds& lp, unsigned char* smem,
                             const TeamShared& ts, int f, int J, int lane)
{
    double* dist = reinterpret_cast<double*>(smem + P::off_dist(lp));
    unsigned char* par = par_base<P>(lp, smem);
    int* best = reinterpret_cast<int*>(smem + P::off_best(lp));
    // (everything the layer loop reads of the argument structs, once in front of it)
    const SerialK K = serial_k(lat, lp, smem);
    const int* const layer_off = pin_sgpr(lat.layer_off);
    const int* const ll = pin_sgpr(in.last_layer) + (size_t)sc.s * LTPL_MAX_LAST_NODES;
    const int* const ln = pin_sgpr(in.last_node) + (size_t)sc.s * LTPL_MAX_LAST_NODES;
    const double* const w_last = pin_sgpr(in.w_last);
    const int L = lat.L, kpad = P::kpad(lp), hm = P::hmax(lp);
    double* d0 = dist + (size_t)(f * 2) * kpad;
    const int K0 = at(layer_off, sc.sl + 1) - at(layer_off, sc.sl);
    const bool ok = sc.sn >= 0 && sc.sn < K0 &&
                    !team_node_removed(K.zone_bits, K.V, sc, ts.cl, ts.cn, f, sc.sl, sc.sn, at(layer_off, sc.sl) + sc.sn);
    for (int n = lane; n < kpad; n += 64) d0[n] = (ok && n == sc.sn) ? 0.0 : INFINITY;
    wave_sync_lds();
    for (int j = 1; j <= J; ++j) {
        int b = sc.sl + j; if (b >= L) b -= L;
        const int v0 = at(layer_off, b), Kb = at(layer_off, b + 1) - v0;
        int fs, fd; double fac;
        team_factor(L, ll, ln, w_last, sc, j, b, fs, fd, fac);
        const double* dprev = dist + (size_t)(f * 2 + ((j - 1) & 1)) * kpad;
        double* dcur = dist + (size_t)(f * 2 + (j & 1)) * kpad;
        (void)team_relax_layer<P>(K, sc, lp, ts.cl, ts.cn, f, j, b, v0, Kb, dprev, dcur,
                                par, ((size_t)par_tab(f) * hm + j) * kpad, lane, fs, fd, fac);
        wave_sync_lds();
    }
    int b = sc.sl + J; if (b >= L) b -= L;
    const int v0 = at(layer_off, b), Kb = at(layer_off, b + 1) - v0;
    const int g = team_goal(lat, dist + (size_t)(f * 2 + (J & 1)) * kpad, v0, Kb, lane);
    if (lane == 0) best[f * P::hmax(lp) + J] = g;
    wave_sync_lds();
}

// ---------------------------------------------------------------------------------------------------------------------
// phase 6: assemble primitive `a` (one wave): backtrack, gather, spline, re-sampling (main_online_path_gen.py:250-328)
// ---------------------------------------------------------------------------------------------------------------------
// Always inlined. An out-of-line copy (the compiler's own choice for the larger plan classes in round 1) takes the kernel-argument
// structs by reference -- they are then copied to private memory -- and, inside the one-wave kernel, ran into wrong results when
// the caller also spilled registers (spill study, DESIGN.md section 4.1: LTPL_NOINLINE_ASSEMBLE reproduces it).
#define LTPL_ASSEMBLE_ATTR __forceinline__
// ---- backtrack: node per layer along the parent tables -> pidx[0 .. J] (LDS path scratch `pw`), for 2-byte parent entries also the
//      in-edge ranks -> pedge[0 .. J-1]; writes out.n_nodes / n_ties of the slot
template <class P>
__device__ __forceinline__ void team_backtrack(const DevPathsOut& out, const TeamLds& lp, unsigned char* smem, int slot, int f, int J,
                                               int jcl, bool share_prefix, int lane, unsigned char* pw, const unsigned char* bt_row)
{
    const int hm = P::hmax(lp);
    const unsigned char* par = par_base<P>(lp, smem);
    const int* best = reinterpret_cast<const int*>(smem + P::off_best(lp));
    double* kx = reinterpret_cast<double*>(pw);
    int* pedge = reinterpret_cast<int*>(kx + 7 * hm); int* pidx = pedge + hm + 1;
    if (bt_row) {
        // the nodes of this path were chased together with those of the scenario's other paths (team_backtrack_all): copy the row
        for (int j = lane; j <= J; j += 64) pidx[j] = bt_row[j];
        wave_sync_lds();
        return;
    }
    bool staged = false;
    if constexpr (P::par_global) {
        // Parent tables in global memory: a lane-0 chase would pay one global round trip per layer. Blocks of 64 table rows
        // are copied into LDS by the whole wave (the rows of a block are contiguous per table; `kx` of the path scratch is
        // not in use yet) and chased there.
        const int kp = P::kpad(lp);
        staged = (size_t)hm * sizeof(double) >= (size_t)64 * kp * 2;
        if (staged) {
            unsigned long long* st64 = reinterpret_cast<unsigned long long*>(kx);
            const uchar2* st = reinterpret_cast<const uchar2*>(kx);
            const int bj = best[f * hm + J];
            int ties = (bj >> 30) & 1;
            int n = bj & 0xffff;
            for (int jhi = J; jhi >= 1; jhi -= 64) {
                const int jlo = jhi - 63 > 1 ? jhi - 63 : 1;
                // two contiguous segments at most: rows below the object layer come from the `default` table
                for (int seg = 0; seg < 2; ++seg) {
                    int ja, jb, pf;
                    const int split = (share_prefix && jcl > jlo && jcl <= jhi) ? jcl : -1;
                    if (split < 0) { if (seg) break; ja = jlo; jb = jhi; pf = (share_prefix && jhi < jcl) ? F_DEF : f; }
                    else if (seg == 0) { ja = jlo; jb = split - 1; pf = F_DEF; }
                    else { ja = split; jb = jhi; pf = f; }
                    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(par + ((size_t)par_tab(pf) * hm + ja) * kp * 2);
                    const int words = (jb - ja + 1) * kp / 4, w0 = (ja - jlo) * kp / 4;          // kpad is a multiple of 4
                    for (int t = lane; t < words; t += 64) st64[w0 + t] = src[t];
                }
                wave_sync_lds();
                if (lane == 0)
                    for (int j = jhi; j >= jlo; --j) {
                        const uchar2 pr = st[(j - jlo) * kp + n];
                        pidx[j] = n;
                        pedge[j - 1] = pr.y & 0x7f;
                        ties += (pr.y >> 7) & 1;
                        n = pr.x;
                    }
                wave_sync_lds();
            }
            if (lane == 0) { pidx[0] = n; out.n_nodes[slot] = J + 1; out.n_ties[slot] = ties; }
        }
    }
    if (lane == 0 && !staged) {
        const int bj = best[f * hm + J];
        int ties = (bj >> 30) & 1;
        int n = bj & 0xffff;
        for (int j = J; j >= 1; --j) {
            const int pf = (share_prefix && j < jcl) ? F_DEF : f;
            const size_t pi = ((size_t)par_tab(pf) * hm + j) * P::kpad(lp) + n;
            pidx[j] = n;                                   // node of layer j (temporarily)
            if constexpr (P::par_entry == 2) {
                const uchar2 pr = reinterpret_cast<const uchar2*>(par)[pi];
                pedge[j - 1] = pr.y & 0x7f;
                ties += (pr.y >> 7) & 1;
                n = pr.x;
            } else {
                const unsigned pr = par[pi];
                ties += pr >> 7;
                n = (int)(pr & 0x7fu);
            }
        }
        pidx[0] = n;
        out.n_nodes[slot] = J + 1;
        out.n_ties[slot] = ties;
    }
    wave_sync_lds();
}

// ---- the backtracks of ALL paths of a scenario at once (one-wave batch form, one-byte parents in LDS): lane a chases the parents of action
//      slot a. The chase is a chain of J dependent LDS round trips with a handful of instructions between them; per path it was the longest
//      stretch of the assembly in which the wave only waits (~2.5 k cycles), and a scenario has 1.8 paths on average -- now once per scenario.
//      Rows: bt[a * LTPL_BT_PITCH + j] = node of layer j; n_nodes / n_ties of the slots are written here.
template <class P>
__device__ __forceinline__ void team_backtrack_all(const DevPathsOut& out, const TeamLds& lp, unsigned char* smem, int s, int my_pk, int my_f,
                                                   bool my_sp, int jcl, int lane, int n_act, unsigned char* bt)
{
    const int hm = P::hmax(lp), kpad = P::kpad(lp);
    const unsigned char* par = par_base<P>(lp, smem);
    const int* best = reinterpret_cast<const int*>(smem + P::off_best(lp));
    if (lane < n_act && (my_pk & 1)) {
        const int J = my_pk >> 8;
        const int bj = best[my_f * hm + J];
        int ties = (bj >> 30) & 1;
        int n = bj & 0xffff;
        unsigned char* row = bt + lane * LTPL_BT_PITCH;
        for (int j = J; j >= 1; --j) {
            const int pf = (my_sp && j < jcl) ? F_DEF : my_f;
            const unsigned pr = par[((size_t)par_tab(pf) * hm + j) * kpad + n];
            row[j] = (unsigned char)n;
            ties += pr >> 7;
            n = (int)(pr & 0x7fu);
        }
        row[0] = (unsigned char)n;
        const int slot = s * LTPL_MAX_ACTIONS + lane;
        out.n_nodes[slot] = J + 1;
        out.n_ties[slot] = ties;
    }
    wave_sync_lds();
}

// ---- everything behind the backtrack: edge look-up, gather, spline, re-sampling (main_online_path_gen.py:260-328). Needs the path's
//      nodes in pidx[0 .. N] (and, with `by_rank`, the in-edge ranks in pedge[0 .. N-1]) and nothing else of the team's LDS. (Round 3
//      ran it as a kernel of its own over the slots of a batch -- one wave per action slot, 78 VGPRs, the one-wave path kernel ending
//      behind the backtrack: 31.5 instead of 33.0 M ticks/s. Fused, the instruction-heavy assembly of one scenario overlaps the
//      latency-bound sweeps of its neighbours on the SIMD; apart, the hand-over and the second launch cost more than the higher
//      occupancy returns.) Returns the number of path samples.
template <int RL = 0>
__device__ __forceinline__ int team_assemble_rest(const DevLat& lat_, const DevPathsIn& in_, const DevPathsOut& out_, int s, int sl, int flags,
                                                  int hm, bool by_rank, int slot, int N, int lane, unsigned char* pw, long long* adbg,
                                                  bool skip_pp, double* vel_kappa, double* vel_len, double* vel_x, double* vel_y, int vtile,
                                                  const double* ts_psi, const int4* lay)
{
    // (RL: argument structs in the kernarg segment, re-read at every stage of the assembly -- see karg_reload)
    const DevLat* latp = LTPL_KARG_LAT(&lat_); const DevPathsIn* inp = LTPL_KARG_IN(&in_); const DevPathsOut* outp = LTPL_KARG_OUT(&out_);
#define lat (*latp)
#define in (*inp)
#define out (*outp)
#define LTPL_KARGS() do { latp = LTPL_KARG_LAT(latp); inp = LTPL_KARG_IN(inp); outp = LTPL_KARG_OUT(outp); } while (0)
    const int L = lat.L;
    double* kx = reinterpret_cast<double*>(pw);
    double* ky = kx + hm; double* el = ky + hm; double* mx = el + hm; double* my = mx + hm;
    double* cpx = my + hm; double* cpy = cpx + hm;
    int* pedge = reinterpret_cast<int*>(cpy + hm); int* pidx = pedge + hm + 1;
    int* o_nodes = out.nodes + (size_t)slot * out.cap_nodes;
    wave_sync_lds();
    for (int i0 = 0; i0 <= N; i0 += 64) {
        const int i = i0 + lane;
        int node = 0, e = 0;
        if (i <= N) node = pidx[i];
        if (i >= 1 && i <= N) {
            int b = sl + i; if (b >= L) b -= L;
            if (by_rank) e = at(lat.in_ptr, lat.layer_off[b] + node) + pedge[i - 1];
            else {
                // the table only holds the source NODE: look the in-edge (source -> node) up in the node's record (DevLat::node_rec: first
                // in-edge + the sources of the first 12 in-edges, sorted by source, ONE 16-byte load; the node's global id from the per-layer
                // table in LDS); longer segments serially
                const int src = pidx[i - 1], gid = lay[i].x + node;
                const int4 nr = at(lat.node_rec, gid);
                e = nr.x;
                const unsigned pat = 0x01010101u * (unsigned)src;
                const unsigned x0 = (unsigned)nr.y ^ pat, x1 = (unsigned)nr.z ^ pat, x2 = (unsigned)nr.w ^ pat;
                // exact zero-byte detector (no false positives from borrows): bytes are < 0x80 or the 0xff padding
                const unsigned z0 = ~(((x0 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x0 | 0x7f7f7f7fu);
                const unsigned z1 = ~(((x1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x1 | 0x7f7f7f7fu);
                const unsigned z2 = ~(((x2 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x2 | 0x7f7f7f7fu);
                int k = z0 ? (__ffs((int)z0) - 1) >> 3 : (z1 ? 4 + ((__ffs((int)z1) - 1) >> 3) : (z2 ? 8 + ((__ffs((int)z2) - 1) >> 3) : LTPL_NODE_REC_SRC));
                if (k >= LTPL_NODE_REC_SRC) {
                    const int e1 = at(lat.in_ptr, gid + 1);
                    while (e + k < e1 - 1 && (int)at(lat.edge_src8, e + k) != src) ++k;
                }
                e += k;
            }
        }
        wave_sync_lds();
        if (i <= N) o_nodes[i] = node;
        if (i >= 1 && i <= N) pedge[i - 1] = e;
    }
    wave_sync_lds();

    dbg_stamp(adbg, 9);
    LTPL_KARGS();
    // gather: rows per edge, node row indices, knots, element lengths (:260-297)
    int run = 0;
    bool seg_dup = true;                                     // (uniform) some segment starts where its predecessor starts: rows -> segments by search
    // Everything of an edge comes from its record (DevLat::edge_rec), requested in ONE round trip: sample range and length, first knot; the
    // first / last segment also take the last knot and the (sin, cos) of the end headings (spline end slopes; parked in rows 0 and N of the
    // right-hand sides, unused until the solve). (Before: samp_ptr -> sx / sy / ssc, two dependent round trips.)
    seg_dup = false;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        int take = 0, k0 = 0, ns = 0;
        dbl2 r01, r23, r45, r67, r89;
        r01.x = r01.y = r23.x = r23.y = r45.x = r45.y = r67.x = r67.y = r89.x = r89.y = 0.0;
        if (i < N) {
            const double* r = lat.edge_rec + (unsigned)pedge[i] * (unsigned)LTPL_EDGE_REC;
            r01 = *reinterpret_cast<const dbl2*>(r); r23 = *reinterpret_cast<const dbl2*>(r + 2);
            if (i == 0) r67 = *reinterpret_cast<const dbl2*>(r + 6);
            if (i == N - 1) { r45 = *reinterpret_cast<const dbl2*>(r + 4); r89 = *reinterpret_cast<const dbl2*>(r + 8); }
            const long long w = __double_as_longlong(r01.x);
            k0 = (int)(unsigned)w; ns = (int)(w >> 32);
            take = (i == N - 1) ? ns : ns - 1;
        }
        seg_dup = seg_dup || __ballot(i < N && take <= 0) != 0ull;
        int tot; const int off = wave_excl_scan(take, lane, tot);
        if (i < N) {
            pidx[i] = run + off;
            kx[i] = r23.x; ky[i] = r23.y; el[i] = r01.y;
            pedge[i] = k0;                                   // from here on: first sample of the segment's edge
            if (i == 0) { mx[0] = r67.x; my[0] = r67.y; }
            if (i == N - 1) { kx[N] = r45.x; ky[N] = r45.y; pidx[N] = run + off + take - 1; mx[N] = r89.x; my[N] = r89.y; }
        }
        run += tot;
    }
    const int n_pts = run;
    wave_sync_lds();
    {
        int* o_idx = out.node_idx + (size_t)slot * out.cap_nodes;
        for (int i = lane; i <= N; i += 64) o_idx[i] = pidx[i];
    }
    if (lane == 0) out.n_pts[slot] = n_pts;

    dbg_stamp(adbg, 10);
    LTPL_KARGS();
    // tph.calc_splines (main_online_path_gen.py:299-309) as the equivalent clamped C2 spline in the cumulated
    // el_lengths parameter: tridiagonal system in the knot slopes m_i, Thomas algorithm. Everything that does not depend
    // on the elimination order is prepared by all lanes (reciprocal segment lengths, diagonal, right-hand sides of x and
    // y); the sequential elimination (lane 0 -> x, lane 1 -> y) is one reciprocal and four fused multiply-adds per row.
    {
        // end slopes: tangent = (cos(psi + pi/2), sin(psi + pi/2)) = (-sin psi, cos psi). Round 4: no sincos per path (165 vector
        // instructions on 64 lanes for two angles) -- the headings of the samples are lattice constants (DevLat::ssc), the start heading of
        // a scenario with a constant path segment is one sincos per SCENARIO (phase 0, TeamShared::psi_sc)
        double sn = 0.0, cs = 1.0;
        if (lane == 0) { if (flags & LTPL_FLAG_HAS_PSI_S) { sn = ts_psi[0]; cs = ts_psi[1]; } else { sn = mx[0]; cs = my[0]; } }
        if (lane == 1) { sn = mx[N]; cs = my[N]; }
        const double sx0 = -readlane_f64(sn, 0), sy0 = readlane_f64(cs, 0), sxN = -readlane_f64(sn, 1), syN = readlane_f64(cs, 1);
        // rows i = 1 .. N-1: a_i = 1/h_{i-1}, c_i = 1/h_i, b_i = 2 (a_i + c_i); stored: cpx <- a_i, cpy <- b_i (scratch),
        // mx / my <- right-hand sides d_i (x / y)
        for (int i = lane; i < N; i += 64) cpx[i] = fast_rcp(el[i]);         // reciprocal segment lengths (reused below)
        wave_sync_lds();
        for (int i = 1 + lane; i <= N - 1; i += 64) {
            const double ai = cpx[i - 1], ci = cpx[i];
            double dx = 3.0 * ((kx[i] - kx[i - 1]) * (ai * ai) + (kx[i + 1] - kx[i]) * (ci * ci));
            double dy = 3.0 * ((ky[i] - ky[i - 1]) * (ai * ai) + (ky[i + 1] - ky[i]) * (ci * ci));
            if (i == 1) { dx -= ai * sx0; dy -= ai * sy0; }
            if (i == N - 1) { dx -= ci * sxN; dy -= ci * syN; }
            mx[i] = dx; my[i] = dy; cpy[i] = 2.0 * (ai + ci);
        }
        wave_sync_lds();
        if (N <= 63) {
            // Parallel cyclic reduction (lane = row i of the system, i = 1 .. N-1 unknown slopes; rows 0 and N and every lane
            // beyond are IDENTITY rows a = c = 0, b = 1, d = 0, which the update leaves unchanged -- no masks anywhere): log2(N)
            // rounds in which every row eliminates its two neighbours at distance 1, 2, 4, ... The serial elimination (Thomas) ran
            // ~30 instructions per ROW on two lanes. The system is strictly diagonally dominant (b = 2 (a + c)), so PCR is as
            // stable as the elimination; the slopes agree to rounding.
            const int i = lane;
            const bool row = i >= 1 && i <= N - 1;
            double ra = 0.0, rb = 1.0, rc = 0.0, rx = 0.0, ry = 0.0;
            if (row) { ra = (i == 1) ? 0.0 : cpx[i - 1]; rc = (i == N - 1) ? 0.0 : cpx[i]; rb = cpy[i]; rx = mx[i]; ry = my[i]; }
            wave_sync_lds();
            for (int st = 1; st < N - 1; st <<= 1) {
                if (i <= N) { cpx[i] = ra; cpy[i] = rb; mx[i] = rx; my[i] = ry; }
                wave_sync_lds();
                const int lo = i - st > 0 ? i - st : 0, hi = i + st < N ? i + st : N;
                const double cm = __shfl(rc, lo), cp = __shfl(rc, hi);
                const double al = -ra * fast_rcp(cpy[lo]), ga = -rc * fast_rcp(cpy[hi]);
                rb = rb + al * cm + ga * cpx[hi];
                rx = rx + al * mx[lo] + ga * mx[hi]; ry = ry + al * my[lo] + ga * my[hi];
                ra = al * cpx[lo]; rc = ga * cp;
                wave_sync_lds();
            }
            if (row) { const double ib = fast_rcp(rb); mx[i] = rx * ib; my[i] = ry * ib; }
            if (lane == 0) { mx[0] = sx0; my[0] = sy0; mx[N] = sxN; my[N] = syN; }
        } else {
        if (lane < 2) {
            double* m = lane == 0 ? mx : my;
            m[0] = lane == 0 ? sx0 : sy0; m[N] = lane == 0 ? sxN : syN;
            // forward elimination: cprime_i = c_i / (b_i - a_i cprime_{i-1}), dprime_i = (d_i - a_i dprime_{i-1}) / (same)
            double cprev = 0.0, dprev_ = 0.0;
            double* cpr = lane == 0 ? kx + 0 : nullptr;   // (unused; cprime is kept in `el`'s neighbour scratch below)
            (void)cpr;
            for (int i = 1; i <= N - 1; ++i) {
                const double ai = cpx[i - 1], ci = (i == N - 1) ? 0.0 : cpx[i], bi = cpy[i];
                const double r = fast_rcp(bi - ai * cprev);     // on the serial path of the wave: 6 instead of ~25 instructions per row
                const double cpi = ci * r, dpi = (m[i] - ai * dprev_) * r;
                m[i] = dpi; cprev = cpi; dprev_ = dpi;
                if (lane == 0) cpy[i] = cpi;                                    // cprime (identical for x and y)
            }
        }
        wave_sync_lds();
        if (lane < 2) {
            double* m = lane == 0 ? mx : my;
            for (int i = N - 2; i >= 1; --i) m[i] = m[i] - cpy[i] * m[i + 1];
        }
        }
    }
    wave_sync_lds();

    dbg_stamp(adbg, 11);
    LTPL_KARGS();
    // coefficients per segment, t in [0, 1]: a0 = k_i, a1 = m_i h, a2 = 3 d - 2 T0 - T1, a3 = -2 d + T0 + T1
    double* o_coeff = out.coeff + (size_t)slot * out.cap_nodes * 8;
    for (int i = lane; i < N; i += 64) {
        const double h = el[i];
        {
            const double T0 = mx[i] * h, T1 = mx[i + 1] * h, dlt = kx[i + 1] - kx[i];
            store2(o_coeff + i * 8 + 0, kx[i], T0);
            store2(o_coeff + i * 8 + 2, 3.0 * dlt - 2.0 * T0 - T1, -2.0 * dlt + T0 + T1);
        }
        {
            const double T0 = my[i] * h, T1 = my[i + 1] * h, dlt = ky[i + 1] - ky[i];
            store2(o_coeff + i * 8 + 4, ky[i], T0);
            store2(o_coeff + i * 8 + 6, 3.0 * dlt - 2.0 * T0 - T1, -2.0 * dlt + T0 + T1);
        }
    }

    dbg_stamp(adbg, 12);
    LTPL_KARGS();
    // tph.interp_splines(stepnum_fixed) + tph.calc_head_curv_an (:311-322); column 4 keeps the offline spacing
    double* o_pp = out.path_param + (size_t)slot * out.cap_pts * 5;
    // (everything the row loop reads of the argument structs, once in front of it)
    const double* const a_slen = pin_sgpr(lat.slen);
    ke_t* a_vke = pin_sgpr(out.vke); double* a_vxy = nullptr;
    if (a_vke) {                                          // planes of the batch velocity stage, blocked by 8 rows (kep_base / kep_row)
        const int nrb = (((out.cap_pts + 7) >> 3) << 3) / KE_RB, nsp = out.n_slots_pad;      // row blocks per tile (plane_rows / KE_RB)
        a_vke += ((size_t)(vtile >> 6) * nrb * 64 + (vtile & 63)) * KE_RB;
        if (vtile >= nsp) {                               // follow job: (x, y) for the lane-per-job follow preparation
            const int fj = vtile - nsp;
            a_vxy = out.vxy + 2 * (((size_t)(fj >> 6) * nrb * 64 + (fj & 63)) * KE_RB);
        }
    }
    // Row -> segment (segment i with pidx[i] <= r < pidx[i+1]; last: <=). Round 5: the first rows of segments 1 .. N-1 are marked in a bit
    // string (LDS, `cpx`: free behind the solve); the segment of row r is the number of marks at or below r -- per block of 64 rows ONE uniform
    // 8-byte read, a running scalar count and a lane-prefix population count. (It was a binary search over pidx per row: five dependent LDS
    // round trips per block. Kept for paths with a segment that contributes no row, whose marks would coincide.)
    unsigned* segw = reinterpret_cast<unsigned*>(cpx);
    if (((n_pts + 63) >> 6) > hm) seg_dup = true;            // (the bit string has to fit `cpx`: hm * 64 rows)
    if (!seg_dup) {
        for (int w = lane; w < ((n_pts + 63) >> 6) * 2; w += 64) segw[w] = 0u;
        wave_sync_lds();
        for (int i = 1 + lane; i <= N - 1; i += 64) { const int q = pidx[i]; atomicOr(&segw[q >> 5], 1u << (q & 31)); }
        wave_sync_lds();
    }
    int seg_carry = 0;
    for (int r = lane; r < n_pts; r += 64) {
        int lo = 0, hi = N;
        if (seg_dup) { while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pidx[mid] <= r) lo = mid; else hi = mid; } }
        else {
            const uint2 mk = *reinterpret_cast<const uint2*>(segw + ((r - lane) >> 5));             // uniform address: the block's 64 marks
            const unsigned m0 = __builtin_amdgcn_readfirstlane(mk.x), m1 = __builtin_amdgcn_readfirstlane(mk.y);
            const unsigned own = lane < 32 ? (m0 >> lane) & 1u : (m1 >> (lane - 32)) & 1u;
            lo = seg_carry + (int)__builtin_amdgcn_mbcnt_hi(m1, __builtin_amdgcn_mbcnt_lo(m0, 0u)) + (int)own;
            seg_carry += __builtin_popcount(m0) + __builtin_popcount(m1);
        }
        const int i = lo, k = r - pidx[i];
        const int n_i = pidx[i + 1] - pidx[i] + 1;
        const double t = (k == n_i - 1) ? 1.0 : (double)k * fast_rcp((double)(n_i - 1));
        const double h = el[i];
        const double Tx0 = mx[i] * h, Tx1 = mx[i + 1] * h, dx_ = kx[i + 1] - kx[i];
        const double Ty0 = my[i] * h, Ty1 = my[i + 1] * h, dy_ = ky[i + 1] - ky[i];
        const double ax0 = kx[i], ax1 = Tx0, ax2 = 3.0 * dx_ - 2.0 * Tx0 - Tx1, ax3 = -2.0 * dx_ + Tx0 + Tx1;
        const double ay0 = ky[i], ay1 = Ty0, ay2 = 3.0 * dy_ - 2.0 * Ty0 - Ty1, ay3 = -2.0 * dy_ + Ty0 + Ty1;
        const double t2 = t * t, t3 = t2 * t;
        double x = ((ax0 + ax1 * t) + ax2 * t2) + ax3 * t3;
        double y = ((ay0 + ay1 * t) + ay2 * t2) + ay3 * t3;
        if (r == n_pts - 1) { x = ((ax0 + ax1) + ax2) + ax3; y = ((ay0 + ay1) + ay2) + ay3; }
        const double xd = ax1 + 2.0 * ax2 * t + 3.0 * ax3 * t2, yd = ay1 + 2.0 * ay2 * t + 3.0 * ay3 * t2;
        const double xdd = 2.0 * ax2 + 6.0 * ax3 * t, ydd = 2.0 * ay2 + 6.0 * ay3 * t;
        const double q = xd * xd + yd * yd;
        double* row = o_pp + (size_t)r * 5;
        // psi = normalize(atan2(y', x') - pi/2) = atan2(-x', y') (rotation by -90 degrees), range [-pi, pi)
        double psi_r = heading_atan2(-xd, yd);
        if (psi_r >= D_PI) psi_r -= 2.0 * D_PI;
        const double kap = (xd * ydd - yd * xdd) * rsqrt_cubed(q);          // q^(-3/2); q = |tangent|^2 > 0
        const double len_r = at(a_slen, pedge[i] + k);
        if (!skip_pp)           // (experiment build: LTPL_ABLATE bit 16 drops the path_param stores, timing only)
        { store2_u(row, x, y); store2_u(row + 2, psi_r, kap); row[4] = len_r; }
        if (vel_kappa) { vel_kappa[r] = kap; vel_len[r] = len_r; }
        if (a_vke) {
            // r = lane + 64 k, so kep_row(r) = kep_row(lane) + 64 (r - lane)
            const size_t ro = (size_t)kep_row(lane) + (size_t)(r - lane) * 64;
            a_vke[ro] = make_ke(fabs(kap), len_r);     // (non-temporal hints on the planes' stores and last loads: -10 % ticks/s, profiles/r06b_ab_bench.txt)
            if (a_vxy) store2(a_vxy + 2 * ro, x, y);
        }
        if (vel_x) { vel_x[r] = x; vel_y[r] = y; }
    }
    dbg_stamp(adbg, 13);
    if (out.job_cnt && lane == 0) out.job_slot[vtile] = make_int2(slot, n_pts);
    wave_sync_lds();
    return n_pts;
#undef LTPL_KARGS
#undef lat
#undef in
#undef out
}

template <class P, int RL = 0>
__device__ LTPL_ASSEMBLE_ATTR WavePath team_assemble(const DevLat& lat, const DevPathsIn& in, const DevPathsOut& out, const Scen& sc,
                                  const TeamLds& lp, unsigned char* smem, int a, int f, int J, int name, int reduced,
                                  int jcl, bool share_prefix, int lane, unsigned char* pw,
                                  double* vel_kappa, double* vel_len, double* vel_x, double* vel_y, int vtile_in, const double* ts_psi,
                                  const unsigned char* bt_row = nullptr)
{
    const int hm = P::hmax(lp), s = sc.s;
    const int slot = s * LTPL_MAX_ACTIONS + a;
    WavePath wp; wp.valid = 1; wp.name = name; wp.reduced = reduced;
    // batch pipeline: take a job of the velocity stage (class 0 = generic profile, 1 = follow); the planes are tiled by job
    // (one-wave batch form: the job indices of all paths of the scenario were reserved with ONE atomic in the decision phase --
    //  vtile_in; the four-wave forms reserve per path here)
    int vtile = vtile_in;
    if (out.job_cnt && vtile_in < 0) {
        if (lane == 0) {
            const int cls = name == LTPL_ACT_FOLLOW ? 1 : 0;
            const int jb = atomicAdd(&out.job_cnt[cls], 1);
            vtile = cls ? out.n_slots_pad + jb : jb;
        }
        vtile = __builtin_amdgcn_readfirstlane(vtile);
    }
    long long* const adbg = a == 0 ? lp.dbg : nullptr;        // experiment build: phase stamps 8 .. 13 of the first primitive's assembly
    dbg_stamp(adbg, 8);
    team_backtrack<P>(out, lp, smem, slot, f, J, jcl, share_prefix, lane, pw, bt_row);
    const int end_node = reinterpret_cast<const int*>(reinterpret_cast<double*>(pw) + 7 * hm)[hm + 1 + J];      // pidx[J]
    const int n_pts = team_assemble_rest<RL>(lat, in, out, s, sc.sl, sc.flags, hm, P::par_entry == 2, slot, J, lane, pw, adbg,
                                         LTPL_ABLATED(lp, 16), vel_kappa, vel_len, vel_x, vel_y, vtile, ts_psi,
                                         reinterpret_cast<const int4*>(smem + P::off_lay(lp)));
    const int L = lat.L;
    wp.n_pts = n_pts; wp.n_nodes = J + 1;
    { int gl = sc.sl + J; if (gl >= L) gl -= L; wp.goal_layer = gl; }
    wp.end_node = end_node;
    wave_sync_lds();
    return wp;
}

// One sweep layer for the (compile-time) filter set ACT, EDGE-PARALLEL: lane = edge of the transition (register image
// `er`, prefetched one layer ahead with coalesced loads), two rounds of LDS atomics per layer:
//   round 0  frontier[dst] = min(frontier[dst], dprev[src] + cost)      ds_min_u64 on the fp64 bit pattern (costs >= 0)
//   round 1  edges that attain the minimum count themselves and elect the first of them in CSC order
// then lane = destination node: parents, reachability, node filters (zone / overtake side), goal of the last layer. The
// exact tie-break of the reference order only matters when the minimum is attained more than once: those nodes (rare)
// re-scan their in-edges serially.
// what the layer loop reads of the lattice / the LDS plan, copied into registers once in front of the sweeps (with the kernel
// arguments re-read from the kernarg segment -- karg_reload -- nothing inside the loop may depend on an argument load)
struct SweepK {
    const double* sw_cost; const unsigned* sw_meta; int E, V;
    const unsigned* blocked_bits; const unsigned* zone_bits;
    bool zone_any;              // uniform: some node of the planning range is removed by a zone (else the node step skips the bit look-up)
};

// Counting and election in ONE LDS atomic (round 4): every edge that attains its destination's minimum adds `CW_ONE | key` to the node's
// word -- the count of such edges in the high byte, and, when it is 1 (the normal case), the winner's election key (source << 8 | rank,
// 16 bits) in the low bits. (Round 3: one atomic add on a counter and one atomic min on the key word per chunk, two arrays to reset and to
// read back; the LDS pipe of a CU is ~78 % busy in this kernel, profiles/r04b_pmc_karg_reload.txt.) With two or more winners the low bits hold
// the SUM of their keys (< 2^23 for 127 in-edges) and are not used: the exact tie-break below elects into `widx`.
#ifndef LTPL_MQ
#define LTPL_MQ 2             // obstacle positions tested per pass over a transition's edges in the mask phase
#endif
#ifndef LTPL_CH_ALWAYS
#define LTPL_CH_ALWAYS 1      // edge chunks of a transition that are processed unconditionally (sentinel edges in unused lanes); later chunks are skipped when
                              // empty. Round 3 kept two: 44 % of Monteblanco's transitions have at most 64 edges, one is 2 % fewer vector and 5 % fewer LDS
                              // instructions (+0.9 % ticks/s, profiles/r04l_*); three costs a spilled register
#endif
#define CW_SHIFT 24
#define CW_ONE (1u << CW_SHIFT)
struct LayerArgs {
    int j, b, v0, Kb, ne, eb, kpad, hm, cur, prv, H;
    int fs, fd, cl_hit, cn;                      // cl_hit: this layer is the closest object's layer
    bool from_def;
    double fac;
};

// (one layer of software pipelining -- the candidate read of layer j + 1 issued with the read-back of layer j -- was built and lost its A/B twice:
//  docs/HISTORY.md, tools/experiments/r05_lost_switches.patch)
// NCHK (round 5): 0 = the number of chunks that hold edges is tested at run time at every site (uniform branches); 1 .. CH = exactly the first
// NCHK chunks hold edges (NCHK < CH: and the transition has no tail) -- the layer loop picks the instance per layer with ONE switch, the body
// carries no chunk tests, and only the LAST chunk can hold lanes beyond the transition.
template <class P, int NW, int CH, unsigned ACT, int NCHK = 0>
__device__ __forceinline__ void team_layer(const SweepK& K, const Scen& sc, const TeamLds& lp, unsigned char* smem,
                                           const LayerArgs& A, const EdgeRegs (&er)[CH], const unsigned blk,
                                           int wave, int lane)
{
    const unsigned* blocked_bits = K.blocked_bits;
    const unsigned* zone_bits = K.zone_bits;
    double* dist = reinterpret_cast<double*>(smem + P::off_dist(lp));
    unsigned char* par = par_base<P>(lp, smem);
    int* best = reinterpret_cast<int*>(smem + P::off_best(lp));
    unsigned* cnt_all = reinterpret_cast<unsigned*>(smem + P::off_cnt(lp));
    unsigned* widx_all = reinterpret_cast<unsigned*>(smem + P::off_widx(lp));
    const int kpad = A.kpad, tid = wave * 64 + lane;
    constexpr int NT = NW * 64;
    // chunks of the register image that hold edges of this transition (uniform; an integer in a scalar register: as a boolean per site the
    // compiler carried lane masks from block to block and inverted them through a vector register)
    const int nch = NCHK > 0 ? NCHK : pin_sgpr((A.ne + NT - 1) / NT);
    // (run-time form: tested through a fresh copy at every site -- one shared boolean crosses basic blocks as a lane mask and is inverted through
    //  a vector register)
    auto absent = [&](int ci) {
        if constexpr (NCHK > 0) return ci >= NCHK;
        else return ci >= LTPL_CH_ALWAYS && ci >= pin_sgpr(nch);
    };
    const bool any_blk = __ballot(blk != 0u) != 0ull;
    int poff[NFILT], coff[NFILT];
#pragma unroll
    for (int f = 0; f < NFILT; ++f) {
        const int fprev = (A.from_def && (f == F_LEFT || f == F_RIGHT)) ? F_DEF : f;
        poff[f] = (fprev * 2 + A.prv) * kpad; coff[f] = (f * 2 + A.cur) * kpad;
    }
    // reset the targets of this layer
    for (int n = tid; n < kpad; n += NT) {
#pragma unroll
        for (int f = 0; f < NFILT; ++f)
            if ((ACT >> f) & 1u) { dist[coff[f] + n] = INFINITY; cnt_all[f * kpad + n] = 0u; }
    }
    team_sync<NW>();
    // the first tail edges of the transition, requested now (P::tail_pf), consumed by tail_edges below
    // (a scalar test the compiler cannot fold into the tail loop's per-lane entry condition: folded, every layer paid the lane compare and the
    //  address arithmetic of a loop that 1 transition in 20 enters)
    auto has_tail_f = [&]() { if constexpr (NCHK > 0 && NCHK < CH) return false; else return pin_sgpr(A.ne) > CH * NT; };
#define has_tail has_tail_f()
    EdgeRegs tl; tl.c = INFINITY; tl.meta = 0u;
    if constexpr (P::tail_pf) {
        if (has_tail) { const int ei = CH * NT + tid; const int e = ei < A.ne ? A.eb + ei : K.E; tl.c = at(K.sw_cost, e); tl.meta = at(K.sw_meta, e); }
    }
    // candidate sums are kept for the active filters only (compile-time compaction keeps the register image small)
    constexpr int NA = ((ACT >> 0) & 1) + ((ACT >> 1) & 1) + ((ACT >> 2) & 1) + ((ACT >> 3) & 1);
    constexpr int SL[NFILT] = {0, (int)((ACT >> 0) & 1), (int)(((ACT >> 0) & 1) + ((ACT >> 1) & 1)), (int)(((ACT >> 0) & 1) + ((ACT >> 1) & 1) + ((ACT >> 2) & 1))};
    // An edge that must not be used (beyond the transition, blocked, unreachable source) simply carries the candidate
    // +inf: inf + cost = inf never wins the atomic min and never matches a finite minimum, so the rounds need no
    // per-lane bookkeeping and no divergent control flow.
    // Lanes beyond the transition get the cost +inf below (their edge word is whatever followed the transition in the table: prefetch), so
    // every lane of a loaded chunk runs the same code: all LDS reads of a round are issued before the first wait. Chunks that hold no
    // edge are not processed (NCHK: compile time; else uniform tests; 111 edges per transition on average).
    double cand[CH][NA];
#pragma unroll
    for (int ci = 0; ci < CH; ++ci) {
        if (absent(ci)) continue;            // uniform: no edges in this chunk
        const int src = sw_src(er[ci].meta);
#pragma unroll
        for (int f = 0; f < NFILT; ++f) if ((ACT >> f) & 1u) cand[ci][SL[f]] = dist[poff[f] + src];
    }
#pragma unroll
    for (int ci = 0; ci < CH; ++ci) {
        if (absent(ci)) continue;
        const int dst = sw_dst(er[ci].meta);
        // planning_range: every edge OF THE TRANSITION -- the lanes of the chunk beyond it hold whatever followed in the table (prefetch)
        // (with NCHK only the last chunk can; one vector compare against the uniform count -- as a scalar lane mask it was eight scalar instructions)
        const double c_pr = (NCHK > 0 && ci < NCHK - 1) || lane < A.ne - (ci * NW + wave) * 64 ? er[ci].c : (double)INFINITY;
        // other filters: unblocked edges (bit ci of `blk`: this lane's edge of chunk ci is blocked; most transitions hold none -- uniform skip)
        double c_np = c_pr;
        if ((ACT & ~(1u << F_PR)) && any_blk) c_np = ((blk >> ci) & 1u) ? (double)INFINITY : c_pr;
#pragma unroll
        for (int f = 0; f < NFILT; ++f) {
            if (!((ACT >> f) & 1u)) continue;
            cand[ci][SL[f]] = cand[ci][SL[f]] + (f == F_PR ? c_pr : c_np);
            // EVERY lane issues the minimum (round 5): +inf changes nothing, and since the chunk loads take whatever follows the transition in the
            // table (prefetch) the unused lanes address scattered nodes instead of meeting in the sentinel's. No compare, no exec-mask save /
            // restore around the atomic, and the atomics of a layer's chunks issue back to back: +2.7 % ticks/s (profiles/r05q_ab_bench.txt).
            // (Rounds 2 - 4: conditional -- the unused lanes all held the sentinel edge, destination node 0, and serialised in the LDS: +18 %
            //  kernel time when it was tried unconditionally then.)
            atomicMin(reinterpret_cast<unsigned long long*>(&dist[coff[f] + dst]), (unsigned long long)__double_as_longlong(cand[ci][SL[f]]));
        }
    }
    // transitions with more edges than the register image: the rest straight from global memory (rare, not prefetched);
    // ROUND = 0: atomic min of the candidate sums, 1: election among the edges that attain it, 2 / 3: exact tie-break
    auto tail_edges = [&](int ROUND) {
        // (keeps the block behind its scalar branch: nothing of the loop is hoisted in front of `has_tail`, and no lane constant of it --
        //  (tid + CH * NT - e_base) was one -- is computed in front of the sweeps, parked in scratch and fetched here)
        const int e_base_t = pin_sgpr(__builtin_amdgcn_readfirstlane(sc.e_base));
        double* dumin = reinterpret_cast<double*>(smem + P::off_dumin(lp));
        for (int ei = CH * NT + tid; ei < A.ne; ei += NT) {
            const int e = A.eb + ei;
            double c; unsigned meta;
            if (P::tail_pf && ROUND == 0 && ei < (CH + 1) * NT) { c = tl.c; meta = tl.meta; }   // (the prefetched first 64 tail edges; the later rounds
                                                                                                //  re-read them -- cache hits now -- instead of holding three registers across the election)
            else { c = at(K.sw_cost, e); meta = at(K.sw_meta, e); }
            const int src = sw_src(meta), dst = sw_dst(meta);
            int el_ = e - e_base_t; if (el_ < 0) el_ += K.E;
            const bool unbl = !((blocked_bits[el_ >> 5] >> (el_ & 31)) & 1u);
            if (A.fs >= 0 && src == A.fs && dst == A.fd) c *= A.fac;
            const unsigned key = elect_key(meta);
#pragma unroll
            for (int f = 0; f < NFILT; ++f) {
                if (!((ACT >> f) & 1u)) continue;
                const double du = dist[poff[f] + src];
                bool ok = du < INFINITY;
                if (f != F_PR) ok = ok && unbl;
                // (rounds 0 / 1 without control flow, like the register chunks: an edge that must not be used carries +inf / adds nothing. The
                //  cost is then needed by every lane, so its load is issued WITH the edge word's at the top of the iteration -- as `if (!ok)
                //  continue` the compiler had sunk it behind the frontier read: two dependent global round trips per round, on 42 % of
                //  Monteblanco's transitions)
                if (ROUND == 0) {
                    const double cd0 = ok ? du + c : (double)INFINITY;
                    atomicMin(reinterpret_cast<unsigned long long*>(&dist[coff[f] + dst]), (unsigned long long)__double_as_longlong(cd0));
                    continue;
                }
                if (ROUND == 1) {
                    const double cd1 = du + c, g1 = dist[coff[f] + dst];
                    const bool win = (g1 == cd1) & ok;                        // (no short circuit: the cost stays an unconditional load)
                    atomicAdd(&cnt_all[f * kpad + dst], win ? (CW_ONE | key) : 0u);
                    continue;
                }
                if (!ok) continue;
                const double cd = du + c;
                if (ROUND == 0) { atomicMin(reinterpret_cast<unsigned long long*>(&dist[coff[f] + dst]), (unsigned long long)__double_as_longlong(cd)); continue; }
                if (dist[coff[f] + dst] != cd) continue;
                if (ROUND == 1) { atomicAdd(&cnt_all[f * kpad + dst], CW_ONE | key); continue; }
                if ((cnt_all[f * kpad + dst] >> CW_SHIFT) < 2u) continue;
                if (ROUND == 2) atomicMin(reinterpret_cast<unsigned long long*>(&dumin[f * kpad + dst]), (unsigned long long)__double_as_longlong(du));
                else if (dumin[f * kpad + dst] == du) atomicMin(&widx_all[f * kpad + dst], key);
            }
        }
    };
    if (has_tail) tail_edges(0);
    team_sync<NW>();
    {
        double got[CH][NA];
#pragma unroll
        for (int ci = 0; ci < CH; ++ci) {
            if (absent(ci)) continue;
            const int dst = sw_dst(er[ci].meta);
#pragma unroll
            for (int f = 0; f < NFILT; ++f) if ((ACT >> f) & 1u) got[ci][SL[f]] = dist[coff[f] + dst];
        }
#pragma unroll
        for (int ci = 0; ci < CH; ++ci) {
            if (absent(ci)) continue;
            const int dst = sw_dst(er[ci].meta);
            const unsigned key = elect_key(er[ci].meta);
#pragma unroll
            for (int f = 0; f < NFILT; ++f) {
                if (!((ACT >> f) & 1u)) continue;
                const bool win = got[ci][SL[f]] == cand[ci][SL[f]] && cand[ci][SL[f]] < INFINITY;
                atomicAdd(&cnt_all[f * kpad + dst], win ? (CW_ONE | key) : 0u);          // (every lane adds, the others add nothing: +0.7 %, r05r_ab_bench.txt)
            }
        }
    }
    if (has_tail) tail_edges(1);
    team_sync<NW>();
    // lane = destination node (this form is only entered with at most 64 nodes in the layer): counters, elected edges and the zone bit
    // of every active filter are fetched in ONE LDS round trip and serve both the tie check and the node step (they used to be
    // read twice with a wait each -- the sweep is a chain of LDS round trips, ~6 per layer)
    const int n = lane;
    const bool nv = n < A.Kb;
    unsigned c_r[NA], w_r[NA];
#pragma unroll
    for (int f = 0; f < NFILT; ++f)
        if ((ACT >> f) & 1u) {
            const unsigned cw = nv ? cnt_all[f * kpad + n] : 0u;       // (read by every lane + select: -1 % ticks/s, profiles/r05u_ab_bench.txt)
            c_r[SL[f]] = cw >> CW_SHIFT; w_r[SL[f]] = cw & (CW_ONE - 1u);
        }
    bool zone_rem = false;
    if (K.zone_any && nv) { int nl = A.v0 + n - sc.n_base; if (nl < 0) nl += K.V; zone_rem = (zone_bits[nl >> 5] >> (nl & 31)) & 1u; }
    // exact tie-break (rare): a node whose minimum is attained by several edges takes, in the reference's order, the
    // predecessor with the smaller distance first, then CSC order. Every wave reads the same counters, so the branch is
    // uniform over the team.
    {
        bool tied = false;
#pragma unroll
        for (int k = 0; k < NA; ++k) tied = tied || c_r[k] >= 2u;
        if (__ballot(tied) != 0ull) {
            double* dumin = reinterpret_cast<double*>(smem + P::off_dumin(lp));
            for (int m = tid; m < A.Kb; m += NT) {
#pragma unroll
                for (int f = 0; f < NFILT; ++f)
                    if (((ACT >> f) & 1u) && (cnt_all[f * kpad + m] >> CW_SHIFT) >= 2u) { dumin[f * kpad + m] = INFINITY; widx_all[f * kpad + m] = 0xffffffffu; }
            }
            team_sync<NW>();
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
#pragma unroll
                for (int ci = 0; ci < CH; ++ci) {
                    if (absent(ci)) continue;
                    const int src = sw_src(er[ci].meta), dst = sw_dst(er[ci].meta);
                    const unsigned key = elect_key(er[ci].meta);
#pragma unroll
                    for (int f = 0; f < NFILT; ++f) {
                        if (!((ACT >> f) & 1u) || !(cand[ci][SL[f]] < INFINITY)) continue;
                        if (dist[coff[f] + dst] != cand[ci][SL[f]] || (cnt_all[f * kpad + dst] >> CW_SHIFT) < 2u) continue;
                        const double du = dist[poff[f] + src];
                        if (round == 0) atomicMin(reinterpret_cast<unsigned long long*>(&dumin[f * kpad + dst]), (unsigned long long)__double_as_longlong(du));
                        else if (dumin[f * kpad + dst] == du) atomicMin(&widx_all[f * kpad + dst], key);
                    }
                }
                if (has_tail) tail_edges(2 + round);
                team_sync<NW>();
            }
#pragma unroll
            for (int f = 0; f < NFILT; ++f) if (((ACT >> f) & 1u) && c_r[SL[f]] >= 2u) w_r[SL[f]] = widx_all[f * kpad + n];      // re-elected (tied nodes only)
        }
    }
    // node step: parents, node filters, reachability
#pragma unroll
    for (int f = 0; f < NFILT; ++f) {
        if (!((ACT >> f) & 1u)) continue;
        if (NW > 1 && (f % NW) != wave) continue;
        bool any = false;
        if (nv) {
            const unsigned c = c_r[SL[f]], w = w_r[SL[f]];
            bool rem = zone_rem;
            if (f == F_LEFT) rem = rem || (A.cl_hit && n >= A.cn);
            if (f == F_RIGHT) rem = rem || (A.cl_hit && n < A.cn);
            const bool fin = c >= 1u && !rem;
            const int bsrc = fin ? (int)((w >> 8) & 255u) : 0, bk = fin ? (int)(w & 255u) : 0, tie = (fin && c >= 2u) ? 1 : 0;   // elect_key layout
            if (rem) dist[coff[f] + n] = INFINITY;
            par_store<P>(par, ((size_t)par_tab(f) * A.hm + A.j) * kpad + n, bsrc, bk, tie);
            any = fin;
        }
        any = __ballot(any) != 0ull;
        if (lane == 0) best[f * A.hm + A.j] = any ? -2 : -1;           // the goal node of the last layer is evaluated after the sweep
    }
}
#undef has_tail

// ---------------------------------------------------------------------------------------------------------------------
// the team body. Returns, for every wave, the result of the LAST primitive the wave assembled (NW = 4: wave a <-> slot a)
// ---------------------------------------------------------------------------------------------------------------------
// RL = true: the four argument structs are references INTO THE KERNARG SEGMENT (paths_kargs()) and are re-read per phase (karg_reload);
// RL = false: plain references to the kernel's by-value arguments, as before.
template <int NW, class P, int RL = 0>
__device__ __forceinline__ WavePath team_paths_body(const DevLat& lat_, const DevPathsIn& in_, const DevPathsOut& out_,
                                                    const TeamLds& lp_, unsigned char* smem, TeamShared& ts,
                                                    double* vel_kappa, double* vel_len, double* vel_x, double* vel_y)
{
    const DevLat* latp = LTPL_KARG_LAT(&lat_); const DevPathsIn* inp = LTPL_KARG_IN(&in_);
    const DevPathsOut* outp = LTPL_KARG_OUT(&out_); const TeamLds* lpp = LTPL_KARG_LP(&lp_);
    // (the body keeps its names: `lat`, `in`, `out`, `lp` are the CURRENT views; LTPL_KARGS() re-reads them at a phase boundary)
#define lat (*latp)
#define in (*inp)
#define out (*outp)
#define lp (*lpp)
#define LTPL_KARGS() do { latp = LTPL_KARG_LAT(latp); inp = LTPL_KARG_IN(inp); outp = LTPL_KARG_OUT(outp); lpp = LTPL_KARG_LP(lpp); } while (0)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int L = lat.L;
    constexpr int NT = NW * 64;

    short* pos_layer = reinterpret_cast<short*>(smem + lp.off_pos_layer);
    unsigned char* pos_veh = smem + lp.off_pos_veh;
    unsigned* blocked_bits = reinterpret_cast<unsigned*>(smem + lp.off_blocked);
    unsigned* zone_bits = reinterpret_cast<unsigned*>(smem + lp.off_zone);
    double* dist = reinterpret_cast<double*>(smem + P::off_dist(lp));
    unsigned char* par = par_base<P>(lp, smem);
    int* best = reinterpret_cast<int*>(smem + P::off_best(lp));
    int4* lay = reinterpret_cast<int4*>(smem + P::off_lay(lp));

    dbg_stamp(lp.dbg, 0);
    if (LTPL_ABLATED(lp, 8)) { WavePath w0; w0.valid = 0; w0.n_pts = 0; w0.n_nodes = 0; w0.name = LTPL_ACT_NONE; w0.reduced = 0; w0.goal_layer = -1; w0.end_node = -1; return w0; }   // timing experiment: launch cost only
    // ---- phase 0: scenario scalars (uniform; the planning range only depends on the start layer and is tabulated at
    //      ltpl_create: gen_local_node_template.py:101-147) -----------------------------------------------------------
    if (LTPL_POISON_ON(lp)) {
        unsigned* w = reinterpret_cast<unsigned*>(smem);
        for (int i = tid; i < lp.total / 4; i += NT) w[i] = lp.poison;
        unsigned* tw = reinterpret_cast<unsigned*>(&ts);
        for (int i = tid; i < (int)(sizeof(TeamShared) / 4); i += NT) tw[i] = lp.poison;
        team_sync<NW>();
    }
    Scen sc;
    sc.s = blockIdx.x;
    if constexpr (NW == 1) { const int* ord = in.order; if (ord) sc.s = ord[blockIdx.x]; }     // large batches: scenarios in the order of their start layers (DevPathsIn::order)
    sc.sl = in.start_layer[sc.s]; sc.sn = in.start_node[sc.s]; sc.flags = in.flags[sc.s];
    sc.veh0 = in.veh_off[sc.s]; sc.n_veh = in.veh_off[sc.s + 1] - sc.veh0;
    sc.n_fac = min(in.n_last[sc.s] - 1, in.n_w_last);
    const int zone0 = in.zone_off[sc.s], zone1 = in.zone_off[sc.s + 1];
    const int const_closest = in.const_closest[sc.s], last_action = in.last_action[sc.s];
    sc.el = at(lat.rng_end, sc.sl);
    sc.pos0 = in.pos_off[sc.veh0]; sc.n_pos = in.pos_off[sc.veh0 + sc.n_veh] - sc.pos0;
    sc.H = sc.el - sc.sl; if (sc.H < 0) sc.H = L - sc.sl + sc.el;
    const int H = sc.H, kpad = P::kpad(lp), hm = P::hmax(lp);

    for (int i = tid; i < lp.words_blocked; i += NT) blocked_bits[i] = 0u;
    for (int i = tid; i < lp.words_zone; i += NT) zone_bits[i] = 0u;
    // per-layer table of the planning range
    // (rows beyond the planning range are harmless: the bound only depends on the lattice, so the loads do not wait for H)
    for (int j = tid; j < hm; j += NT) {
        int b = sc.sl + j; if (b >= L) b -= L;
        const int v0 = at(lat.layer_off, b);
        lay[j] = make_int4(v0, (at(lat.layer_off, b + 1) - v0) | (at(lat.layer_degmax, b) << 16), at(lat.layer_ebase, b), at(lat.layer_ebase, b + 1));
    }
    {
        // pair tid of the previous solution: which transition of the planning range does it span?
        int fj = -1, fs = -1, fd = -1; double fac = 1.0;
        if (tid < sc.n_fac && tid < LTPL_MAX_LAST_NODES) {
            const int* ll = in.last_layer + (size_t)sc.s * LTPL_MAX_LAST_NODES;
            const int* ln = in.last_node + (size_t)sc.s * LTPL_MAX_LAST_NODES;
            const int la = ll[tid], lb = ll[tid + 1];
            int nx = la + 1; if (nx >= L) nx -= L;
            int jj = lb - sc.sl; if (jj < 0) jj += L;
            if (la >= 0 && la < L && lb == nx && jj >= 1 && jj <= sc.H) { fj = jj; fs = ln[tid]; fd = ln[tid + 1]; fac = in.w_last[tid]; }
        }
        if (tid < LTPL_MAX_LAST_NODES) { ts.fac_j[tid] = fj; ts.fac_src[tid] = fs; ts.fac_dst[tid] = fd; ts.fac[tid] = fac; }
        if (wave == 0) {                                   // (every lane of the wave takes part: fj = -1 beyond the list)
            static_assert(LTPL_MAX_LAST_NODES <= 8, "the list lies in the first eight lanes");
            const int mx = oct_max_i32(fj);
            if (tid == 0) ts.fac_jmax = mx;
        }
    }
    if (wave == 0 && (sc.flags & LTPL_FLAG_HAS_PSI_S)) {        // start heading of the constant path segment: one sincos per scenario, not per path
        double sn, cs;
        const double psi0 = in.psi_s[sc.s];                         // (uniform)
        if (fabs(psi0) <= 12.0) heading_sincos(psi0, &sn, &cs); else sincos(psi0, &sn, &cs);
        if (lane == 0) { ts.psi_sc[0] = sn; ts.psi_sc[1] = cs; }
    }
    // vehicle of every position (radius lookup in phase 2)
    for (int k = tid; k < sc.n_veh; k += NT) {
        const int p0 = in.pos_off[sc.veh0 + k] - sc.pos0, p1 = in.pos_off[sc.veh0 + k + 1] - sc.pos0;
        for (int p = p0; p < p1; ++p) pos_veh[p] = (unsigned char)k;
    }
    // reference line -> LDS (aliases the parent table, which is not live before phase 4)
    // (with the closest-layer grid the line is only read by the rare full scan of phase 1, which then takes it from global memory)
    double* refl = reinterpret_cast<double*>(smem + P::off_par(lp));
    const bool ref_staged = lp.ref_lds && lat.lgrid == nullptr;
    if (ref_staged)
        for (int l = tid; l < L; l += NT) { refl[2 * l] = at(lat.ref_x, l); refl[2 * l + 1] = at(lat.ref_y, l); }
    team_sync<NW>();
    {
        // offsets of the planning range from the layer table (no further dependent scalar loads)
        const int4 l0 = lay[0], l1 = lay[1], lH = lay[H];
        sc.e_base = l1.z; sc.n_base = l0.x;
        int NH = lH.x + (lH.y & 0xffff) - sc.n_base; if (NH <= 0) NH += lat.V;
        sc.NH = NH;
    }
    // zone-removed nodes of the "overtaking_zones" filter (gen_local_node_template.py:96; GraphBase.py:713-745)
    {
        const int* const zone_gid = in.zone_gid; const int V = lat.V;
        for (int i = zone0 + tid; i < zone1; i += NT) {
            int nl = zone_gid[i] - sc.n_base; if (nl < 0) nl += V;
            if (nl < sc.NH) atomicOr(&zone_bits[nl >> 5], 1u << (nl & 31));
        }
    }

    dbg_stamp(lp.dbg, 1);
    LTPL_KARGS();
    // ---- phase 1: closest reference-line layer per obstacle position (get_intersec_edges.py:40-51) -----------------
    // Round 5: the closest-layer GRID (layer_grid.hpp, built at ltpl_create) names, for the cell a position lies in, the at most two
    // intervals of layers that can be the argmin for any point of that cell; lane = position evaluates the reference's fp64 distances on
    // those layers only, in ascending order with the full scan's strict '<' (same first minimum). Only when some position of the scenario
    // lies outside the grid or in a "full scan" cell does the scan over ALL layers below run (for every position: it overwrites).
    bool full_scan = true;
    if (lat.lgrid != nullptr) {
        const int4* const g_cells = pin_sgpr(lat.lgrid); const double* const g_rx = pin_sgpr(lat.ref_x); const double* const g_ry = pin_sgpr(lat.ref_y);
        const double gx0 = lat.lg_x0, gy0 = lat.lg_y0, ginv = lat.lg_inv; const int gnx = lat.lg_nx, gny = lat.lg_ny;
        bool miss = false;
        for (int p0 = 0; p0 < sc.n_pos; p0 += NT) {
            const int p = p0 + tid;
            const bool pv = p < sc.n_pos;
            double px = 0.0, py = 0.0; int4 c = make_int4(0, 0, 0, 0);
            if (pv) {
                px = in.pos_x[sc.pos0 + p]; py = in.pos_y[sc.pos0 + p];
                const double fx = (px - gx0) * ginv, fy = (py - gy0) * ginv;
                // (the comparison form also rejects NaN coordinates: they take the full scan like any position off the grid)
                if (fx >= 0.0 && fy >= 0.0 && fx < (double)gnx && fy < (double)gny) c = g_cells[(int)fy * gnx + (int)fx];
                else c.y = -1;
                if (c.y < 0) { miss = true; c.y = 0; c.w = 0; }
            }
            const int tot = c.y + c.w;
            double bd = INFINITY; int bl = 0x7fffffff;
            // four candidates per round trip (the loads of a group are issued before the first compare; a lane beyond its last candidate
            // repeats that one: a repeated distance never wins the strict '<')
            for (int k0 = 0; __ballot(k0 < tot) != 0ull; k0 += 4) {
                int l4[4]; double x4[4], y4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u < tot ? k0 + u : (tot > 0 ? tot - 1 : 0);
                    l4[u] = k < c.y ? c.x + k : c.z + (k - c.y);
                    x4[u] = at(g_rx, l4[u]); y4[u] = at(g_ry, l4[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double dx = x4[u] - px, dy = y4[u] - py;
                    const double d2 = dx * dx + dy * dy;
                    if (k0 < tot && d2 < bd) { bd = d2; bl = l4[u]; }          // (interval 1 lies below interval 2: ascending layers, first minimum)
                }
            }
            if (pv) {
                const int ol = bl, sl = sc.sl, el = sc.el;
                const bool gate = (sl - 1 <= ol && ol <= el + 1) || (sl > el && (sl - 1 <= ol || ol <= el + 1));
                pos_layer[p] = (short)(gate ? ol : -1);
            }
        }
        const bool wave_miss = __ballot(miss) != 0ull;
        if constexpr (NW == 1) full_scan = wave_miss;
        else {
            if (lane == 0) ts.grid_miss[wave] = wave_miss ? 1 : 0;
            team_sync<NW>();
            full_scan = false;
#pragma unroll
            for (int w = 0; w < NW; ++w) full_scan = full_scan || ts.grid_miss[w] != 0;
            if (full_scan) team_sync<NW>();            // (every wave has read the flags before the scan below rewrites pos_layer)
        }
    }
    // The scan over all layers: lane = (layer segment, position), every lane scans its segment of the reference line for its position
    // (strict '<' keeps the first minimum), the segments of a position are then combined by log2(#segments) exchange steps.
    for (int pp0 = 0; pp0 < sc.n_pos && full_scan; pp0 += 64) {
        const int cnt = min(64, sc.n_pos - pp0);
        const int cntw = (cnt - wave + NW - 1) / NW;                 // positions of this wave: q = wave + NW * i
        if (cntw <= 0) continue;
        int np2 = 1; while (np2 < cntw) np2 <<= 1;
        const int nseg = 64 / np2, chunk = (L + nseg - 1) / nseg;
        const int qi = lane & (np2 - 1), seg = lane / np2;
        const int q = wave + NW * qi;
        const bool qv = qi < cntw;
        double px = 0.0, py = 0.0;
        if (qv) { px = in.pos_x[sc.pos0 + pp0 + q]; py = in.pos_y[sc.pos0 + pp0 + q]; }
        double bd = INFINITY; int bl = 0x7fffffff;
        const int l0 = seg * chunk, l1 = min(L, l0 + chunk);
        if (ref_staged) {
            for (int l = l0; l < l1; ++l) {
                const double dx = refl[2 * l] - px, dy = refl[2 * l + 1] - py;
                const double d2 = dx * dx + dy * dy;
                if (d2 < bd) { bd = d2; bl = l; }
            }
        } else {
            const double* const ref_x = lat.ref_x; const double* const ref_y = lat.ref_y;
            for (int l = l0; l < l1; ++l) {
                const double dx = at(ref_x, l) - px, dy = at(ref_y, l) - py;
                const double d2 = dx * dx + dy * dy;
                if (d2 < bd) { bd = d2; bl = l; }
            }
        }
        wave_min2_from(bd, bl, np2);
        if (qv && seg == 0) {
            const int ol = bl, sl = sc.sl, el = sc.el;
            const bool gate = (sl - 1 <= ol && ol <= el + 1) || (sl > el && (sl - 1 <= ol || ol <= el + 1));
            pos_layer[pp0 + q] = (short)(gate ? ol : -1);
        }
    }
    team_sync<NW>();

    dbg_stamp(lp.dbg, 2);
    LTPL_KARGS();
    // ---- phase 2: obstacle x edge-sample mask (GraphBase.get_intersec_edges_in_range, GraphBase.py:567-646) --------
    // Window of a position with closest layer ol = layers [ol-1, ol+1] with the reference's wrap quirks (:597-600):
    // the transitions into layer ol and into layer ol+1 (the latter never across the seam); both end points must lie in
    // the planning range. Transition-major and edge-parallel: positions live in lanes, a ballot selects the positions
    // whose window contains transition j; lane = edge of the transition; an edge whose bounding circle (centre + radius
    // over its samples, tabulated at ltpl_create) cannot reach the obstacle disc is rejected without touching its
    // samples, the others test their samples exactly like the reference (d^2 <= (r + w/2)^2 + step^2 / 4).
    // transitions of the planning range inside SOME obstacle window (bit j; every other transition cannot hold a blocked edge, so the
    // sweep's prefetch skips the bitmap look-up there); planning ranges beyond 63 layers: every transition is looked up
    unsigned long long touched = H <= 63 ? 0ull : ~0ull;
    // (what the loops below read of the argument structs, once)
    const float4* const m_cap = pin_sgpr(lat.edge_cap); const double* const m_sx = pin_sgpr(lat.sx); const double* const m_sy = pin_sgpr(lat.sy);
    const float m_slack = pin_sgpr(lat.cull_slack); const int m_E = pin_sgpr(lat.E), m_shell_cap = pin_sgpr(lp.shell_cap);
    // shell list of this wave: entries (edge in sweep order | query lane << 24, packed sample range of the capsule record)
    uint2* shell = reinterpret_cast<uint2*>(smem + lp.off_shell) + (size_t)(NW == 1 ? 0 : wave) * lp.shell_cap;
    int n_shell = 0;                                               // uniform
    auto shell_push = [&](bool mine, int e, float packed_f, int ql) {
        const unsigned long long mk = __ballot(mine);
        if (mine) shell[n_shell + __popcll(mk & ((1ull << lane) - 1ull))] = make_uint2((unsigned)e | ((unsigned)ql << 24), __float_as_uint(packed_f));
        n_shell += __popcll(mk);
    };
    // exact test of the listed (edge, position) pairs, GraphBase.py:626-643: lane = (entry, sample slot), four entries x 16 samples per
    // pass; positions come from the lanes of the current position batch (mx, my, mr = its x, y and squared threshold)
    auto flush_shell = [&](double mx, double my, double mr) {
        if (n_shell == 0) return;
        wave_sync_lds();
        // SIXTEEN entries per pass, as two independent groups of eight entries x eight sample slots (round 5): a pass is one LDS read and one
        // dependent global round trip (the samples) long whatever it holds, and a scenario with an object in range lists ~45 entries --
        // twelve passes of four entries x 16 slots were the longest stretch of the mask phase (an edge of the reference's tracks has 5 - 7
        // samples). The first LTPL_SHELL_SS samples of both groups are requested before either is evaluated; longer ranges continue per group.
#ifndef LTPL_SHELL_SS
#define LTPL_SHELL_SS 8
#endif
        constexpr int SS = LTPL_SHELL_SS, EPG = 64 / SS;          // sample slots per entry, entries per group
        for (int t0 = 0; t0 < n_shell; t0 += 2 * EPG) {
            const int slot = lane & (SS - 1);
            int e_[2], ql_[2], k0_[2], ns_[2]; bool tv_[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int t = t0 + EPG * g + lane / SS;
                tv_[g] = t < n_shell;
                const uint2 en_ = shell[tv_[g] ? t : t0];
                e_[g] = (int)(en_.x & 0xffffffu); ql_[g] = (int)(en_.x >> 24);
                k0_[g] = (int)(en_.y & 0xffffffu); ns_[g] = (int)(en_.y >> 24);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (ns_[g] == 0) { const int ec_ = at(lat.sw2csc, e_[g]); k0_[g] = at(lat.samp_ptr, ec_); ns_[g] = at(lat.samp_ptr, ec_ + 1) - k0_[g]; }   // (range too large for the packing)
            double sx_[2], sy_[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {                         // first SS samples of both groups: one round trip
                const int k = slot < ns_[g] ? slot : 0;
                sx_[g] = at(m_sx, k0_[g] + k); sy_[g] = at(m_sy, k0_[g] + k);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const double px = __shfl(mx, ql_[g]), py = __shfl(my, ql_[g]), pr = __shfl(mr, ql_[g]);
                bool hit = false;
                { const double dx = sx_[g] - px, dy = sy_[g] - py; hit = slot < ns_[g] && (dx * dx + dy * dy <= pr); }
                for (int k = slot + SS; k < ns_[g]; k += SS) {
                    const double dx = at(m_sx, k0_[g] + k) - px, dy = at(m_sy, k0_[g] + k) - py;
                    hit = hit || (dx * dx + dy * dy <= pr);
                }
                const unsigned long long hm = __ballot(tv_[g] && hit);
                if (tv_[g] && slot == 0 && ((hm >> (lane & (64 - SS))) & ((1ull << SS) - 1ull))) {
                    int el_ = e_[g] - sc.e_base; if (el_ < 0) el_ += m_E;
                    atomicOr(&blocked_bits[el_ >> 5], 1u << (el_ & 31));
                }
            }
        }
        n_shell = 0;
        wave_sync_lds();
    };
    for (int pp0 = 0; pp0 < sc.n_pos && !LTPL_ABLATED(lp, 1); pp0 += 64) {
        const int p = pp0 + lane;
        int ol = -1; double mpx = 0.0, mpy = 0.0, mref = 0.0, msq = 0.0;
        if (p < sc.n_pos) {
            ol = pos_layer[p];
            mpx = in.pos_x[sc.pos0 + p]; mpy = in.pos_y[sc.pos0 + p];
            const double rr = in.veh_radius[sc.veh0 + pos_veh[p]] + lat.veh_width / 2;
            mref = rr * rr;
            mref += (lat.sampled_resolution * lat.sampled_resolution) / 4;
            msq = sqrt(mref);
        }
        unsigned long long gated = __ballot(ol >= 0);
        if (gated == 0ull) continue;
        // transitions of the planning range that any position's window touches (a handful of the H layers): bit j
        const bool sparse = H <= 63;                                    // (long-horizon lattices: every transition is looked at)
        unsigned long long jbits = 0ull;
        if (sparse) {
            int jj = ol - sc.sl; if (jj < 0) jj += L;                 // transition INTO layer ol; INTO ol + 1 is jj + 1 (never across the seam)
            unsigned long long mine = 0ull;
            if (ol >= 0) {
                if (jj >= 1 && jj <= H) mine |= 1ull << jj;
                if (ol + 1 < L && jj + 1 >= 1 && jj + 1 <= H) mine |= 1ull << (jj + 1);
            }
            const unsigned lo = (unsigned)mine, hi = (unsigned)(mine >> 32);
            while (gated) {
                const int src_lane = __ffsll((long long)gated) - 1;
                gated &= gated - 1;
                jbits |= (unsigned long long)__builtin_amdgcn_readlane(lo, src_lane) | ((unsigned long long)__builtin_amdgcn_readlane(hi, src_lane) << 32);
            }
        }
        touched |= jbits;
        // The capsule records of a transition's FIRST chunk are requested one transition ahead (round 5): for the first transition here, in
        // front of the loop -- its round trip passes behind the position loads above --, for every other one while its predecessor is
        // evaluated. (A transition was: broadcast the queries, request the first chunk, wait a full global round trip, evaluate.)
        float4 pf0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pf1 = pf0;
        auto cap_first = [&](int jn) {
            const int4 lyn = lay[jn];
            if (lyn.z + wave * 64 < lyn.w) { const int en_ = min(lyn.z + wave * 64 + lane, lyn.w - 1); pf0 = at(m_cap, 2 * en_); pf1 = at(m_cap, 2 * en_ + 1); }
        };
        // (planning ranges beyond 63 layers walk every transition and find most of them untouched: no look-ahead there)
        { const int j0 = (sparse && jbits) ? __ffsll((long long)jbits) - 1 : 0; if (j0 >= 1 && j0 <= H) cap_first(j0); }
        for (int j = 1; j <= H; ++j) {
            if (sparse) { if (!jbits) break; j = __ffsll((long long)jbits) - 1; jbits &= jbits - 1; }
            int b = sc.sl + j; if (b >= L) b -= L;
            unsigned long long m = __ballot(ol >= 0 && (ol == b || ol + 1 == b));
            const int4 ly = lay[j];
            const int eb = ly.z, ee = ly.w;
            const float4 cur0 = pf0, cur1 = pf1;                 // this transition's first chunk (requested one transition ago)
            { const int jn = (sparse && jbits) ? __ffsll((long long)jbits) - 1 : 0; if (jn >= 1 && jn <= H) cap_first(jn); }
            bool first_round = sparse;
            if (m == 0ull) continue;
            while (m) {
                // up to MQ matching positions per round, broadcast into uniform registers
                constexpr int MQ = LTPL_MQ;
                double qx[MQ], qy[MQ], qr[MQ];
                int qlane[MQ];                                 // lane that holds the query's position (exact test: flush_shell)
                float qxf[MQ], qyf[MQ], qlm[MQ], qlh[MQ];      // fp32 query, thresholds of the two-sided cull (without the edge's own terms)
#pragma unroll
                for (int q = 0; q < MQ; ++q) {
                    if (m) {
                        const int src_lane = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        qlane[q] = src_lane;
                        qx[q] = readlane_f64(mpx, src_lane); qy[q] = readlane_f64(mpy, src_lane);
                        qr[q] = readlane_f64(mref, src_lane);
                        qxf[q] = (float)qx[q]; qyf[q] = (float)qy[q];
                        const float t = (float)readlane_f64(msq, src_lane);          // sqrt of the squared-distance threshold
                        qlm[q] = t * 1.000001f + m_slack;                            // MISS  if dist(q, chord) > qlm + dev
                        qlh[q] = t * 0.999999f - m_slack;                            // HIT   if dist^2 + (gap / 2)^2 <= (qlh - dev)^2
                    } else { qlane[q] = 0; qx[q] = 0.0; qy[q] = 0.0; qr[q] = -1.0; qxf[q] = 0.0f; qyf[q] = 0.0f; qlm[q] = -1.0e30f; qlh[q] = -1.0e30f; }   // always MISS
                }
                // the capsule records of the NEXT chunk of 64 edges are requested before the current chunk is evaluated (round 5: a wide
                // transition -- 245 edges on the C3 oval -- was four dependent load -> evaluate steps per pass)
                float4 nx0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), nx1 = nx0;
                if (first_round) { nx0 = cur0; nx1 = cur1; first_round = false; }
                else
                if (eb + wave * 64 < ee) { const int en_ = min(eb + wave * 64 + lane, ee - 1); nx0 = at(m_cap, 2 * en_); nx1 = at(m_cap, 2 * en_ + 1); }
                for (int e0 = eb + wave * 64; e0 < ee; e0 += NT) {
                    // (predicated, not branched: the list bookkeeping below is wave-uniform)
                    const int e = min(e0 + lane, ee - 1);
                    const float4 c0 = nx0, c1 = nx1;                                  // (Ax, Ay, ABx, ABy), (1 / |AB|^2, dev, hg2, samples)
                    if (e0 + NT < ee) { const int en_ = min(e0 + NT + lane, ee - 1); nx0 = at(m_cap, 2 * en_); nx1 = at(m_cap, 2 * en_ + 1); }
                    int el_ = e - sc.e_base; if (el_ < 0) el_ += m_E;
                    // lanes beyond the transition and edges already blocked by another object take no part
                    const bool live = e0 + lane < ee && !((blocked_bits[el_ >> 5] >> (el_ & 31)) & 1u);
                    // Two-sided cull on the edge's CAPSULE (chord A -> B between its first and last sample, `dev` = largest distance
                    // of a sample from the chord, `hg2` = (half the largest gap between consecutive sample projections)^2, all
                    // tabulated at ltpl_create, rounded so that both decisions are conservative): with d = dist(q, chord),
                    //   every sample is at least d - dev away           -> MISS without touching the samples if d - dev > threshold
                    //   some sample is at most sqrt(d^2 + hg2) + dev away -> HIT without touching the samples if that <= threshold
                    // and only the thin shell in between runs the reference's exact fp64 sample test. (Round 2: with a bounding
                    // CIRCLE nearly every edge of a window was "near" and loaded its samples -- 0.44 ms of a 1.05 ms launch.)
                    bool sure = false;
                    bool unsure_q[MQ];
#pragma unroll
                    for (int q = 0; q < MQ; ++q) {
                        const float ux = qxf[q] - c0.x, uy = qyf[q] - c0.y;
                        float t = (ux * c0.z + uy * c0.w) * c1.x;
                        t = fminf(fmaxf(t, 0.0f), 1.0f);
                        const float dx = ux - t * c0.z, dy = uy - t * c0.w, d2 = dx * dx + dy * dy;
                        const float lm = qlm[q] + c1.y, lh = qlh[q] - c1.y;
                        const bool miss = d2 > lm * lm * 1.000001f || lm < 0.0f;
                        const bool hit_q = lh > 0.0f && (d2 + c1.z) * 1.000001f <= lh * lh;
                        sure = sure || hit_q;
                        unsure_q[q] = !miss && !hit_q;
                    }
                    if (live && sure) atomicOr(&blocked_bits[el_ >> 5], 1u << (el_ & 31));
                    // SHELL edges (neither certain MISS nor certain HIT for some query) need the reference's exact fp64 sample test. They
                    // are few (7 % of the window edges) and scattered over the lanes, so running the sample loop here would drag the whole
                    // wave through ~250 instructions per chunk for a handful of busy lanes: they are APPENDED to a list instead (edge,
                    // query lane, sample range) and tested with lane = (entry, sample) in flush_shell.
                    const bool sh = live && !sure && !LTPL_ABLATED(lp, 32);   // (experiment build: bit 32 treats shell edges as MISS, timing only)
#pragma unroll
                    for (int q = 0; q < MQ; ++q) shell_push(sh && unsure_q[q], e, c1.w, qlane[q]);
                    (void)qx; (void)qy; (void)qr;
                    if (n_shell + MQ * 64 > m_shell_cap) flush_shell(mpx, mpy, mref);
                }
            }
        }
        flush_shell(mpx, mpy, mref);                                 // (the query lanes change with the next batch of positions)
    }

    dbg_stamp(lp.dbg, 3);
    LTPL_KARGS();
    // ---- phase 3: closest object (gen_local_node_template.py:191-213) and action template (mopg.py:124-174) --------
    if (wave == 0) {
        // closest = smallest layer distance of the vehicle's LAST position; first vehicle wins ties
        int key = 0x7fffffff;
        double vpx = 0.0, vpy = 0.0;                       // first position of vehicle `lane` (vehicles 0 .. 63)
        for (int k = lane; k < sc.n_veh; k += 64) {
            const int pfirst = in.pos_off[sc.veh0 + k], pnext = in.pos_off[sc.veh0 + k + 1];
            if (k < 64) { vpx = in.pos_x[pfirst]; vpy = in.pos_y[pfirst]; }
            const int plast = pnext - 1 - sc.pos0;
            const int ol = (plast >= 0 && plast < sc.n_pos && pnext > pfirst) ? (int)pos_layer[plast] : -1;
            if (ol >= 0) {
                int ld = ol - sc.sl; if (ld < 0) ld = L - sc.sl + ol;
                if (ld <= H) { const int kk = ld * 256 + k; if (kk < key) key = kk; }
            }
        }
        key = wave_min_i32(key);
        int ci = -1, cl = -1, have = 0, cn = -1;
        if (key != 0x7fffffff) {
            ci = key & 255; have = 1;
            cl = sc.sl + (key >> 8); if (cl >= L) cl -= L;
            double px, py;
            if (ci < 64) { px = readlane_f64(vpx, ci); py = readlane_f64(vpy, ci); }
            else { const int p = in.pos_off[sc.veh0 + ci]; px = in.pos_x[p]; py = in.pos_y[p]; }
            const int4 ly = lay[key >> 8];
            const int v0 = ly.x, K = ly.y & 0xffff;
            double bd = INFINITY, dummy = 0.0; int bn = 0x7fffffff;
            for (int n = lane; n < K; n += 64) {
                const double dx = at(lat.node_x, v0 + n) - px, dy = at(lat.node_y, v0 + n) - py;
                const double d2 = dx * dx + dy * dy;
                if (d2 < bd) { bd = d2; bn = n; }
            }
            {   // closest node: the squared distance alone first (a tie falls back to the (distance, node) reduction)
                const double mb = wave_min_f64(bd);
                const unsigned long long eq = __ballot(bd == mb);
                if ((eq & (eq - 1ull)) == 0ull && eq != 0ull) bn = __builtin_amdgcn_readlane(bn, __ffsll((long long)eq) - 1);
                else wave_min3(bd, dummy, bn);
            }
            cn = bn;
        }
        if (lane == 0) { ts.closest_idx = ci; ts.cl = cl; ts.cn = cn; ts.have_cn = have; }
    }
    team_sync<NW>();
    if (lp.mask_out) {
        // diagnostics: phase 2's result as it stands (every atomicOr of every wave lies in front of the barrier above)
        unsigned* mo = lp.mask_out + (size_t)sc.s * (size_t)(lp.words_blocked + 2);
        if (tid == 0) { mo[0] = (unsigned)sc.e_base; mo[1] = (unsigned)H; }
        for (int i = tid; i < lp.words_blocked; i += NT) mo[2 + i] = blocked_bits[i];
    }
    const int t_cl = ts.cl, t_cn = ts.cn, t_have = ts.have_cn, fac_jmax = ts.fac_jmax;
    // action template (uniform, every thread)
    int n_act = 0, filt[LTPL_MAX_ACTIONS], nm0[LTPL_MAX_ACTIONS];
    int closest_idx = ts.closest_idx;
    {
        const bool action_sets = sc.flags & LTPL_FLAG_ACTION_SETS, in_const = sc.flags & LTPL_FLAG_OBJ_IN_CONST,
                   besides = sc.flags & LTPL_FLAG_OBJ_BESIDES;
        if (const_closest >= 0) closest_idx = const_closest;
        for (int a = 0; a < LTPL_MAX_ACTIONS; ++a) { filt[a] = F_DEF; nm0[a] = LTPL_ACT_NONE; }
        if (action_sets && (in_const || besides)) {
            filt[n_act] = F_PR; nm0[n_act++] = LTPL_ACT_FOLLOW;
            const int la = last_action;
            if (!in_const && (la == LTPL_ACT_LEFT || la == LTPL_ACT_RIGHT)) { filt[n_act] = F_DEF; nm0[n_act++] = la; }
            else if (!in_const) {
                filt[n_act] = F_DEF; nm0[n_act++] = LTPL_ACT_LEFT;
                filt[n_act] = F_DEF; nm0[n_act++] = LTPL_ACT_RIGHT;
            }
        } else if (action_sets && closest_idx >= 0 && t_have) {
            filt[0] = F_PR; nm0[0] = LTPL_ACT_FOLLOW;
            filt[1] = F_LEFT; nm0[1] = LTPL_ACT_LEFT;
            filt[2] = F_RIGHT; nm0[2] = LTPL_ACT_RIGHT;
            n_act = 3;
        } else {
            filt[0] = F_DEF; nm0[0] = LTPL_ACT_STRAIGHT; n_act = 1;
        }
    }
    unsigned need = 0;
    for (int a = 0; a < n_act; ++a) need |= 1u << filt[a];
    // overtake_left / overtake_right differ from `default` only from the object layer on: branch them off the
    // `default` sweep there (jcl = distance of the object layer from the start layer)
    const bool lr = (need & ((1u << F_LEFT) | (1u << F_RIGHT))) != 0;
    int jcl = 0;
    if (lr) { jcl = t_cl - sc.sl; if (jcl < 0) jcl += L; }
    const bool share_prefix = lr && jcl >= 1;
    if (tid == 0) {
        out.end_layer[sc.s] = sc.el;
        out.closest_obj_index[sc.s] = closest_idx;
        out.closest_obj_node[2 * sc.s] = t_have ? t_cl : -1;
        out.closest_obj_node[2 * sc.s + 1] = t_have ? t_cn : -1;
        out.n_actions[sc.s] = n_act;
    }

    dbg_stamp(lp.dbg, 4);
    LTPL_KARGS();
    // does any zone node lie in the planning range at all? (every wave reads the complete bitmap: the same answer team-wide)
    bool zone_any = true;
    {
        unsigned zw = 0u;
        for (int i = lane; i < lp.words_zone; i += 64) zw |= zone_bits[i];
        zone_any = __ballot(zw != 0u) != 0ull;
    }
    const SweepK swk{pin_sgpr(lat.sw_cost), pin_sgpr(lat.sw_meta), pin_sgpr(lat.E), pin_sgpr(lat.V), blocked_bits, zone_bits, zone_any};
    // ---- phase 4: layered min-plus sweeps (GraphBase.search_graph_layer, GraphBase.py:854-894) ---------------------
    // Edge-parallel (team_layer): the edges of a transition (cost + packed source / destination / rank) are loaded with
    // coalesced loads one layer AHEAD into registers, so that the global latency hides behind the LDS work of the current
    // layer; all active filters share the edge registers.
    // The sweeps run on SWN waves of the team (= all of them). Round 3 tried ONE sweeping wave inside the four-wave latency kernels (a
    // layer = a handful of LDS round trips of one wave instead of four workgroup barriers): SLOWER -- k_tick 77 -> 85 us, C5 (600 layers)
    // 1.69 -> 2.01 ms. A wave that is alone on its SIMD issues an instruction every ~13 cycles; four waves that split edges and filters
    // each run a quarter of the instructions, which outweighs the barriers.
    auto sweeps = [&](auto swn_tag) {
        constexpr int SWN = decltype(swn_tag)::value;
        constexpr int SNT = SWN * 64;
        constexpr int CH = SWN == 1 ? P::ch1 : 1;     // register chunks of 64 * SWN edges prefetched per layer transition
        const int swave = SWN == 1 ? 0 : wave;
        // initial frontier of every filter that starts at layer 0
        for (int f = swave; f < NFILT; f += SWN) {
            const bool own_start = (need >> f) & 1u || (f == F_DEF && share_prefix);
            if (!own_start) continue;
            const int K0 = lay[0].y & 0xffff;
            const bool ok = sc.sn >= 0 && sc.sn < K0 &&
                            !team_node_removed(zone_bits, swk.V, sc, t_cl, t_cn, f, sc.sl, sc.sn, sc.n_base + sc.sn);
            double* d0 = dist + (size_t)(f * 2) * kpad;
            for (int n = lane; n < kpad; n += 64) d0[n] = (ok && n == sc.sn) ? 0.0 : INFINITY;
            if (lane == 0) { ts.start_ok[f] = ok; best[f * hm] = -1; }
        }
        EdgeRegs er[CH], en[CH];
        // (every chunk starts as the sentinel edge: the chunks a transition does not fill are never processed, but they ARE copied by the
        //  rotation and looked at by the discount patch -- reading an indeterminate value is undefined behaviour the optimiser may build on)
#pragma unroll
        for (int ci = 0; ci < CH; ++ci) { er[ci].c = INFINITY; er[ci].meta = 0u; en[ci].c = INFINITY; en[ci].meta = 0u; }
        int4 lyc = make_int4(0, 0, 0, 0), lyn = make_int4(0, 0, 0, 0);     // table rows lay[j] of the current / the prefetched transition (uniform)
        // blocked flags of the lane's edges, bit ci = chunk ci, for the current (bm) and the next (bn) transition. One VECTOR register
        // each: as ballots (one scalar pair per chunk and buffer) they were twelve scalar registers that the compiler kept spilling
        // and reloading inside the layer loop.
        unsigned bm = 0u, bn = 0u;
        // Round 5: the loads of a chunk are addressed as (uniform pointer to the chunk's first edge) + (lane * element size) -- no per-lane
        // index, no redirection: a chunk is read whole, and what lies beyond the transition (the next transition's edges, or the sentinel
        // entries behind the tables: LTPL_SW_PAD) is replaced by the sentinel cost +inf where the layer step CONSUMES the chunk (team_layer:
        // one compare of the lane against the uniform edge count; at load time the select would wait for the load). The edge word of such a
        // lane is arbitrary but addresses nodes of the tables: with the candidate +inf it changes nothing (its atomics are issued all the same). The blocked flags are only
        // formed for a transition that can hold a blocked edge (`touched`), in a block of their own behind the loads.
        auto prefetch = [&](int j, EdgeRegs (&dr)[CH], unsigned& db) {
            const int4 ly = uniform_i4(lay[j]);
            lyn = ly;                                             // (the layer loop takes the next layer's table row from here: LTPL_LY_CARRY)
            // (the lane offsets pass through an empty asm statement HERE: hoisted out of the layer loop as 64-bit values they cost two
            //  register pairs and a 64-bit vector add per load instead of the scalar-base + 32-bit-lane-offset addressing; the chunks of a
            //  transition differ by an immediate offset)
            const int e0 = __builtin_amdgcn_readfirstlane(ly.z + swave * 64);     // uniform: first edge of this wave's first chunk
            const char* pc = reinterpret_cast<const char*>(pin_sgpr(swk.sw_cost + e0));
            const char* pm = reinterpret_cast<const char*>(pin_sgpr(swk.sw_meta + e0));
            unsigned o8 = (unsigned)lane * 8u, o4 = (unsigned)lane * 4u;
            asm volatile("" : "+v"(o8), "+v"(o4));
#pragma unroll
            for (int ci = 0; ci < CH; ++ci) {
                if (ci >= LTPL_CH_ALWAYS && ly.z + ci * SNT >= ly.w) continue;    // uniform: chunk 0 is always loaded, the rest on demand
                dr[ci].c = *reinterpret_cast<const double*>(pc + o8 + ci * SNT * 8);
                dr[ci].meta = *reinterpret_cast<const unsigned*>(pm + o4 + ci * SNT * 4);
            }
            db = 0u;
            if (j > 63 || ((touched >> j) & 1ull)) {              // uniform: can this transition hold a blocked edge at all?
                unsigned bw[CH]; int el[CH];
#pragma unroll
                for (int ci = 0; ci < CH; ++ci) {                // (all reads first, one wait; lanes beyond the transition read its last edge's word)
                    bw[ci] = 0u; el[ci] = ly.w;
                    if (ci >= LTPL_CH_ALWAYS && ly.z + ci * SNT >= ly.w) continue;
                    el[ci] = ly.z + (ci * SWN + swave) * 64 + lane;
                    int el_ = (el[ci] < ly.w ? el[ci] : ly.w - 1) - sc.e_base; if (el_ < 0) el_ += swk.E;
                    bw[ci] = blocked_bits[el_ >> 5] >> (el_ & 31);
                }
#pragma unroll
                for (int ci = 0; ci < CH; ++ci) db |= (el[ci] < ly.w ? (bw[ci] & 1u) : 0u) << ci;
            }
        };
        team_sync<SWN>();
        prefetch(1, er, bm);
        lyc = lyn;
        // `planning_range` (every edge) and `default` (unblocked edges) only differ from the first transition on that holds a
        // blocked edge: in front of it the planning_range sweep is not run, `default`'s frontier, parents and reachability are
        // copied when the sweeps part (one-wave batch form; a transition with edges beyond the register image parts conservatively)
        bool pr_shared = SWN == 1 && !P::par_global && ((need >> F_PR) & 1u) && (((need >> F_DEF) & 1u) || share_prefix);
        auto copy_def_to_pr = [&](int jn, int buf) {          // layers [0, jn) are complete, frontier of layer jn - 1 in buffer `buf`
            for (int n = lane; n < kpad; n += 64) dist[(size_t)(F_PR * 2 + buf) * kpad + n] = dist[(size_t)(F_DEF * 2 + buf) * kpad + n];
            const int roww = kpad * P::par_entry / 4;           // 4-byte words per layer row (kpad is a multiple of 4)
            unsigned* p0 = reinterpret_cast<unsigned*>(par) + (size_t)par_tab(F_PR) * hm * roww;
            const unsigned* p1 = reinterpret_cast<const unsigned*>(par) + (size_t)par_tab(F_DEF) * hm * roww;
            for (int w = roww + lane; w < jn * roww; w += 64) p0[w] = p1[w];
            for (int r = lane; r < jn; r += 64) best[F_PR * hm + r] = best[F_DEF * hm + r];
            if (lane == 0) ts.start_ok[F_PR] = ts.start_ok[F_DEF];
            team_sync<SWN>();
        };
        // The layer loop is split into RUNS of layers that share one configuration (set of advancing filters, `planning_range` still
        // riding on `default` or not): the per-layer decisions (which filters advance, first own layer of left / right, ...) are taken
        // once per run by the driver below, the loop of a run is specialised for its filter set at compile time and carries no template
        // logic. (As one loop with the decisions inside, the uniform booleans lived in scalar register PAIRS across the whole sweep and
        // were spilled: ~20 v_readlane reloads per layer.)
        auto layer_args = [&](int j, const int4& ly, bool from_def, EdgeRegs (&er)[CH]) {
            LayerArgs A;
            int b = sc.sl + j; if (b >= L) b -= L;
            A.j = j; A.b = b; A.v0 = ly.x; A.Kb = ly.y & 0xffff; A.ne = ly.w - ly.z; A.eb = ly.z; A.kpad = kpad; A.hm = hm; A.cur = j & 1; A.prv = (j - 1) & 1;
            A.H = H; A.cl_hit = (b == t_cl) ? 1 : 0; A.cn = t_cn;
            A.fs = -1; A.fd = -1; A.fac = 1.0;
            if (j <= fac_jmax) {                                  // (first layers only; the values are uniform: scalar registers, scalar tests)
                for (int i = 0; i < sc.n_fac; ++i)
                    if (__builtin_amdgcn_readfirstlane(ts.fac_j[i]) == j) {
                        A.fs = __builtin_amdgcn_readfirstlane(ts.fac_src[i]); A.fd = __builtin_amdgcn_readfirstlane(ts.fac_dst[i]);
                        A.fac = uniform_f64(ts.fac[i]); break;
                    }
            }
            A.from_def = from_def;
            if (A.fs >= 0) {
                // previous-solution discount (first layers only): patch the one edge in the register image
#pragma unroll
                for (int ci = 0; ci < CH; ++ci)         // (the sentinel lanes carry +inf: +inf * 0 would be NaN)
                    if (sw_src(er[ci].meta) == A.fs && sw_dst(er[ci].meta) == A.fd && er[ci].c < INFINITY) er[ci].c *= A.fac;
            }
            return A;
        };
        auto rotate = [&]() {
#pragma unroll
            for (int ci = 0; ci < CH; ++ci) er[ci] = en[ci];
            bm = bn; lyc = lyn;
            team_sync<SWN>();
        };
        // layers j0 .. j1 for the compile-time filter set of `act_tag`; stops in front of the first layer that needs something else:
        // why = 1: `planning_range` has to part from `default` here (a blocked edge, or edges beyond the register image),
        // why = 2: more than 64 nodes in the layer (serial form; runtime LDS plans only). Returns the first layer NOT done.
        auto run = [&](auto act_tag, int j0, int j1, bool riding, bool from_def, int& why) -> int {
            constexpr unsigned ACT = decltype(act_tag)::value;
            why = 0;
            int j = j0;
            [[maybe_unused]] constexpr bool one_filter = ACT == (1u << F_DEF) || ACT == (1u << F_PR);
            for (; j <= j1; ++j) {
                const int4 ly = lyc;                               // = lay[j], read with the prefetch of this transition one layer ago
                if (riding) { if (__ballot(bm != 0u) != 0ull || ly.w - ly.z > CH * SNT) { why = 1; break; } }
                if constexpr (!P::fixed) { if ((ly.y & 0xffff) > 64) { why = 2; break; } }
                const LayerArgs A = layer_args(j, ly, from_def && j == j0, er);
                if (j < H) prefetch(j + 1, en, bn);                // global loads in flight during this layer's LDS work
                if constexpr (SWN == 1 && CH == 3) {               // one-wave batch form: the layer body specialised for 1 / 2 / 3 chunks of edges
                    if (A.ne <= 64) team_layer<P, SWN, CH, ACT, 1>(swk, sc, lp, smem, A, er, bm, swave, lane);
                    else if (A.ne <= 128) team_layer<P, SWN, CH, ACT, 2>(swk, sc, lp, smem, A, er, bm, swave, lane);
                    else team_layer<P, SWN, CH, ACT, 3>(swk, sc, lp, smem, A, er, bm, swave, lane);
                } else team_layer<P, SWN, CH, ACT>(swk, sc, lp, smem, A, er, bm, swave, lane);
                rotate();
            }
            return j;
        };
        // one layer in the serial form (lane = node): more than 64 nodes in the layer, or a filter set without a specialisation
        auto serial_layer = [&](int j, unsigned actm, bool from_def) {
            const LayerArgs A = layer_args(j, lay[j], from_def, er);
            if (j < H) prefetch(j + 1, en, bn);
            const SerialK srk = serial_k(lat, lp, smem);          // (rare path: read here, not held across the edge-parallel runs)
            for (int f = swave; f < NFILT; f += SWN) {
                if (!((actm >> f) & 1u)) continue;
                const int fprev = (A.from_def && (f == F_LEFT || f == F_RIGHT)) ? F_DEF : f;
                double* dcur = dist + (size_t)(f * 2 + A.cur) * kpad;
                const bool any = team_relax_layer<P>(srk, sc, lp, t_cl, t_cn, f, j, A.b, A.v0, A.Kb,
                                                  dist + (size_t)(fprev * 2 + A.prv) * kpad, dcur,
                                                  par, ((size_t)par_tab(f) * hm + j) * kpad, lane, A.fs, A.fd, A.fac);
                if (lane == 0) best[f * hm + j] = any ? -2 : -1;
            }
            rotate();
        };
        for (int j = 1; j <= H && !LTPL_ABLATED(lp, 2);) {
            // filters that advance through layer j; left / right read `default`'s frontier in their first own layer (j == jcl)
            const bool from_def = share_prefix && j == jcl;
            if (pr_shared && from_def) { copy_def_to_pr(j, (j - 1) & 1); pr_shared = false; }
            unsigned actm = 0;
            if (((need >> F_PR) & 1u) && !pr_shared) actm |= 1u << F_PR;
            if (((need >> F_DEF) & 1u) || (share_prefix && j < jcl)) actm |= 1u << F_DEF;
            if (((need >> F_LEFT) & 1u) && (!share_prefix || j >= jcl)) actm |= 1u << F_LEFT;
            if (((need >> F_RIGHT) & 1u) && (!share_prefix || j >= jcl)) actm |= 1u << F_RIGHT;
            const int j1 = (share_prefix && j < jcl) ? jcl - 1 : (from_def ? j : H);       // last layer of this configuration
            int why = 0, jn;
            switch (actm) {          // the action templates only produce these filter sets (phase 3); anything else takes the serial form
                case (1u << F_DEF):
                    jn = run(std::integral_constant<unsigned, (1u << F_DEF)>(), j, j1, pr_shared, from_def, why); break;
                case (1u << F_PR):
                    jn = run(std::integral_constant<unsigned, (1u << F_PR)>(), j, j1, false, from_def, why); break;
                case (1u << F_PR) | (1u << F_DEF):
                    jn = run(std::integral_constant<unsigned, (1u << F_PR) | (1u << F_DEF)>(), j, j1, false, from_def, why); break;
                case (1u << F_PR) | (1u << F_LEFT) | (1u << F_RIGHT):
                    jn = run(std::integral_constant<unsigned, (1u << F_PR) | (1u << F_LEFT) | (1u << F_RIGHT)>(), j, j1, false, from_def, why); break;
                default:
                    // riding `planning_range` parts in front of a layer that holds blocked edges here as well
                    if (pr_shared) {
                        const int4 ly = lay[j];
                        if (__ballot(bm != 0u) != 0ull || ly.w - ly.z > CH * SNT) { jn = j; why = 1; break; }
                    }
                    serial_layer(j, actm, from_def); jn = j + 1; break;
            }
            if (why == 1) { copy_def_to_pr(jn, (jn - 1) & 1); pr_shared = false; }
            else if (why == 2) { serial_layer(jn, actm, from_def && jn == j); jn = jn + 1; }   // (the run has already checked that a riding
                                                                                               //  `planning_range` need not part here)
            j = jn;
        }
        if (pr_shared && !LTPL_ABLATED(lp, 2)) copy_def_to_pr(H + 1, H & 1);        // no blocked edge in the whole range: identical sweeps
        // goal node of the last layer for every filter that reached it (virtual goal edges, GraphBase.py:188-194)
        if (!LTPL_ABLATED(lp, 2)) {
            const int4 lyH = lay[H];
            for (int f = swave; f < NFILT; f += SWN)
                if (((need >> f) & 1u) && best[f * hm + H] == -2) {
                    const int g = team_goal(lat, dist + (size_t)(f * 2 + (H & 1)) * kpad, lyH.x, lyH.y & 0xffff, lane);
                    if (lane == 0) best[f * hm + H] = g;
                }
        }
    };
    sweeps(std::integral_constant<int, NW>());
    {
        if (share_prefix && wave == 0 && lane == 0) { ts.start_ok[F_LEFT] = ts.start_ok[F_DEF]; ts.start_ok[F_RIGHT] = ts.start_ok[F_DEF]; }
        team_sync<NW>();
    }

    if constexpr (P::par_global) {
        // the parent tables were written with global stores: drain them before another wave of the team backtracks
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        team_sync<NW>();
    }
    dbg_stamp(lp.dbg, 5);
    LTPL_KARGS();
    // ---- phase 5: search loop with horizon back-off (main_online_path_gen.py:187-248); uniform, every thread ---------
    // per action slot, ONE packed word (valid | reduced << 1 | (name + 1) << 2 | layer distance << 8): these uniform values live across the
    // whole assembly; as four int arrays they ended up as a VGPR tuple in scratch once the kernel was at its 128-register budget
    int slot_pk[LTPL_MAX_ACTIONS];
    int job_base = 0;                                   // lanes 0 / 1: first reserved job index of the generic / follow class
    auto slot_valid = [&](int a) { return slot_pk[a] & 1; };
    auto slot_red = [&](int a) { return (slot_pk[a] >> 1) & 1; };
    auto slot_name = [&](int a) { return ((slot_pk[a] >> 2) & 63) - 1; };
    auto slot_j = [&](int a) { return slot_pk[a] >> 8; };
    {
        const bool in_const = sc.flags & LTPL_FLAG_OBJ_IN_CONST;
        int mod_j = H;
        for (int a = 0; a < LTPL_MAX_ACTIONS; ++a) {
            slot_pk[a] = (LTPL_ACT_NONE + 1) << 2;
            if (a >= n_act) continue;
            const int f = filt[a]; int nm = nm0[a];
            bool found = false;
            for (;;) {
                if (mod_j == 0) break;
                // reachability of layer mod_j under filter f (prefix layers of left / right live in the default table)
                const int pf = (share_prefix && (f == F_LEFT || f == F_RIGHT) && mod_j < jcl) ? F_DEF : f;
                found = ts.start_ok[f] && best[pf * hm + mod_j] != -1;
                if (found || !(nm == LTPL_ACT_FOLLOW || nm == LTPL_ACT_STRAIGHT)) break;
                mod_j -= 1;
            }
            // main_online_path_gen.py:223-224: also when an open track's planning range ends in the last layer
            const bool reduced = mod_j != H || (!lat.closed && sc.el == L - 1);
            int goal = sc.sl + mod_j; if (goal >= L) goal -= L;
            if (reduced) {
                const int cl = t_cl, sl = sc.sl;
                const bool in_mod = t_have && ((sl <= cl && cl <= goal) || (sl > goal && (cl >= sl || cl <= goal)));
                if (!in_const && t_have && !in_mod) {
                    if (nm == LTPL_ACT_FOLLOW || nm == LTPL_ACT_STRAIGHT) nm = LTPL_ACT_STRAIGHT;
                    else found = false;
                }
            }
            slot_pk[a] = (found ? 1 : 0) | (reduced ? 2 : 0) | ((nm + 1) << 2) | (mod_j << 8);
        }
        // batch pipeline, one-wave form: reserve the velocity jobs of ALL valid paths of the scenario with one atomic per class
        // (lane 0: generic profiles, lane 1: follow), issued here so that its round trip (it was 38 us of the launch as a
        // per-path atomic in front of each assembly) passes behind the output stores and the backtrack
        if constexpr (NW == 1) {
            if (out.job_cnt) {
                int n_gen = 0, n_fol = 0;
                for (int a = 0; a < n_act; ++a) if (slot_valid(a)) { if (slot_name(a) == LTPL_ACT_FOLLOW) ++n_fol; else ++n_gen; }
                const int mine = lane == 0 ? n_gen : n_fol;
                if (lane < 2 && mine > 0) job_base = atomicAdd(&out.job_cnt[lane], mine);
            }
        }
        if (tid == 0) {
            for (int a = 0; a < LTPL_MAX_ACTIONS; ++a) {
                const int slot = sc.s * LTPL_MAX_ACTIONS + a;
                if (a >= n_act) {
                    out.action_id[slot] = LTPL_ACT_NONE; out.valid[slot] = 0; out.reduced[slot] = 0; out.goal_layer[slot] = -1;
                    out.n_nodes[slot] = 0; out.n_pts[slot] = 0; out.n_ties[slot] = 0;
                    continue;
                }
                int goal = sc.sl + slot_j(a); if (goal >= L) goal -= L;
                out.action_id[slot] = slot_name(a); out.reduced[slot] = slot_red(a); out.goal_layer[slot] = goal;
                out.valid[slot] = slot_valid(a);
                if (!slot_valid(a)) { out.n_nodes[slot] = 0; out.n_pts[slot] = 0; out.n_ties[slot] = 0; }
            }
        }
    }
    // goal nodes of reduced-horizon paths: re-sweep the filter up to that layer (rare). A left / right path that ends
    // in front of the object layer is a `default` path.
    {
        bool any_resweep = false;
        for (int f = 0; f < NFILT; ++f) {
            int Jf = -1;
            for (int a = 0; a < n_act; ++a) if (slot_valid(a) && filt[a] == f && slot_j(a) != H) Jf = slot_j(a);
            if (Jf < 0) continue;
            any_resweep = true;
            if (wave == f % NW) team_resweep<P>(lat, in, sc, lp, smem, ts, f, Jf, lane);
        }
        if (any_resweep) {
            if constexpr (P::par_global) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // parent stores of the re-sweep
            team_sync<NW>();
        }
    }

    dbg_stamp(lp.dbg, 6);
    LTPL_KARGS();
    // ---- phase 6: wave (a mod NW) assembles primitive a --------------------------------------------------------------
    WavePath wp; wp.valid = 0; wp.n_pts = 0; wp.n_nodes = 0; wp.name = LTPL_ACT_NONE; wp.reduced = 0; wp.goal_layer = -1;
    wp.end_node = -1;
    // one-wave batch form with one-byte parents in LDS: all backtracks of the scenario in one pass (rows in the frontier arrays, which are
    // free behind the goal evaluation and the re-sweeps)
    constexpr int BT_OFF = BtPlace<P, NW>::off;
    constexpr bool BT_ALL = BT_OFF >= 0;
    unsigned char* bt = smem + (BT_ALL ? BT_OFF : 0);
    if constexpr (BT_ALL) {
        if (!LTPL_ABLATED(lp, 4)) {
            int my_pk = 0, my_f = 0;
#pragma unroll
            for (int a = 0; a < LTPL_MAX_ACTIONS; ++a) if (lane == a) { my_pk = slot_pk[a]; my_f = filt[a]; }
            const bool my_sp = share_prefix && (my_f == F_LEFT || my_f == F_RIGHT) && (my_pk >> 8) == H;
            team_backtrack_all<P>(out, lp, smem, sc.s, my_pk, my_f, my_sp, jcl, lane, n_act, bt);
        }
    }
    for (int a = wave; a < n_act; a += NW) {
        wp.name = slot_name(a); wp.reduced = slot_red(a); wp.valid = 0;
        if (!slot_valid(a) || LTPL_ABLATED(lp, 4)) continue;
        unsigned char* pw = smem + P::off_path(lp) + (size_t)(wave < P::n_path_bufs(lp) ? wave : P::n_path_bufs(lp) - 1) * P::path_stride(lp);
        // after a re-sweep the parents of every layer belong to filter f itself
        const bool sp = share_prefix && (filt[a] == F_LEFT || filt[a] == F_RIGHT) && slot_j(a) == H;
        int vtile_in = -1;
        if constexpr (NW == 1) {
            if (out.job_cnt) {
                const int cls = slot_name(a) == LTPL_ACT_FOLLOW ? 1 : 0;
                int k = 0;                                  // valid paths of the same class in front of this one
                for (int b = 0; b < a; ++b) if (slot_valid(b) && (slot_name(b) == LTPL_ACT_FOLLOW) == (cls == 1)) ++k;
                vtile_in = __builtin_amdgcn_readlane(job_base, cls) + k + (cls ? out.n_slots_pad : 0);
            }
        }
        wp = team_assemble<P, RL>(lat, in, out, sc, lp, smem, a, filt[a], slot_j(a), slot_name(a), slot_red(a), jcl, sp, lane, pw,
                               vel_kappa, vel_len, vel_x, vel_y, vtile_in, ts.psi_sc, BT_ALL ? bt + a * LTPL_BT_PITCH : nullptr);
    }
    dbg_stamp(lp.dbg, 7);
    return wp;
#undef LTPL_KARGS
#undef lat
#undef in
#undef out
#undef lp
}
